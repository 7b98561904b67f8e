#!/bin/bash
# A/B of the wide-layer weight-gradient kernel (conv_wgrad_dma_kernel) inside one gpurun call: per-shape table + training step time.
set -u
mkdir -p gpurun_out
for v in 0 1; do
  FX_WGRAD_DMA=$v FX_WGRAD_TABLE=gpurun_out/wgrad_table_dma$v.txt timeout 300 python bench.py --train --steps 10 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']
print('FX_WGRAD_DMA=$v', d['value'], 'img/s', d['ms_per_step'], 'ms; wgrad family', r['ms_per_step'], 'ms serial,', r['achieved'], 'TFLOP/s, frac', r['frac'])"
done
echo "--- per shape, old kernel"; head -16 gpurun_out/wgrad_table_dma0.txt
echo "--- per shape, wide-layer kernel on"; head -16 gpurun_out/wgrad_table_dma1.txt
