#!/bin/bash
# sweep of the wide-layer weight-gradient kernel's workgroup target inside one gpurun call
for w in ${WGS_LIST:-64 96 128 160 64 96 128 160}; do
  FX_WGRAD_DMA_WGS=$w timeout 300 python bench.py --train --steps 10 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']
print('FX_WGRAD_DMA_WGS=$w', d['value'], 'img/s', d['ms_per_step'], 'ms; wgrad family', r['ms_per_step'], 'ms serial,', r['achieved'], 'TFLOP/s')"
done
