#!/bin/bash
# A/B of the mask criterion (scripts/bench_mask_criterion.py: all prediction sets, MaskFormer and BiSeNetFormer training shapes) in one gpurun call
for cfg in "$@"; do
  if [ "$cfg" = "-" ]; then e=""; else e="$cfg"; fi
  echo "[$cfg]"; env $e timeout 300 python scripts/bench_mask_criterion.py --iters 8 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   ', d['workload'][:40], d['gpu_ms_per_call'], 'ms', d['total_loss'])"
done
