#!/bin/bash
# A/B of MaskFormer-L inference (bs=16 800^2) inside one gpurun call: arguments are environment assignments ("-" = default)
for rep in 1 2; do
for cfg in "$@"; do
  if [ "$cfg" = "-" ]; then e=""; else e="$cfg"; fi
  v=$(env $e timeout 300 python bench.py --model fai-mf-l-coco-ins --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])")
  echo "rep$rep [$cfg] img/s ms/step: $v"
done
done
