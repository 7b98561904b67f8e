#!/bin/bash
# A/B of BiSeNetFormer-L inference (bs=32 640^2) inside one gpurun call: arguments are environment assignments ("-" = default)
for rep in 1 2; do
for cfg in "$@"; do
  if [ "$cfg" = "-" ]; then e=""; else e="$cfg"; fi
  v=$(env $e timeout 300 python bench.py --model bisenetformer-l-ade --no-cpu-baseline --no-other-configs --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])")
  echo "rep$rep [$cfg] img/s ms/step: $v"
done
done
