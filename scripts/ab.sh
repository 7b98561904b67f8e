#!/bin/bash
# A/B runs of the headline bench inside ONE gpurun call (box-to-box variation is ~3 %): each argument is an environment
# assignment string (use "-" for the default), e.g.  bash scripts/ab.sh - "FX_PW_CHAIN_MAX_STAGE=2" "FX_STREAMS=1"
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2; do
for cfg in "$@"; do
  if [ "$cfg" = "-" ]; then e=""; else e="$cfg"; fi
  v=$(env $e timeout 300 python bench.py --no-cpu-baseline --no-other-configs --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline']['sum_of_kernel_ms_per_step'])")
  echo "rep$rep [$cfg] img/s ms/step sum_kernel_ms: $v"
done
done
