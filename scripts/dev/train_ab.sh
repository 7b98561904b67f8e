#!/bin/bash
# A/B of the RT-DETR training bench inside ONE gpurun call: each argument an environment assignment string ("-" = default).
# usage: [BENCH_ARGS="--model ..."] bash scripts/dev/train_ab.sh - "FX_ENC_SELECT_ROWS=0"
export TMPDIR=/tmp
for rep in $(seq 1 ${REPS:-2}); do
for cfg in "$@"; do
  if [ "$cfg" = "-" ]; then e=""; else e="$cfg"; fi
  v=$(env $e timeout 300 python bench.py --train --no-cpu-baseline --steps ${STEPS:-20} --warmup ${WARMUP:-5} $BENCH_ARGS 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j.get('final_total_loss'))")
  echo "rep$rep [$cfg] img/s ms/step loss: $v"
done
done
