#!/bin/bash
out=gpurun_out/r02b; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "score_head or pw_chain" > $out/t_k.log 2>&1; echo "kernel tests rc=$?" | tee -a $out/summary.txt
timeout 900 python -m pytest tests/test_gpu_e2e.py -q > $out/t_e2e.log 2>&1; echo "e2e tests rc=$?" | tee -a $out/summary.txt
FX_SCORE_HEAD=0 timeout 900 python -m pytest tests/test_gpu_e2e.py -q -k free_running > $out/t_e2e_nosh.log 2>&1; echo "e2e (old score path) rc=$?" | tee -a $out/summary.txt
for sh in 0 1; do
  FX_PW_CHAIN_MAX_STAGE=1 FX_SCORE_HEAD=$sh timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --per-op $out/perop_sh$sh.txt > $out/bench_sh$sh.json 2> $out/bench_sh$sh.err; echo "bench sh=$sh rc=$?" | tee -a $out/summary.txt
  python -c "
import json;j=json.loads(open('$out/bench_sh$sh.json').read().strip().splitlines()[-1]);print(j['value'],j['ms_per_step'])" | tee -a $out/summary.txt
done
tail -15 $out/t_k.log; grep -E "^E  |passed|failed|Error" $out/t_e2e.log | head -40; grep -E "^E  |passed|failed" $out/t_e2e_nosh.log | head
