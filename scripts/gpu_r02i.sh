#!/bin/bash
out=gpurun_out/r02i; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "row_chain" > $out/t_rc.log 2>&1; echo "rc test rc=$?" | tee -a $out/summary.txt
grep -E "^E  |passed|failed" $out/t_rc.log | head -20
timeout 900 python -m pytest tests/test_gpu_e2e.py -q > $out/t_e2e.log 2>&1; echo "e2e rc=$?" | tee -a $out/summary.txt
grep -E "^E  |passed|failed" $out/t_e2e.log | head -20
run() { name=$1; shift; env "$@" timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --per-op $out/perop_$name.txt > $out/bench_$name.json 2> $out/bench_$name.err; python -c "
import json;j=json.loads(open('$out/bench_$name.json').read().strip().splitlines()[-1]);print('$name',j['value'],j['ms_per_step'])" | tee -a $out/summary.txt; }
run rc0 FX_ROW_CHAIN=0
run rc1 FX_ROW_CHAIN=1
grep "row_chain\|fx_mha\|msda" $out/perop_rc1.txt | cut -c1-120 | head -30
