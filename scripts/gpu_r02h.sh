#!/bin/bash
out=gpurun_out/r02h; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "flat or conv_igemm" > $out/t_flat.log 2>&1; echo "flat tests rc=$?" | tee -a $out/summary.txt
tail -12 $out/t_flat.log
run() { name=$1; shift; env "$@" timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --per-op $out/perop_$name.txt > $out/bench_$name.json 2> $out/bench_$name.err; python -c "
import json;j=json.loads(open('$out/bench_$name.json').read().strip().splitlines()[-1]);print('$name',j['value'],j['ms_per_step'])" | tee -a $out/summary.txt; }
run pw0 FX_PW_NO_LOADER=0
run pw1 FX_PW_FLAT=1
grep "pw_flat" $out/perop_pw1.txt | cut -c1-160
