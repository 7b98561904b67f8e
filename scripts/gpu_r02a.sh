#!/bin/bash
# round-2 GPU batch A: pw_chain kernel parity, e2e parity, bench A/B
out=gpurun_out/r02a; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "pw_chain" > $out/t_pw.log 2>&1; echo "pw tests rc=$?" | tee -a $out/summary.txt
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_mf.py -q -x > $out/t_e2e.log 2>&1; echo "e2e tests rc=$?" | tee -a $out/summary.txt
for cfg in "0 2" "1 1" "1 2"; do set -- $cfg
  FX_PW_CHAIN=$1 FX_PW_CHAIN_MAX_STAGE=$2 timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --per-op $out/perop_c$1_s$2.txt > $out/bench_c$1_s$2.json 2> $out/bench_c$1_s$2.err; echo "bench chain=$1 stage=$2 rc=$?" | tee -a $out/summary.txt
  tail -c 600 $out/bench_c$1_s$2.json | head -c 300 >> $out/summary.txt; echo >> $out/summary.txt
done
tail -5 $out/t_pw.log; tail -5 $out/t_e2e.log; cat $out/summary.txt | cut -c1-400
