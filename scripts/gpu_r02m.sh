#!/bin/bash
# round-2 checkpoint: full GPU test suite, bench line + per-op table, rocprofv3 stats, training bench
out=$PWD/gpurun_out/r02m; mkdir -p $out
ROOT=$PWD
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -q -x --durations=15 ) > $out/t_gpu.log 2>&1; echo "gpu tests rc=$?" | tee -a $out/summary.txt
timeout 600 python bench.py --per-op $out/per_op_hipevent.txt > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" | tee -a $out/summary.txt
timeout 300 python bench.py --train --steps 10 --warmup 3 > $out/train_bench.json 2> $out/train_bench.err; echo "train bench rc=$?" | tee -a $out/summary.txt
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o r02m -- python $ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $out/prof.log 2>&1
cd $ROOT
find $out/prof -name '*kernel_trace.csv' -delete
tail -4 $out/t_gpu.log; head -c 300 $out/bench.json; echo; head -c 300 $out/train_bench.json; echo
