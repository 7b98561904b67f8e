#!/bin/bash
# kernel trace of a few two-part steps (start / end timestamps per hardware queue) -> gpurun_out/<tag>/trace.csv
TAG=${1:-r04t}; shift
out=$PWD/gpurun_out/$TAG; mkdir -p $out
ROOT=$PWD
export TMPDIR=/tmp
cd /tmp
env "$@" timeout 400 rocprofv3 --kernel-trace --output-format csv -d $out/prof -o t -- python $ROOT/bench.py --no-cpu-baseline --no-other-configs --steps 8 --warmup 3 > $out/prof.log 2>&1
cd $ROOT
f=$(find $out/prof -name '*kernel_trace.csv' | head -1); cp $f $out/trace.csv; rm -rf $out/prof
python scripts/dev/trace_timeline.py $out/trace.csv > $out/timeline.txt 2>&1; head -12 $out/timeline.txt
