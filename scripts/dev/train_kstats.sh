#!/bin/bash
# rocprofv3 kernel stats of a few training steps -> gpurun_out/<tag>_train_kernel_stats.csv (top rows printed).  usage: [FX_ENV="A=1 B=2"] train_kstats.sh <tag> [bench args]
TAG=${1:-r04w}; shift
ROOT=$PWD; out=$ROOT/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
env $FX_ENV timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$TAG -o t -- python $ROOT/bench.py --train --no-cpu-baseline --steps 6 --warmup 3 "$@" > $out/prof_$TAG.log 2>&1
cd $ROOT
f=$(find $out/prof_$TAG -name '*kernel_stats.csv' | head -1); cp $f $out/${TAG}_train_kernel_stats.csv; rm -rf $out/prof_$TAG
python - <<PY
import csv
rows = list(csv.DictReader(open("$out/${TAG}_train_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms over the profiled run")
for r in rows[:40]:
    print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms {int(r["Calls"]):6d} calls {float(r["AverageNs"])/1e3:8.1f} us  {float(r["Percentage"]):5.1f} %  {r["Name"][:110]}')
PY
