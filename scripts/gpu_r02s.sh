#!/bin/bash
# round-2 official numbers: default bench line (with other_configs), per-op table, rocprofv3 kernel stats, calibrated PMC HBM traffic
TAG=r02s
out=$PWD/gpurun_out/$TAG; mkdir -p $out
ROOT=$PWD
export TMPDIR=/tmp
( time timeout 600 python bench.py --per-op $out/${TAG}_per_op_hipevent.txt > $out/${TAG}_bench.json 2> $out/${TAG}_bench.err ); echo "bench rc=$?"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o $TAG -- python $ROOT/bench.py --no-cpu-baseline --no-other-configs --steps 10 --warmup 3 > $out/prof.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -o f -- python $ROOT/bench.py --no-cpu-baseline --no-other-configs --steps 3 --warmup 2 > $out/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -o w -- python $ROOT/bench.py --no-cpu-baseline --no-other-configs --steps 3 --warmup 2 > $out/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/cal_fetch -o f -- python $ROOT/scripts/pmc_calibrate.py > $out/cal_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/cal_write -o w -- python $ROOT/scripts/pmc_calibrate.py > $out/cal_write.log 2>&1
cd $ROOT
F=$(find $out/pmc_fetch -name '*counter_collection.csv' | head -1); W=$(find $out/pmc_write -name '*counter_collection.csv' | head -1)
CF=$(find $out/cal_fetch -name '*counter_collection.csv' | head -1); CW=$(find $out/cal_write -name '*counter_collection.csv' | head -1)
python scripts/pmc_summary.py $F $W $out/${TAG}_pmc_hbm.md $out/${TAG}_pmc_hbm.json $CF $CW | head -12
find $out -name '*kernel_trace.csv' -delete; find $out -name '*counter_collection.csv' -size +20M -delete
head -c 400 $out/${TAG}_bench.json; echo; tail -3 $out/${TAG}_bench.err | cut -c1-300
python -c "
import json; j=json.loads(open('$out/${TAG}_bench.json').read().strip().splitlines()[-1])
for k,v in j.get('other_configs',{}).items(): print(k, {a:v.get(a) for a in ('value','ms_per_step','error')} if isinstance(v,dict) else v)"
