#!/bin/bash
# ONE parametrised recipe for every `gpurun` call (replaces the per-experiment scripts/gpu_r0*.sh of rounds 2-5; git history keeps them).
#   gpurun --timeout T -- 'bash scripts/gpu.sh <tag> <step> [<step> ...]'
# Every step writes under gpurun_out/<tag>/ and prints a one-line summary; A/B steps run inside this ONE call (boxes differ by ~3 %).
# Steps:
#   tests:<pytest args>          python -m pytest <args> -q            (e.g. tests:'tests/test_gpu_kernels.py -k topk')
#   suite                        the whole -m gpu directory, no -x
#   probe:<bin>[:args]           scripts/probes/bin/<bin> args          (stand-alone HIP probes)
#   bench[:ENV=V,ENV=V]          headline line (no CPU baseline / other configs), twice, with the per-op table of the first run
#   ab:<cfg>|<cfg>|...           headline A/B, each cfg = "-" or comma-separated ENV=V; two interleaved repetitions
#   benchfull                    the default `python bench.py` line exactly as the driver runs it
#   train[:<bench.py args>]      python bench.py --train <args>
#   model:<name>[:args]          python bench.py --model <name> <args>
#   prof[:ENV=V,...]             rocprofv3 --kernel-trace --stats of the headline (as timed) and with FX_PARTS_SERIAL=1 -> kernel_stats csv
#   pmc                          HBM FETCH_SIZE / WRITE_SIZE passes + calibration -> pmc_hbm.{json,md}
#   py:<script>[:args]           python <script> args
tag=$1; shift
out=$PWD/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
envs() { echo "$1" | tr ',' ' '; }
line() { python - "$1" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = j.get("roofline", {})
    print(f"{j['value']} {j['unit']}  {j['ms_per_step']} ms/step  kernel-sum {r.get('sum_of_kernel_ms_per_step')}  dominant {r.get('kernel')} frac {r.get('frac')}")
except Exception as e:
    print("no bench line:", e)
PY
}
for step in "$@"; do
  kind=${step%%:*}; arg=""; [ "$kind" != "$step" ] && arg=${step#*:}
  case $kind in
    tests)
      n=$(ls $out | grep -c '^tests_'); f=$out/tests_$n.txt
      timeout 1500 python -m pytest $arg -q > $f 2>&1; echo "[tests $arg] rc=$? $(tail -1 $f)";;
    suite)
      timeout 2400 python -m pytest tests -m gpu -q > $out/suite.txt 2>&1; echo "[suite] rc=$? $(tail -1 $out/suite.txt)";;
    probe)
      bin=${arg%%:*}; pa=""; [ "$bin" != "$arg" ] && pa=${arg#*:}
      timeout 600 scripts/probes/bin/$bin $pa > $out/probe_$bin.txt 2>&1; echo "[probe $bin] rc=$?";;
    bench)
      for r in 1 2; do
        env $(envs "$arg") timeout 400 python bench.py --no-cpu-baseline --no-other-configs $([ $r = 1 ] && echo --per-op $out/per_op.txt) > $out/bench_$r.json 2> $out/bench_$r.err
        echo "[bench $arg run $r] $(line $out/bench_$r.json)"
      done;;
    ab)
      IFS='|' read -ra cfgs <<< "$arg"
      for rep in 1 2; do for cfg in "${cfgs[@]}"; do
        e=""; [ "$cfg" != "-" ] && e=$(envs "$cfg")
        f=$out/ab_$(echo "$cfg" | tr -c 'A-Za-z0-9_=\n' '_')_$rep.json
        env $e timeout 400 python bench.py --no-cpu-baseline --no-other-configs --steps 30 --warmup 8 > $f 2>/dev/null
        echo "[ab rep$rep $cfg] $(line $f)"
      done; done;;
    benchfull)
      timeout 1500 python bench.py > $out/bench_full.json 2> $out/bench_full.err; echo "[benchfull] rc=$? $(line $out/bench_full.json)";;
    train)
      n=$(ls $out | grep -c '^train_'); f=$out/train_$n.json
      timeout 900 python bench.py --train $arg > $f 2> $f.err; echo "[train $arg] rc=$? $(line $f)";;
    model)
      name=${arg%%:*}; ma=""; [ "$name" != "$arg" ] && ma=${arg#*:}
      timeout 900 python bench.py --model $name $ma --no-cpu-baseline --no-other-configs > $out/model_$name.json 2> $out/model_$name.err; echo "[model $name $ma] rc=$? $(line $out/model_$name.json)";;
    prof)
      for mode in timed serial; do
        e=$(envs "$arg"); [ $mode = serial ] && e="$e FX_PARTS_SERIAL=1"
        rm -rf /tmp/prof_$mode
        (cd /tmp && env $e timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$mode -o p -- python $OLDPWD/bench.py --no-cpu-baseline --no-other-configs --steps 10 --warmup 5 > $out/prof_$mode.json 2> $out/prof_$mode.err)
        cp $(find /tmp/prof_$mode -name '*kernel_stats.csv' | head -1) $out/${mode}_kernel_stats.csv 2>/dev/null
        echo "[prof $mode] $(line $out/prof_$mode.json) $(head -3 $out/${mode}_kernel_stats.csv | tail -2 | cut -c1-150 | tr '\n' ' ')"
      done;;
    pmc)
      # separate passes per counter, kernel-trace only (MI355X_MICROARCH.md: never combined with other trace domains); calibration kernels of known byte counts
      ROOT=$PWD
      (cd /tmp
       for c in FETCH_SIZE WRITE_SIZE; do
         timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -o p -- python $ROOT/bench.py --no-cpu-baseline --no-other-configs --steps 3 --warmup 2 > $out/pmc_$c.log 2>&1
         timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/cal_$c -o c -- python $ROOT/scripts/pmc_calibrate.py > $out/cal_$c.log 2>&1
       done)
      F=$(find $out/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1); W=$(find $out/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)
      CF=$(find $out/cal_FETCH_SIZE -name '*counter_collection.csv' | head -1); CW=$(find $out/cal_WRITE_SIZE -name '*counter_collection.csv' | head -1)
      python scripts/pmc_summary.py $F $W $out/pmc_hbm.md $out/pmc_hbm.json $CF $CW | head -6
      find $out -name '*kernel_trace.csv' -delete; find $out -name '*counter_collection.csv' -size +20M -delete;;
    py)
      sc=${arg%%:*}; pa=""; [ "$sc" != "$arg" ] && pa=$(echo ${arg#*:} | tr ',' ' ')
      n=$(ls $out | grep -c '^py_'); timeout 1200 python $sc $pa > $out/py_$n.txt 2>&1; echo "[py $sc] rc=$? $(tail -3 $out/py_$n.txt | tr '\n' ' ' | cut -c1-400)";;
    *) echo "unknown step $step";;
  esac
done
