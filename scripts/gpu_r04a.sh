#!/bin/bash
# round 4, first GPU call: the new BASELINE-config parity tests, the touched kernel tests, smoke on two seeds, a short headline bench
out=$PWD/gpurun_out/r04a; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_baseline_configs.py -q -s -x > $out/baseline_tests.txt 2>&1; echo "baseline tests rc=$?"
tail -25 $out/baseline_tests.txt | cut -c1-1500
timeout 600 python -m pytest tests/test_gpu_train_ops.py -q -x > $out/train_ops.txt 2>&1; echo "train_ops rc=$?"; tail -3 $out/train_ops.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; echo "smoke rc=$?"; grep smoke $out/smoke.txt; tail -3 $out/smoke.txt | cut -c1-600
timeout 300 python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; head -c 400 $out/bench.json
