#!/bin/bash
# round 5, call A: lean row-chain programs (two workgroups per CU) - kernel parity, A/B on the headline; the new parity tests
# (training step at the BASELINE shapes, adapter deepcopy / pickle); smoke with the tightened gates
TAG=r05a
out=$PWD/gpurun_out/$TAG; mkdir -p $out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "row_chain" > $out/rc_tests.txt 2>&1; echo "row_chain kernel tests rc=$?"; tail -3 $out/rc_tests.txt
for i in 1 2; do
  for lean in 0 1; do
    FX_RC_LEAN=$lean timeout 200 python bench.py --no-cpu-baseline --no-other-configs --steps 30 --warmup 5 > $out/bench_lean${lean}_$i.json 2> $out/bench_lean${lean}_$i.err
    python - <<PY
import json
try:
    j = json.loads(open("$out/bench_lean${lean}_$i.json").read().strip().splitlines()[-1])
    rc = j["roofline"]["all_conv_variants"].get("row_chain", {})
    print("FX_RC_LEAN=$lean run $i:", j["value"], "img/s", j["ms_per_step"], "ms; row_chain serial", rc.get("ms"), "ms")
except Exception as e:
    print("FX_RC_LEAN=$lean run $i: failed", e)
PY
  done
done
timeout 400 python -m pytest tests/test_gpu_baseline_configs.py -q -x -s -k config1 > $out/config1_parity.txt 2>&1; echo "config1 parity (lean) rc=$?"; tail -4 $out/config1_parity.txt
timeout 300 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_detr_variants.py -q -x > $out/e2e.txt 2>&1; echo "e2e rc=$?"; tail -3 $out/e2e.txt
timeout 400 python -m pytest tests/test_gpu_train_api.py -q -x -k "deepcopy" > $out/adapter.txt 2>&1; echo "adapter deepcopy rc=$?"; tail -5 $out/adapter.txt
( time timeout 900 python -m pytest tests/test_gpu_train_baseline_configs.py -q -s > $out/train_baseline_parity.txt 2>&1 ); echo "train baseline parity rc=$?"; grep -v Warning $out/train_baseline_parity.txt | tail -30 | cut -c1-1500
timeout 400 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; echo "smoke rc=$?"; grep smoke $out/smoke.txt
