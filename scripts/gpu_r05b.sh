#!/bin/bash
# round 5, call B: the fp16 element type (kernel parity, training steps vs the oracle, loss scale), config-3 training parity with the
# sensitivity gate, config 5 in fp16 vs bf16, two cheap schedule A/Bs of the headline
TAG=r05b
out=$PWD/gpurun_out/$TAG; mkdir -p $out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_fp16.py -q -x -s > $out/fp16_tests.txt 2>&1 ); echo "fp16 tests rc=$?"; grep -v Warning $out/fp16_tests.txt | grep -E "fp16|passed|failed|Error|assert" | cut -c1-600 | tail -25
timeout 900 python -m pytest tests/test_gpu_train_baseline_configs.py -q -s -k config3 > $out/train_c3_parity.txt 2>&1; echo "config3 train parity rc=$?"; grep -v Warning $out/train_c3_parity.txt | grep -E "oracle forward|21 losses|configs\[|sensitivity|passed|failed|assert" | cut -c1-900
for dt in fp16 bf16; do
  timeout 400 python bench.py --train --model bisenetformer-l-ade --norm BN --dtype $dt --steps 10 --warmup 4 --no-cpu-baseline > $out/bf_train_bn_$dt.json 2> $out/bf_train_bn_$dt.err; echo "bf train BN $dt rc=$?"
  python - <<PY
import json
try:
    j = json.loads(open("$out/bf_train_bn_$dt.json").read().strip().splitlines()[-1])
    print("  ", j["dtype"], j["value"], "img/s", j["ms_per_step"], "ms", "loss", j.get("final_total_loss"), "scale", j.get("loss_scale"))
except Exception as e:
    print("   failed", e); print(open("$out/bf_train_bn_$dt.err").read()[-1500:])
PY
done
timeout 400 python bench.py --train --model bisenetformer-l-ade --norm FrozenBN --dtype fp16 --steps 10 --warmup 4 --no-cpu-baseline > $out/bf_train_frozen_fp16.json 2> $out/bf_train_frozen_fp16.err; echo "bf train FrozenBN fp16 rc=$?"; head -c 300 $out/bf_train_frozen_fp16.json; echo
timeout 300 python bench.py --train --dtype fp16 --steps 8 --warmup 4 --no-cpu-baseline > $out/detr_train_fp16.json 2> $out/detr_train_fp16.err; echo "detr train fp16 rc=$?"; head -c 300 $out/detr_train_fp16.json; echo
for v in "FX_STREAMS=2" "FX_STREAMS=4" "FX_MULTI_MODE=branches" "FX_STREAMS=2"; do
  env $v timeout 200 python bench.py --no-cpu-baseline --no-other-configs --steps 30 --warmup 5 > $out/ab.json 2> $out/ab.err
  python - <<PY
import json
try:
    j = json.loads(open("$out/ab.json").read().strip().splitlines()[-1]); print("$v:", j["value"], "img/s", j["ms_per_step"], "ms")
except Exception as e:
    print("$v: failed", e)
PY
done
