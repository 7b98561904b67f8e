#!/bin/bash
# round 5, call J: res2's 64 -> 64 3x3 layers on the LDS-resident-filter kernel (conv3x3_c64.hip): parity + A/B (FX_C3_C64=0/1)
TAG=r05j
out=$PWD/gpurun_out/$TAG; mkdir -p $out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -k "conv3x3_flat_halo_kernel" > $out/c64_tests.txt 2>&1; echo "flat-case kernel tests rc=$?"; tail -4 $out/c64_tests.txt | cut -c1-500
for i in 1 2; do
  for on in 0 1; do
    FX_C3_C64=$on timeout 200 python bench.py --no-cpu-baseline --no-other-configs --steps 30 --warmup 5 --per-op $out/per_op_c64_${on}.txt > $out/bench_c64_${on}_$i.json 2> $out/bench_c64_${on}_$i.err
    python - <<PY
import json
try:
    j = json.loads(open("$out/bench_c64_${on}_$i.json").read().strip().splitlines()[-1])
    v = j["roofline"]["all_conv_variants"]
    print("FX_C3_C64=$on run $i:", j["value"], "img/s", j["ms_per_step"], "ms; conv3x3_c64<64>", v.get("conv3x3_c64<64>", {}).get("ms"), "conv3x3_kplane<64>", v.get("conv3x3_kplane<64>", {}).get("ms"))
except Exception as e:
    print("FX_C3_C64=$on run $i: failed", e); print(open("$out/bench_c64_${on}_$i.err").read()[-800:])
PY
  done
done
grep -E "c64|kplane<64>" $out/per_op_c64_1.txt | head -4
timeout 400 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_train_conv.py -q -x -k "config1 or resnet_vd_backward" > $out/parity.txt 2>&1; echo "config1 + resnet backward parity rc=$?"; tail -3 $out/parity.txt | cut -c1-300
