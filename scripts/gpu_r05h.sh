#!/bin/bash
# round 5, call H: fused normalise + conv1_1 + conv1_2 kernel (stem12.hip): parity + A/B on the headline
TAG=r05i
out=$PWD/gpurun_out/$TAG; mkdir -p $out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -k "stem_conv1_1_conv1_2_fused" > $out/stem12_tests.txt 2>&1; echo "stem12 kernel tests rc=$?"; tail -6 $out/stem12_tests.txt | cut -c1-500
for i in 1 2; do
  for fuse in 0 1; do
    FX_STEM12_FUSE=$fuse timeout 200 python bench.py --no-cpu-baseline --no-other-configs --steps 30 --warmup 5 --per-op $out/per_op_fuse${fuse}.txt > $out/bench_fuse${fuse}_$i.json 2> $out/bench_fuse${fuse}_$i.err
    python - <<PY
import json
try:
    j = json.loads(open("$out/bench_fuse${fuse}_$i.json").read().strip().splitlines()[-1])
    v = j["roofline"]["all_conv_variants"]
    print("FX_STEM12_FUSE=$fuse run $i:", j["value"], "img/s", j["ms_per_step"], "ms; stem_c1+c2", v.get("stem_c1+c2", {}).get("ms"), "conv3x3_c32<32>", v.get("conv3x3_c32<32>", {}).get("ms"))
except Exception as e:
    print("FX_STEM12_FUSE=$fuse run $i: failed", e); print(open("$out/bench_fuse${fuse}_$i.err").read()[-800:])
PY
  done
done
grep -E "maxpool|conv1_|stem" $out/per_op_fuse1.txt | head -4
timeout 400 python -m pytest tests/test_gpu_baseline_configs.py -q -x -s -k "config1" > $out/config1.txt 2>&1; echo "config1 parity rc=$?"; grep -E "worst over|passed|failed" $out/config1.txt | cut -c1-500
