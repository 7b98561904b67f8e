#!/bin/bash
# round 5, call D: single-pass form of the fused conv1_3 + max-pool kernel (both weight blocks resident, B fragments shared): parity + A/B
TAG=r05e
out=$PWD/gpurun_out/$TAG; mkdir -p $out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -k "stem_conv_relu_maxpool" > $out/stem_tests.txt 2>&1; echo "fused stem kernel tests rc=$?"; tail -4 $out/stem_tests.txt | cut -c1-400
for i in 1 2; do
  for fuse in 0 1; do
    FX_STEM_FUSE=$fuse timeout 200 python bench.py --no-cpu-baseline --no-other-configs --steps 30 --warmup 5 --per-op $out/per_op_fuse${fuse}.txt > $out/bench_fuse${fuse}_$i.json 2> $out/bench_fuse${fuse}_$i.err
    python - <<PY
import json
try:
    j = json.loads(open("$out/bench_fuse${fuse}_$i.json").read().strip().splitlines()[-1])
    v = j["roofline"]["all_conv_variants"]
    print("FX_STEM_FUSE=$fuse run $i:", j["value"], "img/s", j["ms_per_step"], "ms; stem_c3+pool", v.get("stem_c3+pool", {}).get("ms"), "conv3x3_c32<64>", v.get("conv3x3_c32<64>", {}).get("ms"))
except Exception as e:
    print("FX_STEM_FUSE=$fuse run $i: failed", e)
PY
  done
done
grep -E "maxpool|conv1_|stem" $out/per_op_fuse1.txt | head -4
timeout 300 python bench.py --model fai-mf-l-coco-ins --no-cpu-baseline --steps 10 --warmup 3 > $out/mf.json 2> $out/mf.err; head -c 250 $out/mf.json; echo
