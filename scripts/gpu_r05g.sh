#!/bin/bash
# round 5, call F: persistent loader-wave form of the fused conv1_3 + max-pool kernel: parity + A/B (FX_STEM_POOL_8WAVE=0/1)
TAG=r05g
out=$PWD/gpurun_out/$TAG; mkdir -p $out
export TMPDIR=/tmp
for ps in 1 0; do FX_STEM_POOL_8WAVE=$ps timeout 300 python -m pytest tests/test_gpu_kernels.py -q -k "stem_conv_relu_maxpool" > $out/stem_tests_$ps.txt 2>&1; echo "fused stem kernel tests (persist=$ps) rc=$?"; tail -3 $out/stem_tests_$ps.txt | cut -c1-300; done
for i in 1 2; do
  for ps in 0 1; do
    FX_STEM_POOL_8WAVE=$ps timeout 200 python bench.py --no-cpu-baseline --no-other-configs --steps 30 --warmup 5 --per-op $out/per_op_ps${ps}.txt > $out/bench_ps${ps}_$i.json 2> $out/bench_ps${ps}_$i.err
    python - <<PY
import json
try:
    j = json.loads(open("$out/bench_ps${ps}_$i.json").read().strip().splitlines()[-1])
    v = j["roofline"]["all_conv_variants"]
    print("FX_STEM_POOL_8WAVE=$ps run $i:", j["value"], "img/s", j["ms_per_step"], "ms; stem_c3+pool", v.get("stem_c3+pool", {}).get("ms"))
except Exception as e:
    print("FX_STEM_POOL_8WAVE=$ps run $i: failed", e)
PY
  done
done
grep -E "maxpool|stem" $out/per_op_ps1.txt | head -4
