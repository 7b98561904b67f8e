#!/bin/bash
# RT-DETR: res2's block output not stored by the fused seam launch (FX_SKIP_UNUSED_RES); kernel test, e2e parity, A/B inside one call
out=$PWD/gpurun_out/r05q; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "pw_chain" > $out/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $out/tests.txt
timeout 600 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_odd_sizes.py tests/test_gpu_detr_variants.py -q > $out/e2e.txt 2>&1; echo "e2e rc=$?"; tail -2 $out/e2e.txt
for r in 1 2; do for v in 0 1; do
  FX_SKIP_UNUSED_RES=$v timeout 300 python bench.py --no-cpu-baseline --no-other-configs --per-op $out/per_op_skip$v.txt > $out/bench_skip${v}_$r.json 2> $out/bench_skip${v}_$r.err
  python - <<PY
import json; j=json.loads(open("$out/bench_skip${v}_$r.json").read().strip().splitlines()[-1]); print("FX_SKIP_UNUSED_RES=$v run $r:", j["value"], "img/s", j["ms_per_step"], "ms")
PY
done; done
for v in 0 1; do grep -E "pool:" $out/per_op_skip$v.txt | head -2; done
