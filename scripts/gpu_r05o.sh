#!/bin/bash
# register-resident top-k: exactness tests (every form), A/B of the headline inside one call, per-op rows of the two top-k launches
out=$PWD/gpurun_out/r05o; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train_ops.py -q -k "topk or adamw" > $out/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $out/tests.txt
for r in 1 2; do for v in 0 1; do
  FX_TOPK_REGS=$v timeout 300 python bench.py --no-cpu-baseline --no-other-configs --per-op $out/per_op_regs$v.txt > $out/bench_regs${v}_$r.json 2> $out/bench_regs${v}_$r.err
  python - <<PY
import json; j=json.loads(open("$out/bench_regs${v}_$r.json").read().strip().splitlines()[-1]); print("FX_TOPK_REGS=$v run $r:", j["value"], "img/s", j["ms_per_step"], "ms")
PY
done; done
for v in 0 1; do grep -E "topk" $out/per_op_regs$v.txt | head -2; done
timeout 300 python -m pytest tests/test_gpu_e2e.py -q > $out/e2e.txt 2>&1; echo "e2e rc=$?"; tail -2 $out/e2e.txt
