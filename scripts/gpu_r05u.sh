#!/bin/bash
# MSDA with all 48 taps of a query in flight (one memory phase instead of three): tests, per-op rows, headline
out=$PWD/gpurun_out/r05u; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train_ops.py -q -k "msda" > $out/tests.txt 2>&1; echo "tests rc=$?"; tail -2 $out/tests.txt
for r in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-other-configs --per-op $out/per_op.txt > $out/bench_$r.json 2> $out/bench_$r.err
  python - <<PY
import json; j=json.loads(open("$out/bench_$r.json").read().strip().splitlines()[-1]); print("run $r:", j["value"], "img/s", j["ms_per_step"], "ms")
PY
done
grep -E "fx_msda|fx_mha|post_attn" $out/per_op.txt | head -9
timeout 300 python -m pytest tests/test_gpu_e2e.py -q -x > $out/e2e.txt 2>&1; echo "e2e rc=$?"; tail -2 $out/e2e.txt
