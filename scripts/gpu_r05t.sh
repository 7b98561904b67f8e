#!/bin/bash
# MHA: K/V staging loads batched in front of the LDS writes - kernel tests, per-op rows, headline; MF / BF lines
out=$PWD/gpurun_out/r05t; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train_ops.py tests/test_gpu_mf.py -q -k "mha or attention or attn" > $out/tests.txt 2>&1; echo "tests rc=$?"; tail -2 $out/tests.txt
for r in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-other-configs --per-op $out/per_op.txt > $out/bench_$r.json 2> $out/bench_$r.err
  python - <<PY
import json; j=json.loads(open("$out/bench_$r.json").read().strip().splitlines()[-1]); print("run $r:", j["value"], "img/s", j["ms_per_step"], "ms")
PY
done
grep -E "fx_mha" $out/per_op.txt | head -7
timeout 300 python bench.py --model fai-mf-l-coco-ins --no-cpu-baseline --per-op $out/mf_per_op.txt > $out/mf.json 2> $out/mf.err; head -c 160 $out/mf.json; echo
grep -E "fx_mha" $out/mf_per_op.txt | head -4
timeout 300 python bench.py --model bisenetformer-l-ade --no-cpu-baseline --per-op $out/bf_per_op.txt > $out/bf.json 2> $out/bf.err; head -c 160 $out/bf.json; echo
grep -E "fx_mha" $out/bf_per_op.txt | head -4
timeout 600 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_bf.py -q -x > $out/e2e.txt 2>&1; echo "engines rc=$?"; tail -2 $out/e2e.txt
