#!/bin/bash
# quad-tile pw_chain with the pixel map in an LDS table: kernel tests, engine parity, per-op rows, headline
out=$PWD/gpurun_out/r05s; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "pw_chain" > $out/tests.txt 2>&1; echo "tests rc=$?"; tail -2 $out/tests.txt
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_mf.py tests/test_gpu_backbone_sizes.py -q -x > $out/e2e.txt 2>&1; echo "engines rc=$?"; tail -2 $out/e2e.txt
for r in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-other-configs --per-op $out/per_op.txt > $out/bench_$r.json 2> $out/bench_$r.err
  python - <<PY
import json; j=json.loads(open("$out/bench_$r.json").read().strip().splitlines()[-1]); print("run $r:", j["value"], "img/s", j["ms_per_step"], "ms")
PY
done
grep -E "pool" $out/per_op.txt | head -3
timeout 300 python bench.py --model fai-mf-l-coco-ins --no-cpu-baseline --per-op $out/mf_per_op.txt > $out/mf.json 2> $out/mf.err; head -c 160 $out/mf.json; echo
grep -E "pool" $out/mf_per_op.txt | head -3
