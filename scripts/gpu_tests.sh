#!/bin/bash
# full GPU test suite (no -x: one call shows every failure)
out=$PWD/gpurun_out/${1:-tests}; mkdir -p $out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --durations=10 ${@:2} ) > $out/t_gpu.log 2>&1; echo "gpu tests rc=$?" | tee $out/summary.txt
grep -E "^(FAILED|ERROR)|passed|failed" $out/t_gpu.log | head -40
