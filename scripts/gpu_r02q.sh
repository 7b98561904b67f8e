#!/bin/bash
out=$PWD/gpurun_out/r02q; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py tests/test_gpu_train_api.py -q -k "topk or e2e or postprocess or eval" > $out/t.log 2>&1; grep -E "^E  |passed|failed|Error" $out/t.log | cut -c1-400 | tail -10
timeout 300 python bench.py --no-cpu-baseline --no-other-configs --steps 30 --warmup 5 --per-op $out/per_op.txt 2>/dev/null | head -c 180; echo; grep -E "topk" $out/per_op.txt
