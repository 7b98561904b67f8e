#!/bin/bash
# training-step changes: parity tests, launch census, training bench (with / without the weight-gradient side stream)
mkdir -p gpurun_out/r02y
timeout 1200 python -m pytest -x -q -m gpu tests/test_gpu_train_ops.py tests/test_gpu_train_conv.py tests/test_gpu_train_detr.py tests/test_gpu_train_api.py tests/test_gpu_train_mf.py tests/test_gpu_train_bf.py > gpurun_out/r02y/tests_full.txt 2>&1; grep -E "passed|failed|Error|error|assert" gpurun_out/r02y/tests_full.txt | head -30 > gpurun_out/r02y/tests.txt
cat gpurun_out/r02y/tests.txt
timeout 300 python scripts/dev/train_eager_ops.py 2>&1 | grep -E "3 steps|aten ops|conv weight gradients" 
timeout 400 python bench.py --train --no-cpu-baseline > gpurun_out/r02y/train_bench.json 2> gpurun_out/r02y/train_bench.err
cut -c1-200 gpurun_out/r02y/train_bench.json
FX_WGRAD_STREAM=0 timeout 400 python bench.py --train --no-cpu-baseline 2>/dev/null | cut -c1-200
