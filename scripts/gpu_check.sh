#!/bin/bash
# quick GPU check of the training path: parity tests, launch census, training bench
mkdir -p gpurun_out/chk
timeout 1200 python -m pytest -x -q -m gpu tests/test_gpu_train_ops.py tests/test_gpu_train_conv.py tests/test_gpu_train_detr.py tests/test_gpu_train_api.py > gpurun_out/chk/tests_full.txt 2>&1; grep -E "passed|failed|Error|error|^E " gpurun_out/chk/tests_full.txt | head -30
timeout 300 python scripts/dev/train_eager_ops.py 2>&1 | grep -E "3 steps|aten ops"
timeout 400 python bench.py --train --no-cpu-baseline 2>/dev/null | cut -c1-200
