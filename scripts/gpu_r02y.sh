#!/bin/bash
# fused box loss / two-pass max-pool backward / training convs on the flat kernels: parity tests, launch census, training bench
mkdir -p gpurun_out/r02y
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_criterion.py tests/test_gpu_train_conv.py tests/test_gpu_train_detr.py 2>&1 > gpurun_out/r02y/tests_full.txt 2>&1; grep -E "passed|failed|Error|error|assert" gpurun_out/r02y/tests_full.txt | head -30 > gpurun_out/r02y/tests.txt
cat gpurun_out/r02y/tests.txt
timeout 300 python scripts/dev/train_eager_ops.py 2>&1 | grep -E "3 steps|aten ops" 
timeout 400 python bench.py --train --no-cpu-baseline > gpurun_out/r02y/train_bench.json 2> gpurun_out/r02y/train_bench.err
cut -c1-400 gpurun_out/r02y/train_bench.json
