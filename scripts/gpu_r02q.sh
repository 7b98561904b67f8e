#!/bin/bash
out=$PWD/gpurun_out/r02q; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_train_mf.py tests/test_gpu_train_api.py -q > $out/t_mf.log 2>&1; grep -E "^E  |passed|failed|Error" $out/t_mf.log | cut -c1-600 | tail -12
