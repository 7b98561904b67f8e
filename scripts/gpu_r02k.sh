#!/bin/bash
out=gpurun_out/r02k; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_train_api.py tests/test_gpu_two_streams.py -q > $out/t_api.log 2>&1; echo "api tests rc=$?" | tee -a $out/summary.txt
grep -E "^E  |passed|failed|xfail|Error" $out/t_api.log | head -30
