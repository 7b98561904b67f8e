#!/bin/bash
out=$PWD/gpurun_out/r02q; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_train_api.py -q > $out/t.log 2>&1; grep -E "^E  |passed|failed|Error" $out/t.log | cut -c1-500 | tail -20
