#!/bin/bash
out=$PWD/gpurun_out/r02q; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_train_bf.py -q -s -k losses > $out/t_bf.log 2>&1; grep -E "FrozenBN:|BN:|^quartiles|^zero-grad|^E  |passed|failed" $out/t_bf.log | cut -c1-1200 | tail -30
timeout 900 python -m pytest tests/test_gpu_train_conv.py -q -s -k "bottleneck_pair or batch_stat" > $out/t_tr.log 2>&1; grep -E "parameter tensors|bottleneck pair|^quartiles|^E  |passed|failed" $out/t_tr.log | cut -c1-600 | tail -30
