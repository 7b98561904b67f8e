#!/bin/bash
out=$PWD/gpurun_out/r02q; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_train_seg.py tests/test_gpu_mask_criterion.py tests/test_gpu_train_bf.py -q > $out/t.log 2>&1; grep -E "^E  |passed|failed|Error" $out/t.log | cut -c1-400 | tail -20
timeout 400 python bench.py --train --model bisenetformer-l-ade --norm BN --steps 8 --warmup 2 2>/dev/null | head -c 260; echo
cd /tmp; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_bf_train -o bf_train -- python $OLDPWD/bench.py --train --model bisenetformer-l-ade --norm BN --steps 3 --warmup 1 > $out/prof_bf_train.log 2>&1; cd $OLDPWD
find $out -name '*kernel_trace.csv' -delete
head -14 $out/prof_bf_train/bf_train_kernel_stats.csv | cut -c1-130
