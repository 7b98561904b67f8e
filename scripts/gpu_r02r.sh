#!/bin/bash
out=$PWD/gpurun_out/r02r; mkdir -p $out
ROOT=$PWD
export TMPDIR=/tmp
timeout 300 python bench.py --train --steps 10 --warmup 3 > $out/train_bench.json 2> $out/train_bench.err; echo "detr train rc=$?"
timeout 300 python bench.py --train --norm BN --steps 10 --warmup 3 > $out/train_bn_bench.json 2> $out/train_bn_bench.err; echo "detr train BN rc=$?"
timeout 400 python bench.py --train --model bisenetformer-l-ade --steps 8 --warmup 2 > $out/bf_train_bench.json 2> $out/bf_train_bench.err; echo "bf train rc=$?"
timeout 400 python bench.py --train --model bisenetformer-l-ade --norm BN --steps 8 --warmup 2 > $out/bf_train_bn_bench.json 2> $out/bf_train_bn_bench.err; echo "bf train BN rc=$?"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_bf_train -o bf_train -- python $ROOT/bench.py --train --model bisenetformer-l-ade --norm BN --steps 3 --warmup 1 > $out/prof_bf_train.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_train -o train -- python $ROOT/bench.py --train --steps 4 --warmup 2 > $out/prof_train.log 2>&1
cd $ROOT
find $out -name '*kernel_trace.csv' -delete
for f in train_bench train_bn_bench bf_train_bench bf_train_bn_bench; do head -c 330 $out/$f.json; echo; tail -3 $out/$f.err | cut -c1-300; done
