#!/bin/bash
out=$PWD/gpurun_out/r02q; mkdir -p $out
timeout 400 python bench.py --model fai-mf-l-coco-ins --no-cpu-baseline --per-op $out/mf_per_op.txt > $out/mf_bench.json 2>/dev/null
timeout 400 python bench.py --model bisenetformer-l-ade --no-cpu-baseline --per-op $out/bf_per_op.txt > $out/bf_bench.json 2>/dev/null
head -c 200 $out/mf_bench.json; echo; head -c 200 $out/bf_bench.json; echo
