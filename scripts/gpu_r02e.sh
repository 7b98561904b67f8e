#!/bin/bash
out=gpurun_out/r02e; mkdir -p $out
run() { name=$1; shift; env "$@" timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --per-op $out/perop_$name.txt > $out/bench_$name.json 2> $out/bench_$name.err; python -c "
import json;j=json.loads(open('$out/bench_$name.json').read().strip().splitlines()[-1]);print('$name',j['value'],j['ms_per_step'])" | tee -a $out/summary.txt; }
run base FX_X=0
run dmak256 FX_DMA_MIN_KTOT=256
run dmak512 FX_DMA_MIN_KTOT=512
run bn128 FX_DMA_FORCE_BN=128
run dmak256bn128 FX_DMA_MIN_KTOT=256 FX_DMA_FORCE_BN=128
