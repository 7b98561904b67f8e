#!/bin/bash
# stream-pool check: the MaskFormer leg as first / second engine of a process, then the default bench line
mkdir -p gpurun_out/r02t
PROBE=mf_only timeout 400 python scripts/dev/leg_probe.py 2>&1 | grep -E "MF|DETR" 
PROBE=detr_then_mf timeout 400 python scripts/dev/leg_probe.py 2>&1 | grep -E "MF|DETR"
timeout 900 python bench.py > gpurun_out/r02t/bench_default.json 2> gpurun_out/r02t/bench_default.err
tail -c 3000 gpurun_out/r02t/bench_default.json
