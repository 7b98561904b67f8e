#!/bin/bash
out=gpurun_out/r02c; mkdir -p $out
for c in 0 1; do echo "== FX_PW_CHAIN=$c"; FX_PW_CHAIN=$c FX_PW_CHAIN_MAX_STAGE=1 python scripts/dev/parity_probe.py 2>&1 | tail -22; done | tee $out/probe.txt
