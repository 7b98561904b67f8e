#!/bin/bash
out=$PWD/gpurun_out/r02o; mkdir -p $out; rm -f $out/probe.txt
for cfgs in "FX_PW_CHAIN=0" "FX_PW_CHAIN=1"; do
echo "== $cfgs" | tee -a $out/probe.txt
env $cfgs timeout 300 python -m pytest tests/test_gpu_mf.py -q -x -k "golden" 2>&1 | grep -E "AssertionError:|passed|failed" | tee -a $out/probe.txt
env $cfgs timeout 300 python scripts/dev/mf_det_probe.py 2>&1 | grep -E "found|MISS" | cut -c1-60 | tee -a $out/probe.txt
done
