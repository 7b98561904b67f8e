#!/bin/bash
TAG=r05k
out=$PWD/gpurun_out/$TAG; mkdir -p $out
timeout 300 python scripts/dev/rc_scaling_probe.py > $out/rc_scaling.txt 2>&1; echo "rc=$?"; grep -v amdgpu $out/rc_scaling.txt | cut -c1-400
