#!/bin/bash
# training step: rocprofv3 kernel trace (per-dispatch rows incl. queue) for the stream-overlap analysis
mkdir -p gpurun_out/r02z
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/r02z/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02z/prof -o train -- python bench.py --train --no-cpu-baseline --steps 6 --warmup 3 > gpurun_out/r02z/prof.log 2>&1
find gpurun_out/r02z/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r02z/train_kernel_stats.csv
find gpurun_out/r02z/prof -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} gpurun_out/r02z/train_kernel_trace.csv
rm -rf gpurun_out/r02z/prof
ls -la gpurun_out/r02z
