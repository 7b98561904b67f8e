#!/bin/bash
# training step: launch census + rocprofv3 kernel stats
mkdir -p gpurun_out/r02z
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null

timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02z/prof -o train -- python bench.py --train --no-cpu-baseline --steps 7 --warmup 2 > gpurun_out/r02z/prof.log 2>&1
find gpurun_out/r02z/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r02z/train_kernel_stats.csv
find gpurun_out/r02z/prof -type f ! -name "*kernel_stats.csv" -delete
head -40 gpurun_out/r02z/train_kernel_stats.csv | cut -c1-160
