#!/bin/bash
# copy the judged summaries of a final collection (gpurun_out/<tag>/) into profiles/ under the names earlier rounds used
TAG=${1:-r05z}
src=gpurun_out/$TAG
for f in $src/${TAG}_*.json $src/${TAG}_*.txt $src/${TAG}_*.md; do [ -f "$f" ] && cp "$f" profiles/; done
cp $src/${TAG}_prof_kernel_stats.csv profiles/${TAG}_rocprofv3_kernel_stats.csv
cp $src/${TAG}_prof_serial_kernel_stats.csv profiles/${TAG}_serial_rocprofv3_kernel_stats.csv
cp $src/${TAG}_prof_train_kernel_stats.csv profiles/${TAG}_train_rocprofv3_kernel_stats.csv
cp $src/${TAG}_prof_bf_train_kernel_stats.csv profiles/${TAG}_bf_train_rocprofv3_kernel_stats.csv
cp $src/${TAG}_pmc_hbm.json profiles/pmc_hbm_latest.json
cp $src/${TAG}_train_pmc_hbm.json profiles/pmc_train_hbm_latest.json
ls profiles | grep $TAG | wc -l
