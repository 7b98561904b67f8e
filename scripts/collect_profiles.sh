#!/bin/bash
# Collect the numbers behind bench.py's line on the GPU box (run from the repo root through gpurun):
#   scripts/collect_profiles.sh r01m
# 1. the official bench line (+ per-op HIP-event table), 2. rocprofv3 --kernel-trace --stats of the same command,
# 3. two SEPARATE rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only, as MI355X_MICROARCH.md prescribes),
# 4. the same for the MaskFormer, BiSeNetFormer and training workloads.  Everything lands in gpurun_out/; summaries are copied to profiles/ by hand.
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
python bench.py --per-op $OUT/${TAG}_per_op_hipevent.txt > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python bench.py --model fai-mf-l-coco-ins --cpu-iters 1 --per-op $OUT/${TAG}_mf_per_op_hipevent.txt > $OUT/${TAG}_mf_bench.json 2> $OUT/${TAG}_mf_bench.err
if [ -z "$FX_PROFILE_QUICK" ]; then
python bench.py --model fai-mf-l-coco-ins --no-cpu-baseline --steps 10 --warmup 3 --mf-full-masks > $OUT/${TAG}_mf_bench_fullmasks.json 2>/dev/null
python bench.py --model fai-mf-l-coco-ins --no-cpu-baseline --steps 10 --warmup 3 --mf-masks-d2h > $OUT/${TAG}_mf_bench_masks_d2h.json 2>/dev/null
fi
timeout 300 python bench.py --train --steps 10 --warmup 3 > $OUT/${TAG}_train_bench.json 2> $OUT/${TAG}_train_bench.err
timeout 300 python bench.py --train --norm BN --steps 10 --warmup 3 > $OUT/${TAG}_train_bn_bench.json 2> $OUT/${TAG}_train_bn_bench.err
python bench.py --model bisenetformer-l-ade --cpu-iters 1 --per-op $OUT/${TAG}_bf_per_op_hipevent.txt > $OUT/${TAG}_bf_bench.json 2> $OUT/${TAG}_bf_bench.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_train -o ${TAG}_train -- python $ROOT/bench.py --train --steps 4 --warmup 2 > $OUT/prof_${TAG}_train.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o $TAG -- python $ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $OUT/prof_$TAG.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_mf -o ${TAG}_mf -- python $ROOT/bench.py --model fai-mf-l-coco-ins --no-cpu-baseline --steps 5 --warmup 2 > $OUT/prof_${TAG}_mf.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_bf -o ${TAG}_bf -- python $ROOT/bench.py --model bisenetformer-l-ade --no-cpu-baseline --steps 10 --warmup 3 > $OUT/prof_${TAG}_bf.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$TAG -o f -- python $ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 2 > $OUT/pmc_fetch_$TAG.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_$TAG -o w -- python $ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 2 > $OUT/pmc_write_$TAG.log 2>&1
cd $ROOT
F=$(find $OUT/pmc_fetch_$TAG -name '*counter_collection.csv' | head -1)
W=$(find $OUT/pmc_write_$TAG -name '*counter_collection.csv' | head -1)
python scripts/pmc_summary.py $F $W $OUT/${TAG}_pmc_hbm.md $OUT/${TAG}_pmc_hbm.json > /dev/null 2>&1
# keep the merged-back payload small: the raw traces are large, the stats CSVs are what is committed
find $OUT/prof_$TAG $OUT/prof_${TAG}_mf $OUT/prof_${TAG}_bf $OUT/prof_${TAG}_train -name '*kernel_trace.csv' -delete
find $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG -name '*kernel_trace.csv' -delete
find $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG -name '*counter_collection.csv' -size +20M -delete
head -c 400 $OUT/${TAG}_bench.json; echo; head -c 300 $OUT/${TAG}_mf_bench.json; echo; head -c 300 $OUT/${TAG}_train_bench.json; echo
