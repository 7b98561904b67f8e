#!/bin/bash
# end-of-round collection (what profiles/<TAG>_* is copied from; round 6: TAG=r06z): the default bench line (with other_configs and the CPU baseline) + per-op table,
# rocprofv3 kernel stats and calibrated PMC HBM traffic of the headline, the training steps of the three families (+ kernel stats),
# MaskFormer / BiSeNetFormer inference with per-op tables
TAG=${TAG:-r06z}
out=$PWD/gpurun_out/$TAG; mkdir -p $out
ROOT=$PWD
export TMPDIR=/tmp
( time timeout 700 python bench.py --per-op $out/${TAG}_per_op_hipevent.txt > $out/${TAG}_bench.json 2> $out/${TAG}_bench.err ); echo "bench rc=$?"
timeout 300 python bench.py --train --steps 12 --warmup 4 > $out/${TAG}_train_bench.json 2> $out/${TAG}_train_bench.err; echo "detr train rc=$?"
timeout 300 python bench.py --train --norm BN --steps 12 --warmup 4 > $out/${TAG}_train_bn_bench.json 2> $out/${TAG}_train_bn_bench.err; echo "detr train BN rc=$?"
timeout 400 python bench.py --train --model bisenetformer-l-ade --norm FrozenBN --steps 10 --warmup 3 > $out/${TAG}_bf_train_bench.json 2> $out/${TAG}_bf_train_bench.err; echo "bf train (fp16, FrozenBN) rc=$?"
timeout 400 python bench.py --train --model bisenetformer-l-ade --norm BN --steps 10 --warmup 3 > $out/${TAG}_bf_train_bn_bench.json 2> $out/${TAG}_bf_train_bn_bench.err; echo "bf train (fp16, BN) rc=$?"
timeout 400 python bench.py --train --model bisenetformer-l-ade --norm BN --dtype bf16 --steps 10 --warmup 3 > $out/${TAG}_bf_train_bn_bf16_bench.json 2> $out/${TAG}_bf_train_bn_bf16_bench.err; echo "bf train (bf16, BN) rc=$?"
timeout 400 python bench.py --train --model fai-mf-l-coco-ins --steps 8 --warmup 3 > $out/${TAG}_mf_train_bench.json 2> $out/${TAG}_mf_train_bench.err; echo "mf train rc=$?"
FX_TRAIN_GRAPH=1 timeout 300 python bench.py --train --steps 12 --warmup 4 > $out/${TAG}_train_graph_bench.json 2> $out/${TAG}_train_graph_bench.err; echo "detr train (graphs) rc=$?"
timeout 300 python bench.py --no-cpu-baseline --no-other-configs --pipeline 1 > $out/${TAG}_bench_single_batch.json 2> $out/${TAG}_bench_single_batch.err; echo "headline, one batch in flight rc=$?"
timeout 600 python -m pytest tests/test_gpu_baseline_configs.py -q -s > $out/${TAG}_baseline_config_parity.txt 2>&1; echo "baseline-config parity rc=$?"
timeout 900 python -m pytest tests/test_gpu_train_baseline_configs.py -q -s > $out/${TAG}_train_baseline_config_parity.txt 2>&1; echo "training baseline-config parity rc=$?"
timeout 600 python -m pytest tests/test_gpu_fp16.py -q -s > $out/${TAG}_fp16_tests.txt 2>&1; echo "fp16 tests rc=$?"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/${TAG}_smoke.txt 2>&1; echo "smoke rc=$?"
timeout 300 python bench.py --model fai-mf-l-coco-ins --no-cpu-baseline --per-op $out/${TAG}_mf_per_op_hipevent.txt > $out/${TAG}_mf_bench.json 2> $out/${TAG}_mf_bench.err
timeout 300 python bench.py --model bisenetformer-l-ade --no-cpu-baseline --per-op $out/${TAG}_bf_per_op_hipevent.txt > $out/${TAG}_bf_bench.json 2> $out/${TAG}_bf_bench.err
# kernel stats twice: as timed (three batches in flight: durations under concurrency) and SERIAL (one whole-batch plan on one stream:
# the durations the per-op table / `roofline.achieved` are measured at)
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o $TAG -- python $ROOT/bench.py --no-cpu-baseline --no-other-configs --steps 10 --warmup 3 > $out/prof.log 2>&1
FX_BENCH_PIPELINE=1 FX_STREAMS=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_serial -o ${TAG}_serial -- python $ROOT/bench.py --no-cpu-baseline --no-other-configs --steps 10 --warmup 3 > $out/prof_serial.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_train -o ${TAG}_train -- python $ROOT/bench.py --train --no-cpu-baseline --steps 6 --warmup 3 > $out/prof_train.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_bf_train -o ${TAG}_bf_train -- python $ROOT/bench.py --train --model bisenetformer-l-ade --norm BN --no-cpu-baseline --steps 4 --warmup 2 > $out/prof_bf_train.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -o f -- python $ROOT/bench.py --no-cpu-baseline --no-other-configs --steps 3 --warmup 2 > $out/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -o w -- python $ROOT/bench.py --no-cpu-baseline --no-other-configs --steps 3 --warmup 2 > $out/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/cal_fetch -o f -- python $ROOT/scripts/pmc_calibrate.py > $out/cal_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/cal_write -o w -- python $ROOT/scripts/pmc_calibrate.py > $out/cal_write.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_train_$c -o t -- python $ROOT/bench.py --train --no-cpu-baseline --steps 2 --warmup 1 > $out/pmc_train_$c.log 2>&1; done
FX_BENCH_PIPELINE=1 FX_STREAMS=1 timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $out/pmc_sq -o s -- python $ROOT/bench.py --no-cpu-baseline --no-other-configs --steps 2 --warmup 1 > $out/pmc_sq.log 2>&1
cd $ROOT
python scripts/pmc_sq_summary.py $out/${TAG}_sq_counters.md $(find $out/pmc_sq -name '*counter_collection.csv' | head -1) | head -5
F=$(find $out/pmc_fetch -name '*counter_collection.csv' | head -1); W=$(find $out/pmc_write -name '*counter_collection.csv' | head -1)
CF=$(find $out/cal_fetch -name '*counter_collection.csv' | head -1); CW=$(find $out/cal_write -name '*counter_collection.csv' | head -1)
python scripts/pmc_summary.py $F $W $out/${TAG}_pmc_hbm.md $out/${TAG}_pmc_hbm.json $CF $CW | head -12
python scripts/pmc_summary.py $out/pmc_train_FETCH_SIZE/t_counter_collection.csv $out/pmc_train_WRITE_SIZE/t_counter_collection.csv $out/${TAG}_train_pmc_hbm.md $out/${TAG}_train_pmc_hbm.json $CF $CW | head -8   # -> profiles/pmc_train_hbm_latest.json (run the training bench line after copying it)
for d in prof prof_serial prof_train prof_bf_train; do find $out/$d -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $out/${TAG}_${d}_kernel_stats.csv; done
find $out -name '*kernel_trace.csv' -delete; find $out -name '*counter_collection.csv' -size +20M -delete
head -c 300 $out/${TAG}_bench.json; echo; tail -2 $out/${TAG}_bench.err | cut -c1-300
python -c "
import json; j=json.loads(open('$out/${TAG}_bench.json').read().strip().splitlines()[-1])
for k,v in j.get('other_configs',{}).items(): print(k, {a:v.get(a) for a in ('value','ms_per_step','error')} if isinstance(v,dict) else v)"
for f in train_bench train_bn_bench bf_train_bench bf_train_bn_bench bf_train_bn_bf16_bench mf_train_bench mf_bench bf_bench; do head -c 200 $out/${TAG}_$f.json; echo; done
