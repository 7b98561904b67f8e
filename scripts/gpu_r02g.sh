#!/bin/bash
out=$PWD/gpurun_out/r02g; mkdir -p $out
ROOT=$PWD
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $out/pmc1 -o a -- python $ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $out/pmc1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_SMEM --kernel-trace --output-format csv -d $out/pmc2 -o b -- python $ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $out/pmc2.log 2>&1
cd $ROOT
A=$(find $out/pmc1 -name '*counter_collection.csv' | head -1); B=$(find $out/pmc2 -name '*counter_collection.csv' | head -1)
python scripts/pmc_sq_summary.py $out/sq_summary.md $A $B | cut -c1-400
find $out -name '*kernel_trace.csv' -delete; find $out -name '*counter_collection.csv' -size +10M -delete
tail -3 $out/pmc1.log
