#!/bin/bash
# round 5, call L: row-chain weight prefetch into L2 (FX_RC_PREFETCH=0/1): parity, per-stage stamps, A/B on RT-DETR and BiSeNetFormer
TAG=r05l
out=$PWD/gpurun_out/$TAG; mkdir -p $out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -k "row_chain" > $out/rc_tests.txt 2>&1; echo "row_chain kernel tests rc=$?"; tail -2 $out/rc_tests.txt
for pf in 0 1; do FX_RC_PREFETCH=$pf timeout 300 python scripts/dev/rc_scaling_probe.py > $out/rc_scaling_pf$pf.txt 2>&1; echo "--- FX_RC_PREFETCH=$pf"; grep -E "rows    32|rows  4800|stages" $out/rc_scaling_pf$pf.txt | cut -c1-330; done
for i in 1 2; do
  for pf in 0 1; do
    FX_RC_PREFETCH=$pf timeout 200 python bench.py --no-cpu-baseline --no-other-configs --steps 30 --warmup 5 > $out/bench_pf${pf}_$i.json 2> $out/bench_pf${pf}_$i.err
    python -c "
import json; j=json.loads(open('$out/bench_pf${pf}_$i.json').read().strip().splitlines()[-1]); print('RT-DETR FX_RC_PREFETCH=$pf run $i:', j['value'], 'img/s', j['ms_per_step'], 'ms; row_chain', j['roofline']['all_conv_variants'].get('row_chain',{}).get('ms'))"
  done
done
for pf in 0 1 0 1; do
  FX_RC_PREFETCH=$pf timeout 200 python bench.py --model bisenetformer-l-ade --no-cpu-baseline --steps 30 --warmup 5 > $out/bf_pf${pf}.json 2> $out/bf_pf${pf}.err
  python -c "
import json; j=json.loads(open('$out/bf_pf${pf}.json').read().strip().splitlines()[-1]); print('BiSeNetFormer FX_RC_PREFETCH=$pf:', j['value'], 'img/s', j['ms_per_step'], 'ms')"
done
timeout 400 python -m pytest tests/test_gpu_baseline_configs.py -q -x -k "config1" > $out/parity.txt 2>&1; echo "config1 parity rc=$?"; tail -2 $out/parity.txt | cut -c1-200
