#!/bin/bash
# TrainStep(graphs="auto"): the new test, then eager / replay / auto of the three training lines inside one call
out=$PWD/gpurun_out/r05m; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_train_detr.py -q -k "graphs" > $out/tests.txt 2>&1; echo "graph tests rc=$?"; tail -3 $out/tests.txt
for m in 0 1 auto; do
  timeout 300 python bench.py --train --train-graphs $m --steps 12 --warmup 8 --no-cpu-baseline > $out/detr_$m.json 2> $out/detr_$m.err; echo "detr $m rc=$?"
done
for m in 0 auto; do
  timeout 300 python bench.py --train --model bisenetformer-l-ade --norm BN --train-graphs $m --steps 10 --warmup 8 --no-cpu-baseline > $out/bf_$m.json 2> $out/bf_$m.err; echo "bf $m rc=$?"
  timeout 300 python bench.py --train --model fai-mf-l-coco-ins --train-graphs $m --steps 8 --warmup 8 --no-cpu-baseline > $out/mf_$m.json 2> $out/mf_$m.err; echo "mf $m rc=$?"
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('OUT','gpurun_out/r05m')+'/*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), j['value'], j['ms_per_step'], j['config'].get('graphs'))
    except Exception as e: print(f, 'ERR', e)
PY
