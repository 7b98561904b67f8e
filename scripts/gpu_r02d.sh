#!/bin/bash
out=gpurun_out/r02d; mkdir -p $out
timeout 1500 python -m pytest tests/ -q -m gpu -x > $out/t_all.log 2>&1; echo "all gpu tests rc=$?" | tee -a $out/summary.txt
tail -15 $out/t_all.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $out/summary.txt; tail -5 $out/smoke.log
