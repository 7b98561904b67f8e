#!/usr/bin/env python
"""Per-kernel SQ counters from rocprofv3 --pmc passes (one or more counter_collection CSVs): sums per kernel name over all its
launches, plus the derived fractions used in DESIGN.md: MFMA-busy share of wave time, wave parked (s_waitcnt / barrier),
issue stalls.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles
(MI355X_MICROARCH.md, cycle-constants table).  Usage: pmc_sq_summary.py out.md a.csv [b.csv ...]"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)[:70]


def main(out_md, *paths):
    agg = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    for p in paths:
        for row in csv.DictReader(open(p)):
            k = short(row["Kernel_Name"])
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            launches[k].add((p, row.get("Dispatch_Id", row.get("Correlation_Id", ""))))
    cols = sorted({c for v in agg.values() for c in v})
    lines = ["# SQ counters per kernel (rocprofv3 --pmc, sums over launches)", "", "| kernel | launches | " + " | ".join(cols) + " | mfma_busy/wave_cyc | wait_any | wait_inst |", "|---|---:|" + "---:|" * (len(cols) + 3)]
    for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0)):
        v = agg[k]
        wc = v.get("SQ_WAVE_CYCLES", 0) * 4.0   # quad-cycles -> cycles summed over waves
        busy = v.get("SQ_BUSY_CYCLES", 0)
        mf = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
        d1 = f"{mf / wc:.3f}" if wc else "-"
        d2 = f"{v.get('SQ_WAIT_ANY', 0) / v['SQ_WAVE_CYCLES']:.3f}" if v.get("SQ_WAVE_CYCLES") else "-"
        d3 = f"{v.get('SQ_WAIT_INST_ANY', 0) / v['SQ_WAVE_CYCLES']:.3f}" if v.get("SQ_WAVE_CYCLES") else "-"
        lines.append(f"| `{k}` | {len(launches[k]) // max(len(paths), 1) or len(launches[k])} | " + " | ".join(f"{v.get(c, 0):.4g}" for c in cols) + f" | {d1} | {d2} | {d3} |")
    open(out_md, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:16]))


if __name__ == "__main__":
    main(sys.argv[1], *sys.argv[2:])
