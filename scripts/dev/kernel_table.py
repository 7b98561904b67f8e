#!/usr/bin/env python
"""Per-step kernel table from a rocprofv3 --kernel-trace --stats kernel_stats.csv: python scripts/dev/kernel_table.py <csv> <steps|@kernel> [rows]
(@kernel: divide by the number of calls of that kernel, e.g. @adamw_kernel for a training run)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
div = sys.argv[2]
if div.startswith("@"):
    steps = sum(int(r["Calls"]) for r in rows if r["Name"].startswith(div[1:]))
else:
    steps = int(div)
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / steps
print(f"{steps} steps, {tot:.3f} ms of kernels per step, {sum(int(r['Calls']) for r in rows) / steps:.0f} launches per step")
acc = 0.0
for r in rows[:n]:
    t = float(r["TotalDurationNs"]) / 1e6 / steps
    acc += t
    print(f"{r['Name'][:100]:100s} {int(r['Calls']) / steps:7.1f} {t:8.3f} {float(r['AverageNs']) / 1e3:8.1f} {acc:7.2f}")
