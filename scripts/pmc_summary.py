#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as
MI355X_MICROARCH.md prescribes).  Units: the counters are in KiB (hbm_bytes = (FETCH+WRITE)*1024); on gfx950 FETCH_SIZE
reports exactly half of the bytes of wide coalesced streaming reads, so the read side is doubled (WRITE_SIZE uncalibrated,
taken as is).  Output: markdown table + JSON {kernel: {launches, fetch_bytes_per_launch, write_bytes_per_launch}}."""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*$", "", name)
    return re.sub(r"^void ", "", name)


def load(path, counter):
    agg = defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        a = agg[short(row["Kernel_Name"])]
        a[0] += 1
        a[1] += float(row["Counter_Value"])
    return agg


def main(fetch_csv, write_csv, out_md, out_json):
    f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    res = {}
    lines = ["# HBM traffic per kernel (rocprofv3 PMC, gfx950-corrected)", "",
             "FETCH_SIZE x 1024 x 2 (gfx950: counter reports half of wide coalesced reads), WRITE_SIZE x 1024; per launch averages.", "",
             "| kernel | launches | read MB/launch | write MB/launch | total MB/launch |", "|---|---:|---:|---:|---:|"]
    for k in sorted(f, key=lambda k: -(f[k][1] * 2 + w.get(k, [0, 0])[1])):
        n = f[k][0]
        rb = f[k][1] * 1024 * 2 / n
        wb = (w[k][1] * 1024 / w[k][0]) if k in w and w[k][0] else 0.0
        res[k] = {"launches": n, "fetch_bytes_per_launch": rb, "write_bytes_per_launch": wb}
        lines.append(f"| `{k[:90]}` | {n} | {rb / 1e6:.2f} | {wb / 1e6:.2f} | {(rb + wb) / 1e6:.2f} |")
    open(out_md, "w").write("\n".join(lines) + "\n")
    json.dump(res, open(out_json, "w"), indent=1)
    print("\n".join(lines[:14]))


if __name__ == "__main__":
    main(*sys.argv[1:5])
