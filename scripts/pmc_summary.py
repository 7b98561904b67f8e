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


# known bytes per launch of scripts/pmc_calibrate.py's kernels: (read, write)
CALIB = {"cast_f32_bf16_kernel": ((128 << 20) * 4, (128 << 20) * 2), "add_rows_kernel": ((128 << 20) * 4, (128 << 20) * 2)}


def calibrate(fetch_csv, write_csv):
    """bytes per counter unit from the known-byte kernels (both agree within a few percent on a healthy box); falls back to the
    guide's factors (FETCH_SIZE KiB x 2 on gfx950, WRITE_SIZE KiB) when no calibration CSVs are given."""
    f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    fr, fw, detail = [], [], {}
    for k, (rb, wb) in CALIB.items():
        if k in f and k in w and f[k][0] and w[k][0]:
            r_unit, w_unit = rb / (f[k][1] / f[k][0]), wb / (w[k][1] / w[k][0])
            fr.append(r_unit)
            fw.append(w_unit)
            detail[k] = {"read_bytes_per_count": r_unit, "write_bytes_per_count": w_unit, "launches": f[k][0]}
    if not fr:
        return 2048.0, 1024.0, {}
    return sum(fr) / len(fr), sum(fw) / len(fw), detail


def main(fetch_csv, write_csv, out_md, out_json, calib_fetch=None, calib_write=None):
    f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    ru, wu, detail = calibrate(calib_fetch, calib_write) if calib_fetch and calib_write else (2048.0, 1024.0, {})
    res = {}
    src = (f"calibrated on known-byte kernels (scripts/pmc_calibrate.py): {ru:.1f} B per FETCH_SIZE count, {wu:.1f} B per WRITE_SIZE count"
           if detail else "FETCH_SIZE x 1024 x 2 (gfx950: counter reports half of wide coalesced reads), WRITE_SIZE x 1024 (uncalibrated)")
    lines = ["# HBM traffic per kernel (rocprofv3 PMC)", "", src + "; per launch averages.", "",
             "| kernel | launches | read MB/launch | write MB/launch | total MB/launch |", "|---|---:|---:|---:|---:|"]
    if detail:
        res["__calibration__"] = {"read_bytes_per_count": ru, "write_bytes_per_count": wu, "kernels": detail}
    for k in sorted(f, key=lambda k: -(f[k][1] * 2 + w.get(k, [0, 0])[1])):
        n = f[k][0]
        rb = f[k][1] * ru / n
        wb = (w[k][1] * wu / w[k][0]) if k in w and w[k][0] else 0.0
        res[k] = {"launches": n, "fetch_bytes_per_launch": rb, "write_bytes_per_launch": wb}
        lines.append(f"| `{k[:90]}` | {n} | {rb / 1e6:.2f} | {wb / 1e6:.2f} | {(rb + wb) / 1e6:.2f} |")
    open(out_md, "w").write("\n".join(lines) + "\n")
    json.dump(res, open(out_json, "w"), indent=1)
    print("\n".join(lines[:14]))


if __name__ == "__main__":
    main(*sys.argv[1:7])
