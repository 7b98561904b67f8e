#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite .db) into a per-kernel stats table
(name, calls, total ms, avg us, min us, max us, % of GPU kernel time) -> markdown for profiles/."""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*$", "", name)
    m = re.match(r"void (.*)", name)
    return (m.group(1) if m else name)[:110]


def main(db_path: str, out_path: str, title: str):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, start, end from kernels").fetchall()
    agg = {}
    for n, s, e in rows:
        d = agg.setdefault(short(n), [0, 0.0, 1e30, 0.0])
        dur = (e - s) / 1e3  # us
        d[0] += 1
        d[1] += dur
        d[2] = min(d[2], dur)
        d[3] = max(d[3], dur)
    tot = sum(v[1] for v in agg.values())
    lines = [f"# {title}", "", f"source: rocprofv3 --kernel-trace (rocpd db `{db_path.split('/')[-1]}`), {len(rows)} dispatches, "
             f"{tot / 1e3:.2f} ms total kernel time", "", "| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---:|---:|---:|---:|---:|---:|"]
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"| `{k}` | {v[0]} | {v[1] / 1e3:.3f} | {v[1] / v[0]:.2f} | {v[2]:.2f} | {v[3]:.2f} | {100 * v[1] / tot:.2f} |")
    open(out_path, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:16]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "rocprofv3 kernel stats")
