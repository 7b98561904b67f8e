"""Timeline of one two-part inference step from a rocprofv3 --kernel-trace CSV: per hardware queue busy time, time with 0 / 1 / 2 kernels in
flight, and what each queue runs per 0.5 ms.  usage: python scripts/dev/trace_timeline.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"])))
        for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: r[1])
stems = [r for r in rows if "stem" in r[0]]
# timed two-part steps: two stem launches on different queues within 200 us of each other
pairs = [(a, b) for a, b in zip(stems, stems[1:]) if a[3] != b[3] and b[1] - a[1] < 1000000]
print("two-queue steps found:", len(pairs), "of", len(stems), "stem launches;", [(s[3], round((s[1] - stems[0][1]) / 1e6, 2)) for s in stems[:40]])
if not pairs:
    sys.exit(0)
a, b = pairs[len(pairs) // 2]
nxt = [p for p in pairs if p[0][1] > a[1]][0]
t0, t1 = a[1], nxt[0][1]
step = [r for r in rows if t0 <= r[1] < t1]
print("step wall ms %.3f, kernels %d" % ((t1 - t0) / 1e6, len(step)))
byq = collections.defaultdict(list)
for r in step:
    byq[r[3]].append(r)
for q, l in sorted(byq.items()):
    print("queue %s: %d kernels, busy %.3f ms, last end %.3f ms" % (q, len(l), sum(r[2] - r[1] for r in l) / 1e6, (max(r[2] for r in l) - t0) / 1e6))
ev = sorted([(r[1], 1) for r in step] + [(min(r[2], t1), -1) for r in step])
active, last, hist = 0, t0, collections.Counter()
for t, d in ev:
    hist[active] += t - last
    last = t
    active += d
print("ms with N kernels in flight:", {k: round(v / 1e6, 3) for k, v in sorted(hist.items())})


def name(n):
    for k in ("conv3x3_kplane", "conv_pw_kplane", "conv3x3_flat", "pw_chain", "conv_igemm_dma", "conv_igemm", "row_chain", "msda", "mha", "topk", "stem", "maxpool",
              "avgpool", "score_head", "layernorm"):
        if k in n:
            return k
    return n[:24]


for q, l in sorted(byq.items()):
    print("--- queue", q)
    bins = collections.defaultdict(collections.Counter)
    for r in l:
        bins[int((r[1] - t0) / 0.5e6)][name(r[0])] += (r[2] - r[1]) / 1e6
    for bb in sorted(bins):
        print("  %.1f ms: %s" % (bb * 0.5, {k: round(v, 3) for k, v in bins[bb].most_common(4)}))
# the longest kernels of the step
print("--- 25 longest launches (ms, workgroups, name)")
for r in sorted(step, key=lambda r: r[1] - r[2])[:25]:
    print("  %.3f %6d %s" % ((r[2] - r[1]) / 1e6, r[4], r[0][:90]))
