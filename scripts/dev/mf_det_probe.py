"""Probe for tests/test_gpu_mf.py::test_mf_detections_vs_reference_golden: per strong reference detection, the closest engine detection."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_gpu_mf as T
s = T.setup.__wrapped__() if hasattr(T.setup, "__wrapped__") else T.setup.__pytest_wrapped__.obj()
g, cfg, sd, eng, images, x_u8, forced, *_ = s
pl = eng.forward(x_u8, forced_attn=forced); torch.cuda.synchronize()
tot = 0
for b in range(2):
    n = int(pl.det_count[b])
    mine = [(int(q), float(sc), int(l), bx.tolist()) for q, sc, l, bx in zip(pl.det_query[b, :n].cpu(), pl.det_scores[b, :n].cpu(), pl.det_labels[b, :n].cpu(), pl.det_boxes[b, :n].cpu())]
    conf, cls, bbox = g[f"det{b}_conf"], g[f"det{b}_cls"], g[f"det{b}_bbox"]
    strong = conf > cfg["threshold"] + 0.05
    for c, k, bx in zip(conf[strong], cls[strong], bbox[strong]):
        hit = [v for v in mine if v[2] == int(k) and abs(v[1] - float(c)) <= 5e-2 and max(abs(np.array(v[3]) - bx)) <= 2]
        tot += bool(hit)
        if not hit:
            cand = sorted(mine, key=lambda v: (v[2] != int(k)) * 1e6 + np.abs(np.array(v[3]) - bx).sum())[:2]
            print(f"img{b} MISS ref conf={c:.4f} cls={k} box={bx.tolist()} | nearest: {cand}")
print("found", tot, "n_mine", n)
