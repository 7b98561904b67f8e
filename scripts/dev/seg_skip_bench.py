"""fx_seg_postprocess (x8 cell kernel, 16 x 100 queries, 80x80 -> 640x640: one BiSeNetFormer part) with and without the hopeless-query
test, on two kinds of input: 'flat' - every query competitive everywhere (what random-init weights produce; the bench), and 'peaked' -
10 queries with class score near 1 and compact masks, the rest with score <= 0.05 (what a trained model produces).  One process, the knob
is read per call.  usage: python scripts/dev/seg_skip_bench.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from focoos_amd import _lib  # noqa: E402

DEV = "cuda:0"
lib = _lib.load()
B, Q, h, w, S = 16, 100, 80, 80, 8
H, W = h * S, w * S
g = torch.Generator().manual_seed(0)


def inputs(kind):
    if kind == "flat":
        lo = torch.sigmoid(torch.randn(B, Q, h, w, generator=g) * 0.5)
        score = torch.rand(B, Q, generator=g) * 0.02 + 0.01
    else:
        yy, xx = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing="ij")
        lo = torch.full((B, Q, h, w), 0.02)
        lo += torch.rand(B, Q, h, w, generator=g) * 0.02
        score = torch.rand(B, Q, generator=g) * 0.05
        for b in range(B):
            for q in torch.randperm(Q, generator=g)[:10].tolist():
                cy, cx, r = (float(v) for v in (torch.rand(3, generator=g) * torch.tensor([h, w, 20.0]) + torch.tensor([0, 0, 8.0])))
                lo[b, q] = torch.sigmoid((r - ((yy - cy) ** 2 + (xx - cx) ** 2).sqrt()) * 1.5).clamp_min(0.02)
                score[b, q] = 0.8 + 0.2 * float(torch.rand(1, generator=g))
    return lo.to(DEV), score.to(DEV), torch.randint(0, 150, (B, Q), generator=g).int().to(DEV)


def run(lo, score, label, skip, reps=50):
    os.environ["FX_SEG_SKIP"] = str(skip)
    nb = lib.fx_seg_postprocess_workspace_bytes(B, Q, h, w, H, W)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    cnt = torch.zeros(B, dtype=torch.int32, device=DEV)
    dq, dl, da = (torch.zeros(B, Q, dtype=torch.int32, device=DEV) for _ in range(3))
    ds = torch.zeros(B, Q, dtype=torch.float32, device=DEV)
    db = torch.zeros(B, Q, 4, dtype=torch.int32, device=DEV)
    words = torch.zeros(B, Q, H, W // 32, dtype=torch.int32, device=DEV)
    winner = torch.zeros(B, H, W, dtype=torch.uint8, device=DEV)
    st = torch.cuda.current_stream().cuda_stream

    def call():
        rc = lib.fx_seg_postprocess(lo.data_ptr(), h, w, H, W, score.data_ptr(), label.data_ptr(), B, Q, 0.5, 0, ws.data_ptr(), nb, cnt.data_ptr(),
                                    dq.data_ptr(), ds.data_ptr(), dl.data_ptr(), db.data_ptr(), da.data_ptr(), words.data_ptr(), winner.data_ptr(), st)
        assert rc == 0, rc

    for _ in range(5):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, winner.clone()


for kind in ("flat", "peaked"):
    lo, score, label = inputs(kind)
    res = {}
    for rep in range(2):
        for skip in (0, 1):
            t, win = run(lo, score, label, skip)
            res.setdefault(skip, []).append(t)
            if skip == 0:
                ref = win
            else:
                assert torch.equal(win, ref)
    print(f"{kind:7s}: skip=0 {min(res[0]):7.1f} us   skip=1 {min(res[1]):7.1f} us   (winner maps identical)")
