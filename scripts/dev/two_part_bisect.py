"""Which kernels of the OTHER queue make a part's decoder go wrong?  Part 0 runs completely (eager, stream s0) while a chosen subset of
part 1's front launches is replayed on stream s1; part 0's outputs are compared bitwise with its serial result.  GPU only."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from focoos_amd.model import FAIDetr  # noqa: E402
from focoos_amd.registry import ModelRegistry  # noqa: E402
from focoos_amd.synth import synth_image_structured as sis  # noqa: E402

cfg = ModelRegistry.get_model_info("fai-detr-l-obj365")["config"]
model = FAIDetr(cfg, device="cuda:0", seed=0)
eng = model.engine
B, N = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 12
imgs = torch.from_numpy(np.stack([sis(100 + i) for i in range(B)])).to("cuda:0")
pl = eng.plan(B, 640, 640, False, 2)
p0, p1 = pl.parts
pl.input.copy_(imgs)
pl.sizes.copy_(torch.tensor([[640, 640]] * B, dtype=torch.int32))
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
keys = [n for n in p0.bufs if n.startswith("dec") or n.startswith("encbb") or n in ("target", "logits", "memory", "output_memory")]


def snap():
    d = {n: p0.bufs[n].t.clone() for n in keys}
    d["probs"] = pl.probs[: B // 2].clone()
    d["refs"] = torch.stack([r.clone() for r in p0.refs])
    return d


p0._launch(p0.ops, s0.cuda_stream, 0.3)
p1._launch(p1.ops, s0.cuda_stream, 0.3)
torch.cuda.synchronize()
ref = snap()
groups = {}
for i, (fn, a) in enumerate(p1.ops[: p1.split_at]):
    m = p1.meta.get(i)
    key = (m["variant"].split("<")[0] if m else fn.__name__)
    groups.setdefault(key, []).append(i)
groups["ALL_FRONT"] = list(range(p1.split_at))
groups["NONE"] = []
FENCED = os.environ.get("FX_BISECT_FENCED", "0") == "1"
s2 = torch.cuda.Stream()


def launch_victim():
    """FENCED: every decoder launch is followed by a round trip through another queue (event -> s2 -> event), which forces the
    runtime to emit barrier packets with full release / acquire between consecutive victim kernels."""
    if not FENCED:
        p0._launch(p0.ops[p0.split_at:], s0.cuda_stream, 0.3)
        return
    for fn, a in p0.ops[p0.split_at:]:
        fn(*p0.patch_args(fn, a, 0.3), s0.cuda_stream)
        e1 = torch.cuda.Event()
        e1.record(s0)
        s2.wait_event(e1)
        e2 = torch.cuda.Event()
        e2.record(s2)
        s0.wait_event(e2)


if FENCED:
    groups = {k: groups[k] for k in ("pw_chain", "conv_igemm", "ALL_FRONT", "NONE")}
for name, idxs in groups.items():
    bad, firsts = 0, {}
    for it in range(N):
        torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            for rep in range(3 if len(idxs) < 40 else 1):
                for i in idxs:
                    fn, a = p1.ops[i]
                    fn(*p1.patch_args(fn, a, 0.3), torch.cuda.current_stream().cuda_stream)
        launch_victim()                                            # part 0's decoder only (its front outputs are in place)
        torch.cuda.synchronize()
        cur = snap()
        d = [k for k in ref if not torch.equal(ref[k], cur[k])]
        if d:
            bad += 1
            firsts[d[0]] = firsts.get(d[0], 0) + 1
    print(f"aggressor {name:28s} launches {len(idxs):3d}: part-0 decoder wrong in {bad}/{N}  first differing buffer: {firsts}")

if os.environ.get("FX_BISECT_VICTIM", "0") == "1":
    # which victim launch is the first to go wrong?  run only the first K decoder launches beside the aggressor
    idxs = groups["ALL_FRONT"] if "ALL_FRONT" in groups else list(range(p1.split_at))
    names = [(fn.__name__, (p0.meta.get(p0.split_at + i) or {}).get("name", "")) for i, (fn, a) in enumerate(p0.ops[p0.split_at:])]
    # restore the reference state first
    p0._launch(p0.ops, s0.cuda_stream, 0.3)
    torch.cuda.synchronize()
    for K in range(1, min(len(names), 14) + 1):
        bad = 0
        which = {}
        for it in range(8):
            torch.cuda.synchronize()
            with torch.cuda.stream(s1):
                for i in idxs:
                    fn, a = p1.ops[i]
                    fn(*p1.patch_args(fn, a, 0.3), torch.cuda.current_stream().cuda_stream)
            p0._launch(p0.ops[p0.split_at:p0.split_at + K], s0.cuda_stream, 0.3)
            torch.cuda.synchronize()
            cur = snap()
            d = [k for k in ref if not torch.equal(ref[k], cur[k])]
            if d:
                bad += 1
                which[d[0]] = which.get(d[0], 0) + 1
            p0._launch(p0.ops[p0.split_at:], s0.cuda_stream, 0.3)   # restore
            torch.cuda.synchronize()
        print(f"victim prefix K={K:2d} (last launch {names[K - 1]}): wrong in {bad}/8 {which}")

if os.environ.get("FX_BISECT_DETAIL", "0") == "1":
    idxs = list(range(p1.split_at))
    p0._launch(p0.ops, s0.cuda_stream, 0.3)
    torch.cuda.synchronize()
    r0 = p0.refs[0].clone()
    hb_ref = p0.bufs["encbb.1"].t.clone()
    for it in range(4):
        torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            for i in idxs:
                fn, a = p1.ops[i]
                fn(*p1.patch_args(fn, a, 0.3), torch.cuda.current_stream().cuda_stream)
        p0._launch(p0.ops[p0.split_at:p0.split_at + 4], s0.cuda_stream, 0.3)
        torch.cuda.synchronize()
        cur = p0.refs[0]
        ne = (cur != r0)
        rows = ne.any(1).nonzero().flatten()
        print(f"trial {it}: refs[0] differing elements {int(ne.sum())} of {ne.numel()}, rows {rows.numel()} (first {rows[:8].tolist()}, last {rows[-4:].tolist()}), "
              f"max|d| {float((cur - r0).abs().max()):.3e}; hb equal {bool(torch.equal(hb_ref, p0.bufs['encbb.1'].t))}; NaN {int(torch.isnan(cur).sum())}")
        if rows.numel():
            r = int(rows[0])
            print("   row", r, "ref", r0[r].tolist(), "cur", cur[r].tolist(), "idx", int(p0.enc_topk.flatten()[r]))
        p0._launch(p0.ops[p0.split_at:], s0.cuda_stream, 0.3)
        torch.cuda.synchronize()

if os.environ.get("FX_BISECT_DETAIL", "0") == "2":
    idxs = list(range(p1.split_at))
    p0._launch(p0.ops, s0.cuda_stream, 0.3)
    torch.cuda.synchronize()
    wl, bl = eng.bbox_last["enc"]
    W0, b0, an0, ix0 = wl.clone(), bl.clone(), p0.anchors.clone(), p0.enc_topk.clone()
    r0, un0 = p0.refs[0].clone(), p0.ref_unact.clone()
    for it in range(3):
        torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            for i in idxs:
                fn, a = p1.ops[i]
                fn(*p1.patch_args(fn, a, 0.3), torch.cuda.current_stream().cuda_stream)
        p0._launch(p0.ops[p0.split_at:p0.split_at + 4], s0.cuda_stream, 0.3)
        torch.cuda.synchronize()
        hb = p0.bufs["encbb.1"].t.float().view(-1, 256)
        exp_u = hb @ wl.t() + bl + p0.anchors[p0.enc_topk.flatten().long()]
        cur_u = p0.ref_unact
        print(f"trial {it}: inputs unchanged: W {bool(torch.equal(W0, wl))} b {bool(torch.equal(b0, bl))} anchors {bool(torch.equal(an0, p0.anchors))} idx {bool(torch.equal(ix0, p0.enc_topk))}")
        d_cur = (cur_u - exp_u).abs()
        d_ref = (un0 - exp_u).abs()
        rows = (cur_u != un0).any(1).nonzero().flatten()
        print(f"   rows differing {rows.numel()}; |cur - host| max {float(d_cur.max()):.3e}, |serial - host| max {float(d_ref.max()):.3e}")
        for r in rows[:3].tolist():
            print(f"   row {r}: host {exp_u[r].tolist()} serial {un0[r].tolist()} concurrent {cur_u[r].tolist()}")
        p0._launch(p0.ops[p0.split_at:], s0.cuda_stream, 0.3)
        torch.cuda.synchronize()
