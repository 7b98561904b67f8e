"""Find the first buffer (in launch order) that differs between a concurrent 2-part replay and the single-part reference."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from focoos_amd.model import FAIDetr
from focoos_amd.registry import ModelRegistry
from focoos_amd.synth import synth_image_structured as sis
cfg = ModelRegistry.get_model_info("fai-detr-l-obj365")["config"]
model = FAIDetr(cfg, device="cuda:0", seed=0)
eng = model.engine
B = 32
imgs = torch.from_numpy(np.stack([sis(100 + i) for i in range(B)])).to("cuda:0")
def run(x, ns):
    pl = eng.plan(B, 640, 640, False, ns)
    with torch.cuda.stream(eng.stream):
        pl.input.copy_(x); pl.sizes.copy_(torch.tensor([[640, 640]] * B, dtype=torch.int32))
        pl.run(eng.stream.cuda_stream, 0.3, None, True)
    eng.stream.synchronize()
    return pl
ref = run(imgs, 1)
names = list(ref.bufs.keys())
refb = {n: ref.bufs[n].t.clone() for n in names}
ref_probs = ref.probs.clone()
hb = B // 2
for it in range(40):
    pl = run(imgs, 2)
    if torch.equal(pl.probs, ref_probs):
        continue
    print("mismatch at replay", it)
    for n in names:
        r = ref.bufs[n]
        full = refb[n]
        per = full.numel() // B if r.B == B else None
        for pi, p in enumerate(pl.parts):
            if n not in p.bufs: continue
            t = p.bufs[n].t
            if r.B == B:        # NHWC buffer with batch leading
                seg = full[pi * hb * per:(pi + 1) * hb * per]
            elif r.B % B == 0:  # rows = B * something
                rows_per = full.numel() // B
                seg = full[pi * hb * rows_per:(pi + 1) * hb * rows_per]
            else:
                continue
            if seg.numel() != t.numel():
                continue
            neq = (seg.view(torch.int16) != t.view(torch.int16)) if t.dtype == torch.bfloat16 else (seg != t)
            if neq.any():
                idx = neq.nonzero().flatten()
                print(f"  diff buffer: {n} part {pi}: {int(neq.sum())} of {t.numel()} elements, first flat idx {int(idx[0])}, last {int(idx[-1])}")
    for k in ("enc_topk", "enc_scores", "enc_topk_val"):
        print("  ", k, bool(torch.equal(getattr(pl, k), getattr(ref, k))))
    for pi, p in enumerate(pl.parts):
        for li, (a, b) in enumerate(zip(p.refs, ref.refs)):
            seg = b[pi * hb * 300:(pi + 1) * hb * 300]
            if not torch.equal(a, seg):
                print(f"   refs[{li}] part {pi} differs: rows", (a != seg).any(1).nonzero().flatten()[:5].tolist())
        seg = ref.ref_unact[pi * hb * 300:(pi + 1) * hb * 300]
        print("   ref_unact equal", pi, bool(torch.equal(p.ref_unact, seg)))
    break
else:
    print("no mismatch in 40 replays")
