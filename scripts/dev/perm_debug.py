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
perm = torch.randperm(B, generator=torch.Generator().manual_seed(0)).to("cuda:0")
def run(x, ns, graph=True):
    pl = eng.plan(B, 640, 640, False, ns)
    with torch.cuda.stream(eng.stream):
        pl.input.copy_(x); pl.sizes.copy_(torch.tensor([[640, 640]] * B, dtype=torch.int32))
        pl.run(eng.stream.cuda_stream, 0.3, None, graph)
    eng.stream.synchronize()
    return pl
for ns in (1, 2):
    for graph in (False, True):
        pl = run(imgs, ns, graph); a = {k: getattr(pl, k).clone() for k in ("probs", "boxes", "enc_topk", "enc_scores")}
        feats = {}
        parts = pl.parts if ns > 1 else [pl]
        for name in ("res3", "res5", "enc_s8", "memory", "output_memory", "dec0.out", "dec5.out"):
            feats[name] = torch.cat([p.bufs[name].torch_view().float().reshape(p.B, -1).clone() for p in parts], 0)
        pl = run(imgs, ns, graph); a2 = {k: getattr(pl, k).clone() for k in a}
        print(f"ns={ns} graph={graph} idempotent:", {k: bool(torch.equal(a[k], a2[k])) for k in a})
        pl = run(imgs[perm].contiguous(), ns, graph); b = {k: getattr(pl, k).clone() for k in a}
        parts = pl.parts if ns > 1 else [pl]
        print("   perm-equal:", {k: bool(torch.equal(a[k][perm], b[k])) for k in a})
        for name in feats:
            fb = torch.cat([p.bufs[name].torch_view().float().reshape(p.B, -1) for p in parts], 0)
            d = (feats[name][perm] - fb).abs().max().item()
            print("     ", name, d)
        nd = (a["probs"][perm] != b["probs"]).sum().item(); print("   differing prob elements", nd, "max", (a["probs"][perm] - b["probs"]).abs().max().item())
        bad = ((a["probs"][perm] != b["probs"]).flatten(1).any(1)).nonzero().flatten().tolist(); print("   images differing", bad)
