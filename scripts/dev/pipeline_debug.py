import sys
import numpy as np, torch
from focoos_amd.model import BisenetFormer, FAIMaskFormer
from focoos_amd.registry import ModelRegistry
from focoos_amd.synth import synth_image_structured as sis
fam = sys.argv[1] if len(sys.argv) > 1 else "bf"
if fam == "bf":
    cfg, cls, B, H, W = ModelRegistry.get_model_info("bisenetformer-l-ade")["config"], BisenetFormer, 8, 384, 512
else:
    cfg, cls, B, H, W = ModelRegistry.get_model_info("fai-mf-l-coco-ins")["config"], FAIMaskFormer, 4, 320, 384
eng = cls(cfg, device="cuda:0", seed=0).engine
xs = [torch.from_numpy(np.stack([sis(500 + 20 * j + i, H, W) for i in range(B)])).to("cuda:0") for j in range(2)]
sizes = torch.tensor([[H, W]] * B, dtype=torch.int32, device="cuda:0")
st = eng.stream
keys = ("probs", "mask_probs", "det_count", "det_scores", "det_boxes")
def run(pl, x):
    with torch.cuda.stream(st):
        pl.input.copy_(x); pl.sizes.copy_(sizes); pl.run(st.cuda_stream, 0.3, None, True)
    st.synchronize()
    return {k: getattr(pl, k).clone() for k in keys}, {n: (b.t if hasattr(b, "t") else b).clone() for n, b in pl.bufs.items() if isinstance((b.t if hasattr(b, "t") else b), torch.Tensor)}
A = eng.plan(B, H, W, False, None, 1)
a0, ab0 = run(A, xs[0]); a1, ab1 = run(A, xs[1]); a0b, ab0b = run(A, xs[0])
eng.plans.clear()
Bp = eng.plan(B, H, W, False, None, 1)
b1, bb1 = run(Bp, xs[1])
print("same plan, batch0 first vs third run:", {k: torch.equal(a0[k], a0b[k]) for k in keys})
print("batch1: second run of plan A vs first run of a fresh plan:", {k: torch.equal(a1[k], b1[k]) for k in keys})
bad = [n for n in ab1 if n in bb1 and ab1[n].shape == bb1[n].shape and not torch.equal(ab1[n], bb1[n])]
print("buffers that differ (batch1, used plan vs fresh plan):", bad[:12], len(bad), "of", len(ab1))
