"""Concurrent batch parts vs the SAME two parts run one after the other (same kernels, same buffers): any difference is a
cross-queue hazard, not a numerics difference between batch sizes.  GPU only."""
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
B, N = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 60
imgs = torch.from_numpy(np.stack([sis(100 + i) for i in range(B)])).to("cuda:0")
sizes = torch.tensor([[640, 640]] * B, dtype=torch.int32)
pl = eng.plan(B, 640, 640, False, 2)
st = eng.stream
keys = ("probs", "boxes", "enc_topk", "enc_scores", "det_scores", "det_labels", "det_boxes", "det_count")


def snapshot():
    out = {k: getattr(pl, k).clone() for k in keys}
    for pi, p in enumerate(pl.parts):
        for n, nt in p.bufs.items():
            out[f"{pi}:{n}"] = nt.t.clone()
    return out


with torch.cuda.stream(st):
    pl.input.copy_(imgs)
    pl.sizes.copy_(sizes)
    for p in pl.parts:              # serial reference: the two parts back to back on ONE stream, eager launches
        p._launch(p.ops, st.cuda_stream, 0.3)
st.synchronize()
ref = snapshot()
bad = 0
first = None
for it in range(N):
    with torch.cuda.stream(st):
        pl.run(st.cuda_stream, 0.3, None, True)   # concurrent: one graph per part, two streams
    st.synchronize()
    cur = snapshot()
    diff = [k for k in ref if not torch.equal(ref[k].view(torch.uint8) if ref[k].dtype != torch.bool else ref[k], cur[k].view(torch.uint8) if cur[k].dtype != torch.bool else cur[k])]
    if diff:
        bad += 1
        if first is None:
            first = (it, diff[:12])
print(f"two-part concurrent vs serial: {bad} of {N} replays differ", first or "")
