"""Prints the stage-by-stage error of the bf16 engine against the fp32 oracle (teacher-forced query set) - where the final
probability / box error comes from.  GPU only.  Usage: python scripts/dev/parity_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from focoos_amd.model import FAIDetr  # noqa: E402
from focoos_amd.registry import ModelRegistry  # noqa: E402
from focoos_amd.synth import synth_image, synth_image_structured, synth_state_dict  # noqa: E402
from oracle import detr_oracle as O  # noqa: E402
from tests.helpers import load_golden, rel_l2  # noqa: E402

g = load_golden("detr_l_obj365_b2.npz")
cfg = ModelRegistry.get_model_info("fai-detr-l-obj365")["config"]
sd = synth_state_dict(cfg, int(g["seed"]))
model = FAIDetr(cfg, device="cuda:0", seed=int(g["seed"]))
images = [synth_image(0), synth_image_structured(1)]
x = torch.from_numpy(np.stack(images)).to("cuda:0")
forced = torch.from_numpy(g["enc_topk"]).long()
col = {}
with torch.no_grad():
    po, bo = O.detr_forward(sd, cfg, O.get_torch_batch(images, (640, 640)), forced_topk=forced, collect=col)
out = model.forward(x, forced_topk=forced, use_graph=False)
torch.cuda.synchronize()
pl = model.last_plan


def nchw(nt):
    return nt.torch_view().float().cpu().permute(0, 3, 1, 2)


for k in ("res3", "res4", "res5", "enc_s32", "enc_s16", "enc_s8"):
    print(f"{k:10s} rel_l2 {rel_l2(nchw(pl.bufs[k]), col[k]):.5f}")
print(f"memory     rel_l2 {rel_l2(pl.bufs['memory'].t.float().cpu().view(2, -1, 256), col['memory']):.5f}")
se = (pl.enc_scores.cpu() - col["enc_scores"]).abs()
print(f"enc_scores max|d| {se.max():.4f} mean|d| {se.mean():.4f}  oracle std {col['enc_scores'].std():.3f}")
print(f"target     rel_l2 {rel_l2(pl.bufs['target'].t.float().cpu().view(2, 300, 256), col['target']):.5f}")
for i in range(6):
    e = rel_l2(pl.bufs[f"dec{i}.out"].t.float().cpu().view(2, 300, 256), col[f"dec{i}_out"])
    r = (pl.refs[i + 1].cpu().view(2, 300, 4) - col[f"dec{i}_ref"]).abs().max()
    print(f"dec{i}.out   rel_l2 {e:.5f}   ref max|d| {r:.5f}")
dp = (out.logits.cpu() - po).abs()
print(f"probs max|d| {dp.max():.5f} mean|d| {dp.mean():.6f}; boxes max|d| {(out.boxes.cpu() - bo).abs().max():.5f}")
lo = torch.logit(po.clamp(1e-6, 1 - 1e-6))
le = torch.logit(out.logits.cpu().clamp(1e-6, 1 - 1e-6))
print(f"logit err max {((le - lo).abs() * (po > 0.05)).max():.4f}  logits std {lo.std():.3f}")
