"""Print per-stage errors of the MaskFormer engine vs the oracle (teacher-forced and free-running)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from focoos_amd.engine_mf import MfEngine
from focoos_amd.registry import ModelRegistry
from focoos_amd.synth import synth_image_structured, synth_state_dict
from oracle import mf_oracle as M
from oracle.detr_oracle import get_torch_batch
from tests.helpers import load_golden, rel_l2

g = load_golden("mf_l_coco_ins_b2.npz")
cfg = ModelRegistry.get_model_info("fai-mf-l-coco-ins")["config"]
sd = synth_state_dict(cfg, int(g["seed"]), family="fai_mf")
eng = MfEngine(cfg, sd, device="cuda:0", full_masks=True)
h, w = (int(v) for v in g["hw"])
images = [synth_image_structured(i, h, w) for i in range(2)]
forced = [torch.from_numpy(np.unpackbits(g[f"attn_mask{i}"], axis=-1)[..., : int(g[f"attn_mask{i}_len"])].astype(bool)) for i in range(9)]
col = {}
with torch.no_grad():
    x = get_torch_batch(images, None)
    probs_o, masks_o = M.mf_forward(sd, cfg, x, forced_attn=forced, collect=col)
x_u8 = torch.from_numpy(np.stack(images)).to("cuda:0")
pl = eng.forward(x_u8, forced_attn=forced)
torch.cuda.synchronize()
nchw = lambda nt: nt.torch_view().float().cpu().permute(0, 3, 1, 2)
for name in ("res2", "res3", "res4", "res5", "msf0", "msf1", "msf2", "fpn_s4", "mask_features"):
    print(name, rel_l2(nchw(pl.bufs[name]), col[name]))
B, L, Cc = col["enc_tokens"].shape
print("enc_tokens", rel_l2(pl.bufs["enc_tokens"].torch_view().float().cpu().reshape(B, L, Cc), col["enc_tokens"]))
for i in range(9):
    got = pl.bufs[f"dec{i}.out"].torch_view().float().cpu().reshape(B, -1, 256)
    print(f"dec{i}", rel_l2(got, col[f"dec{i}_out"]), float(col[f"dec{i}_out"].abs().mean()))
print("probs maxabs", float((pl.probs.cpu() - probs_o).abs().max()))
lo_o = torch.sigmoid(col["mask_logits"])
d = (pl.mask_probs.cpu() - lo_o).abs()
print("mask lo mean", float(d.mean()), "agree", float(((pl.mask_probs.cpu() >= 0.5) == (lo_o >= 0.5)).float().mean()))
print("mask full mean", float((pl.masks.cpu() - masks_o).abs().mean()))
print("det counts", pl.det_count.tolist(), [len(g[f"det{i}_conf"]) for i in range(2)])
