import sys; sys.path.insert(0,"/root/repo")
import numpy as np, torch
from focoos_amd.engine_mf import MfEngine
from focoos_amd.registry import ModelRegistry
from focoos_amd.synth import synth_image_structured, synth_state_dict
from oracle import mf_oracle as M
from oracle.detr_oracle import get_torch_batch
from tests.helpers import rel_l2
DEV="cuda:0"
cfg = ModelRegistry.get_model_info("fai-mf-l-ade")["config"]
for seed in (13, 3):
    sd = synth_state_dict(cfg, seed, family="fai_mf")
    eng = MfEngine(cfg, sd, device=DEV, full_masks=False)
    images = [synth_image_structured(60 + i, 192, 256) for i in range(2)]
    col = {}
    with torch.no_grad():
        probs_o, masks_o = M.mf_forward(sd, cfg, get_torch_batch(images, None), collect=col, upsample=False)
    pl = eng.forward(torch.from_numpy(np.stack(images)).to(DEV), forced_attn=col["attn_masks"])
    torch.cuda.synchronize()
    nchw = lambda nt: nt.torch_view().float().cpu().permute(0, 3, 1, 2)
    print("seed", seed, {n: round(rel_l2(nchw(pl.bufs[n]), col[n]), 4) for n in ("res2", "res5", "msf0", "msf1", "msf2", "fpn_s4", "mask_features")})
    print("   dec", [round(rel_l2(pl.bufs[f"dec{i}.out"].torch_view().float().cpu().reshape(2, -1, 256), col[f"dec{i}_out"]), 4) for i in range(6)],
          "norms", [round(float(col[f"dec{i}_out"].norm() / col[f"dec{i}_out"].numel() ** 0.5), 2) for i in range(6)])
    print("   dprob", float((pl.probs.cpu() - probs_o).abs().max()), "dmask mean", float((pl.mask_probs.cpu() - masks_o).abs().mean()),
          "agree", float(((pl.mask_probs.cpu() >= 0.5) == (masks_o >= 0.5)).float().mean()))
