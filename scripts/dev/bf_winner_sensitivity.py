"""Which rounding moves BiSeNetFormer's per-pixel winner map ("carry mask_embed and the 1/8-res logits of the final
head in fp32").  CPU, fp32 oracle: the final head's mask logits are recomputed with (a) the mask embedding rounded to bf16, (b) the mask
features rounded to bf16, (c) both - what fp32 operands in fx_query_pixel_logits would remove - and (d) both inputs of the head carrying
the relative error the bf16 network arrives with (stage rel-L2 ~1 %).  Winner = argmax_q score_q * sigmoid(mask logit) at 1/8 resolution.
usage: python scripts/dev/bf_winner_sensitivity.py"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from focoos_amd.registry import ModelRegistry  # noqa: E402
from focoos_amd.synth import synth_image_structured as sis  # noqa: E402
from focoos_amd.synth import synth_state_dict  # noqa: E402
from oracle import bf_oracle as B  # noqa: E402
from oracle import mf_oracle as M  # noqa: E402

torch.set_num_threads(16)
cfg = ModelRegistry.get_model_info("bisenetformer-l-ade")["config"]
sd = synth_state_dict(cfg, 0, family="bisenetformer")
img = torch.from_numpy(np.stack([sis(200 + i, 384, 512) for i in range(2)])).permute(0, 3, 1, 2).float()
col = {}
H = "head.predictor.forward_prediction_heads"
with torch.no_grad():
    probs, _ = B.bf_forward(sd, cfg, img, None, col, upsample=False)
    x, mf = col["dec5_out"], col["mask_features"]
    score = probs.max(-1).values                                  # [B,Q]

    def winner(xx, ff, round_emb=False, round_feat=False):
        dec = M.layer_norm(sd, f"{H}.decoder_norm", xx)
        emb = M.mlp(sd, f"{H}.mask_classifier", dec, 3)
        if round_emb:
            emb = emb.bfloat16().float()
        if round_feat:
            ff = ff.bfloat16().float()
        logit = torch.einsum("bqc,bchw->bqhw", emb, ff)
        prod = score[:, :, None, None] * logit.sigmoid()
        top2 = prod.topk(2, dim=1).values
        return prod.argmax(1), (top2[:, 0] - top2[:, 1]) / top2[:, 0].clamp_min(1e-12)

    w0, margin = winner(x, mf)
    print(f"pixels with a top-2 margin below 10 %: {100 * (margin < 0.1).float().mean():.1f} %")

    def report(name, w):
        agree = (w == w0).float()
        print(f"{name:62s} agreement {100 * agree.mean():.2f} % of all pixels, {100 * agree[margin >= 0.1].mean():.2f} % where the margin is >= 10 %")

    report("(a) mask embedding rounded to bf16", winner(x, mf, True, False)[0])
    report("(b) mask features rounded to bf16", winner(x, mf, False, True)[0])
    report("(c) both rounded (what fp32 operands would remove)", winner(x, mf, True, True)[0])
    g = torch.Generator().manual_seed(1)
    for rel in (0.005, 0.01, 0.02):
        nx, nf = torch.randn(x.shape, generator=g), torch.randn(mf.shape, generator=g)
        xp, fp = x + nx * (rel * x.norm() / nx.norm()), mf + nf * (rel * mf.norm() / nf.norm())
        report(f"(d) decoder state and mask features + {100 * rel:.1f} % relative error", winner(xp, fp)[0])
