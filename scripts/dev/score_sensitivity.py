"""Where the encoder-score error of the bf16 engine comes from ("keep `memory` in fp32 for the score path, then TOL_SCORE
<= 2e-2").  CPU only, fp32 oracle: the scores of the top-300 selection are recomputed (a) from `memory` rounded to bf16 - the only rounding an
fp32 side copy of the level projections would remove - and (b) from `memory` with a relative perturbation of the size the bf16 backbone +
hybrid encoder deliver (stage rel-L2 0.5-0.9 %, profiles/r03_parity_probe.txt).   usage: python scripts/dev/score_sensitivity.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from focoos_amd.registry import ModelRegistry  # noqa: E402
from focoos_amd.synth import synth_image_structured as sis  # noqa: E402
from focoos_amd.synth import synth_state_dict  # noqa: E402
from oracle import detr_oracle as O  # noqa: E402

torch.set_num_threads(16)
cfg = ModelRegistry.get_model_info("fai-detr-l-obj365")["config"]
sd = synth_state_dict(cfg, 0, family="fai_detr")
img = torch.from_numpy(np.stack([sis(100, 320, 320)])).permute(0, 3, 1, 2).float()
col = {}
P = "head.predictor"
with torch.no_grad():
    O.detr_forward(sd, cfg, img, None, col)
    mem, sc = col["memory"], col["enc_scores"]

    def scores(m):
        _, valid = O.generate_anchors([[40, 40], [20, 20], [10, 10]])
        om = O.layer_norm(sd, f"{P}.enc_output.1", O.linear(sd, f"{P}.enc_output.0", valid.to(m.dtype) * m))
        return O.linear(sd, f"{P}.enc_score_classifier", om).max(-1).values

    s0 = scores(mem)
    assert (s0 - sc).abs().max() == 0
    print(f"score std {sc.std():.3f}; memory std {mem.std():.1f}")
    s1 = scores(mem.bfloat16().float())
    print(f"(a) memory rounded to bf16, fp32 head:           max |dscore| {(s1 - s0).abs().max():.4f}  rms {(s1 - s0).pow(2).mean().sqrt():.4f}")
    for rel in (0.003, 0.005, 0.007, 0.009):
        noise = torch.randn(mem.shape, generator=torch.Generator().manual_seed(1))
        s2 = scores(mem + noise * (rel * mem.norm() / noise.norm()))
        print(f"(b) memory + {100 * rel:.1f} % relative perturbation:       max |dscore| {(s2 - s0).abs().max():.4f}  rms {(s2 - s0).pow(2).mean().sqrt():.4f}")
    v, _ = torch.sort(sc[0], descending=True)
    print(f"300th score {v[299]:.4f}; tokens within +-0.02 of the cut: {int(((sc[0] - v[299]).abs() < 0.02).sum())}, within +-0.085: {int(((sc[0] - v[299]).abs() < 0.085).sum())}; "
          f"gap 300th-301st {v[299] - v[300]:.6f}")
