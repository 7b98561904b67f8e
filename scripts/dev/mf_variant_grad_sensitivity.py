"""How well conditioned is a MaskFormer variant's training gradient at random-init weights?  (CPU, oracle only - dev tool, not shipped.)

Runs the fp32 training oracle twice on the same inputs with attention masks, Hungarian matches and point-sampling draws teacher-forced from
the first run: once with fp32 weights, once with the weights rounded to bf16 (what the engine's MFMA kernels consume; activations stay fp32).
Prints the per-parameter gradient relative-L2 between the two - the floor any bf16 engine can be asked to meet on that configuration.

  python scripts/dev/mf_variant_grad_sensitivity.py fai-mf-m-ade [fai-mf-l-coco-ins@50 ...]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from focoos_amd.registry import ModelRegistry  # noqa: E402
from focoos_amd.synth import synth_image_structured, synth_state_dict  # noqa: E402
from oracle import detr_oracle as O  # noqa: E402
from oracle import train_oracle as T  # noqa: E402
from tests.helpers import rel_l2  # noqa: E402
from oracle.mask_criterion_oracle import RandStream  # noqa: E402
from tests.test_gpu_train_bf import _DrawAndRecord  # noqa: E402


def run(name, scale_dec=1.0):
    depth = None
    if "@" in name:
        name, depth = name.split("@")
    cfg = dict(ModelRegistry.get_model_info(name)["config"], criterion_num_points=2048)
    if depth:
        cfg["backbone_config"] = dict(cfg["backbone_config"], depth=int(depth))
    sd = synth_state_dict(cfg, 41, family="fai_mf")
    for k in sd:
        if ".transformer.encoder.layers." in k and k.endswith("self_attn.in_proj_weight"):
            sd[k] = sd[k].clone()
            sd[k][:512] *= 0.05
    damp = float(os.environ.get("DAMP", "1"))
    if damp != 1.0:   # keep the decoder's attention logits O(1) too (they see un-normalised FPN maps when there is no pixel-decoder encoder)
        for k in sd:
            if k.startswith("head.predictor.") and k.endswith("in_proj_weight"):
                sd[k] = sd[k].clone()
                sd[k][:512] *= damp
    imgs = [synth_image_structured(160 + i, 192, 256) for i in range(2)]
    labels, masks = T.synth_mask_targets(7, 2, int(cfg["num_classes"]), (192, 256), counts=(3, 5, 2, 4))
    x = O.get_torch_batch(imgs, None)

    def trainable(k, v):
        if not (v.dtype == torch.float32 and v.dim() > 0) or any(t in k for t in ("running_", "empty_weight")):
            return False
        is_bn = (k.endswith((".norm.weight", ".norm.bias")) and ".transformer." not in k) or k.endswith(
            (".bn.weight", ".bn.bias", ".avd_layer.1.weight", ".avd_layer.1.bias"))
        return not is_bn

    def grads(sd_in, forced=None, fixed=None, rand=None):
        sdg = {k: (v.clone().requires_grad_(True) if trainable(k, v) else v.clone()) for k, v in sd_in.items()}
        col = {}
        outs = T.mf_train_outputs(sdg, cfg, x, collect=col, **({"forced_attn": forced} if forced is not None else {}))
        rs = rand if rand is not None else _DrawAndRecord(78)
        losses, matches = T.bf_criterion(outs, labels, masks, rs, cfg, **({"fixed_matches": fixed} if fixed is not None else {}))
        sum(losses.values()).backward()
        return sdg, col, rs, matches, losses, outs

    a, col, rs, matches, la, oa = grads(sd)
    sdb = {k: (v.bfloat16().float() if v.dtype == torch.float32 and v.dim() >= 2 else v) for k, v in sd.items()}
    b, _, _, _, lb, ob = grads(sdb, forced=col["attn_masks"], fixed=matches, rand=RandStream(rs.rec))
    errs = sorted(((rel_l2(b[k].grad, a[k].grad), k) for k in a if isinstance(a[k], torch.Tensor) and a[k].requires_grad and a[k].grad is not None), reverse=True)
    n = len(errs)
    print(f"{name}: mask logits rel-L2 {rel_l2(ob['pred_masks'].detach(), oa['pred_masks'].detach()):.4f}; worst loss dev "
          f"{max(abs(float(la[k]) - float(lb[k])) / (abs(float(la[k])) + 1e-3) for k in la):.4f}")
    print(f"  {n} tensors; worst {[(round(e, 3), k) for e, k in errs[:4]]}")
    print(f"  quartiles {[round(errs[n * q // 4][0], 4) for q in (1, 2, 3)]}")


if __name__ == "__main__":
    torch.manual_seed(0)
    for nm in sys.argv[1:] or ["fai-mf-m-ade"]:
        run(nm)
