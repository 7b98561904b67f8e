"""with live BatchNorm the engine's backbone gradients deviate 32-41 % (median rel-L2) from the fp32 oracle.
Which rounding is responsible?  This CPU experiment runs the fp32 TRAINING ORACLE itself (oracle/train_oracle.py, batch statistics) and
injects bf16 rounding at chosen points of every conv-norm layer - forward value AND the gradient flowing back through the same point, the
way a bf16-storage engine rounds them - then compares every parameter gradient with the pure-fp32 run (discrete choices teacher-forced):
  z   : the conv output before BatchNorm          (the engine stores it in bf16: train_nn.ConvNormLayer, fx_bn_stats_bf16 reads it back)
  y   : the layer output after BatchNorm + activation
  w   : conv weights rounded to bf16 (the MFMA operand)
  zy  : both activations, zyw: everything (what the engine does), y_w: an engine with an fp32 `z`
No GPU, no product code: test infrastructure only.  usage: python scripts/dev/bn_grad_sensitivity.py [modes...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import torch.nn.functional as F

from focoos_amd.registry import ModelRegistry
from focoos_amd.synth import synth_image_structured, synth_state_dict
from oracle import detr_oracle as O
from oracle import train_oracle as T


class _RoundBoth(torch.autograd.Function):
    """bf16 rounding of the value in the forward AND of the gradient in the backward (a bf16 tensor in HBM in both directions)."""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


RDT = {"bf16": torch.bfloat16, "fp16": torch.float16}[os.environ.get("SENS_DTYPE", "bf16")]   # fp16: what the reference's own AMP autocast stores


class _RoundBoth(torch.autograd.Function):  # noqa: F811  (dtype-parametrised form of the class above)
    @staticmethod
    def forward(ctx, x):
        return x.to(RDT).float()

    @staticmethod
    def backward(ctx, g):
        return g.to(RDT).float()


rb = _RoundBoth.apply
MODE = {"z": False, "y": False, "w": False}


def conv_bn_rounded(sd, prefix, x, stride=1, act=None, conv="conv", norm="norm", padding=None):
    w = sd[f"{prefix}.{conv}.weight"]
    if MODE["w"]:
        w = w + (w.to(RDT).float() - w).detach()      # bf16 operand, fp32 master: the gradient reaches the master unrounded
    k = w.shape[-1]
    pad = (k - 1) // 2 if padding is None else padding
    z = F.conv2d(x, w, None, stride=stride, padding=pad)
    if MODE["z"]:
        z = rb(z)
    y = O.apply_act(O.batch_norm(sd, f"{prefix}.{norm}", z), act)
    if MODE["y"]:
        y = rb(y)
    return y


def run(mode_name, sd, cfg, x, labels, boxes, forced=None):
    for k in MODE:
        MODE[k] = k in mode_name.split("_")[0] if "_" not in mode_name else k in mode_name.replace("_", "")
    sdg = {k: (v.clone().requires_grad_(True) if (v.dtype == torch.float32 and v.dim() > 0 and not any(t in k for t in ("running_", "empty_weight", "mask_features"))) else v.clone())
           for k, v in sd.items()}
    orig = O.conv_bn
    O.conv_bn = conv_bn_rounded if mode_name != "fp32" else orig
    O.BN_TRAINING[0] = True
    try:
        outs = T.detr_train_outputs(sdg, cfg, x, forced_topk=None if forced is None else forced[0])
    finally:
        O.BN_TRAINING[0] = False
        O.conv_bn = orig
    losses, matches = T.criterion(outs, labels, boxes, fixed_matches=None if forced is None else forced[1])
    sum(losses.values()).backward()
    grads = {k: v.grad.clone() for k, v in sdg.items() if isinstance(v, torch.Tensor) and v.requires_grad and v.grad is not None}
    return grads, (outs["topk_ind"], matches), {k: float(v) for k, v in losses.items()}


def group(name):
    if name.startswith("pixel_decoder.backbone.conv1"):
        return "stem"
    for i in range(4):
        if name.startswith(f"pixel_decoder.backbone.res_layers.{i}."):
            return f"res{i + 2}"
    if name.startswith("pixel_decoder."):
        return "encoder"
    if ".decoder.layers." in name:
        return "decoder"
    return "head-other"


def main():
    torch.manual_seed(0)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    cfg = ModelRegistry.get_model_info("fai-detr-l-coco")["config"]
    sd = synth_state_dict(cfg, 21)
    k_qk = "pixel_decoder.encoder.0.layers.0.self_attn.in_proj_weight"   # as tests/test_gpu_train_detr.py: keep the AIFI logits O(1)
    sd[k_qk] = sd[k_qk].clone()
    sd[k_qk][:512] *= 0.05
    nimg, (ih, iw) = 4, (160, 192)
    imgs = [synth_image_structured(80 + i, ih, iw) for i in range(nimg)]
    labels, boxes = T.synth_targets(2, nimg, 80, counts=(4, 6, 2, 5))
    x = O.get_torch_batch(imgs, None)
    ref, forced, ref_losses = run("fp32", sd, cfg, x, labels, boxes)
    modes = sys.argv[1:] or ["z", "y", "w", "zy", "y_w", "zyw"]
    print(f"rounding dtype {os.environ.get('SENS_DTYPE', 'bf16')}; model fai-detr-l-coco seed 21, {nimg} images {ih}x{iw}, batch-statistics BatchNorm; parameter-gradient rel-L2 vs the fp32 run: median per group")
    for m in modes:
        g, _, losses = run(m, sd, cfg, x, labels, boxes, forced)
        rows = {}
        floor = 1e-3 * float(np.median([float(v.norm()) for v in ref.values()]))
        for k, v in ref.items():
            if float(v.norm()) < floor:
                continue
            e = float((g[k] - v).norm() / v.norm())
            rows.setdefault(group(k), []).append(e)
        allv = sorted(e for v in rows.values() for e in v)
        dl = max(abs(losses[k] - ref_losses[k]) / (abs(ref_losses[k]) + 1e-3) for k in ref_losses)
        print(f"rounding {m:5s}: " + "  ".join(f"{gk} {np.median(rows[gk]):.3f}" for gk in ("stem", "res2", "res3", "res4", "res5", "encoder", "decoder", "head-other") if gk in rows)
              + f"  | all: median {allv[len(allv) // 2]:.3f} p90 {allv[int(0.9 * len(allv))]:.3f}  max loss dev {dl:.4f}", flush=True)


if __name__ == "__main__":
    main()
