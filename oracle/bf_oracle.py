"""TEST INFRASTRUCTURE — CPU fp32 restatement of the BiSeNetFormer inference path (bisenetformer-l-ade).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path (focoos_amd/)
never does.  Each function cites the reference lines it restates (paths relative to /root/reference):
  * STDC backbone, CatBottleneck                 focoos/nn/backbone/stdc.py:15-31, 108-166, 282-320
  * AttentionRefinementModule / ContextPath      focoos/models/bisenetformer/modelling.py:149-212
  * FeatureFusionModule / BiseNet                focoos/models/bisenetformer/modelling.py:215-279
  * TransformerDecoder (two levels) + heads      :375-447 / :68-113 (restated once in oracle/mf_oracle.masked_decoder)
  * MaskFormerHead.forward tail, BisenetFormer.forward   :487-510, :594-609
  * BisenetFormerProcessor.postprocess           focoos/models/bisenetformer/processor.py:176-300 (oracle/mf_oracle.postprocess)
Pinned against the real reference (imported from /root/reference where present: tests/test_oracle_vs_reference.py) and against
the committed golden vectors it produced (tests/golden/bf_l_ade_b2.npz, scripts/make_golden.py).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import mf_oracle as M

SD = Dict[str, torch.Tensor]


def conv_x(sd: SD, prefix: str, x: torch.Tensor, stride: int = 1, relu: bool = True) -> torch.Tensor:
    """ConvX / ConvBNReLU: conv (no bias, padding k//2) + BatchNorm2d (eval) + ReLU — stdc.py:15-31, modelling.py:125-146."""
    w = sd[f"{prefix}.conv.weight"]
    y = F.conv2d(x, w, None, stride=stride, padding=w.shape[-1] // 2)
    y = _bn(sd, f"{prefix}.bn", y)
    return F.relu(y) if relu else y


def _bn(sd: SD, prefix: str, y: torch.Tensor) -> torch.Tensor:
    """nn.BatchNorm2d: running statistics, or (detr_oracle.BN_TRAINING set by the training oracle: model.train()) batch statistics
    with the in-place running-statistics update."""
    from .detr_oracle import BN_TRAINING

    return F.batch_norm(y, sd[f"{prefix}.running_mean"], sd[f"{prefix}.running_var"], sd[f"{prefix}.weight"], sd[f"{prefix}.bias"], BN_TRAINING[0],
                        0.1 if BN_TRAINING[0] else 0.0, 1e-5)


def cat_bottleneck(sd: SD, prefix: str, x: torch.Tensor, stride: int) -> torch.Tensor:
    """CatBottleneck.forward (block_num 4) — stdc.py:146-166: [skip(out1) | conv1 | conv2 | conv3] concatenated; for stride 2 the
    first branch goes through a depthwise 3x3 stride-2 conv + BN and the skip through AvgPool2d(3, 2, 1)."""
    out1 = conv_x(sd, f"{prefix}.conv_list.0", x)
    cur = out1
    if stride == 2:
        w = sd[f"{prefix}.avd_layer.0.weight"]
        cur = _bn(sd, f"{prefix}.avd_layer.1", F.conv2d(out1, w, None, stride=2, padding=1, groups=w.shape[0]))
        out1 = F.avg_pool2d(out1, kernel_size=3, stride=2, padding=1)
    outs = [out1]
    for j in (1, 2, 3):
        cur = conv_x(sd, f"{prefix}.conv_list.{j}", cur)
        outs.append(cur)
    return torch.cat(outs, dim=1)


def stdc(sd: SD, prefix: str, x: torch.Tensor, layers: Sequence[int] = (4, 5, 3)) -> Dict[str, torch.Tensor]:
    """STDC.forward — stdc.py:313-320 with _make_layers :282-311 (outputs after features 1 and the last block of each stage)."""
    x = conv_x(sd, f"{prefix}.features.0", x, 2)
    x = conv_x(sd, f"{prefix}.features.1", x, 2)
    outs = {"res2": x}
    idx = 2
    for i, n in enumerate(layers):
        for j in range(n):
            x = cat_bottleneck(sd, f"{prefix}.features.{idx}", x, 2 if j == 0 else 1)
            idx += 1
        outs[f"res{i + 3}"] = x
    return outs


def attention_refinement(sd: SD, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """AttentionRefinementModule.forward — modelling.py:159-167."""
    feat = conv_x(sd, f"{prefix}.conv", F.conv2d(x, sd[f"{prefix}.proj.weight"]))
    att = feat.mean(dim=(2, 3), keepdim=True)
    att = torch.sigmoid(_bn(sd, f"{prefix}.bn_atten", F.conv2d(att, sd[f"{prefix}.conv_atten.weight"])))
    return feat * att


def context_path(sd: SD, prefix: str, feat8: torch.Tensor, feat16: torch.Tensor, feat32: torch.Tensor):
    """ContextPath.forward (out4=False) — modelling.py:186-212.  Returns (x8, x16, x32)."""
    avg = conv_x(sd, f"{prefix}.conv_avg", feat32.mean(dim=(2, 3), keepdim=True))
    f32_sum = attention_refinement(sd, f"{prefix}.arm32", feat32) + avg
    f32_up = conv_x(sd, f"{prefix}.conv_head32", F.interpolate(f32_sum, size=feat16.shape[-2:], mode="bilinear"))
    f16_sum = attention_refinement(sd, f"{prefix}.arm16", feat16) + f32_up
    f16_up = conv_x(sd, f"{prefix}.conv_head16", F.interpolate(f16_sum, size=feat8.shape[-2:], mode="bilinear"))
    return f16_up, f16_sum, f32_sum


def feature_fusion(sd: SD, prefix: str, fsp: torch.Tensor, fcp: torch.Tensor) -> torch.Tensor:
    """FeatureFusionModule.forward — modelling.py:226-237."""
    s = F.conv2d(fsp, sd[f"{prefix}.proj1.weight"], sd[f"{prefix}.proj1.bias"]) + F.conv2d(fcp, sd[f"{prefix}.proj2.weight"], sd[f"{prefix}.proj2.bias"])
    feat = conv_x(sd, f"{prefix}.convblk", s)
    att = F.adaptive_avg_pool2d(feat, 1)
    att = torch.sigmoid(F.conv2d(F.relu(F.conv2d(att, sd[f"{prefix}.conv1.weight"])), sd[f"{prefix}.conv2.weight"]))
    return feat * att + feat


def bisenet(sd: SD, feats: Dict[str, torch.Tensor], collect: Optional[dict] = None):
    """BiseNet.forward_features — modelling.py:272-279.  Returns (mask_features [B,out_dim,H/8,W/8], (cp32, cp16, cp8))."""
    P = "pixel_decoder"
    cp8, cp16, cp32 = context_path(sd, f"{P}.cp", feats["res3"], feats["res4"], feats["res5"])
    fuse = feature_fusion(sd, f"{P}.ffm", feats["res3"], cp8)
    out = conv_x(sd, f"{P}.conv_out", fuse)
    if collect is not None:
        collect.update(cp8=cp8, cp16=cp16, cp32=cp32, ffm=fuse, mask_features=out)
    return out, (cp32, cp16, cp8)


def bf_forward(sd: SD, cfg: Dict, images: torch.Tensor, forced_attn: Optional[Sequence[torch.Tensor]] = None,
               collect: Optional[dict] = None, upsample: bool = True):
    """BisenetFormer.forward (eval) — modelling.py:594-609 + MaskFormerHead.forward :487-510.  ``images`` [B,3,H,W] float32 on the
    0..255 scale.  Returns (class probabilities [B,Q,K], mask probabilities [B,Q,H,W] - sigmoid at 1/8 resolution, bilinearly
    upsampled to the image size; [B,Q,H/8,W/8] if not ``upsample``)."""
    mean = torch.tensor(cfg.get("pixel_mean", [123.675, 116.28, 103.53]), dtype=torch.float32).view(-1, 1, 1)
    std = torch.tensor(cfg.get("pixel_std", [58.395, 57.12, 57.375]), dtype=torch.float32).view(-1, 1, 1)
    x = (images - mean) / std
    feats = stdc(sd, "pixel_decoder.backbone", x, tuple(cfg["backbone_config"].get("layers", (4, 5, 3))))
    if collect is not None:
        collect.update(feats)
    mask_features, msf = bisenet(sd, feats, collect)
    cls, masks = M.masked_decoder(sd, list(msf[:-1]), mask_features, cfg, forced_attn, collect, max_levels=2)   # F1, F2 only (:383)
    if collect is not None:
        collect.update(cls_logits=cls, mask_logits=masks)
    probs = cls.sigmoid()[..., :-1] if cfg.get("cls_sigmoid", False) else F.softmax(cls, dim=-1)[..., :-1]
    mp = masks.sigmoid()
    if upsample:
        mp = F.interpolate(mp, size=images.shape[2:], mode="bilinear", align_corners=False)
    return probs, mp


def postprocess(probs, mask_pred, image_sizes, cfg: Dict, threshold: Optional[float] = None):
    """BisenetFormerProcessor.postprocess with the config's switches (predict_all_pixels, use_mask_score, thresholds)."""
    return M.postprocess(probs, mask_pred, image_sizes, mask_threshold=float(cfg.get("mask_threshold", 0.5)),
                         threshold=float(cfg.get("threshold", 0.5)) if threshold is None else threshold,
                         use_mask_score=bool(cfg.get("use_mask_score", False)), predict_all_pixels=bool(cfg.get("predict_all_pixels", False)))
