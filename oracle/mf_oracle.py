"""TEST INFRASTRUCTURE — CPU fp32 restatement of the reference's MaskFormer hot path.

The checker, never the product: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.  Functional PyTorch-CPU fp32 code over a ``state_dict`` with the reference's key names, restating
``FAIMaskFormer.forward`` in eval mode and ``MaskFormerProcessor.postprocess`` up to (not including) the cv2/PNG/base64
tail.  Citations are relative to /root/reference.

Parity pinning: the reference's tests hold no golden vector for this path except ``test_masks_to_xyxy``
(tests/utils/test_vision.py:185-205, restated in tests/test_mf_oracle.py); the oracle is pinned against outputs of the
reference itself run in the build container (``scripts/make_golden.py`` -> ``tests/golden/mf_*.npz``;
``tests/test_oracle_vs_reference.py`` compares live when /root/reference is present).  Arithmetic underneath
(conv2d, batch_norm, nn.MultiheadAttention, F.interpolate) is PyTorch's (reference pins torch~=2.7.1, image has 2.10.0).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .detr_oracle import RESNET_BLOCKS, layer_norm, linear, mlp, resnet_vd

SD = Dict[str, torch.Tensor]


def position_embedding_sine_normalized(h: int, w: int, num_pos_feats: int, temperature: float = 10000.0,
                                       scale: float = 2 * math.pi, eps: float = 1e-6) -> torch.Tensor:
    """PositionEmbeddingSine(normalize=True).forward — focoos/nn/layers/position_encoding.py:52-81.
    cumsum WITHOUT the -1 of the DETR variant, normalised by the last row/col, sin/cos interleaved per pair,
    channel order [y(0..npf), x(0..npf)].  Returns [1, h*w, 2*npf] (token-major)."""
    not_mask = torch.ones(1, h, w, dtype=torch.bool)
    y_embed = not_mask.cumsum(1, dtype=torch.float32)
    x_embed = not_mask.cumsum(2, dtype=torch.float32)
    y_embed = y_embed / (y_embed[:, -1:, :] + eps) * scale
    x_embed = x_embed / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / num_pos_feats)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).view(1, h, w, -1)
    pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).view(1, h, w, -1)
    return torch.cat((pos_y, pos_x), dim=3).view(1, h * w, 2 * num_pos_feats)


def mha(sd: SD, prefix: str, q_in, k_in, v_in, nhead: int, attn_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.MultiheadAttention forward (dropout 0), batch-major restatement; ``attn_mask`` bool [B, Lq, Lk], True = key
    not allowed (same mask for all heads — modelling.py:514 repeats it per head)."""
    B, Lq, C = q_in.shape
    Lk = k_in.shape[1]
    W, bias = sd[f"{prefix}.in_proj_weight"], sd[f"{prefix}.in_proj_bias"]
    d = C // nhead
    q = F.linear(q_in, W[:C], bias[:C]).view(B, Lq, nhead, d).transpose(1, 2)
    k = F.linear(k_in, W[C:2 * C], bias[C:2 * C]).view(B, Lk, nhead, d).transpose(1, 2)
    v = F.linear(v_in, W[2 * C:], bias[2 * C:]).view(B, Lk, nhead, d).transpose(1, 2)
    s = (q / math.sqrt(d)) @ k.transpose(-1, -2)
    if attn_mask is not None:
        s = s.masked_fill(attn_mask[:, None], float("-inf"))
    o = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, Lq, C)
    return linear(sd, f"{prefix}.out_proj", o)


def conv2d_norm_act(sd: SD, prefix: str, x: torch.Tensor, padding: int, relu: bool) -> torch.Tensor:
    """focoos.nn.layers.conv.Conv2d.forward — conv.py:33-69 (conv [+bias] -> BN eval if present -> activation)."""
    y = F.conv2d(x, sd[f"{prefix}.weight"], sd.get(f"{prefix}.bias"), padding=padding)
    if f"{prefix}.norm.weight" in sd:
        from .detr_oracle import BN_TRAINING   # set by the training oracle only (model.train(): batch statistics + running-stat update)

        y = F.batch_norm(y, sd[f"{prefix}.norm.running_mean"], sd[f"{prefix}.norm.running_var"],
                         sd[f"{prefix}.norm.weight"], sd[f"{prefix}.norm.bias"], training=BN_TRAINING[0], momentum=0.1 if BN_TRAINING[0] else 0.0, eps=1e-5)
    return F.relu(y) if relu else y


def transformer_fpn(sd: SD, feats: Dict[str, torch.Tensor], cfg: Dict, collect: Optional[dict] = None):
    """TransformerFPN.forward_features — fai_mf/modelling.py:347-369, with TransformerEncoderOnly.forward :177-198 and the
    pre-norm TransformerEncoderLayer.forward (ReLU FFN) — nn/layers/transformer.py:583-601, TransformerEncoder :478-498.
    Returns (mask_features [B,od,H/4,W/4], [stride32, stride16, stride8])."""
    P = "pixel_decoder"
    fd = int(cfg.get("pixel_decoder_feat_dim", 256))
    nhead = int(cfg.get("pixel_decoder_transformer_nheads", 8))
    n_enc = int(cfg.get("pixel_decoder_transformer_layers", 0))
    x = feats["res5"]
    if n_enc > 0:
        x = F.conv2d(x, sd[f"{P}.input_proj.weight"], sd[f"{P}.input_proj.bias"])
        B, C, h, w = x.shape
        pos = position_embedding_sine_normalized(h, w, fd // 2)
        src = x.flatten(2).permute(0, 2, 1)
        for li in range(n_enc):
            p = f"{P}.transformer.encoder.layers.{li}"
            s2 = layer_norm(sd, f"{p}.norm1", src)
            src = src + mha(sd, f"{p}.self_attn", s2 + pos, s2 + pos, s2, nhead)
            s2 = layer_norm(sd, f"{p}.norm2", src)
            src = src + linear(sd, f"{p}.linear2", F.relu(linear(sd, f"{p}.linear1", s2)))
        src = layer_norm(sd, f"{P}.transformer.encoder.norm", src)
        if collect is not None:
            collect["enc_tokens"] = src
        x = src.permute(0, 2, 1).reshape(B, C, h, w)
    y = conv2d_norm_act(sd, f"{P}.layer_4", x, 1, True)
    msf = [y]
    for idx, name in ((3, "res4"), (2, "res3"), (1, "res2")):
        cur = conv2d_norm_act(sd, f"{P}.adapter_{idx}", feats[name], 0, False)
        y = cur + F.interpolate(y, size=cur.shape[-2:], mode="nearest")
        y = conv2d_norm_act(sd, f"{P}.layer_{idx}", y, 1, True)
        if len(msf) < 3:
            msf.append(y)
    mask_features = F.conv2d(y, sd[f"{P}.mask_features.weight"], sd[f"{P}.mask_features.bias"], padding=1)
    if collect is not None:
        collect.update(msf0=msf[0], msf1=msf[1], msf2=msf[2], fpn_s4=y, mask_features=mask_features)
    return mask_features, msf


def prediction_heads(sd: SD, x: torch.Tensor, mask_features: torch.Tensor, size: Optional[Tuple[int, int]]):
    """PredictionHeads.forward — fai_mf/modelling.py:71-113.  x: [B,Q,C].  Returns class logits [B,Q,K+1], mask logits
    [B,Q,H4,W4] and the boolean attention mask [B,Q,h*w] (True = masked: bilinear-resized mask logit < 0)."""
    H = "head.predictor.forward_prediction_heads"
    dec = layer_norm(sd, f"{H}.decoder_norm", x)
    cls = linear(sd, f"{H}.classifier", dec)
    emb = mlp(sd, f"{H}.mask_classifier", dec, 3)
    masks = torch.einsum("bqc,bchw->bqhw", emb, mask_features)
    attn = None
    if size is not None:
        attn = F.interpolate(masks, size=tuple(size), mode="bilinear", align_corners=False).flatten(2) < 0
    return cls, masks, attn


def masked_decoder(sd: SD, msf: List[torch.Tensor], mask_features: torch.Tensor, cfg: Dict,
                   forced_attn: Optional[Sequence[torch.Tensor]] = None, collect: Optional[dict] = None, max_levels: int = 3,
                   all_heads: Optional[list] = None):
    """MultiScaleMaskedTransformerDecoder.forward — fai_mf/modelling.py:453-549 (pre_norm=True, enforce_input_project=True,
    use_attn_masks=True; cross-attn first, then self-attn, then FFN; layers cycle over the 3 levels).
    ``forced_attn``: optional list (one per decoder layer) of boolean masks [B,Q,Lk] to teacher-force the masked attention
    (the discrete step whose flips near logit 0 are the H1-style hazard of this model).
    ``max_levels`` = 2 gives BiSeNetFormer's TransformerDecoder.forward (bisenetformer/modelling.py:375-447), the same
    layer sequence cycling over two levels.  ``all_heads``: list that receives (class logits, mask logits) of every prediction head
    (the learnable queries + one per layer) - the deep-supervision outputs of the training forward."""
    H = "head.predictor"
    nl = int(cfg.get("transformer_predictor_dec_layers", 6))
    hd = int(cfg.get("transformer_predictor_hidden_dim", 256))
    nhead = 8  # fai_mf/modelling.py:689
    nlev = min(max_levels, nl)
    src, pos, sizes = [], [], []
    for i in range(nlev):
        f = msf[i]
        sizes.append((f.shape[2], f.shape[3]))
        pos.append(position_embedding_sine_normalized(f.shape[2], f.shape[3], hd // 2))
        src.append(F.conv2d(f, sd[f"{H}.input_proj.{i}.weight"], sd[f"{H}.input_proj.{i}.bias"]).flatten(2).permute(0, 2, 1))
    B = src[0].shape[0]
    qe = sd[f"{H}.query_embed.weight"].unsqueeze(0).repeat(B, 1, 1)
    out = sd[f"{H}.query_feat.weight"].unsqueeze(0).repeat(B, 1, 1)
    cls, masks, attn = prediction_heads(sd, out, mask_features, sizes[0])
    if all_heads is not None:   # training: every head is supervised (predictions_class / predictions_mask, modelling.py:489-545)
        all_heads.append((cls, masks))
    used_masks = []
    for i in range(nl):
        lvl = i % nlev
        if forced_attn is not None:
            attn = forced_attn[i]
        # a query whose mask is empty attends everywhere (:509-512)
        attn = attn & (attn.sum(-1, keepdim=True) != attn.shape[-1])
        used_masks.append(attn)
        p = f"{H}.transformer_cross_attention_layers.{i}"
        t2 = layer_norm(sd, f"{p}.norm", out)
        out = out + mha(sd, f"{p}.multihead_attn", t2 + qe, src[lvl] + pos[lvl], src[lvl], nhead, attn)
        p = f"{H}.transformer_self_attention_layers.{i}"
        t2 = layer_norm(sd, f"{p}.norm", out)
        out = out + mha(sd, f"{p}.self_attn", t2 + qe, t2 + qe, t2, nhead)
        p = f"{H}.transformer_ffn_layers.{i}"
        t2 = layer_norm(sd, f"{p}.norm", out)
        out = out + linear(sd, f"{p}.linear2", F.relu(linear(sd, f"{p}.linear1", t2)))
        if collect is not None:
            collect[f"dec{i}_out"] = out
        cls, masks, attn = prediction_heads(sd, out, mask_features, sizes[(i + 1) % nlev])
        if all_heads is not None:
            all_heads.append((cls, masks))
    if collect is not None:
        collect["attn_masks"] = used_masks
    return cls, masks


def backbone_features(sd: SD, cfg: Dict, x: torch.Tensor) -> Dict[str, torch.Tensor]:
    """res2..res5 of the configured backbone: ResNet-vd (fai-mf-l-*, focoos/nn/backbone/resnet.py:252-266) or STDC (fai-mf-m-ade,
    focoos/nn/backbone/stdc.py:313-320) - load_backbone dispatches on backbone_config.model_type (focoos/nn/backbone/build.py:4-8)."""
    bb = cfg["backbone_config"]
    if bb.get("model_type", "resnet") == "stdc":
        from . import bf_oracle as BF

        return BF.stdc(sd, "pixel_decoder.backbone", x, tuple(bb.get("layers", (4, 5, 3))))
    return resnet_vd(sd, "pixel_decoder.backbone", x, RESNET_BLOCKS[int(bb.get("depth", 50))])


def mf_forward(sd: SD, cfg: Dict, images: torch.Tensor, forced_attn: Optional[Sequence[torch.Tensor]] = None,
               collect: Optional[dict] = None, upsample: bool = True):
    """FAIMaskFormer.forward (eval) — fai_mf/modelling.py:712-725 + MaskFormerHead.forward :599-617.
    ``images`` [B,3,H,W] float32 on the 0..255 scale.  Returns (class probabilities [B,Q,K] — softmax with the no-object
    column dropped —, mask probabilities [B,Q,H,W] (sigmoid, bilinearly upsampled x4; [B,Q,H/4,W/4] if not ``upsample``))."""
    mean = torch.tensor(cfg.get("pixel_mean", [123.675, 116.28, 103.53]), dtype=torch.float32).view(-1, 1, 1)
    std = torch.tensor(cfg.get("pixel_std", [58.395, 57.12, 57.375]), dtype=torch.float32).view(-1, 1, 1)
    x = (images - mean) / std
    feats = backbone_features(sd, cfg, x)
    if collect is not None:
        collect.update(feats)
    mask_features, msf = transformer_fpn(sd, feats, cfg, collect)
    cls, masks = masked_decoder(sd, msf, mask_features, cfg, forced_attn, collect)
    if collect is not None:
        collect.update(cls_logits=cls, mask_logits=masks)
    if cfg.get("cls_sigmoid", False):
        probs = cls.sigmoid()[..., :-1]
    else:
        probs = F.softmax(cls, dim=-1)[..., :-1]
    mp = masks.sigmoid()
    if upsample:
        mp = F.interpolate(mp, size=images.shape[2:], mode="bilinear", align_corners=False)
    return probs, mp


def masks_to_xyxy(masks: np.ndarray) -> np.ndarray:
    """focoos/utils/vision.py:344-370 — [x_min, y_min, x_max, y_max] (inclusive) per bool mask, zeros when empty."""
    n = masks.shape[0]
    xyxy = np.zeros((n, 4), dtype=int)
    for i, m in enumerate(masks):
        rows, cols = np.any(m, axis=1), np.any(m, axis=0)
        if np.any(rows) and np.any(cols):
            y0, y1 = np.where(rows)[0][[0, -1]]
            x0, x1 = np.where(cols)[0][[0, -1]]
            xyxy[i, :] = [x0, y0, x1, y1]
    return xyxy


def postprocess(probs: torch.Tensor, mask_pred: torch.Tensor, image_sizes: Sequence[Tuple[int, int]],
                mask_threshold: float = 0.5, threshold: float = 0.5, use_mask_score: bool = True, predict_all_pixels: bool = False):
    """MaskFormerProcessor.postprocess — fai_mf/processor.py:168-306 (and the line-for-line identical
    BisenetFormerProcessor.postprocess, bisenetformer/processor.py:176-300), stopping before the cv2 PNG/base64 encoding (:296).  The reference's gather indexing (:237-262) only works for batch 1 (index tensors are
    [1, n]); this restates the batch-1 behaviour and applies it to each image independently.
    Per image returns (scores f32 [n], labels i64 [n], query index i64 [n], boxes int [n,4], masks bool [n,H_img,W_img])."""
    res = []
    for i in range(probs.shape[0]):
        scores, labels = probs[i].max(-1)
        mp = mask_pred[i]
        if predict_all_pixels:   # :215-229 / bisenetformer/processor.py:215-229: every pixel goes to the query maximising score x probability
            winner = (scores.view(-1, 1, 1) * mp).argmax(dim=0)
            binm = winner.unsqueeze(0) == torch.arange(mp.shape[0]).view(-1, 1, 1)
        else:
            binm = mp >= mask_threshold
        keep = (binm.sum(dim=(-2, -1)) > 1).nonzero(as_tuple=True)[0]          # :232 (strictly more than one pixel)
        scores, labels, binm, mp, q = scores[keep], labels[keep], binm[keep], mp[keep], keep
        if use_mask_score:
            bs = binm.int() * 1e-3                                             # :247-250 (float32 after promotion)
            mscore = (bs * mp).sum(-1).sum(-1) / (bs.sum(-1).sum(-1) + 1e-5)
            scores = scores * mscore
            binm_f = bs
        else:
            binm_f = binm
        if threshold > 0:
            f = (scores > threshold).nonzero(as_tuple=True)[0]
            scores, labels, binm_f, q = scores[f], labels[f], binm_f[f], q[f]
        if len(binm_f) == 0:
            res.append((scores, labels, q, np.zeros((0, 4), dtype=int), np.zeros((0,) + tuple(image_sizes[i]), bool)))
            continue
        # NOTE :276-278: the *scaled* (x1e-3) mask is bilinearly resized, then .bool() — any non-zero value is True
        resized = F.interpolate(binm_f.float().unsqueeze(0), size=tuple(image_sizes[i]), mode="bilinear",
                                align_corners=False)[0].bool().numpy()
        res.append((scores, labels, q, masks_to_xyxy(resized), resized))
    return res
