"""TEST INFRASTRUCTURE — CPU fp32 restatement of the reference's RT-DETR hot path.

This file is the *checker*, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product path (``focoos_amd``) never routes through it.

It restates, as plain functional PyTorch-CPU fp32 code operating on a
``state_dict`` with the reference's key names, what the reference computes for
``FAIDetr.forward`` in eval mode and ``DETRProcessor.postprocess``.  Each function
cites the reference file:line it follows (paths relative to /root/reference).

Parity pinning: the reference's own tests hold no golden vector for this path
(SURVEY §0.6, §8c), so the oracle is pinned against *outputs of the reference
itself run in the build container* (``oracle/ref_import.py`` +
``scripts/make_golden.py`` -> ``tests/golden/*.npz``; ``tests/test_oracle_vs_reference.py``
compares live when /root/reference is present, ``tests/test_oracle_golden.py``
compares against the committed fixtures anywhere).  The arithmetic underneath
(conv2d, batch_norm, grid_sample, topk ...) is PyTorch's, a third-party dependency
of the reference pinned at torch~=2.7.1 (pyproject.toml:55; this image has 2.10.0).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
RESNET_BLOCKS = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3]}  # focoos/nn/backbone/resnet.py:17-23


# ----------------------------------------------------------------------------- building blocks
# nn.BatchNorm2d's mode: False = running statistics (eval / freeze_bn), True = batch statistics with the in-place
# running-statistics update (model.train(); used by the training oracle only - see oracle/train_oracle.py).
BN_TRAINING = [False]


def batch_norm(sd: SD, prefix: str, y: torch.Tensor) -> torch.Tensor:
    return F.batch_norm(y, sd[f"{prefix}.running_mean"], sd[f"{prefix}.running_var"], sd[f"{prefix}.weight"], sd[f"{prefix}.bias"],
                        training=BN_TRAINING[0], momentum=0.1, eps=1e-5)


def conv_bn(sd: SD, prefix: str, x: torch.Tensor, stride: int = 1, act: Optional[str] = None,
            conv: str = "conv", norm: str = "norm", padding: Optional[int] = None) -> torch.Tensor:
    """ConvNormLayer.forward — focoos/nn/layers/conv.py:78-98 (conv, BN eval, act)."""
    w = sd[f"{prefix}.{conv}.weight"]
    k = w.shape[-1]
    pad = (k - 1) // 2 if padding is None else padding
    y = F.conv2d(x, w, None, stride=stride, padding=pad)
    y = batch_norm(sd, f"{prefix}.{norm}", y)
    return apply_act(y, act)


def apply_act(y: torch.Tensor, act: Optional[str]) -> torch.Tensor:
    """get_activation_fn — focoos/nn/layers/base.py:8-28."""
    if act is None:
        return y
    if act == "relu":
        return F.relu(y)
    if act == "silu":
        return F.silu(y)
    if act == "gelu":
        return F.gelu(y)
    raise ValueError(act)


def linear(sd: SD, prefix: str, x: torch.Tensor) -> torch.Tensor:
    return F.linear(x, sd[f"{prefix}.weight"], sd[f"{prefix}.bias"])


def layer_norm(sd: SD, prefix: str, x: torch.Tensor) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[f"{prefix}.weight"], sd[f"{prefix}.bias"], 1e-5)


def mlp(sd: SD, prefix: str, x: torch.Tensor, num_layers: int) -> torch.Tensor:
    """MLP.forward — focoos/nn/layers/base.py:31-61 (ReLU between layers, none after last)."""
    for i in range(num_layers):
        x = linear(sd, f"{prefix}.layers.{i}", x)
        if i < num_layers - 1:
            x = F.relu(x)
    return x


def mha(sd: SD, prefix: str, q_in: torch.Tensor, k_in: torch.Tensor, v_in: torch.Tensor, nhead: int) -> torch.Tensor:
    """nn.MultiheadAttention(batch_first=True, dropout=0) forward, restated
    (used at modelling.py:901,938 and transformer.py:567,589)."""
    B, Lq, C = q_in.shape
    Lk = k_in.shape[1]
    W, bias = sd[f"{prefix}.in_proj_weight"], sd[f"{prefix}.in_proj_bias"]
    q = F.linear(q_in, W[:C], bias[:C])
    k = F.linear(k_in, W[C:2 * C], bias[C:2 * C])
    v = F.linear(v_in, W[2 * C:], bias[2 * C:])
    d = C // nhead
    q = q.view(B, Lq, nhead, d).transpose(1, 2)
    k = k.view(B, Lk, nhead, d).transpose(1, 2)
    v = v.view(B, Lk, nhead, d).transpose(1, 2)
    att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(d), dim=-1)
    o = (att @ v).transpose(1, 2).reshape(B, Lq, C)
    return linear(sd, f"{prefix}.out_proj", o)


def inverse_sigmoid(x: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """focoos/nn/layers/functional.py:4-6."""
    x = x.clip(min=0.0, max=1.0)
    return torch.log(x.clip(min=eps) / (1 - x).clip(min=eps))


# ----------------------------------------------------------------------------- backbone
def resnet_vd(sd: SD, prefix: str, x: torch.Tensor, blocks: Sequence[int]) -> Dict[str, torch.Tensor]:
    """ResNet.forward (variant d, bottleneck) — focoos/nn/backbone/resnet.py:252-266, 72-121."""
    x = conv_bn(sd, f"{prefix}.conv1.conv1_1", x, 2, "relu")
    x = conv_bn(sd, f"{prefix}.conv1.conv1_2", x, 1, "relu")
    x = conv_bn(sd, f"{prefix}.conv1.conv1_3", x, 1, "relu")
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    outs = {}
    for si, n in enumerate(blocks):
        for bi in range(n):
            p = f"{prefix}.res_layers.{si}.blocks.{bi}"
            stride = 2 if (bi == 0 and si != 0) else 1
            out = conv_bn(sd, f"{p}.branch2a", x, 1, "relu")          # variant d: stride on 3x3
            out = conv_bn(sd, f"{p}.branch2b", out, stride, "relu")
            out = conv_bn(sd, f"{p}.branch2c", out, 1, None)
            if bi == 0:
                if stride == 2:
                    short = F.avg_pool2d(x, 2, 2, 0, ceil_mode=True)
                    short = conv_bn(sd, f"{p}.short.conv", short, 1, None)
                else:
                    short = conv_bn(sd, f"{p}.short", x, 1, None)
            else:
                short = x
            x = F.relu(out + short)
        outs[f"res{si + 2}"] = x
    return outs


# ----------------------------------------------------------------------------- hybrid encoder
def position_embedding_sine(h: int, w: int, num_pos_feats: int, temperature: float = 10000.0) -> torch.Tensor:
    """PositionEmbeddingSine.forward (normalize=False) — modelling.py:148-179. Returns [1, h*w, 2*npf]."""
    not_mask = torch.ones(1, h, w, dtype=torch.bool)
    y_embed = not_mask.cumsum(1, dtype=torch.float32) - 1
    x_embed = not_mask.cumsum(2, dtype=torch.float32) - 1
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / num_pos_feats)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    pos_x_sin = pos_x[:, :, :, 0::2].sin().view(1, h * w, -1)
    pos_x_cos = pos_x[:, :, :, 1::2].cos().view(1, h * w, -1)
    pos_y_sin = pos_y[:, :, :, 0::2].sin().view(1, h * w, -1)
    pos_y_cos = pos_y[:, :, :, 1::2].cos().view(1, h * w, -1)
    return torch.cat((pos_y_sin, pos_y_cos, pos_x_sin, pos_x_cos), dim=2)


def encoder_layer(sd: SD, prefix: str, src: torch.Tensor, pos: torch.Tensor, nhead: int) -> torch.Tensor:
    """TransformerEncoderLayer.forward (post-norm, GELU) — focoos/nn/layers/transformer.py:583-601."""
    q = k = src + pos
    a = mha(sd, f"{prefix}.self_attn", q, k, src, nhead)
    src = layer_norm(sd, f"{prefix}.norm1", src + a)
    f = linear(sd, f"{prefix}.linear2", F.gelu(linear(sd, f"{prefix}.linear1", src)))
    return layer_norm(sd, f"{prefix}.norm2", src + f)


def rep_vgg_block(sd: SD, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """RepVggBlock.forward (unfused branches) — modelling.py:39-45."""
    y = conv_bn(sd, f"{prefix}.conv1", x, 1, None, padding=1) + conv_bn(sd, f"{prefix}.conv2", x, 1, None, padding=0)
    return F.silu(y)


def csp_rep_layer(sd: SD, prefix: str, x: torch.Tensor, num_blocks: int = 3) -> torch.Tensor:
    """CSPRepLayer.forward (expansion 1.0 -> conv3 = Identity) — modelling.py:103-107."""
    x1 = conv_bn(sd, f"{prefix}.conv1", x, 1, "silu")
    for b in range(num_blocks):
        x1 = rep_vgg_block(sd, f"{prefix}.bottlenecks.{b}", x1)
    x2 = conv_bn(sd, f"{prefix}.conv2", x, 1, "silu")
    return x1 + x2


def hybrid_encoder(sd: SD, feats: List[torch.Tensor], cfg: Dict, collect: Optional[dict] = None) -> List[torch.Tensor]:
    """Encoder.forward — modelling.py:297-347. ``feats`` = [res3, res4, res5].
    Returns outs[::-1] = [stride32, stride16, stride8]; the discarded mask_features conv (:347) is skipped."""
    P = "pixel_decoder"
    nhead = int(cfg.get("pixel_decoder_nhead", 8))
    fd = int(cfg.get("pixel_decoder_feat_dim", 256))
    proj = []
    for i, f in enumerate(feats):
        y = F.conv2d(f, sd[f"{P}.input_proj.{i}.0.weight"])
        y = batch_norm(sd, f"{P}.input_proj.{i}.1", y)
        proj.append(y)
    n_enc = int(cfg.get("pixel_decoder_num_encoder_layers", 1))
    if n_enc > 0:
        B, C, h, w = proj[2].shape
        src = proj[2].flatten(2).permute(0, 2, 1)
        pos = position_embedding_sine(h, w, fd // 2)
        for li in range(n_enc):
            src = encoder_layer(sd, f"{P}.encoder.0.layers.{li}", src, pos, nhead)
        proj[2] = src.permute(0, 2, 1).reshape(B, fd, h, w).contiguous()
        if collect is not None:
            collect["aifi"] = src
    inner = [proj[2]]
    for idx in (2, 1):
        high = conv_bn(sd, f"{P}.lateral_convs.{2 - idx}", inner[0], 1, "silu")
        inner[0] = high
        low = proj[idx - 1]
        up = F.interpolate(high, size=low.shape[-2:], mode="bilinear")
        inner.insert(0, csp_rep_layer(sd, f"{P}.fpn_blocks.{2 - idx}", torch.cat([up, low], 1)))
    outs = [inner[0]]
    for idx in range(2):
        down = F.interpolate(outs[-1], size=inner[idx + 1].shape[-2:], mode="bilinear")
        down = conv_bn(sd, f"{P}.downsample_convs.{idx}", down, 1, "silu")
        outs.append(csp_rep_layer(sd, f"{P}.pan_blocks.{idx}", torch.cat([down, inner[idx + 1]], 1)))
    return outs[::-1]


# ----------------------------------------------------------------------------- predictor
def generate_anchors(spatial_shapes: Sequence[Sequence[int]], grid_size: float = 0.05, eps: float = 1e-2):
    """TransformerPredictor._generate_anchors — modelling.py:1169-1189."""
    anchors = []
    for lvl, (h, w) in enumerate(spatial_shapes):
        gy, gx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
        gxy = torch.stack([gx, gy], -1)
        gxy = (gxy.unsqueeze(0) + 0.5) / torch.tensor([w, h]).to(torch.float32)
        wh = torch.ones_like(gxy) * grid_size * (2.0 ** (2 - lvl))
        anchors.append(torch.concat([gxy, wh], -1).reshape(-1, h * w, 4))
    anchors = torch.concat(anchors, 1)
    valid = ((anchors > eps) * (anchors < 1 - eps)).all(-1, keepdim=True)
    anchors = torch.log(anchors / (1 - anchors))
    anchors = torch.where(valid, anchors, 0.0)
    return anchors, valid


def ms_deform_attn_core(value: torch.Tensor, shapes: Sequence[Sequence[int]], loc: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """ms_deform_attn_core_pytorch — focoos/nn/layers/deformable.py:10-35 (the B4 seam)."""
    N_, S_, M_, D_ = value.shape
    _, Lq_, _, L_, P_, _ = loc.shape
    value_list = value.split([h * w_ for h, w_ in shapes], dim=1)
    grids = 2 * loc - 1
    samples = []
    for lid, (H_, W_) in enumerate(shapes):
        v = value_list[lid].flatten(2).transpose(1, 2).reshape(N_ * M_, D_, H_, W_)
        g = grids[:, :, :, lid].transpose(1, 2).flatten(0, 1)
        samples.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False))
    w = w.transpose(1, 2).reshape(N_ * M_, 1, Lq_, L_ * P_)
    out = (torch.stack(samples, dim=-2).flatten(-2) * w).sum(-1).view(N_, M_ * D_, Lq_)
    return out.transpose(1, 2).contiguous()


def ms_deformable_attention(sd: SD, prefix: str, query, ref_points, memory, shapes, nhead=8, n_levels=3, n_points=4):
    """MSDeformableAttention.forward (4-d reference branch) — modelling.py:831-884."""
    bs, Lq, C = query.shape
    Lv = memory.shape[1]
    value = linear(sd, f"{prefix}.value_proj", memory).reshape(bs, Lv, nhead, C // nhead)
    off = linear(sd, f"{prefix}.sampling_offsets", query).reshape(bs, Lq, nhead, n_levels, n_points, 2)
    aw = linear(sd, f"{prefix}.attention_weights", query).reshape(bs, Lq, nhead, n_levels * n_points)
    aw = F.softmax(aw, dim=-1).reshape(bs, Lq, nhead, n_levels, n_points)
    loc = ref_points[:, :, None, :, None, :2] + off / n_points * ref_points[:, :, None, :, None, 2:] * 0.5
    out = ms_deform_attn_core(value, shapes, loc, aw)
    return linear(sd, f"{prefix}.output_proj", out)


def decoder_layer(sd: SD, prefix: str, tgt, ref_input, memory, shapes, qpos, nhead=8):
    """TransformerDecoderLayer.forward — modelling.py:924-958."""
    q = k = tgt + qpos
    tgt = layer_norm(sd, f"{prefix}.norm1", tgt + mha(sd, f"{prefix}.self_attn", q, k, tgt, nhead))
    t2 = ms_deformable_attention(sd, f"{prefix}.cross_attn", tgt + qpos, ref_input, memory, shapes, nhead)
    tgt = layer_norm(sd, f"{prefix}.norm2", tgt + t2)
    t2 = linear(sd, f"{prefix}.linear2", F.relu(linear(sd, f"{prefix}.linear1", tgt)))
    return layer_norm(sd, f"{prefix}.norm3", tgt + t2)


def predictor(sd: SD, feats: List[torch.Tensor], cfg: Dict, forced_topk: Optional[torch.Tensor] = None,
              collect: Optional[dict] = None):
    """TransformerPredictor.forward in eval mode — modelling.py:1234-1263 with
    _get_encoder_input :1145-1167, _get_decoder_input :1191-1232, TransformerDecoder.forward :969-1020."""
    P = "head.predictor"
    nq = int(cfg.get("num_queries", 300))
    nl = int(cfg.get("transformer_predictor_dec_layers", 6))
    nhead = int(cfg.get("transformer_predictor_nhead", 8))
    flat, shapes = [], []
    for i, f in enumerate(feats):
        y = conv_bn(sd, f"{P}.input_proj.{i}", f, 1, None)
        shapes.append([y.shape[2], y.shape[3]])
        flat.append(y.flatten(2).permute(0, 2, 1))
    memory = torch.concat(flat, 1)
    anchors, valid = generate_anchors(shapes)
    mem_v = valid.to(memory.dtype) * memory
    output_memory = layer_norm(sd, f"{P}.enc_output.1", linear(sd, f"{P}.enc_output.0", mem_v))
    enc_class = linear(sd, f"{P}.enc_score_classifier", output_memory)
    enc_coord_unact = mlp(sd, f"{P}.enc_bbox_classifier", output_memory, 3) + anchors
    scores = enc_class.max(-1).values
    if forced_topk is None:
        _, topk_ind = torch.topk(scores, nq, dim=1)
    else:
        topk_ind = forced_topk
    ref_unact = enc_coord_unact.gather(1, topk_ind.unsqueeze(-1).repeat(1, 1, 4))
    target = output_memory.gather(1, topk_ind.unsqueeze(-1).repeat(1, 1, output_memory.shape[-1]))
    if collect is not None:
        collect.update(memory=memory, enc_scores=scores, topk_ind=topk_ind, target=target, ref_unact=ref_unact)
    ref = torch.sigmoid(ref_unact)
    out = target
    logits = boxes = None
    eval_idx = nl - 1
    for i in range(nl):
        qpos = mlp(sd, f"{P}.query_pos_head", ref, 2)
        out = decoder_layer(sd, f"{P}.decoder.layers.{i}", out, ref.unsqueeze(2), memory, shapes, qpos, nhead)
        new_ref = torch.sigmoid(mlp(sd, f"{P}.dec_bbox_classifier.{i}", out, 3) + inverse_sigmoid(ref))
        if collect is not None:
            collect[f"dec{i}_out"] = out
            collect[f"dec{i}_ref"] = new_ref
        if i == eval_idx:
            logits = linear(sd, f"{P}.dec_score_classifier.{i}", out)
            boxes = new_ref
            break
        ref = new_ref
    return logits, boxes


def box_cxcywh_to_xyxy(x: torch.Tensor) -> torch.Tensor:
    """focoos/utils/box.py:14-17."""
    xc, yc, w, h = x.unbind(-1)
    return torch.stack([xc - 0.5 * w, yc - 0.5 * h, xc + 0.5 * w, yc + 0.5 * h], dim=-1)


# ----------------------------------------------------------------------------- whole model
def backbone_features(sd: SD, cfg: Dict, x: torch.Tensor) -> Dict[str, torch.Tensor]:
    """res2..res5 of the configured backbone: ResNet-vd (fai-detr-l-*, focoos/nn/backbone/resnet.py:252-266) or STDC (fai-detr-m-coco,
    focoos/nn/backbone/stdc.py:313-320) - load_backbone dispatches on backbone_config.model_type (focoos/nn/backbone/build.py:4-8)."""
    bb = cfg["backbone_config"]
    if bb.get("model_type", "resnet") == "stdc":
        from . import bf_oracle as BF

        return BF.stdc(sd, "pixel_decoder.backbone", x, tuple(bb.get("layers", (4, 5, 3))))
    return resnet_vd(sd, "pixel_decoder.backbone", x, RESNET_BLOCKS[int(bb.get("depth", 50))])


def detr_forward(sd: SD, cfg: Dict, images: torch.Tensor, forced_topk: Optional[torch.Tensor] = None,
                 collect: Optional[dict] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """FAIDetr.forward (eval) — modelling.py:1344-1358 + DETRHead.forward :386-401.
    ``images``: [B,3,H,W] float32 on the 0..255 scale (un-normalised).  Returns (probabilities [B,Q,K],
    boxes xyxy in [0,1] [B,Q,4])."""
    mean = torch.tensor(cfg.get("pixel_mean", [123.675, 116.28, 103.53]), dtype=torch.float32).view(-1, 1, 1)
    std = torch.tensor(cfg.get("pixel_std", [58.395, 57.12, 57.375]), dtype=torch.float32).view(-1, 1, 1)
    x = (images - mean) / std
    feats = backbone_features(sd, cfg, x)
    if collect is not None:
        collect.update({k: v for k, v in feats.items()})
    enc = hybrid_encoder(sd, [feats["res3"], feats["res4"], feats["res5"]], cfg, collect)
    if collect is not None:
        for n, e in zip(("enc_s32", "enc_s16", "enc_s8"), enc):
            collect[n] = e
    logits, boxes = predictor(sd, enc, cfg, forced_topk, collect)
    return torch.sigmoid(logits), box_cxcywh_to_xyxy(boxes)


# ----------------------------------------------------------------------------- processor
def get_torch_batch(inputs: Sequence[np.ndarray], target_size: Optional[Tuple[int, int]]) -> torch.Tensor:
    """Processor.get_torch_batch for a list of HWC uint8 arrays — focoos/processor/base_processor.py:223-296."""
    out = []
    for inp in inputs:
        t = torch.from_numpy(np.ascontiguousarray(inp)).unsqueeze(0).permute(0, 3, 1, 2).to(torch.float32)
        if target_size is not None:
            t = F.interpolate(t, size=target_size, mode="bilinear", align_corners=False)
        out.append(t.squeeze(0))
    return torch.stack(out, 0)


def postprocess(probs: torch.Tensor, boxes: torch.Tensor, image_sizes: Sequence[Tuple[int, int]],
                top_k: int = 300, threshold: float = 0.5):
    """DETRProcessor.postprocess / _get_predictions — focoos/models/fai_detr/processor.py:146-151,183-197.
    Returns per image (scores f32 [n], labels i64 [n], query index i64 [n], boxes i32 [n,4])."""
    res = []
    K = probs.shape[-1]
    for i in range(probs.shape[0]):
        s, index = torch.topk(probs[i].flatten(0), top_k, dim=-1)
        labels = index % K
        q = index // K
        bp = boxes[i].gather(0, q.unsqueeze(-1).repeat(1, 4))
        m = s > threshold
        bp, s, labels, q = bp[m].clone(), s[m], labels[m], q[m]
        bp[:, 0::2] = bp[:, 0::2] * image_sizes[i][1]
        bp[:, 1::2] = bp[:, 1::2] * image_sizes[i][0]
        res.append((s, labels, q, bp.round().to(torch.int32)))
    return res
