"""TEST INFRASTRUCTURE — CPU fp32 restatement of the RT-DETR *training* forward and criterion,
differentiable through torch autograd so that gradients of every parameter can be compared with the HIP autograd graph.

Restates (paths relative to /root/reference), with BatchNorm in eval mode (the ``freeze_bn`` variant of SURVEY config 4):
  * FAIDetr.forward, training branch            focoos/models/fai_detr/modelling.py:1344-1358
  * TransformerPredictor.forward (training)     :1234-1263 with _get_decoder_input :1191-1232 (target / reference points
                                                detached) and TransformerDecoder.forward :969-1020 (all layers supervised,
                                                reference points detached between layers)
  * SetCriterion.forward                        :553-612 (7 prediction sets: last layer, 5 auxiliary layers, encoder top-k)
Pinned against the real reference in train mode with its BatchNorm modules in eval mode: tests/test_oracle_vs_reference.py.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import criterion_oracle as CO
from . import detr_oracle as O

SD = Dict[str, torch.Tensor]


def predictor_train(sd: SD, feats: List[torch.Tensor], cfg: Dict, forced_topk: Optional[torch.Tensor] = None):
    P = "head.predictor"
    nq = int(cfg.get("num_queries", 300))
    nl = int(cfg.get("transformer_predictor_dec_layers", 6))
    nhead = int(cfg.get("transformer_predictor_nhead", 8))
    flat, shapes = [], []
    for i, f in enumerate(feats):
        y = O.conv_bn(sd, f"{P}.input_proj.{i}", f, 1, None)
        shapes.append([y.shape[2], y.shape[3]])
        flat.append(y.flatten(2).permute(0, 2, 1))
    memory = torch.concat(flat, 1)
    anchors, valid = O.generate_anchors(shapes)
    output_memory = O.layer_norm(sd, f"{P}.enc_output.1", O.linear(sd, f"{P}.enc_output.0", valid.to(memory.dtype) * memory))
    enc_class = O.linear(sd, f"{P}.enc_score_classifier", output_memory)
    enc_coord_unact = O.mlp(sd, f"{P}.enc_bbox_classifier", output_memory, 3) + anchors
    topk_ind = torch.topk(enc_class.max(-1).values, nq, dim=1)[1] if forced_topk is None else forced_topk
    ref_unact = enc_coord_unact.gather(1, topk_ind.unsqueeze(-1).repeat(1, 1, 4))
    enc_topk_bboxes = torch.sigmoid(ref_unact)
    enc_topk_logits = enc_class.gather(1, topk_ind.unsqueeze(-1).repeat(1, 1, enc_class.shape[-1]))
    target = output_memory.gather(1, topk_ind.unsqueeze(-1).repeat(1, 1, output_memory.shape[-1])).detach()
    out = target
    ref_detach = torch.sigmoid(ref_unact.detach())
    ref = ref_detach
    logits, boxes = [], []
    for i in range(nl):
        qpos = O.mlp(sd, f"{P}.query_pos_head", ref_detach, 2)
        out = O.decoder_layer(sd, f"{P}.decoder.layers.{i}", out, ref_detach.unsqueeze(2), memory, shapes, qpos, nhead)
        delta = O.mlp(sd, f"{P}.dec_bbox_classifier.{i}", out, 3)
        inter = torch.sigmoid(delta + O.inverse_sigmoid(ref_detach))
        logits.append(O.linear(sd, f"{P}.dec_score_classifier.{i}", out))
        boxes.append(inter if i == 0 else torch.sigmoid(delta + O.inverse_sigmoid(ref)))
        ref, ref_detach = inter, inter.detach()
    return {"pred_logits": logits[-1], "pred_boxes": boxes[-1],
            "aux_outputs": [{"pred_logits": a, "pred_boxes": b} for a, b in zip(logits[:-1], boxes[:-1])]
            + [{"pred_logits": enc_topk_logits, "pred_boxes": enc_topk_bboxes}], "topk_ind": topk_ind}


def detr_train_outputs(sd: SD, cfg: Dict, images: torch.Tensor, forced_topk: Optional[torch.Tensor] = None):
    mean = torch.tensor(cfg.get("pixel_mean", [123.675, 116.28, 103.53]), dtype=torch.float32).view(-1, 1, 1)
    std = torch.tensor(cfg.get("pixel_std", [58.395, 57.12, 57.375]), dtype=torch.float32).view(-1, 1, 1)
    x = (images - mean) / std
    feats = O.backbone_features(sd, cfg, x)
    enc = O.hybrid_encoder(sd, [feats["res3"], feats["res4"], feats["res5"]], cfg)
    return predictor_train(sd, enc, cfg, forced_topk)


def criterion(outputs, tgt_labels: Sequence[torch.Tensor], tgt_boxes: Sequence[torch.Tensor], fixed_matches=None):
    """SetCriterion.forward: dict of weighted losses over the 7 sets + the matches used (list per set of per-image index pairs)."""
    num_boxes = max(float(sum(len(t) for t in tgt_labels)), 1.0)
    sets = [("", outputs)] + [(f"_{i}", a) for i, a in enumerate(outputs["aux_outputs"])]
    losses, matches = {}, []
    for j, (suffix, o) in enumerate(sets):
        if fixed_matches is None:
            with torch.no_grad():
                costs = CO.matcher_cost(o["pred_logits"].detach(), o["pred_boxes"].detach(), tgt_labels, tgt_boxes)
                idx = CO.hungarian(costs)
        else:
            idx = fixed_matches[j]
        matches.append(idx)
        l = CO.set_criterion_losses(o["pred_logits"], o["pred_boxes"], tgt_labels, tgt_boxes, idx, num_boxes)
        for k, v in l.items():
            losses[k + suffix] = v
    return losses, matches


def synth_targets(seed: int, B: int, K: int, counts=(3, 0, 7, 1)):
    """COCO-shaped random targets: centres U(0.2,0.8), sizes U(0.05,0.35)."""
    rs = np.random.RandomState(seed)
    labels, boxes = [], []
    for b in range(B):
        t = counts[b % len(counts)]
        labels.append(torch.from_numpy(rs.randint(0, K, (t,)).astype(np.int64)))
        boxes.append(torch.from_numpy(np.concatenate([rs.uniform(0.2, 0.8, (t, 2)), rs.uniform(0.05, 0.35, (t, 2))], -1).astype(np.float32)))
    return labels, boxes


# ================================================================================================ BiSeNetFormer (BASELINE config 5)
def bf_train_outputs(sd: SD, cfg: Dict, images: torch.Tensor, forced_attn: Optional[Sequence[torch.Tensor]] = None, collect: Optional[dict] = None):
    """BisenetFormer.forward in training mode up to the criterion (focoos/models/bisenetformer/modelling.py:594-609, TransformerDecoder.forward
    :375-447): ``pred_logits`` / ``pred_masks`` of the last head + ``aux_outputs`` of the learnable-query head and the first
    dec_layers - 1 layers (:438-447).  BatchNorm follows detr_oracle.BN_TRAINING (False = frozen / eval statistics).  The boolean attention
    masks carry no gradient (``< 0`` of a detached tensor, :104-106)."""
    from . import bf_oracle as BF
    from . import mf_oracle as M

    mean = torch.tensor(cfg.get("pixel_mean", [123.675, 116.28, 103.53]), dtype=torch.float32).view(-1, 1, 1)
    std = torch.tensor(cfg.get("pixel_std", [58.395, 57.12, 57.375]), dtype=torch.float32).view(-1, 1, 1)
    x = (images - mean) / std
    feats = BF.stdc(sd, "pixel_decoder.backbone", x, tuple(cfg["backbone_config"].get("layers", (4, 5, 3))))
    if collect is not None:
        collect.update(feats)
    mask_features, msf = BF.bisenet(sd, feats, collect)
    heads: list = []
    M.masked_decoder(sd, list(msf[:-1]), mask_features, cfg, forced_attn, collect, max_levels=2, all_heads=heads)
    return {"pred_logits": heads[-1][0], "pred_masks": heads[-1][1], "aux_outputs": [{"pred_logits": a, "pred_masks": b} for a, b in heads[:-1]]}


def mf_train_outputs(sd: SD, cfg: Dict, images: torch.Tensor, forced_attn: Optional[Sequence[torch.Tensor]] = None, collect: Optional[dict] = None):
    """FAIMaskFormer.forward in training mode up to the criterion (focoos/models/fai_mf/modelling.py:712-725, TransformerFPN :347-369,
    MultiScaleMaskedTransformerDecoder.forward :453-549 with every prediction head supervised :489-549)."""
    from . import mf_oracle as M

    mean = torch.tensor(cfg.get("pixel_mean", [123.675, 116.28, 103.53]), dtype=torch.float32).view(-1, 1, 1)
    std = torch.tensor(cfg.get("pixel_std", [58.395, 57.12, 57.375]), dtype=torch.float32).view(-1, 1, 1)
    x = (images - mean) / std
    feats = M.backbone_features(sd, cfg, x)
    if collect is not None:
        collect.update(feats)
    mask_features, msf = M.transformer_fpn(sd, feats, cfg, collect)
    heads: list = []
    M.masked_decoder(sd, msf, mask_features, cfg, forced_attn, collect, all_heads=heads)
    return {"pred_logits": heads[-1][0], "pred_masks": heads[-1][1], "aux_outputs": [{"pred_logits": a, "pred_masks": b} for a, b in heads[:-1]]}


def bf_criterion(outputs, tgt_labels: Sequence[torch.Tensor], tgt_masks: Sequence[torch.Tensor], rand, cfg: Dict, fixed_matches=None):
    """SetCriterion.forward of the mask families (MaskFormer and BiSeNetFormer share it) with the registry's weights
    (bisenetformer/modelling.py:551-575 == fai_mf/modelling.py:657-681): dict of weighted losses
    + the matches; ``rand`` = mask_criterion_oracle.RandStream of the torch.rand draws in the reference's order."""
    from . import mask_criterion_oracle as MC

    return MC.criterion(outputs, tgt_labels, tgt_masks, rand, int(cfg["num_classes"]), int(cfg.get("criterion_num_points", 12544)),
                        weights=(float(cfg.get("weight_dict_loss_ce", 2)), float(cfg.get("weight_dict_loss_mask", 5)), float(cfg.get("weight_dict_loss_dice", 5))),
                        cost_weights=(float(cfg.get("matcher_cost_class", 2)), float(cfg.get("matcher_cost_mask", 5)), float(cfg.get("matcher_cost_dice", 5))),
                        eos_coef=float(cfg.get("criterion_eos_coef", 0.1)), fixed_matches=fixed_matches)


def synth_mask_targets(seed: int, B: int, K: int, hw, counts=(3, 5)):
    """Seeded blob-shaped instance targets at image resolution: per image labels i64 [T] and masks bool [T, H, W]."""
    rs = np.random.RandomState(seed)
    H, W = hw
    yy, xx = np.mgrid[0:H, 0:W]
    labels, masks = [], []
    for b in range(B):
        t = counts[b % len(counts)]
        labels.append(torch.from_numpy(rs.randint(0, K, (t,)).astype(np.int64)))
        m = np.zeros((t, H, W), bool)
        for i in range(t):
            cy, cx, ry, rx = rs.uniform(0, H), rs.uniform(0, W), rs.uniform(H / 8, H / 3), rs.uniform(W / 8, W / 3)
            m[i] = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
        masks.append(torch.from_numpy(m))
    return labels, masks
