"""TEST INFRASTRUCTURE — CPU restatement of the mask-classification training criterion shared by the MaskFormer and
BiSeNetFormer families.  Only tests / smoke / bench's cpu_baseline may import this.

Restates (paths relative to /root/reference; focoos/models/bisenetformer/loss.py is a line-for-line copy):
  * point_sample                                     focoos/nn/layers/point_rend.py:29-52  (F.grid_sample, bilinear, zero padding,
                                                     align_corners=False, on [0,1]^2 coordinates)
  * get_uncertain_point_coords_with_randomness       point_rend.py:73-128 with calculate_uncertainty  fai_mf/loss.py:27-43
  * MaskHungarianMatcher.memory_efficient_forward    fai_mf/loss.py:661-723 (batch_sigmoid_ce_loss :297-317, batch_dice_loss :277-292)
  * SetCriterion.loss_labels (ce_loss branch) / loss_masks / forward          :411-431, :463-523, :545-592
The reference draws its sample points with unseeded torch.rand calls inside the matcher and the loss (SURVEY H-note on A16):
here every function takes the random tensors as ARGUMENTS, in the order the reference draws them, so that the restatement can
be pinned against the reference (tests record the reference's torch.rand stream) and the HIP kernels can be compared with it
on identical points.  Third-party arithmetic: torch CPU ops (grid_sample, cross_entropy, topk) and SciPy's
linear_sum_assignment (see oracle/criterion_oracle.py: lsa_crouse / hungarian).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .criterion_oracle import hungarian


def point_sample(inp: torch.Tensor, coords: torch.Tensor) -> torch.Tensor:
    """inp [N,C,H,W], coords [N,P,2] in [0,1]^2 (x, y) -> [N,C,P]."""
    return F.grid_sample(inp, 2.0 * coords.unsqueeze(2) - 1.0, align_corners=False).squeeze(3)


def matcher_cost(logits: torch.Tensor, pred_masks: torch.Tensor, tgt_labels: torch.Tensor, tgt_masks: torch.Tensor, coords: torch.Tensor,
                 w_class: float = 2.0, w_mask: float = 5.0, w_dice: float = 5.0, cls_sigmoid: bool = False) -> torch.Tensor:
    """One image: logits [Q,K+1], pred_masks [Q,h,w] (logits), tgt_labels [T], tgt_masks [T,H,W] (0/1), coords [1,P,2] (the
    image's shared torch.rand draw) -> cost [Q,T] float32 (fai_mf/loss.py:672-712)."""
    prob = logits.sigmoid() if cls_sigmoid else logits.softmax(-1)
    cost_class = -prob[:, tgt_labels.long()]
    T, Q = tgt_masks.shape[0], pred_masks.shape[0]
    tgt = point_sample(tgt_masks[:, None].to(pred_masks), coords.repeat(T, 1, 1)).squeeze(1).float()
    out = point_sample(pred_masks[:, None], coords.repeat(Q, 1, 1)).squeeze(1).float()
    hw = out.shape[1]
    pos = F.binary_cross_entropy_with_logits(out, torch.ones_like(out), reduction="none")
    neg = F.binary_cross_entropy_with_logits(out, torch.zeros_like(out), reduction="none")
    cost_mask = (torch.einsum("nc,mc->nm", pos, tgt) + torch.einsum("nc,mc->nm", neg, 1 - tgt)) / hw
    s = out.sigmoid()
    cost_dice = 1 - (2 * torch.einsum("nc,mc->nm", s, tgt) + 1) / (s.sum(-1)[:, None] + tgt.sum(-1)[None, :] + 1)
    return w_mask * cost_mask + w_class * cost_class + w_dice * cost_dice


def importance_points(src_masks: torch.Tensor, rand_over: torch.Tensor, rand_extra: Optional[torch.Tensor], num_points: int,
                      importance_sample_ratio: float = 0.75) -> torch.Tensor:
    """get_uncertain_point_coords_with_randomness with uncertainty = -|logit|: src_masks [N,1,h,w], rand_over [N, int(P*ratio_over), 2]
    (first torch.rand draw), rand_extra [N, P - int(ratio*P), 2] (second draw) -> coords [N,P,2]."""
    n = src_masks.shape[0]
    logits = point_sample(src_masks, rand_over)
    unc = -logits.abs()
    k = int(importance_sample_ratio * num_points)
    idx = torch.topk(unc[:, 0, :], k=k, dim=1)[1]
    pts = torch.gather(rand_over, 1, idx.unsqueeze(-1).expand(n, k, 2))
    if num_points - k > 0:
        pts = torch.cat([pts, rand_extra], dim=1)
    return pts


def label_loss(logits: torch.Tensor, tgt_labels: Sequence[torch.Tensor], indices, num_classes: int, eos_coef: float = 0.1) -> torch.Tensor:
    """SetCriterion.loss_labels, ce_loss branch (:411-431): cross entropy over K+1 classes with weight eos_coef on "no object"."""
    B, Q = logits.shape[:2]
    target = torch.full((B, Q), num_classes, dtype=torch.int64)
    for b, (i, j) in enumerate(indices):
        target[b, torch.as_tensor(i, dtype=torch.int64)] = tgt_labels[b][torch.as_tensor(j, dtype=torch.int64)].long()
    w = torch.ones(num_classes + 1)
    w[-1] = eos_coef
    return F.cross_entropy(logits.float().transpose(1, 2), target, w)


def mask_losses(pred_masks: torch.Tensor, tgt_masks: Sequence[torch.Tensor], indices, num_masks: float, num_points: int, rand_over: torch.Tensor,
                rand_extra: Optional[torch.Tensor], importance_sample_ratio: float = 0.75) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """SetCriterion.loss_masks (:463-523): matched (prediction, target) pairs in image order; returns (loss_mask, loss_dice,
    the point coordinates used [N,P,2])."""
    src, tgt = [], []
    for b, (i, j) in enumerate(indices):
        src.append(pred_masks[b][torch.as_tensor(i, dtype=torch.int64)])
        tgt.append(tgt_masks[b][torch.as_tensor(j, dtype=torch.int64)].to(pred_masks))
    src, tgt = torch.cat(src)[:, None], torch.cat(tgt)[:, None]
    with torch.no_grad():
        coords = importance_points(src, rand_over, rand_extra, num_points, importance_sample_ratio)
        labels = point_sample(tgt, coords).squeeze(1)
    logits = point_sample(src, coords).squeeze(1)
    loss_mask = F.binary_cross_entropy_with_logits(logits, labels, reduction="none").mean(1).sum() / num_masks
    s = logits.sigmoid()
    loss_dice = (1 - (2 * (s * labels).sum(-1) + 1) / (s.sum(-1) + labels.sum(-1) + 1)).sum() / num_masks
    return loss_mask, loss_dice, coords


class RandStream:
    """The reference's torch.rand draws, replayed in order (shapes are checked)."""

    def __init__(self, tensors: Sequence[torch.Tensor]):
        self.t, self.i = list(tensors), 0

    def take(self, *shape) -> torch.Tensor:
        t = self.t[self.i]
        assert tuple(t.shape) == tuple(shape), (self.i, tuple(t.shape), shape)
        self.i += 1
        return t


def criterion(outputs: Dict, tgt_labels: Sequence[torch.Tensor], tgt_masks: Sequence[torch.Tensor], rand: RandStream, num_classes: int,
              num_points: int, weights=(2.0, 5.0, 5.0), cost_weights=(2.0, 5.0, 5.0), eos_coef: float = 0.1, oversample_ratio: float = 3.0,
              importance_sample_ratio: float = 0.75, fixed_matches=None):
    """SetCriterion.forward (:545-592) with deep supervision: the main prediction set, then every aux set; per set the matcher
    (one torch.rand(1,P,2) per image), then loss_labels, then loss_masks (torch.rand(N, 3P, 2), torch.rand(N, P - 0.75P, 2)).
    Returns (dict of weighted losses named like the reference's, list of per-set matches)."""
    w_ce, w_mask, w_dice = weights
    num_masks = max(float(sum(len(t) for t in tgt_labels)), 1.0)
    sets = [("", outputs)] + [(f"_{i}", a) for i, a in enumerate(outputs.get("aux_outputs", []))]
    losses, matches = {}, []
    B = outputs["pred_logits"].shape[0]
    k_imp = int(importance_sample_ratio * num_points)
    for si, (suffix, o) in enumerate(sets):
        costs = []
        for b in range(B):
            coords = rand.take(1, num_points, 2)
            costs.append(matcher_cost(o["pred_logits"][b].detach(), o["pred_masks"][b].detach(), tgt_labels[b], tgt_masks[b], coords, *cost_weights))
        idx = hungarian(costs) if fixed_matches is None else fixed_matches[si]
        matches.append(idx)
        losses["loss_ce" + suffix] = w_ce * label_loss(o["pred_logits"], tgt_labels, idx, num_classes, eos_coef)
        n = sum(len(i) for i, _ in idx)
        r_over = rand.take(n, int(num_points * oversample_ratio), 2)
        r_extra = rand.take(n, num_points - k_imp, 2) if num_points - k_imp > 0 else None
        lm, ld, _ = mask_losses(o["pred_masks"], tgt_masks, idx, num_masks, num_points, r_over, r_extra, importance_sample_ratio)
        losses["loss_mask" + suffix] = w_mask * lm
        losses["loss_dice" + suffix] = w_dice * ld
    return losses, matches


def synth_mask_predictions_and_targets(seed: int = 0, B: int = 2, Q: int = 100, K: int = 80, hw=(40, 48), scale: int = 4, counts=(6, 3), n_aux: int = 2):
    """Seeded mask-classification predictions (main + aux sets) and blob-shaped targets at `scale` x the prediction resolution."""
    rs = np.random.RandomState(seed)
    h, w = hw
    H, W = h * scale, w * scale
    yy, xx = np.mgrid[0:H, 0:W]
    labels, masks = [], []
    for b in range(B):
        t = counts[b % len(counts)]
        labels.append(torch.from_numpy(rs.randint(0, K, (t,)).astype(np.int64)))
        m = np.zeros((t, H, W), np.float32)
        for i in range(t):
            cy, cx, ry, rx = rs.uniform(0, H), rs.uniform(0, W), rs.uniform(H / 10, H / 3), rs.uniform(W / 10, W / 3)
            m[i] = (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0).astype(np.float32)
        masks.append(torch.from_numpy(m))

    def one_set():
        logits = torch.from_numpy(rs.standard_normal((B, Q, K + 1)).astype(np.float32) * 2.0)
        pm = torch.from_numpy(rs.standard_normal((B, Q, h, w)).astype(np.float32) * 3.0)
        pm = F.avg_pool2d(pm, 5, 1, 2) * 3.0   # spatially smooth logits
        for b in range(B):  # a few queries resemble the targets, so the matching is not arbitrary
            for i in range(len(labels[b])):
                q = int(rs.randint(0, Q))
                small = F.interpolate(masks[b][i][None, None], size=(h, w), mode="bilinear", align_corners=False)[0, 0]
                pm[b, q] = (small - 0.5) * 8.0 + pm[b, q] * 0.3
                logits[b, q, labels[b][i]] += 4.0
        return {"pred_logits": logits, "pred_masks": pm}

    out = one_set()
    out["aux_outputs"] = [one_set() for _ in range(n_aux)]
    return out, labels, masks
