"""TEST INFRASTRUCTURE — CPU restatement of the RT-DETR training criterion's matcher and losses
(SURVEY §8a rows A14/A15).  Only tests / smoke / bench's cpu_baseline may import this.

Restates (paths relative to /root/reference):
  * BoxHungarianMatcher.forward        focoos/models/fai_detr/modelling.py:693-758
  * box_iou / generalized_box_iou       focoos/utils/box.py:27-64  (torchvision box_area = (x1-x0)*(y1-y0))
  * SetCriterion.loss_labels_vfl        focoos/models/fai_detr/modelling.py:464-497
  * SetCriterion.loss_boxes             focoos/models/fai_detr/modelling.py:513-530
Third-party arithmetic on this path that is NOT under /root/reference: SciPy
``scipy.optimize.linear_sum_assignment`` (reference pins scipy~=1.14.1, pyproject.toml:44; this image has
1.15.3), called at modelling.py:750 on each image's [num_queries, T_i] float32 cost block.  ``lsa_crouse``
below restates its published algorithm (D. F. Crouse, "On implementing 2D rectangular assignment
algorithms", IEEE TAES 2016 — the modified Jonker-Volgenant shortest-augmenting-path method of
scipy/optimize/rectangular_lsap, including its transposition rule and tie-breaking) and is pinned against
SciPy itself in tests/test_criterion_oracle.py; golden vectors from the real reference's matcher/criterion
are in tests/golden/detr_criterion.npz (scripts/make_golden.py).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


def box_cxcywh_to_xyxy(x: torch.Tensor) -> torch.Tensor:
    xc, yc, w, h = x.unbind(-1)
    return torch.stack([xc - 0.5 * w, yc - 0.5 * h, xc + 0.5 * w, yc + 0.5 * h], dim=-1)


def box_iou(b1: torch.Tensor, b2: torch.Tensor):
    """focoos/utils/box.py:27-40."""
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    lt = torch.max(b1[:, None, :2], b2[:, :2])
    rb = torch.min(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    union = a1[:, None] + a2 - inter
    return inter / union, union


def generalized_box_iou(b1: torch.Tensor, b2: torch.Tensor) -> torch.Tensor:
    """focoos/utils/box.py:43-64."""
    iou, union = box_iou(b1, b2)
    lt = torch.min(b1[:, None, :2], b2[:, :2])
    rb = torch.max(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    area = wh[:, :, 0] * wh[:, :, 1]
    return iou - (area - union) / (area + 1e-5)


def matcher_cost(logits: torch.Tensor, boxes: torch.Tensor, tgt_labels: Sequence[torch.Tensor], tgt_boxes: Sequence[torch.Tensor],
                 w_class: float = 2.0, w_bbox: float = 5.0, w_giou: float = 2.0, alpha: float = 0.25, gamma: float = 2.0) -> List[torch.Tensor]:
    """Per-image cost blocks [Q, T_i] of BoxHungarianMatcher (modelling.py:714-750, focal branch).  The reference builds the
    full [B*Q, sum(T)] matrix and splits it; the per-image blocks on the diagonal are all it uses."""
    out = []
    for b in range(logits.shape[0]):
        p = torch.sigmoid(logits[b])[:, tgt_labels[b].long()]
        neg = (1 - alpha) * (p ** gamma) * (-(1 - p + 1e-8).log())
        pos = alpha * ((1 - p) ** gamma) * (-(p + 1e-8).log())
        c_class = pos - neg
        c_bbox = torch.cdist(boxes[b], tgt_boxes[b], p=1)
        c_giou = -generalized_box_iou(box_cxcywh_to_xyxy(boxes[b]), box_cxcywh_to_xyxy(tgt_boxes[b]))
        out.append(w_bbox * c_bbox + w_class * c_class + w_giou * c_giou)
    return out


def lsa_crouse(cost: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Rectangular linear sum assignment, restating scipy.optimize.linear_sum_assignment (minimisation):
    float64 arithmetic, transpose when there are more rows than columns, rows processed in order, shortest augmenting
    path with SciPy's tie rule (an equal-cost column replaces the current choice only if it is unassigned), and the
    `remaining` list filled in reverse order with swap-removal.  Returns (row_ind ascending, col_ind)."""
    c = np.asarray(cost, dtype=np.float64)
    nr, nc = c.shape
    if nr == 0 or nc == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    transpose = nc < nr
    if transpose:
        c = np.ascontiguousarray(c.T)
        nr, nc = nc, nr
    u, v = np.zeros(nr), np.zeros(nc)
    col4row = -np.ones(nr, np.int64)
    row4col = -np.ones(nc, np.int64)
    path = -np.ones(nc, np.int64)
    for cur in range(nr):
        spc = np.full(nc, np.inf)
        SR = np.zeros(nr, bool)
        SC = np.zeros(nc, bool)
        remaining = [nc - it - 1 for it in range(nc)]
        num_remaining = nc
        min_val, i, sink = 0.0, cur, -1
        while sink == -1:
            index, lowest = -1, np.inf
            SR[i] = True
            for it in range(num_remaining):
                j = remaining[it]
                r = min_val + c[i, j] - u[i] - v[j]
                if r < spc[j]:
                    path[j] = i
                    spc[j] = r
                if spc[j] < lowest or (spc[j] == lowest and row4col[j] == -1):
                    lowest = spc[j]
                    index = it
            min_val = lowest
            if min_val == np.inf:
                raise ValueError("cost matrix is infeasible")
            j = remaining[index]
            if row4col[j] == -1:
                sink = j
            else:
                i = row4col[j]
            SC[j] = True
            num_remaining -= 1
            remaining[index] = remaining[num_remaining]
        u[cur] += min_val
        for r_ in range(nr):
            if SR[r_] and r_ != cur:
                u[r_] += min_val - spc[col4row[r_]]
        for j in range(nc):
            if SC[j]:
                v[j] -= min_val - spc[j]
        j = sink
        while True:
            i = path[j]
            row4col[j] = i
            col4row[i], j = j, col4row[i]
            if i == cur:
                break
    if transpose:
        order = np.argsort(col4row)
        return col4row[order].astype(np.int64), order.astype(np.int64)
    return np.arange(nr, dtype=np.int64), col4row.astype(np.int64)


def hungarian(costs: Sequence[torch.Tensor]) -> List[Tuple[np.ndarray, np.ndarray]]:
    """modelling.py:749-750 with SciPy (the reference's own dependency)."""
    from scipy.optimize import linear_sum_assignment

    return [tuple(np.asarray(a, np.int64) for a in linear_sum_assignment(c.cpu().numpy())) for c in costs]


def set_criterion_losses(logits: torch.Tensor, boxes: torch.Tensor, tgt_labels, tgt_boxes, indices, num_boxes: float,
                         focal_alpha: float = 0.75, focal_gamma: float = 2.0, w_vfl: float = 1.0, w_bbox: float = 5.0, w_giou: float = 2.0):
    """loss_labels_vfl + loss_boxes for ONE prediction set (modelling.py:464-497, 513-530, weights applied as in :576-579)."""
    B, Q, K = logits.shape
    bidx = torch.cat([torch.full((len(i),), b, dtype=torch.int64) for b, (i, _) in enumerate(indices)])
    sidx = torch.cat([torch.as_tensor(i, dtype=torch.int64) for i, _ in indices])
    src_boxes = boxes[bidx, sidx]
    tb = torch.cat([tgt_boxes[b][torch.as_tensor(j, dtype=torch.int64)] for b, (_, j) in enumerate(indices)], 0)
    ious = torch.diag(box_iou(box_cxcywh_to_xyxy(src_boxes), box_cxcywh_to_xyxy(tb))[0]).detach() if len(sidx) else torch.zeros(0)  # :471 detached
    tco = torch.cat([tgt_labels[b][torch.as_tensor(j, dtype=torch.int64)].long() for b, (_, j) in enumerate(indices)])
    target_classes = torch.full((B, Q), K, dtype=torch.int64)
    target_classes[bidx, sidx] = tco
    target = F.one_hot(target_classes, K + 1)[..., :-1]
    tso = torch.zeros(B, Q)
    tso[bidx, sidx] = ious
    target_score = tso.unsqueeze(-1) * target
    pred = torch.sigmoid(logits).detach()  # :487 detached
    weight = focal_alpha * pred.pow(focal_gamma) * (1 - target) + target_score
    loss = F.binary_cross_entropy_with_logits(logits, target_score, weight=weight, reduction="none")
    l_vfl = loss.mean(1).sum() * Q / num_boxes
    l_bbox = F.l1_loss(src_boxes, tb, reduction="none").sum() / num_boxes
    l_giou = (1 - torch.diag(generalized_box_iou(box_cxcywh_to_xyxy(src_boxes), box_cxcywh_to_xyxy(tb)))).sum() / num_boxes if len(sidx) else torch.zeros(())
    return {"loss_vfl": w_vfl * l_vfl, "loss_bbox": w_bbox * l_bbox, "loss_giou": w_giou * l_giou}


def synth_predictions_and_targets(seed: int = 0, B: int = 4, Q: int = 300, K: int = 80, counts=(7, 0, 20, 1)):
    """Seeded synthetic decoder outputs + COCO-shaped targets (centres U(0.2,0.8), sizes U(0.05,0.35))."""
    rs = np.random.RandomState(seed)
    logits = torch.from_numpy((rs.standard_normal((B, Q, K)) * 2.0 - 4.0).astype(np.float32))
    cxcy = rs.uniform(0.1, 0.9, (B, Q, 2))
    wh = rs.uniform(0.02, 0.5, (B, Q, 2))
    boxes = torch.from_numpy(np.concatenate([cxcy, wh], -1).astype(np.float32))
    labels, tboxes = [], []
    for b in range(B):
        t = counts[b % len(counts)]
        labels.append(torch.from_numpy(rs.randint(0, K, (t,)).astype(np.int64)))
        tboxes.append(torch.from_numpy(np.concatenate([rs.uniform(0.2, 0.8, (t, 2)), rs.uniform(0.05, 0.35, (t, 2))], -1).astype(np.float32)))
    return logits, boxes, labels, tboxes
