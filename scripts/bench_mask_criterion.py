#!/usr/bin/env python
"""Timing of the mask-classification criterion on the GPU: matcher + losses over all prediction sets, for the
MaskFormer training shape (bs=16, 100 queries, 80 classes, masks 200x200 -> targets 800x800, 10 sets) and the BiSeNetFormer one
(bs=16, 150 classes, masks 80x80 -> 640x640, 7 sets), 12544 points.  Prints one JSON line per shape; `--cpu` also times the oracle
(the reference's arithmetic on the host CPU) on one prediction set."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from focoos_amd.mask_criterion import MaskHungarianMatcher, SetCriterion  # noqa: E402
from focoos_amd.ports import MaskFormerTargets  # noqa: E402
from oracle import mask_criterion_oracle as MC  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cpu", action="store_true")
ap.add_argument("--iters", type=int, default=5)
args = ap.parse_args()
dev = "cuda:0"
P = 12544
for name, B, K, hw, scale, n_aux in (("fai-mf-l-coco-ins", 16, 80, (200, 200), 4, 9), ("bisenetformer-l-ade", 16, 150, (80, 80), 8, 6)):
    counts = tuple(3 + (7 * i) % 18 for i in range(B))
    out, labels, masks = MC.synth_mask_predictions_and_targets(1, B=B, Q=100, K=K, hw=hw, scale=scale, counts=counts, n_aux=n_aux)
    d = {"pred_logits": out["pred_logits"].to(dev), "pred_masks": out["pred_masks"].to(dev),
         "aux_outputs": [{k: v.to(dev) for k, v in a.items()} for a in out["aux_outputs"]]}
    tg = [MaskFormerTargets(labels=l.to(dev), masks=m.bool().to(dev)) for l, m in zip(labels, masks)]
    crit = SetCriterion(K, MaskHungarianMatcher(2, 5, 5, num_points=P), {"loss_ce": 2, "loss_mask": 5, "loss_dice": 5}, num_points=P,
                        importance_sample_ratio=0.75)
    crit(d, tg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        losses = crit(d, tg)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / args.iters
    res = {"workload": f"{name} criterion: {n_aux + 1} prediction sets, bs={B}, Q=100, K={K}, masks {hw[0]}x{hw[1]}, targets x{scale}, {sum(counts)} targets, "
                       f"{P} points", "gpu_ms_per_call": round(ms, 2), "total_loss": round(float(sum(losses.values())), 4)}
    if args.cpu:
        gen = torch.Generator().manual_seed(0)
        n = sum(counts)
        draws = [torch.rand(1, P, 2, generator=gen) for _ in range(B)] + [torch.rand(n, 3 * P, 2, generator=gen), torch.rand(n, P - int(0.75 * P), 2, generator=gen)]
        one = {"pred_logits": out["pred_logits"], "pred_masks": out["pred_masks"]}
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        t0 = time.perf_counter()
        MC.criterion(one, labels, masks, MC.RandStream(draws), K, P)
        res["cpu_oracle_ms_per_set"] = round(1e3 * (time.perf_counter() - t0), 1)
        res["cpu_oracle_ms_all_sets_extrapolated"] = round(res["cpu_oracle_ms_per_set"] * (n_aux + 1), 1)
    print(json.dumps(res))
