"""Host mirrors of the mask-classification training criterion (MaskFormer and BiSeNetFormer share it) for the gfx950 kernels
(values and gradients: the losses are torch.autograd nodes whose backward is fx_mask_set_loss_bwd_f32):

  MaskHungarianMatcher.forward   focoos/models/fai_mf/loss.py:661-742 (== bisenetformer/loss.py)  -> fx_point_sample_f32 +
                                                                                                   fx_mask_match_cost_f32 + fx_lsa_f32
  SetCriterion.forward           focoos/models/fai_mf/loss.py:545-592                              -> fx_mask_set_loss_f32 per prediction set

Same argument / return structure as the reference (``outputs`` dict with pred_logits / pred_masks / aux_outputs, targets with
``labels`` / ``masks``; the matcher returns a list of (index_i, index_j) int64 tensors; the criterion the dict of weighted losses
with ``_{i}`` suffixes for the auxiliary sets).  The reference draws its sample points with ``torch.rand`` inside; here the draws
come from a ``rand`` callable (default: ``torch.rand`` on the device, i.e. the same distribution) that tests replace to inject the
reference's own draws, in the reference's order: per set, one [1,P,2] per image for the matcher, then [N, 3P, 2] and [N, P - 0.75P, 2]
for the loss.  Nothing leaves the GPU: no cost-matrix D2H copy, no host SciPy call."""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import check


class _MaskTargets:
    """Packed targets in HBM: labels i32 [sumT], masks u8 or f32 [sumT,H,W] (all images padded to one size, like
    nested_tensor_from_tensor_list, loss.py:71-94), offsets i32 [B+1]."""

    def __init__(self, targets: Sequence, device):
        sizes = [int(len(t.labels)) for t in targets]
        self.off_host = np.zeros(len(targets) + 1, np.int32)
        self.off_host[1:] = np.cumsum(sizes)
        self.n = int(self.off_host[-1])
        self.tmax = max(sizes + [0])
        self.offsets = torch.from_numpy(self.off_host).to(device)
        H = max([int(t.masks.shape[-2]) for t in targets if len(t.labels)] + [1])
        W = max([int(t.masks.shape[-1]) for t in targets if len(t.labels)] + [1])
        self.H, self.W = H, W
        if self.n:
            self.labels = torch.cat([t.labels.to(torch.int32) for t in targets]).to(device).contiguous()
            exact = all(t.masks.dtype in (torch.bool, torch.uint8) for t in targets if len(t.labels))
            dt = torch.uint8 if exact else torch.float32
            m = torch.zeros(self.n, H, W, dtype=dt, device=device)
            for b, t in enumerate(targets):
                if len(t.labels):
                    m[self.off_host[b]:self.off_host[b + 1], : t.masks.shape[-2], : t.masks.shape[-1]] = t.masks.to(device=device, dtype=dt)
            self.masks, self.is_u8 = m, int(exact)
        else:
            self.labels = torch.zeros(1, dtype=torch.int32, device=device)
            self.masks, self.is_u8 = torch.zeros(1, 1, 1, dtype=torch.uint8, device=device), 1


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class MaskHungarianMatcher:
    def __init__(self, cost_class: float = 1, cost_mask: float = 1, cost_dice: float = 1, num_points: int = 0, cls_sigmoid: bool = False,
                 rand: Optional[Callable] = None):
        assert cost_class != 0 or cost_mask != 0 or cost_dice != 0, "all costs cant be 0"
        if num_points <= 0:
            raise NotImplementedError("the matcher samples num_points > 0 points per image (criterion_num_points of every registry model)")
        self.cost_class, self.cost_mask, self.cost_dice, self.num_points, self.cls_sigmoid = cost_class, cost_mask, cost_dice, num_points, cls_sigmoid
        self.rand = rand or (lambda *shape, device: torch.rand(*shape, device=device))

    def match_packed(self, logits: torch.Tensor, pred_masks: torch.Tensor, tg: _MaskTargets):
        lib = _lib.load()
        B, Q, K1 = logits.shape
        dev, P = logits.device, self.num_points
        h, w = pred_masks.shape[-2:]
        pi = torch.empty(max(tg.n, 1), dtype=torch.int32, device=dev)
        ti = torch.empty(max(tg.n, 1), dtype=torch.int32, device=dev)
        # one shared set of points per image (loss.py:687), drawn image by image like the reference
        coords = torch.cat([self.rand(1, P, 2, device=dev).float() for _ in range(B)]).contiguous()
        if tg.n:
            st = _stream(dev)
            img_of_pred = torch.arange(B, dtype=torch.int32, device=dev).repeat_interleave(Q)
            img_of_tgt = torch.repeat_interleave(torch.arange(B, dtype=torch.int32, device=dev), torch.from_numpy(np.diff(tg.off_host)).to(dev))
            pp = torch.empty(B * Q, P, dtype=torch.float32, device=dev)
            tp = torch.empty(tg.n, P, dtype=torch.float32, device=dev)
            check(lib.fx_point_sample_f32(pred_masks.data_ptr(), 0, h, w, None, coords.data_ptr(), img_of_pred.data_ptr(), pp.data_ptr(), B * Q, P, st),
                  "fx_point_sample_f32")
            check(lib.fx_point_sample_f32(tg.masks.data_ptr(), tg.is_u8, tg.H, tg.W, None, coords.data_ptr(), img_of_tgt.data_ptr(), tp.data_ptr(), tg.n, P,
                                          st), "fx_point_sample_f32")
            cost = torch.empty(B, Q, tg.tmax, dtype=torch.float32, device=dev)
            nws = int(lib.fx_mask_match_cost_workspace_bytes(B, Q, tg.tmax))      # 0: more than 64 targets in one image - the untiled kernel
            ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=dev)
            check(lib.fx_mask_match_cost_ws_f32(logits.data_ptr(), K1, pp.data_ptr(), tp.data_ptr(), tg.labels.data_ptr(), tg.offsets.data_ptr(), B, Q, K1 - 1,
                                                P, tg.tmax, float(self.cost_class), float(self.cost_mask), float(self.cost_dice), int(self.cls_sigmoid),
                                                cost.data_ptr(), ws.data_ptr(), nws, st), "fx_mask_match_cost_ws_f32")
            from .criterion import lsa_status

            check(lib.fx_lsa_status_f32(cost.data_ptr(), B, Q, tg.tmax, tg.offsets.data_ptr(), pi.data_ptr(), ti.data_ptr(), lsa_status(dev).data_ptr(), st),
                  "fx_lsa_status_f32")
            self.last_cost = cost
        return pi, ti

    @torch.no_grad()
    def forward(self, outputs: Dict[str, torch.Tensor], targets: Sequence) -> List[Tuple[torch.Tensor, torch.Tensor]]:
        dev = outputs["pred_logits"].device
        tg = _MaskTargets(targets, dev)
        pi, ti = self.match_packed(outputs["pred_logits"].float().contiguous(), outputs["pred_masks"].float().contiguous(), tg)
        pi, ti, o = pi.cpu().long(), ti.cpu().long(), tg.off_host
        return [(pi[o[b]:o[b + 1]], ti[o[b]:o[b + 1]]) for b in range(len(targets))]

    __call__ = forward


class _MaskSetLossFn(torch.autograd.Function):
    """(loss_ce, loss_mask, loss_dice) of one prediction set, weighted: fx_mask_set_loss_f32; backward fx_mask_set_loss_bwd_f32 with the
    forward's random draws, matches and workspace (pair sums).  The Hungarian matches and the sample points carry no gradient, as in
    the reference (matcher under no_grad loss.py:661; point selection under no_grad :487-497)."""

    @staticmethod
    def forward(ctx, logits, pm, crit, tg, pi, ti, num_masks):
        lib = _lib.load()
        B, Q, K1 = logits.shape
        dev, P = logits.device, crit.num_points
        h, w = pm.shape[-2:]
        n_over = int(P * crit.oversample_ratio)
        n_extra = P - int(crit.importance_sample_ratio * P)
        r_over = crit.rand(max(tg.n, 1), n_over, 2, device=dev).float().contiguous() if tg.n else torch.zeros(1, n_over, 2, device=dev)
        r_extra = (crit.rand(max(tg.n, 1), n_extra, 2, device=dev).float().contiguous() if (tg.n and n_extra > 0)
                   else torch.zeros(1, max(n_extra, 1), 2, device=dev))
        ws = torch.empty(lib.fx_mask_set_loss_workspace_bytes(B, Q, tg.n, n_over) // 8 + 1, dtype=torch.float64, device=dev)
        out3 = torch.empty(3, dtype=torch.float32, device=dev)
        wd = crit.weight_dict
        args = (logits.data_ptr(), K1, pm.data_ptr(), h, w, tg.masks.data_ptr(), tg.is_u8, tg.H, tg.W, tg.labels.data_ptr(), tg.offsets.data_ptr(), tg.n,
                pi.data_ptr(), ti.data_ptr(), r_over.data_ptr(), n_over, r_extra.data_ptr(), n_extra, P, B, Q, K1 - 1, float(crit.eos_coef),
                float(num_masks), float(wd.get("loss_ce", 1.0)), float(wd.get("loss_mask", 1.0)), float(wd.get("loss_dice", 1.0)), ws.data_ptr(),
                ws.numel() * 8)
        check(lib.fx_mask_set_loss_f32(*args, out3.data_ptr(), _stream(dev)), "fx_mask_set_loss_f32")
        ctx.args, ctx.keep = args, (tg, pi, ti, r_over, r_extra, ws)
        ctx.save_for_backward(logits, pm)
        return out3

    @staticmethod
    def backward(ctx, g3):
        lib = _lib.load()
        logits, pm = ctx.saved_tensors
        g3 = g3.float().contiguous()
        dlogits = torch.empty_like(logits)
        dmasks = torch.zeros_like(pm)
        check(lib.fx_mask_set_loss_bwd_f32(*ctx.args, g3.data_ptr(), dlogits.data_ptr(), logits.shape[-1], dmasks.data_ptr(), _stream(logits.device)),
              "fx_mask_set_loss_bwd_f32")
        return dlogits, dmasks, None, None, None, None, None


class SetCriterion:
    def __init__(self, num_classes: int, matcher: MaskHungarianMatcher, weight_dict: Dict[str, float], losses=("labels", "masks"), eos_coef: float = 0.1,
                 num_points: int = 0, oversample_ratio: float = 3.0, importance_sample_ratio: float = 0.0, deep_supervision: bool = True,
                 loss_class_type: str = "ce_loss", cls_sigmoid: bool = False, rand: Optional[Callable] = None):
        if sorted(losses) != ["labels", "masks"] or loss_class_type != "ce_loss" or num_points <= 0:
            raise NotImplementedError("engine criterion covers losses=['labels','masks'], loss_class_type='ce_loss', num_points > 0 (registry models)")
        self.num_classes, self.matcher, self.weight_dict, self.eos_coef = num_classes, matcher, weight_dict, eos_coef
        self.num_points, self.oversample_ratio, self.importance_sample_ratio = num_points, oversample_ratio, importance_sample_ratio
        self.deep_supervision = deep_supervision
        self.rand = rand or matcher.rand

    def _one_set(self, out, tg: _MaskTargets, num_masks: float, fixed=None) -> torch.Tensor:
        logits, pm = out["pred_logits"].float().contiguous(), out["pred_masks"].float().contiguous()
        if fixed is None:
            with torch.no_grad():
                pi, ti = self.matcher.match_packed(logits.detach(), pm.detach(), tg)
        else:
            pi, ti = fixed
            for _ in range(logits.shape[0]):   # keep the stream of random draws aligned with the reference's order (the matcher's points)
                self.matcher.rand(1, self.matcher.num_points, 2, device=logits.device)
        out3 = _MaskSetLossFn.apply(logits, pm, self, tg, pi, ti, float(num_masks))
        self.last_matches = (pi, ti)
        return out3

    def forward(self, outputs: Dict, targets: Sequence, fixed_matches=None) -> Dict[str, torch.Tensor]:
        """SetCriterion.forward (loss.py:545-592).  ``fixed_matches``: optional per-set (pred_idx, tgt_idx) int32 device tensors in the
        packed target order (tests teacher-force the Hungarian matches with them)."""
        dev = outputs["pred_logits"].device
        tg = _MaskTargets(targets, dev)
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            num = torch.tensor([float(tg.n)], device=dev)
            torch.distributed.all_reduce(num)  # loss.py:559-561
            num_masks = max(float(num.item()) / torch.distributed.get_world_size(), 1.0)
        else:
            num_masks = max(float(tg.n), 1.0)   # single process: a host integer, the launch queue keeps running ahead
        losses = {}
        sets = [("", {k: v for k, v in outputs.items() if k != "aux_outputs"})]
        if self.deep_supervision:
            sets += [(f"_{i}", a) for i, a in enumerate(outputs.get("aux_outputs", []))]
        self.all_matches = []
        for j, (suffix, o) in enumerate(sets):
            l3 = self._one_set(o, tg, num_masks, None if fixed_matches is None else fixed_matches[j])
            self.all_matches.append(self.last_matches)
            losses[f"loss_ce{suffix}"], losses[f"loss_mask{suffix}"], losses[f"loss_dice{suffix}"] = l3[0], l3[1], l3[2]
        return losses

    __call__ = forward
