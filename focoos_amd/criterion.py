"""Host mirrors of the RT-DETR training criterion for the gfx950 kernels (forward values; no autograd this round):

  BoxHungarianMatcher.forward   focoos/models/fai_detr/modelling.py:693-758   -> fx_detr_match_cost_f32 + fx_lsa_f32
  SetCriterion.forward          focoos/models/fai_detr/modelling.py:553-612   -> fx_detr_set_loss_f32 per prediction set

Same argument / return structure as the reference (``outputs`` dict with pred_logits / pred_boxes / aux_outputs, targets
with ``labels`` / ``boxes``; matcher returns a list of (index_i, index_j) int64 tensors; criterion returns the dict of
weighted losses with ``_{i}`` suffixes for the auxiliary sets).  Unlike the reference nothing leaves the GPU: no
cost-matrix D2H copy, no host SciPy call, no ``.item()`` sync (num_boxes is computed on the host from the target list,
as the reference does before its all-reduce)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import check


class _PinnedRing:
    """Small host->device transfers without stalling the launch queue: a pageable-memory H2D copy blocks the host until every
    kernel queued before it has run (measured: ~11 ms per training step), a copy from pinned memory does not.  A ring of
    pinned staging buffers keeps the source alive until the asynchronous copy has certainly been consumed."""

    def __init__(self, slots: int = 64, capacity: int = 8192):
        self.bufs = [torch.empty(capacity, dtype=torch.int32).pin_memory() for _ in range(slots)] if torch.cuda.is_available() else []
        self.i = 0

    def to_device(self, arr: np.ndarray, device) -> torch.Tensor:
        arr = np.ascontiguousarray(arr, dtype=np.int32)
        if not self.bufs or arr.size > self.bufs[0].numel():
            return torch.from_numpy(arr).to(device)
        b = self.bufs[self.i]
        self.i = (self.i + 1) % len(self.bufs)
        b[: arr.size].numpy()[...] = arr.reshape(-1)
        return b[: arr.size].to(device, non_blocking=True).view(arr.shape)


_RING = None


def h2d_i32(arr: np.ndarray, device) -> torch.Tensor:
    global _RING
    if _RING is None:
        _RING = _PinnedRing()
    return _RING.to_device(arr, device)


class _Targets:
    """Packed targets in HBM: labels i32 [sumT], boxes f32 [sumT,4], offsets i32 [B+1]."""

    def __init__(self, targets: Sequence, device):
        sizes = [int(len(t.labels)) for t in targets]
        self.off_host = np.zeros(len(targets) + 1, np.int32)
        self.off_host[1:] = np.cumsum(sizes)
        self.n = int(self.off_host[-1])
        self.tmax = max(sizes + [0])
        self.offsets = h2d_i32(self.off_host, device)
        if self.n:
            self.labels = torch.cat([t.labels for t in targets]).to(device=device, dtype=torch.int32).contiguous()
            self.boxes = torch.cat([t.boxes for t in targets]).to(device=device, dtype=torch.float32).contiguous()
        else:
            self.labels = torch.zeros(1, dtype=torch.int32, device=device)
            self.boxes = torch.zeros(1, 4, device=device)


_LSA_STATUS: Dict[torch.device, torch.Tensor] = {}


def lsa_status(device) -> torch.Tensor:
    """The per-device sticky status word of the Hungarian solves (fx_lsa_status_f32: bit 0 = an assignment was infeasible, bit 1 = NaN / -inf costs)."""
    device = torch.device(device)
    if device not in _LSA_STATUS:
        _LSA_STATUS[device] = torch.zeros(1, dtype=torch.int32, device=device)
    return _LSA_STATUS[device]


def raise_if_infeasible(device, all_ranks: bool = False) -> None:
    """Reads (one 4-byte D2H copy: a synchronisation - call it where the host waits anyway) and clears the status word; raises what
    SciPy's ``linear_sum_assignment`` raises inside the reference matcher (fai_detr/modelling.py:749-750) when the costs are inf / NaN.
    ``all_ranks``: under data parallelism the word is first OR-all-reduced, so that EVERY rank raises in the same step - a rank raising
    alone would leave the others blocked in the next gradient all-reduce (every rank must make this call at the same point)."""
    import torch.distributed as dist

    st = _LSA_STATUS.get(torch.device(device))
    multi = all_ranks and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if multi:
        st = lsa_status(device)          # a rank that never matched still takes part in the collective
        dist.all_reduce(st, op=dist.ReduceOp.BOR)   # a bitfield: bitwise OR (bit 0 infeasible, bit 1 invalid entries - different ranks may set different bits)
    if st is not None:
        bits = int(st.item())
        if bits:
            st.zero_()
            raise ValueError("matrix contains invalid numeric entries" if bits & 2 else "cost matrix is infeasible")


class BoxHungarianMatcher:
    def __init__(self, cost_class: float = 2, cost_bbox: float = 5, cost_giou: float = 2, use_focal_loss: bool = True, alpha: float = 0.25,
                 gamma: float = 2.0):
        assert cost_class != 0 or cost_bbox != 0 or cost_giou != 0, "all costs cant be 0"
        if not use_focal_loss:
            raise NotImplementedError("only the focal-cost branch (matcher_use_focal_loss=True, all registry models) is implemented")
        self.cost_class, self.cost_bbox, self.cost_giou, self.alpha, self.gamma = cost_class, cost_bbox, cost_giou, alpha, gamma

    def match_packed(self, logits: torch.Tensor, boxes: torch.Tensor, tg: _Targets):
        lib = _lib.load()
        B, Q, K = logits.shape
        dev = logits.device
        logits, boxes = logits.float().contiguous(), boxes.float().contiguous()
        # Identity-like defaults instead of uninitialised memory: if the assignment is infeasible (NaN / inf costs after a diverged
        # step) the solver leaves an image's slots untouched, and the criterion would otherwise index boxes[slot, garbage].  The
        # reference raises from SciPy in that case; here the kernel sets the device status word and the host raises the same error at its
        # next synchronisation point (raise_if_infeasible: the matcher's public forward, TrainStep.check - every `check_every` steps - and the trainer's log points).
        pi = torch.zeros(max(tg.n, 1), dtype=torch.int32, device=dev)
        ti = torch.zeros(max(tg.n, 1), dtype=torch.int32, device=dev)
        if tg.n:
            cost = torch.empty(B, Q, tg.tmax, dtype=torch.float32, device=dev)
            st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            check(lib.fx_detr_match_cost_f32(logits.data_ptr(), K, boxes.data_ptr(), tg.labels.data_ptr(), tg.boxes.data_ptr(), tg.offsets.data_ptr(), B, Q,
                                             K, tg.tmax, float(self.cost_class), float(self.cost_bbox), float(self.cost_giou), float(self.alpha),
                                             float(self.gamma), cost.data_ptr(), st), "fx_detr_match_cost_f32")
            check(lib.fx_lsa_status_f32(cost.data_ptr(), B, Q, tg.tmax, tg.offsets.data_ptr(), pi.data_ptr(), ti.data_ptr(), lsa_status(dev).data_ptr(), st),
                  "fx_lsa_status_f32")
        return pi, ti

    def match_packed_sets(self, logits_list, boxes_list, tg: _Targets):
        """match_packed for S prediction sets of one shape against the SAME targets (the main output + the auxiliary decoder / encoder
        sets of SetCriterion.forward, modelling.py:572-611) as ONE cost launch and ONE assignment launch over S*B virtual images: the
        solver is one wave per image, so S launches of B waves each leave the GPU empty S times in a row (7 x 88 us per RT-DETR step).
        Returns [(pi, ti)] * S - the slices are exactly what S match_packed calls return (indices are local to an image)."""
        S = len(logits_list)
        if S == 1 or tg.n == 0:
            return [self.match_packed(l, b, tg) for l, b in zip(logits_list, boxes_list)]
        lib = _lib.load()
        B, Q, K = logits_list[0].shape
        dev = logits_list[0].device
        logits = torch.stack([l.float() for l in logits_list]).contiguous()       # [S,B,Q,K]
        boxes = torch.stack([b.float() for b in boxes_list]).contiguous()
        sup = getattr(tg, "_super", None)
        if sup is None or sup[0] != S:
            off = np.concatenate([tg.off_host[:-1] + s * tg.n for s in range(S)] + [np.array([S * tg.n], np.int32)]).astype(np.int32)
            sup = (S, h2d_i32(off, dev), tg.labels.repeat(S).contiguous(), tg.boxes.repeat(S, 1).contiguous())
            tg._super = sup
        _, offsets, labels, tboxes = sup
        pi = torch.zeros(S * tg.n, dtype=torch.int32, device=dev)
        ti = torch.zeros(S * tg.n, dtype=torch.int32, device=dev)
        cost = torch.empty(S * B, Q, tg.tmax, dtype=torch.float32, device=dev)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        check(lib.fx_detr_match_cost_f32(logits.data_ptr(), K, boxes.data_ptr(), labels.data_ptr(), tboxes.data_ptr(), offsets.data_ptr(), S * B, Q,
                                         K, tg.tmax, float(self.cost_class), float(self.cost_bbox), float(self.cost_giou), float(self.alpha),
                                         float(self.gamma), cost.data_ptr(), st), "fx_detr_match_cost_f32")
        check(lib.fx_lsa_status_f32(cost.data_ptr(), S * B, Q, tg.tmax, offsets.data_ptr(), pi.data_ptr(), ti.data_ptr(), lsa_status(dev).data_ptr(), st),
              "fx_lsa_status_f32")
        return [(pi[s * tg.n:(s + 1) * tg.n], ti[s * tg.n:(s + 1) * tg.n]) for s in range(S)]

    @torch.no_grad()
    def forward(self, outputs: Dict[str, torch.Tensor], targets: Sequence) -> List[Tuple[torch.Tensor, torch.Tensor]]:
        tg = _Targets(targets, outputs["pred_logits"].device)
        pi, ti = self.match_packed(outputs["pred_logits"], outputs["pred_boxes"], tg)
        raise_if_infeasible(outputs["pred_logits"].device)     # the reference raises here (SciPy); this call synchronises anyway (.cpu() below)
        pi, ti, o = pi.cpu().long(), ti.cpu().long(), tg.off_host
        return [(pi[o[b]:o[b + 1]], ti[o[b]:o[b + 1]]) for b in range(len(targets))]

    __call__ = forward


class SetCriterion:
    def __init__(self, num_classes: int, matcher: BoxHungarianMatcher, weight_dict: Dict[str, float], losses=("vfl", "boxes"), focal_alpha: float = 0.75,
                 focal_gamma: float = 2.0, deep_supervision: bool = True, world_size: int = 1):
        if sorted(losses) != ["boxes", "vfl"]:
            raise NotImplementedError("criterion_losses must be ['vfl', 'boxes'] (the registry configuration)")
        self.num_classes, self.matcher, self.weight_dict = num_classes, matcher, weight_dict
        self.focal_alpha, self.focal_gamma, self.deep_supervision, self.world_size = focal_alpha, focal_gamma, deep_supervision, world_size

    def _one_set(self, out, tg: _Targets, num_boxes: float) -> torch.Tensor:
        lib = _lib.load()
        logits, boxes = out["pred_logits"].float().contiguous(), out["pred_boxes"].float().contiguous()
        B, Q, K = logits.shape
        dev = logits.device
        pi, ti = self.matcher.match_packed(logits, boxes, tg)
        ws = torch.empty(lib.fx_detr_set_loss_workspace_bytes(B, Q, tg.n) // 8 + 1, dtype=torch.float64, device=dev)
        out3 = torch.empty(3, dtype=torch.float32, device=dev)
        check(lib.fx_detr_set_loss_f32(logits.data_ptr(), K, boxes.data_ptr(), tg.labels.data_ptr(), tg.boxes.data_ptr(), tg.offsets.data_ptr(), pi.data_ptr(),
                                       ti.data_ptr(), B, Q, K, tg.n, float(num_boxes), float(self.focal_alpha), float(self.focal_gamma),
                                       float(self.weight_dict.get("loss_vfl", 1.0)), float(self.weight_dict.get("loss_bbox", 1.0)),
                                       float(self.weight_dict.get("loss_giou", 1.0)), ws.data_ptr(), out3.data_ptr(),
                                       C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "fx_detr_set_loss_f32")
        return out3

    @torch.no_grad()
    def forward(self, outputs: Dict, targets: Sequence) -> Dict[str, torch.Tensor]:
        dev = outputs["pred_logits"].device
        tg = _Targets(targets, dev)
        num = torch.tensor([float(tg.n)], device=dev)
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.all_reduce(num)  # modelling.py:568-570 (the 4-byte num_boxes all-reduce, C3)
            num = num / torch.distributed.get_world_size()
        num_boxes = max(float(num.item()), 1.0)
        losses = {}
        sets = [("", {k: v for k, v in outputs.items() if k != "aux_outputs"})]
        if self.deep_supervision:
            sets += [(f"_{i}", a) for i, a in enumerate(outputs.get("aux_outputs", []))]
        for suffix, o in sets:
            l3 = self._one_set(o, tg, num_boxes)
            losses[f"loss_vfl{suffix}"], losses[f"loss_bbox{suffix}"], losses[f"loss_giou{suffix}"] = l3[0], l3[1], l3[2]
        return losses

    __call__ = forward
