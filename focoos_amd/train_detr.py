"""RT-DETR as a trainable HIP autograd graph (BASELINE config 4 minus the batch-statistics BatchNorm):
backbone + hybrid encoder (``train_nn``) + ``TransformerPredictor`` in training mode + the set criterion, i.e. what
``FAIDetr.forward(images, targets)`` computes under ``model.train()`` (fai_detr/modelling.py:1344-1358 with
``TransformerPredictor.forward`` :1234-1263, ``TransformerDecoder.forward`` :969-1020, ``SetCriterion.forward`` :553-612).

Kernels: every GEMM / conv / attention / normalisation / deformable sampling / VFL loss, forward and backward, is a
libfocoos_amd.so call.  PyTorch supplies the autograd tape and a handful of *glue* ops on small tensors, listed here so that
nothing hides: ``torch.cat`` of the three memory levels, the 0/1 valid-mask multiply, ``softmax`` over the 12 sampling
weights, sampling-location / sigmoid / inverse-sigmoid arithmetic on [B,300,4], ``gather`` of the encoder top-k rows,
max/top-k index selection (no gradient), and the L1 / GIoU losses on the <= sum(T) matched box pairs.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import nn

from . import _lib
from ._lib import check
from .criterion import h2d_i32, BoxHungarianMatcher, _Targets
from .train import ValueGradSink, ms_deform_attn_core, ms_deform_attn_grouped, ms_deform_attn_grouped_raw
from .train_nn import (ConvNormLayer, HybridEncoder, LayerNorm, Linear, MultiheadAttention, ResNetVd, _AddFn, _Layers, _LinearGroupFn, _PackedLinearGroup,
                       _stream)


class MLP(nn.Module):
    """focoos/nn/layers/base.py:31-61 (ReLU between layers)."""

    def __init__(self, lib, cin, hidden, cout, n):
        super().__init__()
        dims = [cin] + [hidden] * (n - 1) + [cout]
        self.layers = nn.ModuleList([Linear(lib, dims[i], dims[i + 1], act="relu" if i < n - 1 else None) for i in range(n)])

    def forward(self, x):
        for l in self.layers:
            x = l(x)
        return x


INV_SIGMOID_EPS = 1e-5   # focoos/nn/layers/functional.py:4
# encoder score / box heads of the training graph on the selected rows only (TransformerPredictor.forward); 0: over all tokens, then gather
SELECT_ROWS = [os.environ.get("FX_ENC_SELECT_ROWS", "1") != "0"]
RAW_MSDA = [os.environ.get("FX_MSDA_RAW", "1") != "0"]   # decoder cross-attention from the raw projections (train._MSDAGroupRawFunction); 0: the torch-op form


class _BoxRefineFn(torch.autograd.Function):
    """sigmoid(delta + inverse_sigmoid(ref)) - the iterative box refinement of the decoder (fai_detr/modelling.py:1003, 1010) - as one
    launch each way (fx_box_refine_f32) instead of ~8 elementwise launches forward and ~15 backward.  delta bf16 [..., 4] (the bbox
    head's output), ref fp32 [..., 4] (may be detached); returns fp32."""

    @staticmethod
    def forward(ctx, delta, ref):
        lib = _lib.load()
        delta, ref = delta.contiguous(), ref.contiguous()
        box = torch.empty_like(ref)
        check(lib.fx_box_refine_f32(delta.data_ptr(), ref.data_ptr(), box.data_ptr(), ref.numel(), INV_SIGMOID_EPS, _stream(ref.device)),
              "fx_box_refine_f32")
        ctx.save_for_backward(box, ref)
        return box

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        box, ref = ctx.saved_tensors
        g = g.contiguous()
        d_delta = torch.empty(box.shape, dtype=_lib.act_dtype(), device=box.device)
        d_ref = torch.empty_like(box) if ctx.needs_input_grad[1] else None
        check(lib.fx_box_refine_bwd_f32(g.data_ptr(), box.data_ptr(), ref.data_ptr(), d_delta.data_ptr(), d_ref.data_ptr() if d_ref is not None else None,
                                        box.numel(), INV_SIGMOID_EPS, _stream(box.device)), "fx_box_refine_bwd_f32")
        return d_delta, d_ref


class MSDeformableAttention(nn.Module):
    """fai_detr/modelling.py:760-884 (4-d reference branch); the sampling core is fx_msda_f32_fwd/bwd."""

    def __init__(self, lib, c=256, heads=8, levels=3, points=4):
        super().__init__()
        self.c, self.h, self.l, self.p = c, heads, levels, points
        self.sampling_offsets = Linear(lib, c, heads * levels * points * 2)
        self.attention_weights = Linear(lib, c, heads * levels * points)
        self.value_proj = Linear(lib, c, c)
        self.output_proj = Linear(lib, c, c)

    def forward(self, query, ref_points, memory, shapes, residual, value_all=None, sink=None, g=0):
        """``value_all`` / ``sink`` / ``g``: the value projections of all decoder layers computed together by the predictor (bf16
        [B,S,G*256]; this layer reads column slice g); without them the layer projects ``memory`` itself (the reference's form)."""
        B, Q, _ = query.shape
        S = memory.shape[1]
        if (value_all is not None and RAW_MSDA[0] and ref_points.shape[-1] == 4 and ref_points.shape[2] == 1 and not ref_points.requires_grad
                and query.dtype == _lib.act_dtype()):
            # softmax + location arithmetic inside the sampling node (fx_msda_prep_bf16): ~15 elementwise launches per layer less
            out = ms_deform_attn_grouped_raw(value_all, sink, g, shapes, self.sampling_offsets(query), self.attention_weights(query),
                                             ref_points.reshape(B, Q, 4), self.h, self.l, self.p)
            return self.output_proj(out, residual=residual)
        off = self.sampling_offsets(query).float().view(B, Q, self.h, self.l, self.p, 2)
        aw = torch.softmax(self.attention_weights(query).float().view(B, Q, self.h, self.l * self.p), -1).view(B, Q, self.h, self.l, self.p)
        loc = ref_points[:, :, None, :, None, :2] + off / self.p * ref_points[:, :, None, :, None, 2:] * 0.5
        if value_all is not None:
            out = ms_deform_attn_grouped(value_all, sink, g, shapes, loc, aw)
        else:
            value = self.value_proj(memory).view(B, S, self.h, self.c // self.h)
            out = ms_deform_attn_core(value, shapes, loc, aw).to(_lib.act_dtype())
        return self.output_proj(out, residual=residual)


class TransformerDecoderLayer(nn.Module):
    """fai_detr/modelling.py:887-958."""

    def __init__(self, lib, c=256, ffn=1024):
        super().__init__()
        self.lib = lib
        self.self_attn = MultiheadAttention(lib, c)
        self.norm1 = LayerNorm(lib, c)
        self.cross_attn = MSDeformableAttention(lib, c)
        self.norm2 = LayerNorm(lib, c)
        self.linear1 = Linear(lib, c, ffn, act="relu")
        self.linear2 = Linear(lib, ffn, c)
        self.norm3 = LayerNorm(lib, c)

    def forward(self, tgt, ref_input, memory, shapes, qpos, value_all=None, sink=None, g=0):
        qk = _AddFn.apply(tgt, qpos, self.lib)
        tgt = self.norm1(self.self_attn(qk, qk, tgt, residual=tgt))
        tgt = self.norm2(self.cross_attn(_AddFn.apply(tgt, qpos, self.lib), ref_input, memory, shapes, residual=tgt, value_all=value_all, sink=sink, g=g))
        return self.norm3(self.linear2(self.linear1(tgt), residual=tgt))


class _Seq2(nn.Module):
    """nn.Sequential(Linear, LayerNorm) with keys ``0`` / ``1`` (enc_output)."""

    def __init__(self, a, b):
        super().__init__()
        self.add_module("0", a)
        self.add_module("1", b)
        object.__setattr__(self, "_a", a)
        object.__setattr__(self, "_b", b)

    def forward(self, x):
        return self._b(self._a(x))


class TransformerPredictor(nn.Module):
    """fai_detr/modelling.py:1023-1263, training branch (all decoder layers + the encoder top-k set are supervised)."""

    def __init__(self, lib, nc: int, c=256, nq=300, nl=6, ffn=1024, in_dim=None):
        super().__init__()
        self.lib, self.nc, self.c, self.nq, self.nl = lib, nc, c, nq, nl
        # in_dim: channels of the encoder's maps (pixel_decoder_out_dim: 256, or 128 for fai-detr-m-coco) projected to the decoder width
        self.input_proj = nn.ModuleList([ConvNormLayer(lib, in_dim or c, c, 1, 1, None) for _ in range(3)])
        self.decoder = _Layers([TransformerDecoderLayer(lib, c, ffn) for _ in range(nl)])
        self.query_pos_head = MLP(lib, 4, 2 * c, c, 2)
        self.enc_output = _Seq2(Linear(lib, c, c), LayerNorm(lib, c))
        self.enc_score_classifier = Linear(lib, c, nc)
        self.enc_bbox_classifier = MLP(lib, c, c, 4, 3)
        self.dec_score_classifier = nn.ModuleList([Linear(lib, c, nc) for _ in range(nl)])
        self.dec_bbox_classifier = nn.ModuleList([MLP(lib, c, c, 4, 3) for _ in range(nl)])
        self._anchor_cache = {}
        self._value_group = _PackedLinearGroup()   # the six value projections as one GEMM (what the inference plan calls value_all)
        self._sink = None
        self._sel_state = None

    def _anchors(self, shapes, dev, grid_size=0.05, eps=1e-2):
        key = (tuple(shapes), dev)
        if key not in self._anchor_cache:
            out = []
            for lvl, (h, w) in enumerate(shapes):
                gy, gx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
                xy = (torch.stack([gx, gy], -1) + 0.5) / torch.tensor([w, h], dtype=torch.float32)
                wh = torch.ones_like(xy) * grid_size * (2.0 ** (2 - lvl))
                out.append(torch.cat([xy, wh], -1).reshape(h * w, 4))
            a = torch.cat(out, 0)
            valid = ((a > eps) & (a < 1 - eps)).all(-1, keepdim=True)
            a = torch.where(valid, torch.log(a / (1 - a)), torch.zeros_like(a))
            self._anchor_cache[key] = (a.to(dev), valid.to(dev))
        return self._anchor_cache[key]

    def _selection_scores(self, memory: torch.Tensor, valid: torch.Tensor) -> torch.Tensor:
        """max over the classes of enc_score_classifier(enc_output(valid * memory)) for every token, f32 [B, S] (no gradient).  The fused
        launch needs 256 channels and <= 384 classes; other shapes take the layers one by one."""
        B, S, c = memory.shape
        lin, norm, cls = self.enc_output._a, self.enc_output._b, self.enc_score_classifier
        ncp = (self.nc + 127) // 128 * 128
        if c != 256 or ncp > 384:
            om = norm(lin(memory * valid.to(memory.dtype)))
            return cls(om).float().max(-1).values
        lib, dev = self.lib, memory.device
        lin._pack.sync(lib, lin.weight, lin.bias, 0, c)
        cls._pack.sync(lib, cls.weight, cls.bias, 0, self.nc)
        st = self._sel_state
        ver = (lin._pack.ver, cls._pack.ver)
        if st is None or st["dev"] != dev:
            st = self._sel_state = {"dev": dev, "ver": None, "w2_frag": torch.empty(ncp * c, dtype=_lib.act_dtype(), device=dev),
                                    "b2": torch.full((ncp,), -3e38, dtype=torch.float32, device=dev),
                                    "valid": None, "valid_src": None}
        # once per optimizer step: the class weights in fragment order, the bias with -3e38 in the padding (inside a capture always - the
        # launches must be part of the graph that replays the step)
        if st["ver"] != ver or torch.cuda.is_current_stream_capturing():
            assert cls._pack.w_fwd.numel() == ncp * c   # the bf16 image is already [ncp, 256] with zero padding rows
            check(lib.fx_pack_frag_bf16(cls._pack.w_fwd.data_ptr(), st["w2_frag"].data_ptr(), ncp, c, _stream(dev)), "fx_pack_frag_bf16")
            st["b2"][: self.nc].copy_(cls.bias.detach())
            st["ver"] = ver
        if st["valid_src"] is not valid:   # the anchors' validity mask of THIS set of level shapes (cached per shapes in _anchors: identity is enough)
            st["valid"], st["valid_src"] = valid.view(-1).to(torch.uint8).contiguous(), valid
        mem = memory.contiguous()
        om = torch.empty(B * S, c, dtype=_lib.act_dtype(), device=dev)
        scores = torch.empty(B, S, dtype=torch.float32, device=dev)
        check(lib.fx_enc_score_head_bf16(mem.data_ptr(), c, st["valid"].data_ptr(), S, lin._pack.w_fwd_frag.data_ptr(), lin.bias.data_ptr(),
                                         norm.weight.data_ptr(), norm.bias.data_ptr(), C.c_float(1e-5), st["w2_frag"].data_ptr(), st["b2"].data_ptr(),
                                         ncp, om.data_ptr(), c, scores.data_ptr(), B * S, _stream(dev)), "fx_enc_score_head_bf16")
        return scores

    def forward(self, feats: List[torch.Tensor], forced_topk: Optional[torch.Tensor] = None):
        B = feats[0].shape[0]
        proj = [p(f) for p, f in zip(self.input_proj, feats)]
        shapes = [(t.shape[1], t.shape[2]) for t in proj]
        memory = torch.cat([t.reshape(B, -1, self.c) for t in proj], 1)  # [B, S, 256]
        anchors, valid = self._anchors(shapes, memory.device)
        if not SELECT_ROWS[0]:
            # the reference's literal order (modelling.py:1202-1232): both heads over all S tokens, then the gathers - kept switchable as the
            # A/B form of the row-selected path below (tests/test_gpu_train_detr.py::test_encoder_heads_on_selected_rows_equal_all_rows)
            mem_v = memory * valid.to(memory.dtype)
            output_memory = self.enc_output(mem_v)
            enc_class = self.enc_score_classifier(output_memory)                    # [B, S, nc] bf16
            enc_coord_unact = self.enc_bbox_classifier(output_memory).float() + anchors
            with torch.no_grad():
                topk_ind = torch.topk(enc_class.float().max(-1).values, self.nq, dim=1).indices if forced_topk is None else forced_topk
            ref_unact = enc_coord_unact.gather(1, topk_ind.unsqueeze(-1).expand(-1, -1, 4))
            enc_topk_logits = enc_class.gather(1, topk_ind.unsqueeze(-1).expand(-1, -1, self.nc))
            target = output_memory.gather(1, topk_ind.unsqueeze(-1).expand(-1, -1, self.c)).detach()
        else:
            # Every operation between `memory` and the three gathers is ROW-LOCAL (mask, Linear, LayerNorm, Linear / MLP) and only the nq selected
            # rows of an image reach a loss, so the gathers commute with them: the selection scores of all S tokens come from the inference
            # plan's fused launch (fx_enc_score_head_bf16: enc_output -> enc_score -> max, no [B,S,nc] tensor, nothing saved), and the
            # differentiable heads run on the [B, nq] selected rows - same values, same gradients (rows that are not selected have zero
            # gradient in the reference's graph too), 1/28 of the rows in the forward and the backward of four layers.
            with torch.no_grad():
                topk_ind = torch.topk(self._selection_scores(memory, valid), self.nq, dim=1).indices if forced_topk is None else forced_topk
            sel = memory.gather(1, topk_ind.unsqueeze(-1).expand(-1, -1, self.c)) * valid.view(-1)[topk_ind].unsqueeze(-1).to(memory.dtype)
            output_sel = self.enc_output(sel)                                       # [B, nq, 256]
            enc_topk_logits = self.enc_score_classifier(output_sel)
            ref_unact = self.enc_bbox_classifier(output_sel).float() + anchors[topk_ind]
            target = output_sel.detach()
        enc_topk_bboxes = torch.sigmoid(ref_unact)
        out = target
        ref_detach = torch.sigmoid(ref_unact.detach())
        ref = ref_detach
        logits, boxes = [], []
        # every layer's cross-attention projects the same memory: one GEMM with 6 x 256 outputs; the layers read column slices of it and
        # accumulate the value gradient into slices of one fp32 buffer (train.ValueGradSink)
        if self._sink is not None and self._sink.count:
            raise _lib.FocoosAmdError("the previous backward pass did not reach every decoder layer's deformable attention: value gradient incomplete")
        vps = [l.cross_attn.value_proj for l in self.decoder.layers]
        value_all = _LinearGroupFn.apply(memory, self._value_group, self.lib, *[v.weight for v in vps], *[v.bias for v in vps])
        sink = self._sink = ValueGradSink(len(vps))
        for i, layer in enumerate(self.decoder.layers):
            qpos = self.query_pos_head(ref_detach.to(_lib.act_dtype()))
            out = layer(out, ref_detach.unsqueeze(2), memory, shapes, qpos, value_all=value_all, sink=sink, g=i)
            delta = self.dec_bbox_classifier[i](out)
            inter = _BoxRefineFn.apply(delta, ref_detach)
            logits.append(self.dec_score_classifier[i](out))
            boxes.append(inter if i == 0 else _BoxRefineFn.apply(delta, ref))
            ref, ref_detach = inter, inter.detach()
        return {"pred_logits": logits[-1], "pred_boxes": boxes[-1],
                "aux_outputs": [{"pred_logits": a, "pred_boxes": b} for a, b in zip(logits[:-1], boxes[:-1])]
                + [{"pred_logits": enc_topk_logits, "pred_boxes": enc_topk_bboxes}], "topk_ind": topk_ind}


# ------------------------------------------------------------------------------------------------ differentiable criterion
class _VFLFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, cls, score, alpha, gamma, scale):
        lib = _lib.load()
        B, Q, K = logits.shape
        lg = logits.contiguous()
        loss = torch.zeros(1, dtype=torch.float32, device=lg.device)
        dl = torch.empty_like(lg)
        check(lib.fx_vfl_loss_bf16(lg.data_ptr(), K, cls.data_ptr(), score.data_ptr(), float(alpha), float(gamma), float(scale), loss.data_ptr(),
                                   dl.data_ptr(), K, B * Q, K, _stream(lg.device)), "fx_vfl_loss_bf16")
        ctx.save_for_backward(dl)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g.to(dl.dtype), None, None, None, None, None


class _BoxLossFn(torch.autograd.Function):
    """loss_bbox, loss_giou of one prediction set (weights and 1 / num_boxes applied) through fx_detr_box_loss_f32; also fills the
    per-query VFL targets ``cls`` / ``score`` (no gradient: the IoU target is detached in the reference, modelling.py:474-476)."""

    @staticmethod
    def forward(ctx, boxes, tg: _Targets, pi, ti, K, scale_bbox, scale_giou, cls, score):
        lib = _lib.load()
        B, Q, _ = boxes.shape
        bx = boxes.contiguous()
        dev = bx.device
        loss2 = torch.empty(2, dtype=torch.float32, device=dev)
        pair_grad = torch.empty(max(tg.n, 1), 8, dtype=torch.float32, device=dev)
        check(lib.fx_detr_box_loss_f32(bx.data_ptr(), tg.labels.data_ptr(), tg.boxes.data_ptr(), tg.offsets.data_ptr(), pi.data_ptr(), ti.data_ptr(),
                                       B, Q, K, tg.n, float(scale_bbox), float(scale_giou), cls.data_ptr(), score.data_ptr(), loss2.data_ptr(),
                                       pair_grad.data_ptr(), _stream(dev)), "fx_detr_box_loss_f32")
        ctx.tg, ctx.pi, ctx.shape = tg, pi, (B, Q)
        ctx.save_for_backward(pair_grad)
        return loss2[0], loss2[1]

    @staticmethod
    def backward(ctx, g_bbox, g_giou):
        lib = _lib.load()
        (pair_grad,) = ctx.saved_tensors
        B, Q = ctx.shape
        dev = pair_grad.device
        dboxes = torch.empty(B, Q, 4, dtype=torch.float32, device=dev)
        g1 = None if g_bbox is None else g_bbox.float().contiguous()
        g2 = None if g_giou is None else g_giou.float().contiguous()
        check(lib.fx_detr_box_loss_bwd_f32(pair_grad.data_ptr(), ctx.tg.offsets.data_ptr(), ctx.pi.data_ptr(), B, Q, ctx.tg.n,
                                           None if g1 is None else g1.data_ptr(), None if g2 is None else g2.data_ptr(), dboxes.data_ptr(),
                                           _stream(dev)), "fx_detr_box_loss_bwd_f32")
        return dboxes, None, None, None, None, None, None, None, None


class SetCriterionTrain(nn.Module):
    """SetCriterion.forward (fai_detr/modelling.py:553-612) with gradients: Hungarian matching per prediction set on the GPU
    (fx_detr_match_cost_f32 + fx_lsa_f32), VFL through fx_vfl_loss_bf16, box losses on the matched pairs."""

    def __init__(self, nc, weight_dict=None, alpha=0.75, gamma=2.0):
        super().__init__()
        self.nc = nc
        self.matcher = BoxHungarianMatcher()
        self.w = weight_dict or {"loss_vfl": 1.0, "loss_bbox": 5.0, "loss_giou": 2.0}
        self.alpha, self.gamma = alpha, gamma
        self.register_buffer("empty_weight", torch.ones(nc + 1))  # checkpoint key head.criterion.empty_weight

    def _one_set(self, out, tg: _Targets, num_boxes, fixed=None):
        logits, boxes = out["pred_logits"], out["pred_boxes"].float()
        B, Q, K = logits.shape
        dev = logits.device
        if fixed is None:
            pi, ti = self.matcher.match_packed(logits.detach(), boxes.detach(), tg)
        else:
            pi, ti = (t.to(torch.int32).contiguous() for t in fixed)
        losses = {}
        cls = torch.empty(B * Q, dtype=torch.int32, device=dev)      # filled by the kernel: matched label or K (no object)
        score = torch.empty(B * Q, dtype=torch.float32, device=dev)  # IoU of the matched pair or 0
        losses["loss_bbox"], losses["loss_giou"] = _BoxLossFn.apply(boxes, tg, pi, ti, K, self.w["loss_bbox"] / num_boxes,
                                                                    self.w["loss_giou"] / num_boxes, cls, score)
        losses["loss_vfl"] = _VFLFn.apply(logits, cls, score, self.alpha, self.gamma, self.w["loss_vfl"] / num_boxes)
        return losses, (pi, ti)

    def forward(self, outputs, targets: Sequence, fixed_matches=None):
        dev = outputs["pred_logits"].device
        tg = _Targets(targets, dev)
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            num = torch.tensor([float(tg.n)], device=dev)
            torch.distributed.all_reduce(num)    # modelling.py:568-570 (4-byte all-reduce; its .item() is the reference's sync too)
            num_boxes = max(float(num.item()) / torch.distributed.get_world_size(), 1.0)
        else:
            num_boxes = max(float(tg.n), 1.0)     # single process: a host integer - no device round trip, the launch queue keeps running ahead
        sets = [("", outputs)] + [(f"_{i}", a) for i, a in enumerate(outputs.get("aux_outputs", []))]
        if fixed_matches is None and os.environ.get("FX_LSA_BATCHED", "1") != "0" and len({tuple(o["pred_logits"].shape) for _, o in sets}) == 1:
            # the assignments of all prediction sets in one launch (they are independent, and each is only B waves of work)
            fixed_matches = self.matcher.match_packed_sets([o["pred_logits"].detach() for _, o in sets], [o["pred_boxes"].detach() for _, o in sets], tg)
        losses, matches = {}, []
        for j, (suffix, o) in enumerate(sets):
            l, m = self._one_set(o, tg, num_boxes, None if fixed_matches is None else fixed_matches[j])
            matches.append(m)
            for k, v in l.items():
                losses[k + suffix] = v
        self.last_matches = matches
        return losses


class FAIDetrTrainable(nn.Module):
    """Reference-compatible parameter tree (``pixel_decoder.backbone.*``, ``pixel_decoder.*``, ``head.predictor.*``,
    ``head.criterion.empty_weight``) whose forward(images, targets) returns the dict of weighted losses."""

    def __init__(self, config: Dict, norm: str = "FrozenBN"):
        """``norm``: "FrozenBN" (the reference's freeze_bn: running statistics, fixed affine), "BN" (batch statistics under
        .train(), trainable affine, running-statistics updates) or "SyncBN" (BN with the statistics all-reduced over the
        data-parallel group - what the reference converts to for multi-GPU training)."""
        super().__init__()
        if not torch.cuda.is_available():
            raise _lib.FocoosAmdError("focoos_amd needs a ROCm GPU (gfx950); no CPU fallback exists")
        lib = _lib.load()
        self.config = dict(config)
        nc = int(config["num_classes"])
        bb = config["backbone_config"]
        mean, std = config.get("pixel_mean", (123.675, 116.28, 103.53)), config.get("pixel_std", (58.395, 57.12, 57.375))
        fd, n_enc = int(config.get("pixel_decoder_feat_dim", 256)), int(config.get("pixel_decoder_num_encoder_layers", 1))
        if fd not in (128, 256) or int(config.get("pixel_decoder_out_dim", fd)) != fd or (n_enc > 0 and fd != 256):
            raise _lib.FocoosAmdError("hybrid encoder: 256 channels, or 128 without the AIFI layer (attention / LayerNorm kernels: 8 heads of 32 channels)")
        if bb.get("model_type") == "stdc":      # fai-detr-m-coco: the BiSeNetFormer backbone under the hybrid encoder
            from .train_bf import STDC

            base = int(bb.get("base", 64))
            backbone = STDC(lib, base, tuple(bb.get("layers", (4, 5, 3))), mean, std)
            in_channels = (base * 4, base * 8, base * 16)
        else:
            backbone = ResNetVd(int(bb.get("depth", 50)), mean, std)
            in_channels = (512, 1024, 2048)
        self.pixel_decoder = HybridEncoder(lib, in_channels=in_channels, c=fd, ffn=int(config.get("pixel_decoder_dim_feedforward", 1024)), n_enc=n_enc)
        self.pixel_decoder.backbone = backbone
        self.head = nn.Module()
        self.head.criterion = SetCriterionTrain(nc)
        self.head.predictor = TransformerPredictor(lib, nc, nq=int(config.get("num_queries", 300)), nl=int(config.get("transformer_predictor_dec_layers", 6)),
                                                   ffn=int(config.get("transformer_predictor_dim_feedforward", 1024)), in_dim=fd)
        from .train_nn import set_norm_mode
        set_norm_mode(self, norm)

    grad_ready = None   # callable(segment_name) set by TrainStep: "head" / "encoder" gradients are final (overlapped all-reduce)

    def _notify_when_all_grads(self, tensors, name: str):
        if self.grad_ready is not None:
            from .train import notify_when_all_grads

            notify_when_all_grads(tensors, self.grad_ready, name)

    def forward_outputs(self, images: torch.Tensor, forced_topk=None):
        """Everything before the criterion: the prediction sets (static shapes - what a captured training step replays, TrainStep graphs)."""
        f = self.pixel_decoder.backbone(images)
        # Segment boundaries [backbone | encoder | head].  Each boundary tensor is handed on as an ALIAS (view_as: no kernel, no copy, its own
        # autograd node): res4 is computed from res3 and the PAN outputs from one another, so the raw tensors are not an antichain of the
        # graph - a staged backward that stops at them (TrainStep._staged_backward) would have to run the path between two of them in the
        # earlier stage.  The aliases have no path to one another: stopping at them cuts the graph cleanly.
        feats = [f[k].view_as(f[k]) for k in ("res3", "res4", "res5")]
        self._notify_when_all_grads(feats, "encoder")
        enc = [e.view_as(e) for e in self.pixel_decoder(feats)]
        self._notify_when_all_grads(enc, "head")
        self.segment_boundaries = {"head": enc, "encoder": feats}
        out = self.head.predictor(enc, forced_topk)
        self.last_outputs = out
        return out

    def forward(self, images: torch.Tensor, targets: Sequence, forced_topk=None, fixed_matches=None):
        return self.head.criterion(self.forward_outputs(images, forced_topk), targets, fixed_matches)


class TrainStep:
    """One optimisation step of TrainerLoop.run_step (focoos/trainer/trainer.py:723-773) for the engine: forward + losses +
    backward on the HIP autograd graph, data-parallel gradient averaging (RCCL all-reduce of one flat fp32 buffer in 64 MiB
    buckets), global-norm clipping + AdamW in one fused kernel.  Parameters and their gradients live in the optimizer's flat
    buffers (the gradient kernels accumulate straight into the flat gradient views); per-parameter lr / weight decay follow
    build_optimizer (solver/build.py:39-138: backbone lr x0.1, weight_decay_norm on the parameters of normalisation modules).
    The step is launched eagerly (round 3: ~1 700 launches, 20-22 ms of host issue against 24-26 ms of GPU time on two streams - DESIGN.md
    §5 / §7): capturing forward+backward in a hipGraph through torch.cuda.make_graphed_callables was tried in round 1 and dead-locked inside
    the capture on this stack, so it is not used."""

    def __init__(self, model: FAIDetrTrainable, lr: float = 1e-4, backbone_multiplier: float = 0.1, weight_decay: float = 1e-4,
                 weight_decay_norm: float = 0.0, weight_decay_embed: float = 0.0, max_grad_norm: float = 0.1, ema_decay: Optional[float] = None, ema_warmups: int = 2000, scheduler: Optional[str] = None,
                 max_iters: int = 0, scheduler_extra: Optional[Dict] = None, check_every: int = 16, graphs: Optional[bool] = None,
                 loss_scale: Optional[float] = None, staged: Optional[bool] = None):
        """The step computes in the 16-bit element type that is current when it is built (_lib.compute_dtype(): "bf16", or "fp16" =
        the reference's amp training, trainer/trainer.py:645,735-773) and pins it at the top of every step.  Under fp16 the loss is
        multiplied by a dynamic scale before backward (default initial value 2**10 like the reference's GradScaler; ``loss_scale`` overrides),
        the optimizer launch unscales, skips the update on inf / NaN gradients and adapts the scale - all on the device."""
        from . import train_nn
        from .train import BucketedGradAllReduce, FlatAdamW

        self.model, self._nn = model, train_nn
        self.dtype_name = _lib.compute_dtype()
        if loss_scale is None and self.dtype_name == "fp16":
            loss_scale = 1024.0
        self.check_every = int(check_every)   # Hungarian status word polled every N steps (0 = only through check()): see check()
        # hipGraph replay of the step (see _step_graphed); default from FX_TRAIN_GRAPH.  OFF by default: measured on one MI355X the
        # replayed step is 8 % SLOWER than the eager one (RT-DETR 26.6 vs 24.5 ms, BiSeNetFormer 32.3 vs 29.5, MaskFormer 90.6 vs 83.8;
        # profiles/r04_train_graph.txt) - the step is GPU-bound, not host-bound: on ONE stream eager and replay take the same 27.9 ms, and
        # the weight-gradient side stream buys the eager step 3.5 ms of overlap but the replayed graph only 1.2 ms (the runtime runs the
        # forked branches of one graph with less concurrency than two live queues).  The capture removes ~20 ms of host work per step,
        # which matters only where the host is the bottleneck (many ranks per host, slow cores).
        # "auto" (round 5): which of the two wins depends on the BOX (one collection: eager 26.7 ms / replay 24.9 ms where the host cores
        # are slow, profiles/r05z_train*_bench.json; another: eager 23.4 / replay slower), so the step can time both itself: steps 1-2 eager
        # (lazy packing), 3-4 eager timed, 5 capture + first replay, 6-7 replay timed, then the faster form stays.  Several ranks (round 6): the same, with the timings MAX-all-reduced so
        # that every rank keeps the same form - the replayed step overlaps its all-reduces as well (one backward graph per stage, _capture).
        mode = os.environ.get("FX_TRAIN_GRAPH", "0") if graphs is None else graphs
        self._auto, self.graph_choice = None, None
        if isinstance(mode, str) and mode.strip().lower() == "auto":
            self.use_graphs, self._auto = False, {"phase": 0, "n": 0}
        else:
            self.use_graphs = bool(int(mode)) if isinstance(mode, str) else bool(mode)
        self._graph_state, self._eager_steps = None, 0

        from .train import dp_segment_of as seg_of

        # Flat layout = FORWARD order [backbone | pixel decoder | head], whatever order the modules were registered in (the backbone is
        # assigned to pixel_decoder after pixel_decoder's own layers exist, so named_parameters() lists it LAST among them - round 4: with
        # the registration order the three-segment layout below was never recognised and the overlapped all-reduce silently fell back to
        # one segment launched after backward; tests/test_dp_trainstep_cpu.py now drives this constructor on two gloo ranks).  The sort is
        # stable, so parameters keep their module order inside a segment; state_dict() / checkpoints are unaffected (they follow the modules).
        named = sorted(((n, p) for n, p in model.named_parameters() if p.requires_grad), key=lambda np_: seg_of(np_[0]))
        dev = named[0][1].device
        from .state_spec import state_spec
        from .train_data import optimizer_hyperparams

        kinds = state_spec(model.config, getattr(model, "family", "fai_detr"))   # FAIDetrTrainable / train_bf.BisenetFormerTrainable
        spec = []
        for n, p in named:   # per-parameter lr / weight decay exactly as get_optimizer_params (solver/build.py:39-101; pinned in tests/test_train_data_cpu.py)
            plr, pwd = optimizer_hyperparams(n, kinds[n][1], lr, weight_decay, weight_decay_norm, weight_decay_embed, backbone_multiplier)
            spec.append((n, tuple(p.shape), plr, pwd))
        self.spec = spec
        self.opt = FlatAdamW(spec, dev, max_grad_norm=max_grad_norm, loss_scale=loss_scale)
        with torch.no_grad():
            for n, p in named:
                self.opt.params[n].copy_(p.data)
                p.data = self.opt.params[n]
                p.grad = self.opt.grads[n]
        self.named = named
        # gradient all-reduce overlapped with backward: the flat buffer is [backbone | hybrid encoder | predictor] in parameter order;
        # backward finalises the predictor's gradients first (hook on the encoder outputs), then the encoder's (hook on res3..5)
        base = self.opt.flat_p.data_ptr()
        segs = [seg_of(n) for n, _ in named]
        offs = [(self.opt.params[n].data_ptr() - base) // 4 for n, _ in named]
        segments = None
        if segs == sorted(segs) and offs == sorted(offs) and set(segs) == {0, 1, 2}:   # the flat layout really is [backbone | encoder | head]
            b1, b2 = offs[segs.index(1)], offs[segs.index(2)]
            segments = [(0, b1), (b1, b2), (b2, self.opt.flat_g.numel())]
        self._seg_index = {"backbone": 0, "encoder": 1, "head": 2} if segments else {}
        self.reducer = BucketedGradAllReduce(self.opt.flat_g, segments=segments)
        # side stream for the weight-gradient launches (FX_WGRAD_STREAM=0: everything on one stream); from the per-device pool the
        # engines use, so that it maps to a hardware queue of its own
        self.wgrad_stream = None
        if os.environ.get("FX_WGRAD_STREAM", "1") != "0" and torch.device(self.opt.dev).type == "cuda" and _lib.two_queue_safe():
            from .engine import _device_stream

            self.wgrad_stream = _device_stream(torch.device(self.opt.dev), 1)
        # The step runs on a stream of its own, never on the legacy default stream: autograd ties every parameter's AccumulateGrad node to
        # the stream of its first use, and a backward CAPTURE that has to synchronise with the legacy default stream crashes in
        # hipStreamEndCapture (ROCm 7.2; PyTorch's CUDA-graph notes demand the same: warm-up on a side stream).  Stream 0 of the per-device
        # pool = the engines' main stream (training and inference of one process do not overlap).
        self.stream = None
        if torch.device(self.opt.dev).type == "cuda":
            from .engine import _device_stream

            self.stream = _device_stream(torch.device(self.opt.dev), 0)
        self.packer = self._nn.WeightPacker(model)
        # Staged backward (round 6; FX_DP_STAGED=1 or staged=True): backward runs as three autograd calls head -> encoder -> backbone that stop
        # at the model's segment boundaries, and a segment's all-reduce is launched BETWEEN two stages instead of from an autograd hook.  It is
        # the form a CAPTURED step needs (a replayed graph cannot launch a collective from a hook: _capture records one backward graph per
        # stage) and it is what makes the overlap testable on CPU ranks; gradients are those of the one-call backward, bit for bit.
        self.staged = bool(int(os.environ.get("FX_DP_STAGED", "0"))) if staged is None else bool(staged)
        self.stage_log: List[Tuple[str, object]] = []     # ("stage", name) / ("segment", i) in issue order (tests)
        self.reducer.before_collective = self._join_wgrads   # a segment's gradients are final only once its queued wgrads have run
        import os as _os

        if self._seg_index and int(_os.environ.get("FX_DP_OVERLAP", "1")):
            model.grad_ready = lambda name: self.reducer.launch_segment(self._seg_index[name])
        # what DistributedDataParallel does at construction (utils/distributed/dist.py:152): every rank starts from rank 0's parameters
        # and buffers (the engine's parameters are created with torch.empty; ranks that loaded different checkpoints would drift silently)
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.broadcast(self.opt.flat_p, 0)
            for _, b in model.named_buffers():
                if b.is_floating_point() or b.dtype in (torch.int64, torch.int32):
                    dist.broadcast(b, 0)
        # learning-rate schedule (trainer/solver/lr_scheduler.py): one device-side multiply of the per-chunk lr table per change
        self.scheduler, self.max_iters, self.scheduler_extra = scheduler, max_iters, dict(scheduler_extra or {})
        self.iteration, self._lr_factor = 0, 1.0
        self._base_lrs = self.opt.chunk_lr.clone()
        self.ema = None
        if ema_decay is not None:   # the trainer's EMA hook (trainer/solver/ema.py): one pass over the flat parameter buffer per step
            from .train_data import FlatEMA

            # EMAState covers chain(named_parameters, named_buffers) (solver/ema.py): the frozen parameters (FrozenBN affine) and every buffer
            # (running statistics, criterion.empty_weight) ride along as constants so that an EMA checkpoint loads in the reference
            bufs = {n: b for n, b in model.named_buffers()}
            bufs.update({n: p for n, p in model.named_parameters() if not p.requires_grad})
            self.ema = FlatEMA(self.opt.flat_p, {n: self.opt.params[n] for n, _ in named}, bufs, decay=ema_decay, warmups=ema_warmups)

    def _join_wgrads(self):
        self._nn.wgrad_join(self.opt.dev)

    # ------------------------------------------------------------------------------------------------ staged backward
    STAGES = (("head", 2), ("encoder", 1), ("backbone", 0))

    def _stage_params(self, seg: int) -> List[torch.Tensor]:
        from .train import dp_segment_of

        return [p for n, p in self.named if dp_segment_of(n) == seg]

    def _staged_backward(self, roots: Sequence[torch.Tensor], root_grads: Sequence[Optional[torch.Tensor]], between=None, wrap=None):
        """Backward of the model part below ``roots`` as three autograd calls.  Stage k runs from the previous boundary's gradients to the
        parameters of segment k and to the next boundary: ``torch.autograd.grad(roots, inputs = segment parameters + boundary tensors)`` - the
        engine executes exactly the nodes that lead to those inputs, and a NON-LEAF input is a capture point: its producer is not executed
        (``backward(inputs=...)`` would execute and free it - the next stage then finds its saved tensors gone).  Parameter gradients: the HIP
        nodes write straight into the flat views (``p.grad``, DIRECT_GRAD) and hand autograd nothing; what torch glue does return is added to the
        view here.  After a stage (and the join of its weight-gradient launches) ``between(name, segment)`` runs - the step launches that
        segment's all-reduce there, i.e. BEFORE the next stage's kernels are issued: DistributedDataParallel's overlap
        (utils/distributed/dist.py:138-157) without a hook.  ``wrap(name, fn)`` runs a stage's body (the capture wraps it in a graph
        capture); default: call it.  Gradients equal the one-call backward's bit for bit (same nodes, same order inside a segment)."""
        bnd = getattr(self.model, "segment_boundaries", None)
        carry: Dict[str, List] = {"roots": [t for t, g in zip(roots, root_grads) if t.requires_grad],
                                  "grads": [g for t, g in zip(roots, root_grads) if t.requires_grad]}
        if bnd and self._seg_index:
            cuts = {"head": [t for t in bnd["head"] if t.requires_grad], "encoder": [t for t in bnd["encoder"] if t.requires_grad]}
            plan = [("head", 2, cuts["head"]), ("encoder", 1, cuts["encoder"]), ("backbone", 0, [])]
        else:       # no three-segment layout (a model the constructor could not segment): one stage over every parameter
            plan = [("all", None, [])]
        for name, seg, stop in plan:
            def body(stop=stop, seg=seg):
                params = self._stage_params(seg) if seg is not None else [p for _, p in self.named]
                r, g = carry["roots"], carry["grads"]
                if r:
                    res = torch.autograd.grad(r, params + stop, g, allow_unused=True)
                    for prm, gr in zip(params, res[:len(params)]):
                        if gr is not None:      # torch glue (the HIP nodes returned None and accumulated into prm.grad themselves)
                            if prm.grad is None:
                                prm.grad = gr
                            else:
                                prm.grad.add_(gr)
                    pairs = [(t, gr) for t, gr in zip(stop, res[len(params):]) if gr is not None]
                    carry["roots"], carry["grads"] = [t for t, _ in pairs], [gr for _, gr in pairs]
                self._join_wgrads()
            self.stage_log.append(("stage", name))
            if wrap is not None:
                wrap(name, body)
            else:
                body()
            if between is not None:
                between(name, seg)

    def _launch_segment_logged(self, name: str, seg: int):
        self.stage_log.append(("segment", seg))
        self.reducer.launch_segment(seg)

    # ------------------------------------------------------------------------------------------------ captured step
    # TrainerLoop.run_step (trainer/trainer.py:723-773) as TWO hipGraph replays around an eager criterion (the eager
    # step issued ~1 700 launches from Python, 20-22 ms of host time in a 24 ms step; BiSeNetFormer's 1 956 launches were host-bound
    # outright).  What is static is captured: graph 1 = zero the flat gradients + weight re-packing + the model up to its prediction sets
    # (`forward_outputs`), graph 2 = the backward of exactly that part from the prediction sets' gradients (weight gradients on the side
    # stream, forked and joined inside the capture).  What depends on the step's TARGETS - Hungarian matching, pair counts, num_boxes,
    # the losses and their gradient w.r.t. the prediction sets - runs eagerly in between on detached leaves (tens of launches instead
    # of ~1 700): no target-dependent host scalar is frozen into a graph.  The optimizer (its bias-correction step count is a host scalar)
    # and the data-parallel all-reduce stay eager as well; round 6: the backward is THREE graphs (head | encoder | backbone) and segment k's
    # all-reduce is enqueued between stage k's and stage k+1's replay, so the collectives overlap the replayed backward as the eager hooks do.  The mechanism is torch.cuda.make_graphed_callables' (static input / output /
    # gradient buffers in one private pool), written out because the step owns state that must sit inside the captures (gradient
    # zero-fill, arena, weight packing, stream pinning, side-stream join).  Falls back to the eager step for SyncBN (collectives inside
    # the forward) and on CPU tensors (the gloo tests).
    def _graphs_applicable(self, images) -> bool:
        if not self.use_graphs or not images.is_cuda:
            return False
        return not any(getattr(m, "norm_mode", None) == "SyncBN" for m in self.model.modules())

    @staticmethod
    def _flatten(obj, out: List):
        if isinstance(obj, torch.Tensor):
            out.append(obj)
            return ("t", len(out) - 1)
        if isinstance(obj, dict):
            return ("d", {k: TrainStep._flatten(v, out) for k, v in obj.items()})
        if isinstance(obj, (list, tuple)):
            return ("l" if isinstance(obj, list) else "u", [TrainStep._flatten(v, out) for v in obj])
        return ("c", obj)

    @staticmethod
    def _rebuild(spec, tensors):
        kind, v = spec
        if kind == "t":
            return tensors[v]
        if kind == "d":
            return {k: TrainStep._rebuild(x, tensors) for k, x in v.items()}
        if kind in ("l", "u"):
            r = [TrainStep._rebuild(x, tensors) for x in v]
            return r if kind == "l" else tuple(r)
        return v

    def _capture(self, images: torch.Tensor):
        nn_ = self._nn
        dev = torch.device(self.opt.dev)
        st = {"key": (tuple(images.shape), images.dtype), "images": images.clone()}
        for n, p in self.named:
            p.grad = self.opt.grads[n]
        hooks, self.model.grad_ready = self.model.grad_ready, None     # the reducer's hooks belong to the eager path (collectives are not captured)
        torch.cuda.synchronize(dev)
        pool = torch.cuda.graph_pool_handle()
        g_fwd = torch.cuda.CUDAGraph()
        nn_.DIRECT_GRAD[0] = True
        if self.wgrad_stream is not None:
            nn_.WGRAD_STREAM[dev] = self.wgrad_stream
        try:
            with torch.cuda.graph(g_fwd, pool=pool, stream=self.stream):
                nn_.pin_stream(dev, True)          # the capture stream (pinned inside the context: launches must land on it)
                self.opt.zero_grad()
                nn_.ARENA.arm(self.opt.numel + (8 << 20), dev)
                self.packer.pack(dev)
                out = self.model.forward_outputs(st["images"])
                nn_.pin_stream(dev, False)
            tensors: List[torch.Tensor] = []
            st["spec"] = self._flatten(out, tensors)
            st["outs"] = tensors
            req = [t for t in tensors if t.requires_grad]
            st["gouts"] = [torch.zeros_like(t) for t in req]
            # (the backward is captured on THIS thread - no launches from the autograd engine's worker thread into a capture)
            # ONE GRAPH PER STAGE (round 6): head | encoder | backbone, cut at the model's segment boundaries (_staged_backward), each with its
            # weight-gradient fork / join inside - so that the replayed step can launch segment k's all-reduce between stage k and stage k+1
            # and the collective runs beside the next stage's replay (the single backward graph of round 4 could only be followed by all of them)
            stage_graphs: Dict[str, torch.cuda.CUDAGraph] = {}

            def wrap(name, body):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool, stream=self.stream):
                    nn_.pin_stream(dev, True)
                    body()
                    nn_.pin_stream(dev, False)
                stage_graphs[name] = g

            with torch.autograd.set_multithreading_enabled(False):
                self._staged_backward(req, st["gouts"], wrap=wrap)
            st["cuts"] = dict(self.model.segment_boundaries)     # the boundary tensors stay alive with the graphs (their gradients live in the pool)
            del self.stage_log[:]
        finally:
            nn_.DIRECT_GRAD[0] = False
            nn_.pin_stream(dev, False)
            nn_.WGRAD_STREAM.pop(dev, None)
            self.model.grad_ready = hooks
        st["fwd"], st["bwd_stages"] = g_fwd, [(name, seg, stage_graphs[name]) for name, seg in (self.STAGES if "head" in stage_graphs else (("all", None),))]
        self._graph_state = st

    def _step_graphed(self, images: torch.Tensor, targets: Sequence) -> Dict[str, torch.Tensor]:
        st = self._graph_state
        if st is None or st["key"] != (tuple(images.shape), images.dtype):
            self._capture(images)
            st = self._graph_state
        st["images"].copy_(images, non_blocking=True)
        st["fwd"].replay()
        leaves = [t.detach().requires_grad_(True) if t.requires_grad else t for t in st["outs"]]
        out = self._rebuild(st["spec"], leaves)
        self.model.last_outputs = out
        losses = self.model.head.criterion(out, targets)
        total = torch.stack(list(losses.values())).sum()
        if self.opt.scaler is not None:
            total = total * self.opt.scale.detach()
        total.backward()
        for g, l in zip(st["gouts"], [l for l in leaves if l.requires_grad]):
            if l.grad is None:
                g.zero_()
            else:
                g.copy_(l.grad)
        for name, seg, g in st["bwd_stages"]:
            self.stage_log.append(("stage", name))
            g.replay()
            if seg is not None:
                self._launch_segment_logged(name, seg)      # segment k's all-reduce is enqueued before stage k+1's replay (no-op on one rank)
        del self.stage_log[:-8]
        return losses

    def step(self, images: torch.Tensor, targets: Sequence) -> Dict[str, torch.Tensor]:
        with _lib.using_compute_dtype(self.dtype_name):     # the step's own element type, restored when it returns or raises
            if self.stream is None or not images.is_cuda:
                return self._step(images, targets)
            cur = torch.cuda.current_stream(images.device)
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                losses = self._step(images, targets)
            cur.wait_stream(self.stream)
            return losses

    def _auto_tick(self, images: torch.Tensor) -> None:
        """State machine of graphs="auto", called before every step (see __init__)."""
        import time

        import torch.distributed as dist

        a = self._auto
        # several ranks (round 6): the replayed step overlaps its all-reduces too (one backward graph per stage, a segment's collective enqueued
        # between two replays), so it takes part in the comparison; the two timings are MAX-all-reduced so that every rank makes the same choice
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        applicable = images.is_cuda and not any(getattr(m, "norm_mode", None) == "SyncBN" for m in self.model.modules())
        key = (tuple(images.shape), images.dtype)
        if not applicable or a.setdefault("key", key) != key:   # CPU / SyncBN / changing shapes: the eager step
            self.use_graphs, self._auto, self._graph_state = False, None, None
            self.graph_choice = {"graphs": False, "why": "not applicable"}
            return
        dev = images.device

        def now():
            torch.cuda.synchronize(dev)
            return time.perf_counter()

        if a["phase"] == 0 and self._eager_steps >= 2:
            a.update(phase=1, n=0, t0=now())
        elif a["phase"] == 1 and a["n"] == 2:
            a["eager_ms"] = (now() - a["t0"]) * 500.0
            self.use_graphs = True
            a.update(phase=2, n=0)
        elif a["phase"] == 2 and a["n"] == 1:
            a.update(phase=3, n=0, t0=now())
        elif a["phase"] == 3 and a["n"] == 2:
            a["graph_ms"] = (now() - a["t0"]) * 500.0
            eager_ms, graph_ms = a["eager_ms"], a["graph_ms"]
            if multi:
                tm = torch.tensor([eager_ms, graph_ms], dtype=torch.float64, device=dev)
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                eager_ms, graph_ms = float(tm[0]), float(tm[1])
            self.use_graphs = graph_ms < 0.98 * eager_ms
            self.graph_choice = {"graphs": self.use_graphs, "eager_ms": round(eager_ms, 3), "graph_ms": round(graph_ms, 3)}
            if not self.use_graphs:
                self._graph_state = None   # gives the private pool back
            self._auto = None
            return
        a["n"] += 1

    def _step(self, images: torch.Tensor, targets: Sequence) -> Dict[str, torch.Tensor]:
        nn_ = self._nn
        if self._auto is not None:
            self._auto_tick(images)
        if self._graphs_applicable(images) and self._eager_steps >= 2:   # two eager steps first: lazy packing + first-use kernel attributes, then the packer's device table
            if self.scheduler is not None:
                from .train_data import lr_factor

                f = lr_factor(self.scheduler, self.iteration, self.max_iters, **self.scheduler_extra)
                if f != self._lr_factor:
                    self.opt.set_lr_scale(f, self._base_lrs)
                    self._lr_factor = f
            self.iteration += 1
            losses = self._step_graphed(images, targets)
            self.reducer.launch()
            self.reducer.wait()
            self.opt.step()
            nn_.WEIGHTS_EPOCH[0] += 1
            if self.ema is not None:
                self.ema.update()
            if self.check_every > 0 and self.iteration % self.check_every == 0:
                self.check()
            return losses
        self._eager_steps += 1
        if self.scheduler is not None:
            from .train_data import lr_factor

            f = lr_factor(self.scheduler, self.iteration, self.max_iters, **self.scheduler_extra)
            if f != self._lr_factor:
                self.opt.set_lr_scale(f, self._base_lrs)
                self._lr_factor = f
        self.iteration += 1
        losses = self._forward_backward(images, targets)
        self.reducer.launch()
        self.reducer.wait()
        self.opt.step()
        nn_.WEIGHTS_EPOCH[0] += 1
        if self.ema is not None:
            self.ema.update()
        if self.check_every > 0 and self.iteration % self.check_every == 0:
            self.check()
        return losses

    def _forward_backward(self, images: torch.Tensor, targets: Sequence, **forced) -> Dict[str, torch.Tensor]:
        """The eager step up to the gradients: zeroed flat gradient buffer, weight images re-packed in one launch, forward + criterion +
        backward with the gradient kernels accumulating straight into the flat views and the weight gradients on the side stream (joined
        before returning).  ``forced``: teacher-forcing arguments of the model's forward (forced_topk / forced_attn / fixed_matches)."""
        nn_ = self._nn
        assert _lib.compute_dtype() == self.dtype_name    # pinned by step() / forward_backward()
        self.opt.zero_grad()
        for n, p in self.named:  # gradient kernels / autograd accumulate in place into these views
            p.grad = self.opt.grads[n]
        nn_.ARENA.arm(self.opt.numel + (8 << 20), self.opt.dev)   # staging for the 3x3 weight gradients + padded heads
        nn_.DIRECT_GRAD[0] = True
        nn_.pin_stream(self.opt.dev, True)   # one stream-handle lookup per step instead of one per launch (backward runs on this stream too)
        dev = torch.device(self.opt.dev)
        self.packer.pack(dev)   # every weight image the optimizer step made stale, one launch (layers not seen yet pack themselves lazily)
        if self.wgrad_stream is not None:
            nn_.WGRAD_STREAM[dev] = self.wgrad_stream   # weight gradients overlap the input-gradient chain (train_nn._wgrad_fork)
        try:
            if self.staged and self._seg_index:
                # staged form: criterion on detached leaves (as the captured step does), then the model's backward in three stages with the
                # segment all-reduces launched in between; no autograd hooks
                hooks, self.model.grad_ready = self.model.grad_ready, None
                try:
                    out = self.model.forward_outputs(images, **{k: v for k, v in forced.items() if k != "fixed_matches"})
                finally:
                    self.model.grad_ready = hooks
                tensors: List[torch.Tensor] = []
                spec = self._flatten(out, tensors)
                leaves = [t.detach().requires_grad_(True) if t.requires_grad else t for t in tensors]
                lout = self._rebuild(spec, leaves)
                self.model.last_outputs = lout
                losses = self.model.head.criterion(lout, targets, forced["fixed_matches"]) if forced.get("fixed_matches") is not None \
                    else self.model.head.criterion(lout, targets)
                total = torch.stack(list(losses.values())).sum()
                if self.opt.scaler is not None:
                    total = total * self.opt.scale.detach()
                total.backward()
                pairs = [(t, l.grad) for t, l in zip(tensors, leaves) if t.requires_grad and l.grad is not None]
                self._staged_backward([t for t, _ in pairs], [g for _, g in pairs],
                                      between=lambda name, seg: self._launch_segment_logged(name, seg) if seg is not None else None)
                del self.stage_log[:-8]
            else:
                losses = self.model(images, targets, **forced)
                total = torch.stack(list(losses.values())).sum()   # 2 launches instead of one add per loss term
                if self.opt.scaler is not None:
                    total = total * self.opt.scale.detach()        # GradScaler.scale(loss): the gradients carry the factor until the optimizer launch
                total.backward()
        finally:
            nn_.DIRECT_GRAD[0] = False
            nn_.pin_stream(self.opt.dev, False)
            self._join_wgrads()
            nn_.WGRAD_STREAM.pop(dev, None)
        return losses

    def forward_backward(self, images: torch.Tensor, targets: Sequence, **forced) -> Dict[str, torch.Tensor]:
        """``step()`` WITHOUT the all-reduce / optimizer / EMA: the production forward + backward (same routing, streams and buffers) leaving
        the gradients in ``self.opt.grads`` - what the parity tests at the BASELINE shapes compare with the oracle's gradients
        (tests/test_gpu_train_baseline_configs.py)."""
        with _lib.using_compute_dtype(self.dtype_name):
            if self.stream is None or not images.is_cuda:
                return self._forward_backward(images, targets, **forced)
            cur = torch.cuda.current_stream(images.device)
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                losses = self._forward_backward(images, targets, **forced)
            cur.wait_stream(self.stream)
            return losses

    def check(self) -> None:
        """Poll and clear the Hungarian solver's device status word (criterion.raise_if_infeasible): a step whose matching costs were
        NaN / inf - where the reference raises from SciPy inside the matcher (fai_detr/modelling.py:749-750) - raises the same ValueError
        here, on EVERY rank (the word is OR-all-reduced first: a rank raising alone would leave the others blocked in the next gradient
        all-reduce).  One 4-byte D2H read = one host synchronisation, so step() calls it every `check_every` steps (default 16: at most
        15 optimizer steps run on identity assignments before the run ends) instead of every step; direct users of TrainStep that want the
        reference's per-step behaviour pass check_every=1 or call check() themselves."""
        from .criterion import raise_if_infeasible

        raise_if_infeasible(self.opt.dev, all_ranks=True)
        # The un-scaled (bf16) optimizer launch SKIPS an update whose gradients hold an inf / NaN and reports total_norm = inf
        # (fx_adamw_step_f32); nothing else would tell a run that stopped training from one that trains.  Read at the same poll:
        # the all-reduced gradients - hence the norm - are identical on every rank, so every rank raises together.  Under a loss scale
        # (fp16) skipped steps are the scaler's normal operation and are counted on the device instead (FlatAdamW.scaler_state()).
        if self.opt.scaler is None and self.opt.step_count > 0:
            tn = float(self.opt.total_norm)
            if tn != tn or tn == float("inf"):
                raise FloatingPointError(f"focoos_amd.TrainStep: the gradients of optimizer step {self.opt.step_count} hold an inf / NaN (total norm {tn}); "
                                         "the fused AdamW launch skipped the update (parameters and moments untouched). The run has diverged - "
                                         "lower the learning rate or train under the fp16 loss scale")
