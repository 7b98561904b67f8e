"""Training-side host pieces shared by the training step (train_detr.TrainStep) and the integration seams:

* ``ms_deform_attn_core`` — autograd-aware drop-in for the function-pointer seam
  ``MSDeformableAttention.ms_deformable_attn_core`` (fai_detr/modelling.py:806; deformable.py:10-35): fx_msda_f32_fwd/bwd.
* ``FlatAdamW`` — fused multi-tensor AdamW + full-model gradient-norm clipping (trainer/solver/build.py:29-138 semantics:
  per-tensor lr / weight decay, one global clip) on a single flat fp32 buffer: fx_adamw_step_f32.
* ``BucketedGradAllReduce`` — data-parallel gradient averaging over torch.distributed (RCCL on MI355X, gloo in the CPU tests):
  flat buckets, asynchronous all-reduce launched as buckets fill, one wait before the optimizer step (DDP's C1 collective,
  focoos/utils/distributed/dist.py:138-157) plus the 4-byte ``num_boxes`` all-reduce lives in criterion.SetCriterion.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import check


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class _MSDAFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, value, shapes_t, starts_t, loc, attn):
        lib = _lib.load()
        B, S, M, D = value.shape
        Q, L, P = loc.shape[1], loc.shape[3], loc.shape[4]
        v = value.float().contiguous()
        lc, aw = loc.float().contiguous(), attn.float().contiguous()
        out = torch.empty(B, Q, M * D, dtype=torch.float32, device=value.device)
        check(lib.fx_msda_f32_fwd(v.data_ptr(), shapes_t.data_ptr(), starts_t.data_ptr(), L, P, lc.data_ptr(), aw.data_ptr(), out.data_ptr(), B, S, Q, M,
                                  _stream(value.device)), "fx_msda_f32_fwd")
        ctx.save_for_backward(v, shapes_t, starts_t, lc, aw)
        ctx.dims = (B, S, Q, M, D, L, P)
        ctx.in_dtypes = (value.dtype, loc.dtype, attn.dtype)
        return out.to(value.dtype)

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        v, shapes_t, starts_t, lc, aw = ctx.saved_tensors
        B, S, Q, M, D, L, P = ctx.dims
        go = grad_out.float().contiguous()
        gv, gl, ga = torch.empty_like(v), torch.empty_like(lc), torch.empty_like(aw)
        check(lib.fx_msda_f32_bwd(v.data_ptr(), shapes_t.data_ptr(), starts_t.data_ptr(), L, P, lc.data_ptr(), aw.data_ptr(), go.data_ptr(), gv.data_ptr(),
                                  gl.data_ptr(), ga.data_ptr(), B, S, Q, M, _stream(v.device)), "fx_msda_f32_bwd")
        d0, d1, d2 = ctx.in_dtypes
        return gv.view(B, S, M, D).to(d0), None, None, gl.to(d1), ga.to(d2)


class ValueGradSink:
    """Shared fp32 gradient buffer of a value projection that G deformable-attention layers read as column slices ([B,S,G*256]): every
    layer's backward accumulates into its slice; the G-th (last) one hands the whole buffer to autograd, the others return None - no
    per-layer fp32 buffers, zero-fills, casts or G-1 memory-sized gradient additions."""

    def __init__(self, G: int):
        self.G, self.count, self.buf, self.slab = G, 0, None, None   # slab: the backward form (binning / atomics) this pass's buffer was allocated for


class _MSDAGroupFunction(torch.autograd.Function):
    """ms_deform_attn_core on slice ``g`` of a shared bf16 value tensor [B,S,G*M*32] (fx_msda_train_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, value_all, sink: ValueGradSink, g: int, shapes_t, starts_t, loc, attn, shapes_host=None):
        lib = _lib.load()
        B, S, Nt = value_all.shape
        M, D = 8, 32
        Q, L, P = loc.shape[1], loc.shape[3], loc.shape[4]
        assert value_all.dtype == _lib.act_dtype() and value_all.is_contiguous() and Nt == sink.G * M * D
        lc, aw = loc.float().contiguous(), attn.float().contiguous()
        out = torch.empty(B, Q, M * D, dtype=torch.float32, device=value_all.device)
        check(lib.fx_msda_train_fwd(value_all.data_ptr() + g * M * D * 2, 1, Nt, shapes_t.data_ptr(), starts_t.data_ptr(), L, P, lc.data_ptr(),
                                    aw.data_ptr(), out.data_ptr(), B, S, Q, M, _stream(value_all.device)), "fx_msda_train_fwd")
        ctx.save_for_backward(value_all, shapes_t, starts_t, lc, aw)
        ctx.sink, ctx.g, ctx.dims = sink, g, (B, S, Q, M, D, L, P, Nt)
        ctx.shapes_host = shapes_host
        ctx.in_dtypes = (loc.dtype, attn.dtype)
        return out.to(_lib.act_dtype())

    @staticmethod
    def backward(ctx, grad_out):
        value_all, shapes_t, starts_t, lc, aw = ctx.saved_tensors
        gv, gl, ga = _msda_group_backward(value_all, shapes_t, starts_t, lc, aw, grad_out, ctx.sink, ctx.g, ctx.dims, ctx.shapes_host)
        d1, d2 = ctx.in_dtypes
        return gv, None, None, None, None, gl.to(d1), ga.to(d2), None


def _msda_group_backward(value_all, shapes_t, starts_t, lc, aw, grad_out, sink, g, dims, sh):
    """Backward core of the grouped deformable attention: (value gradient or None, grad_loc f32, grad_attn f32)."""
    lib = _lib.load()
    B, S, Q, M, D, L, P, Nt = dims
    go_bf16 = grad_out.dtype == _lib.act_dtype()
    slab = (sh is not None and os.environ.get("FX_MSDA_BWD_SLAB", "1") != "0"
            and lib.fx_msda_bwd_slab_supported(sh.ctypes.data, L, P, Q, M, int(go_bf16)) == 1)
    # The path is decided ONCE per sink (by the first layer of a backward pass to arrive): the shared buffer is bf16 [B,S,Nt] for the
    # binning form and fp32 for the atomic form, and a layer taking the other path would write fp32 rows into a bf16 buffer of half
    # the size.  A later layer that cannot follow the sink's choice fails loudly instead.
    if sink.buf is None:
        sink.slab = slab
    elif sink.slab != slab:
        if sink.slab and not go_bf16:   # the binning form was chosen for bf16 gradients; an fp32 grad_out of a later layer is cast, not re-routed
            grad_out, go_bf16, slab = grad_out.to(_lib.act_dtype()), True, True
        else:
            raise _lib.FocoosAmdError("grouped deformable-attention backward: the layers of one value group must all take the same "
                                      f"path (sink holds a {'bf16 slab' if sink.slab else 'fp32 atomic'} buffer, this layer asked for the other)")
    assert sink.buf is None or sink.buf.dtype == (_lib.act_dtype() if slab else torch.float32)
    gl, ga = torch.empty_like(lc), torch.empty_like(aw)
    if slab:
        go = grad_out.contiguous() if go_bf16 else grad_out.float().contiguous()
        if sink.buf is None:   # every layer overwrites its 256 columns: no zero-fill, no fp32 image, no cast
            sink.buf = torch.empty(B, S, Nt, dtype=_lib.act_dtype(), device=value_all.device)
        check(lib.fx_msda_train_bwd_slab(value_all.data_ptr() + g * M * D * 2, 1, Nt, shapes_t.data_ptr(), starts_t.data_ptr(), sh.ctypes.data, L, P,
                                         lc.data_ptr(), aw.data_ptr(), go.data_ptr(), int(go_bf16), sink.buf.data_ptr() + g * M * D * 2, Nt,
                                         gl.data_ptr(), ga.data_ptr(), B, S, Q, M, _stream(value_all.device)), "fx_msda_train_bwd_slab")
    else:
        go = grad_out.float().contiguous()
        if sink.buf is None:
            sink.buf = torch.zeros(B, S, Nt, dtype=torch.float32, device=value_all.device)
        check(lib.fx_msda_train_bwd(value_all.data_ptr() + g * M * D * 2, 1, Nt, shapes_t.data_ptr(), starts_t.data_ptr(), L, P, lc.data_ptr(),
                                    aw.data_ptr(), go.data_ptr(), sink.buf.data_ptr() + g * M * D * 4, Nt, 0, gl.data_ptr(), ga.data_ptr(), B, S, Q, M,
                                    _stream(value_all.device)), "fx_msda_train_bwd")
    sink.count += 1
    gv = None
    if sink.count == sink.G:   # every layer has delivered its slice
        gv = sink.buf if sink.buf.dtype == _lib.act_dtype() else sink.buf.to(_lib.act_dtype())
        sink.buf, sink.count = None, 0
    return gv, gl, ga


class _MSDAGroupRawFunction(torch.autograd.Function):
    """_MSDAGroupFunction fed with the layer's RAW projections: sampling offsets bf16 [B,Q,M*L*P*2], attention logits bf16 [B,Q,M*L*P] and the
    (detached) reference boxes f32 [B,Q,4] - softmax and location arithmetic of MSDeformableAttention.forward (fai_detr/modelling.py:866-879)
    happen in fx_msda_prep_bf16 / fx_msda_prep_bwd_bf16: 2 launches forward and 3 backward instead of ~9 and ~11."""

    @staticmethod
    def forward(ctx, value_all, sink: ValueGradSink, g: int, shapes_t, starts_t, off, logit, ref, shapes_host, M: int, L: int, P: int):
        lib = _lib.load()
        B, S, Nt = value_all.shape
        D = 32
        Q = off.shape[1]
        dev = value_all.device
        assert value_all.dtype == _lib.act_dtype() and value_all.is_contiguous() and Nt == sink.G * M * D
        off, logit, ref = off.contiguous(), logit.contiguous(), ref.float().contiguous()
        assert off.dtype == _lib.act_dtype() and logit.dtype == _lib.act_dtype() and off.shape[-1] == M * L * P * 2 and logit.shape[-1] == M * L * P
        lc = torch.empty(B, Q, M, L, P, 2, dtype=torch.float32, device=dev)
        aw = torch.empty(B, Q, M, L, P, dtype=torch.float32, device=dev)
        st = _stream(dev)
        check(lib.fx_msda_prep_bf16(off.data_ptr(), M * L * P * 2, logit.data_ptr(), M * L * P, ref.data_ptr(), lc.data_ptr(), aw.data_ptr(), B * Q, M, L, P,
                                    st), "fx_msda_prep_bf16")
        out = torch.empty(B, Q, M * D, dtype=torch.float32, device=dev)
        check(lib.fx_msda_train_fwd(value_all.data_ptr() + g * M * D * 2, 1, Nt, shapes_t.data_ptr(), starts_t.data_ptr(), L, P, lc.data_ptr(),
                                    aw.data_ptr(), out.data_ptr(), B, S, Q, M, st), "fx_msda_train_fwd")
        ctx.save_for_backward(value_all, shapes_t, starts_t, lc, aw, ref)
        ctx.sink, ctx.g, ctx.dims, ctx.shapes_host = sink, g, (B, S, Q, M, D, L, P, Nt), shapes_host
        return out.to(_lib.act_dtype())

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        value_all, shapes_t, starts_t, lc, aw, ref = ctx.saved_tensors
        B, S, Q, M, D, L, P, Nt = ctx.dims
        gv, gl, ga = _msda_group_backward(value_all, shapes_t, starts_t, lc, aw, grad_out, ctx.sink, ctx.g, ctx.dims, ctx.shapes_host)
        g_off = torch.empty(B, Q, M * L * P * 2, dtype=_lib.act_dtype(), device=value_all.device)
        g_logit = torch.empty(B, Q, M * L * P, dtype=_lib.act_dtype(), device=value_all.device)
        check(lib.fx_msda_prep_bwd_bf16(gl.data_ptr(), ga.data_ptr(), aw.data_ptr(), ref.data_ptr(), g_off.data_ptr(), M * L * P * 2, g_logit.data_ptr(),
                                        M * L * P, B * Q, M, L, P, _stream(value_all.device)), "fx_msda_prep_bwd_bf16")
        return gv, None, None, None, None, g_off, g_logit, None, None, None, None, None


def ms_deform_attn_grouped_raw(value_all: torch.Tensor, sink: ValueGradSink, g: int, value_spatial_shapes, offsets: torch.Tensor, logits: torch.Tensor,
                               ref_boxes: torch.Tensor, heads: int, levels: int, points: int) -> torch.Tensor:
    """ms_deform_attn_grouped from the raw projections of the layer (see _MSDAGroupRawFunction); ref_boxes [B,Q,4] carries no gradient."""
    st, ss = _shape_tensors(value_spatial_shapes, value_all.device)
    return _MSDAGroupRawFunction.apply(value_all, sink, g, st, ss, offsets, logits, ref_boxes.detach(), _shape_host(value_spatial_shapes), heads, levels,
                                       points)


def ms_deform_attn_grouped(value_all: torch.Tensor, sink: ValueGradSink, g: int, value_spatial_shapes, sampling_locations: torch.Tensor,
                           attention_weights: torch.Tensor) -> torch.Tensor:
    """ms_deform_attn_core for layer ``g`` of G layers whose value projections were computed together: value_all bf16 [B,S,G*256]; returns
    bf16 [B,Q,256].  All G layers must take part in a backward pass (the last one to run delivers the value gradient)."""
    st, ss = _shape_tensors(value_spatial_shapes, value_all.device)
    return _MSDAGroupFunction.apply(value_all, sink, g, st, ss, sampling_locations, attention_weights, _shape_host(value_spatial_shapes))


def _shape_host(value_spatial_shapes) -> np.ndarray:
    key = tuple((int(h), int(w)) for h, w in value_spatial_shapes)
    if key not in _SHAPE_HOST:
        _SHAPE_HOST[key] = np.ascontiguousarray(np.array(key, dtype=np.int32).reshape(-1))
    return _SHAPE_HOST[key]


_SHAPE_HOST: Dict = {}
# value gradient of the grouped form by binning in LDS (fx_msda_train_bwd_slab: no floating-point atomics, bf16 out) wherever
# fx_msda_bwd_slab_supported says the shapes fit; FX_MSDA_BWD_SLAB=0 keeps the fp32-atomic form (fx_msda_train_bwd) for A/B runs.


def _shape_tensors(value_spatial_shapes, dev):
    shapes = [(int(h), int(w)) for h, w in value_spatial_shapes]
    starts, acc = [], 0
    for h, w in shapes:
        starts.append(acc)
        acc += h * w
    key = (tuple(shapes), str(dev))
    if key not in _SHAPE_CACHE:  # cached: no host->device copy per call
        _SHAPE_CACHE[key] = (torch.tensor(shapes, dtype=torch.int32, device=dev), torch.tensor(starts, dtype=torch.int32, device=dev))
    return _SHAPE_CACHE[key]


def ms_deform_attn_core(value: torch.Tensor, value_spatial_shapes, sampling_locations: torch.Tensor, attention_weights: torch.Tensor) -> torch.Tensor:
    """Same signature and semantics as the reference's ``ms_deform_attn_core_pytorch``; differentiable w.r.t. value,
    sampling_locations and attention_weights.  value [B,S,M,D=32], locations [B,Q,M,L,P,2] in [0,1], weights [B,Q,M,L,P]."""
    st, ss = _shape_tensors(value_spatial_shapes, value.device)
    return _MSDAFunction.apply(value, st, ss, sampling_locations, attention_weights)


_SHAPE_CACHE: Dict = {}


class FlatAdamW:
    """AdamW over a list of (name, tensor, lr, weight_decay) held in ONE flat fp32 buffer on the device.
    ``params`` / ``grads`` are exposed as per-tensor views of the flat buffers (so a backward pass can write grads in place)."""

    CHUNK = 65536

    def __init__(self, named_shapes: Sequence[Tuple[str, Sequence[int], float, float]], device, betas=(0.9, 0.999), eps: float = 1e-8,
                 max_grad_norm: float = 0.1, loss_scale: Optional[float] = None, growth_factor: float = 2.0, backoff_factor: float = 0.5,
                 growth_interval: int = 2000):
        """``loss_scale`` (fp16 element type): initial value of a DYNAMIC loss scale kept on the device - torch.amp.GradScaler's state
        and arithmetic (the reference: GradScaler(init_scale=2**10), trainer/trainer.py:645; growth 2.0 / backoff 0.5 / interval 2000 are
        GradScaler's defaults): ``scale`` is the tensor the loss is multiplied by, ``step()`` unscales, skips on inf / NaN and updates the
        scale inside the optimizer launches (fx_adamw_step_scaled_f32)."""
        self.lib = _lib.load()
        self.dev = torch.device(device)
        self.betas, self.eps, self.max_grad_norm = betas, eps, max_grad_norm
        offs, total = [], 0
        for _, shape, _, _ in named_shapes:
            n = 1
            for s in shape:
                n *= int(s)
            offs.append((total, n))
            total += n
        self.numel = total
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=self.dev)
        self.flat_g = torch.zeros_like(self.flat_p)
        self.flat_m = torch.zeros_like(self.flat_p)
        self.flat_v = torch.zeros_like(self.flat_p)
        self.params: Dict[str, torch.Tensor] = {}
        self.grads: Dict[str, torch.Tensor] = {}
        cs, cl, clr, cwd = [], [], [], []
        for (name, shape, lr, wd), (o, n) in zip(named_shapes, offs):
            self.params[name] = self.flat_p[o:o + n].view(*shape)
            self.grads[name] = self.flat_g[o:o + n].view(*shape)
            for c0 in range(0, n, self.CHUNK):
                cs.append(o + c0)
                cl.append(min(self.CHUNK, n - c0))
                clr.append(lr)
                cwd.append(wd)
        self.nchunks = len(cs)
        self.chunk_start = torch.tensor(cs, dtype=torch.int64, device=self.dev)
        self.chunk_len = torch.tensor(cl, dtype=torch.int32, device=self.dev)
        self.chunk_lr = torch.tensor(clr, dtype=torch.float32, device=self.dev)
        self.chunk_wd = torch.tensor(cwd, dtype=torch.float32, device=self.dev)
        self.ws = torch.empty(self.lib.fx_adamw_workspace_bytes() // 8, dtype=torch.float64, device=self.dev)
        self.total_norm = torch.zeros(1, dtype=torch.float32, device=self.dev)
        self.step_count = 0
        self.scaler = None
        if loss_scale is not None:
            # fx_loss_scale_state {float scale; int32 growth_tracker, good_steps, skipped_steps}: one 16-byte device record, two typed views
            self._scale_state = torch.zeros(4, dtype=torch.int32, device=self.dev)
            self.scale = self._scale_state.view(torch.float32)[0:1]
            self.scale.fill_(float(loss_scale))
            self.scaler = (float(growth_factor), float(backoff_factor), int(growth_interval))

    def scaler_state(self) -> Dict[str, float]:
        """(host synchronisation) the loss-scale record: scale, growth_tracker, good_steps, skipped_steps."""
        st = self._scale_state.cpu()
        return {"scale": float(st.view(torch.float32)[0]), "growth_tracker": int(st[1]), "good_steps": int(st[2]), "skipped_steps": int(st[3])}

    def set_lr_scale(self, scale: float, base_lrs: torch.Tensor):
        self.chunk_lr.copy_(base_lrs * scale)

    def step(self):
        self.step_count += 1
        if self.scaler is not None:
            check(self.lib.fx_adamw_step_scaled_f32(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.flat_m.data_ptr(), self.flat_v.data_ptr(), self.numel,
                                                    self.chunk_start.data_ptr(), self.chunk_len.data_ptr(), self.chunk_lr.data_ptr(), self.chunk_wd.data_ptr(),
                                                    self.nchunks, float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.max_grad_norm),
                                                    self.ws.data_ptr(), self.total_norm.data_ptr(), self._scale_state.data_ptr(), self.scaler[0],
                                                    self.scaler[1], self.scaler[2], _stream(self.dev)), "fx_adamw_step_scaled_f32")
            return
        check(self.lib.fx_adamw_step_f32(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.flat_m.data_ptr(), self.flat_v.data_ptr(), self.numel,
                                         self.chunk_start.data_ptr(), self.chunk_len.data_ptr(), self.chunk_lr.data_ptr(), self.chunk_wd.data_ptr(),
                                         self.nchunks, self.step_count, float(self.betas[0]), float(self.betas[1]), float(self.eps),
                                         float(self.max_grad_norm), self.ws.data_ptr(), self.total_norm.data_ptr(), _stream(self.dev)), "fx_adamw_step_f32")

    def zero_grad(self):
        self.flat_g.zero_()


def notify_when_all_grads(tensors, callback, name):
    """Autograd hooks on the activations that separate two parameter segments: once the gradient of EVERY tensor in ``tensors`` has
    been computed, every layer after them has finished its backward (the autograd engine runs ready nodes latest-created first), so
    those layers' parameter gradients are final and ``callback(name)`` may start their all-reduce."""
    live = [t for t in tensors if t.requires_grad]
    left = [len(live)]

    def hook(_g):
        left[0] -= 1
        if left[0] == 0:
            callback(name)

    for t in live:
        t.register_hook(hook)


def dp_segment_of(name: str) -> int:
    """Segment of a parameter in the flat gradient buffer: 0 backbone, 1 pixel decoder, 2 head - forward order, so backward finalises
    them 2 -> 1 -> 0 (the three families share the prefixes ``pixel_decoder.backbone.`` / ``pixel_decoder.``)."""
    return 0 if name.startswith("pixel_decoder.backbone.") else (1 if name.startswith("pixel_decoder.") else 2)


def dp_plan(config: Dict, family: str, norm: str, world: int, per_rank_batch: int, bucket_bytes: int = 64 << 20, grad_bytes: int = 4) -> Dict:
    """What a data-parallel training step moves, computed from the state spec alone (no GPU, no model): trainable parameters per segment,
    the buckets ``BucketedGradAllReduce`` cuts (never across a segment), bytes all-reduced per step and per rank, the batch split of
    ``TrainerArgs.batch_size`` (loaders.py:61-65).  ``bench.py --train --dry-run`` prints it; tests/test_dp_trainstep_cpu.py checks it against
    the layout the real ``TrainStep`` builds.  ``grad_bytes`` = 2 describes the bf16 bucket option (FX_DP_BF16)."""
    import math

    from .state_spec import state_spec

    kinds = ("conv_w", "lin_w", "lin_b", "ln_w", "ln_b", "emb") + (() if norm == "FrozenBN" else ("bn_w", "bn_b"))
    seg_numel = [0, 0, 0]
    for name, (shape, kind) in state_spec(config, family).items():
        if kind in kinds and not (family == "fai_detr" and ".mask_features." in name):   # RT-DETR's dead mask_features conv is frozen (train_nn.HybridEncoder)
            seg_numel[dp_segment_of(name)] += math.prod(shape) if shape else 1
    per = max(1, bucket_bytes // 4)     # bucket boundaries are cut on the fp32 flat buffer
    buckets, lo = [], 0
    for n in seg_numel:
        buckets += [min(per, lo + n - s) for s in range(lo, lo + n, per)]
        lo += n
    total = sum(seg_numel)
    # SyncBN (the reference's multi-GPU conversion, trainer/trainer.py:333-334): every BatchNorm layer issues one small all-reduce in the
    # forward ([2, C] sums) and one in the backward ([sum da, sum da xhat]) - serialised, latency-bound collectives.  A layer's
    # statistics are a data dependency of its own output, so only SIBLINGS can share a collective: layers that read the same input (forward)
    # / receive the same output gradient (backward) - a bottleneck's shortcut conv with branch2a (forward) / branch2c (backward), the
    # conv1 | conv2 halves of a CSP layer and the 3x3 | 1x1 branches of a RepVGG block, the level projections whose inputs exist together.
    spec = state_spec(config, family)
    bn_layers = [n[: -len(".weight")] for n, (_, kind) in spec.items() if kind == "bn_w"]
    pairs = repvgg_pairs = 0
    if family == "fai_detr":
        parents = {}
        for n in bn_layers:
            parts = n.split(".")
            if "conv1" in parts or "conv2" in parts:    # CSPRepLayer.conv1 | conv2, RepVggBlock.conv1 | conv2
                i = parts.index("conv1") if "conv1" in parts else parts.index("conv2")
                if parts[i - 1] != "backbone":
                    parents.setdefault(".".join(parts[:i]), set()).add(parts[i])
        full = [k for k, v in parents.items() if v == {"conv1", "conv2"}]
        repvgg_pairs = sum(1 for k in full if ".bottlenecks." in k)
        pairs = len(full) + sum(1 for n in bn_layers if ".short." in n)
    # BUILT (round 6, train_nn._SiblingConvBnFn): every pair shares its FORWARD collective; the shortcut | branch2a and CSP conv1 | conv2 pairs
    # share the backward one too (the node waits for both output gradients); a RepVGG block's 3x3 branch gets its gradient out of the 1x1
    # branch's post-collective pass, so its two backward collectives stay apart
    fwd = len(bn_layers) - pairs
    bwd = len(bn_layers) - (pairs - repvgg_pairs)
    syncbn = {"batchnorm_layers": len(bn_layers), "collectives_per_step": 2 * len(bn_layers),
              "sibling_pairs_that_could_share_a_collective": pairs,
              "collectives_per_step_with_siblings_coalesced": 2 * (len(bn_layers) - pairs),
              "collectives_per_step_as_built": fwd + bwd, "forward_collectives_as_built": fwd, "backward_collectives_as_built": bwd,
              "note": "nn.SyncBatchNorm issues one all-reduce per layer and direction; siblings = layers whose conv outputs exist together (same "
                      "input forward / gradients that meet in one autograd node backward) share ONE all-reduce of their concatenated [2, C] sums "
                      "(FX_BN_SIBLINGS, default on; bit-identical results); everything else is a chain of data dependencies and cannot be batched "
                      "without changing the arithmetic"}
    # ring all-reduce moves 2 (N-1)/N of the buffer per rank and direction; xGMI is point-to-point (7 links x ~153 GB/s per GPU)
    ring = 2.0 * (world - 1) / max(world, 1) * total * grad_bytes
    return {"world_size": world, "per_rank_batch": per_rank_batch, "global_batch": per_rank_batch * world, "trainable_parameters": total,
            "segment_parameters": {"backbone": seg_numel[0], "pixel_decoder": seg_numel[1], "head": seg_numel[2]},
            "launch_order": ["head", "pixel_decoder", "backbone"], "bucket_elements": buckets, "n_buckets": len(buckets),
            "allreduce_bytes_per_step": total * grad_bytes, "ring_bytes_sent_per_rank": int(ring),
            "other_collectives": ["num_boxes / num_masks: one 4-byte all-reduce per step"] + (
                ["SyncBN: two [2, C] fp32 all-reduces per BatchNorm layer (forward statistics, backward sums)"] if norm == "SyncBN" else []),
            "syncbn": syncbn if norm == "SyncBN" else None,
            "gradient_dtype": "fp32" if grad_bytes == 4 else "bf16"}


class BucketedGradAllReduce:
    """Average a flat gradient buffer across data-parallel ranks in buckets with asynchronous all-reduce, OVERLAPPED with the rest
    of backward - what DistributedDataParallel's reducer does for the reference (utils/distributed/dist.py:138-157).
    One process per GPU, ``torch.distributed`` backend "nccl" (= RCCL over xGMI) on MI355X, "gloo" in the CPU tests.
    xGMI is point-to-point (7 links/GPU), ring all-reduce is per-link bound: few large buckets (default 64 MiB) rather than
    DDP's 25 MiB NVSwitch-tuned default.

    ``segments``: element ranges [start, end) of the flat buffer whose gradients become final TOGETHER during backward, e.g.
    [backbone | hybrid encoder | predictor head] for RT-DETR (parameter order = forward order, so backward finalises them last to
    first).  Buckets never straddle a segment; ``launch_segment(i)`` - called from an autograd hook on the activation that separates
    segment i from the layers before it - starts that segment's collectives while backward continues into the earlier layers.
    ``launch()`` starts whatever has not been started (end of backward), ``wait()`` completes the step."""

    def __init__(self, flat_grad: torch.Tensor, bucket_bytes: int = 64 << 20, group=None, segments: Optional[Sequence[Tuple[int, int]]] = None,
                 bf16: Optional[bool] = None):
        """``bf16`` (default: FX_DP_BF16=1 in the environment, else off): the buckets travel as bfloat16 - half the bytes on the xGMI
        links (86.7 instead of 173.4 MB per step for RT-DETR-L) at the price of rounding every rank's gradient to 8 mantissa bits
        before the sum (the sum itself is accumulated by the collective in bf16 as well).  Each bucket is cast into a staging tensor
        when its segment is launched and written back, averaged, in wait().  Off by default: the reference's DDP reduces fp32 buckets."""
        import torch.distributed as dist

        self.dist, self.group = dist, group
        self.flat = flat_grad
        n = flat_grad.numel()
        per = max(1, bucket_bytes // flat_grad.element_size())
        self.segments = [(int(a), int(b)) for a, b in (segments or [(0, n)])]
        assert self.segments[0][0] == 0 and self.segments[-1][1] == n and all(self.segments[i][1] == self.segments[i + 1][0] for i in range(len(self.segments) - 1))
        self.seg_buckets: List[List[Tuple[int, int]]] = [[(s, min(per, hi - s)) for s in range(lo, hi, per)] for lo, hi in self.segments]
        self.buckets = [bk for sb in self.seg_buckets for bk in sb]
        self.handles: List = []
        self.launched = [False] * len(self.segments)
        self.log: List[Tuple[str, int]] = []      # ("segment", i) / ("backward_end", -1) in launch order: what the overlap test reads
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.before_collective = None   # optional callable: orders the current stream after gradient producers on other streams
        self.bf16 = bool(int(os.environ.get("FX_DP_BF16", "0"))) if bf16 is None else bool(bf16)
        self._staged: List[Tuple[int, int, torch.Tensor]] = []   # (start, numel, bf16 staging tensor) of the buckets in flight

    def launch_segment(self, i: int):
        """Start the all-reduce of segment i's buckets (idempotent within a step): its gradients are final."""
        if self.launched[i]:
            return
        self.launched[i] = True
        self.log.append(("segment", i))
        if self.world == 1:
            return
        if self.before_collective is not None:
            self.before_collective()
        for s, n in self.seg_buckets[i]:
            buf = self.flat[s:s + n]
            if self.bf16:
                buf = buf.to(torch.bfloat16)
                self._staged.append((s, n, buf))
            self.handles.append(self.dist.all_reduce(buf, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))

    def launch(self, first: int = 0, last: Optional[int] = None):
        """End of backward: start every segment that no hook has started yet."""
        self.log.append(("backward_end", -1))
        for i in range(len(self.segments) - 1, -1, -1):
            self.launch_segment(i)

    def wait(self):
        for h in self.handles:
            h.wait()
        self.handles.clear()
        self.launched = [False] * len(self.segments)
        if self.world > 1:
            if self.bf16:
                for s, n, buf in self._staged:
                    self.flat[s:s + n].copy_(buf)
                self._staged.clear()
            self.flat.div_(self.world)
