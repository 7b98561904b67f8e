"""Training-path modules (first slice): the ResNet-vd backbone of the reference as a PyTorch autograd
graph whose nodes are libfocoos_amd.so kernels — forward AND backward.  This is the architecture BASELINE north_star asks
for: "the Python host calling PyTorch-ROCm for autograd glue only and the actual compute as hand-written CDNA4 HIP kernels".

Reference being replaced: ``ConvNormLayer`` (focoos/nn/layers/conv.py:78-98), ``BottleNeck`` / ``ResNet``
(focoos/nn/backbone/resnet.py:72-121,164-266) under ``loss.backward()`` (focoos/trainer/trainer.py:737-760), with
BatchNorm **frozen** (eval statistics, no affine update — the ``freeze_bn`` variant of SURVEY config 4); train-mode batch
statistics / SyncBN are the next slice.

Numerics: bf16 activations and gradients, fp32 master weights and weight gradients (the reference trains under fp16
autocast with fp32 masters — same structure, wider exponent).  Parameter names are the reference's, so a reference
checkpoint loads with ``load_state_dict`` and ``state_dict()`` writes one.

autograd glue = PyTorch sums the gradients of a tensor with two consumers and calls our ``backward``s in topological
order; every FLOP and every byte moved inside a node is a HIP kernel of this repo.
"""
from __future__ import annotations

import ctypes as C
import math
import weakref
import os
from typing import Dict, List, Optional

import torch
from torch import nn

from . import _lib
from ._lib import FX_ACT, FxConvDesc, FxPackEntry, check
from .state_spec import RESNET_BLOCKS

BN_EPS = 1e-5
# Bumped by the training loop after every optimizer step: the fused AdamW kernel updates the fp32 masters through raw
# pointers (tensor._version does not move), and the bf16 weight images must be rebuilt from them.
WEIGHTS_EPOCH = [0]
# Set by train_detr.TrainStep: parameter gradients are accumulated by the kernels straight into the (pre-zeroed) flat
# gradient views held in ``param.grad`` and the backward returns None for them - no per-parameter zero-fill / add launches.
DIRECT_GRAD = [False]
# ResNet bottlenecks as single autograd nodes with fused backward epilogues (off: one node per layer - the form the live-BatchNorm
# path always uses); kept switchable so that tests can compare the two.
FUSED_BLOCKS = [True]


class ZeroArena:
    """One big pre-zeroed fp32 buffer per step for the accumulate-into temporaries (weight-gradient staging, bias sums):
    a single memset instead of hundreds of small fill launches.  Falls back to torch.zeros when not armed (tests)."""

    def __init__(self):
        self.buf = None
        self.used = 0

    def arm(self, numel: int, device):
        if self.buf is None or self.buf.numel() < numel or self.buf.device != torch.device(device):
            self.buf = torch.empty(numel, dtype=torch.float32, device=device)
        self.buf.zero_()
        self.used = 0

    def disarm(self):
        self.buf = None

    def zeros(self, shape, device) -> torch.Tensor:
        n = 1
        for d in shape:
            n *= int(d)
        n_al = (n + 63) // 64 * 64
        if self.buf is None or self.buf.device != torch.device(device) or self.used + n_al > self.buf.numel():
            return torch.zeros(*shape, dtype=torch.float32, device=device)
        out = self.buf[self.used:self.used + n].view(*shape)
        self.used += n_al
        return out


ARENA = ZeroArena()


# Host-side launch cost matters here: a training step issues ~5 000 eager launches and is host-bound once the kernels are
# fast.  Two things are therefore cached: the stream handle (TrainStep pins it for the duration of a step) and the ctypes conv
# descriptors (a fresh 24-field Structure per call costs ~8 us; shapes are static, only the three pointers change).
_STREAM_PIN: Dict[torch.device, C.c_void_p] = {}


def pin_stream(dev, on: bool) -> None:
    dev = torch.device(dev)
    if on:
        _STREAM_PIN[dev] = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    else:
        _STREAM_PIN.pop(dev, None)


def _stream(dev) -> C.c_void_p:
    h = _STREAM_PIN.get(dev)
    return h if h is not None else C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


# Weight gradients are off backward's critical path (nothing in backward consumes them): with a side stream installed here (TrainStep
# does, for the duration of a step) they are launched on it and overlap the input-gradient chain on the main stream - the same effect
# as the two concurrent batch parts of the inference step: a ~250 TFLOP/s wgrad kernel and a bandwidth-bound dgrad / elementwise kernel
# fill each other's idle units.  Only when the kernels accumulate straight into the flat gradient views (DIRECT_GRAD): a gradient
# handed back to autograd would be consumed on the main stream.  The owner joins the streams before it reads the gradients.
WGRAD_STREAM: Dict[torch.device, "torch.cuda.Stream"] = {}


_WGRAD_KEEP: List = []   # operands of the launches queued on the side stream, released when the owner joins the streams


def _wgrad_fork(dev, *tensors):
    """The side stream for a weight-gradient launch (None: stay on the main stream), ordered after everything queued on the main
    stream so far.  ``tensors`` (temporaries / saved activations the launch reads) are kept referenced until the owner joins the
    streams (wgrad_join) - cheaper on the host than the caching allocator's per-tensor record_stream, at the price of the temporaries
    of one backward pass staying allocated until its end."""
    side = WGRAD_STREAM.get(dev)
    if side is None:
        return None
    check(_lib.load().fx_stream_fork(_stream(dev), C.c_void_p(side.cuda_stream)), "fx_stream_fork")
    _WGRAD_KEEP.append(tensors)
    return side


def wgrad_join(dev) -> None:
    """Order the current stream of ``dev`` after the weight-gradient side stream and release the operands held for it."""
    side = WGRAD_STREAM.get(torch.device(dev))
    if side is not None:
        torch.cuda.current_stream(torch.device(dev)).wait_stream(side)
    _WGRAD_KEEP.clear()


# ---- all weight images in one launch -------------------------------------------------------------------------------------------
# Every layer rebuilds its bf16 images lazily (sync_packed / _PackedLinear.sync) when its master weight changed.  A training step
# changes ALL of them, and ~190 small pack launches per step cost more on the host (Python launch path) and on the GPU (launch latency)
# than the conversion itself.  TrainStep therefore owns a WeightPacker: at the top of a step ONE fx_pack_weights_many_f32 launch over a
# device table of the model's stale layers, whose versions are then stamped so that the lazy path finds nothing to do.  The table is
# cached as long as the pointers in it are unchanged.
class WeightPacker:
    def __init__(self, model: nn.Module):
        self.model = model
        self.items: List = []
        self.table = None     # (key, device table, entries, workgroups)

    def _collect(self):
        out = []
        for m in self.model.modules():
            if hasattr(m, "pack_fields"):
                out.append(m)
            for v in vars(m).values():
                if isinstance(v, (_PackedLinear, _PackedLinearGroup)):
                    out.append(v)
        return out

    def pack(self, dev) -> int:
        """Rebuild every stale weight image of the model on ``dev`` with one launch; returns the number of layers packed (0: nothing
        stale, or the layers have not been through their first lazy packing yet)."""
        dev = torch.device(dev)
        if not self.items:
            self.items = self._collect()
        pend = []
        for o in self.items:
            f = o.pack_fields(dev)
            if f is not None:
                pend.append((o, f))
        if len(pend) < 4:   # a handful: the lazy per-layer path
            return 0
        key = tuple(f[1] for _, f in pend)
        if self.table is None or self.table[0] != key:
            # one table entry per master tensor: a layer contributes one field tuple, a group of masters sharing images a tuple of them
            entries = [t for _, (_, fs) in pend for t in (fs if isinstance(fs[0], tuple) else (fs,))]
            arr = (FxPackEntry * len(entries))()
            blocks = 0
            lib = _lib.load()
            for e, t in zip(arr, entries):
                w, scale, bias, w_fwd, w_dgrad, f_fwd, f_dgrad, bias_out, N, Cc, KH, KW, ld_fwd, ld_dgrad = t[:14]
                e.w, e.scale, e.bias, e.w_fwd, e.w_dgrad, e.w_fwd_frag, e.w_dgrad_frag, e.bias_out = w, scale, bias, w_fwd, w_dgrad, f_fwd, f_dgrad, bias_out
                e.N, e.C, e.KH, e.KW, e.ld_fwd, e.ld_dgrad, e.first_block = N, Cc, KH, KW, ld_fwd, ld_dgrad, blocks
                e.n_offset, e.n_total = (t[14], t[15]) if len(t) > 14 else (0, N)
                nb = int(lib.fx_pack_entry_blocks(N, Cc, KH, KW))
                if nb <= 0:
                    raise _lib.FocoosAmdError(f"fx_pack_entry_blocks({N}, {Cc}, {KH}, {KW}) = {nb}")
                blocks += nb
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            self.table = (key, host.to(dev), len(entries), blocks)
        _, table, n, blocks = self.table
        check(_lib.load().fx_pack_weights_many_f32(table.data_ptr(), n, blocks, _stream(dev)), "fx_pack_weights_many_f32")
        for o, (ver, _) in pend:
            o.pack_stamp(ver)
        return n


_DESC_CACHE: Dict[tuple, tuple] = {}

# Kernel census of a training step (tests / profiles): set to a dict and every convolution launch of the autograd graph - forward, input
# gradient, weight gradient - is counted under the label of the kernel the library routes it to (fx_conv2d_variant /
# fx_conv2d_wgrad_variant: the launch's own routing function).  None (default): no bookkeeping on the hot path.
VARIANT_CENSUS: List[Optional[Dict[str, int]]] = [None]


def _census(label: str) -> None:
    c = VARIANT_CENSUS[0]
    c[label] = c.get(label, 0) + 1



def _conv_call(lib, x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], N: int, KH: int, KW: int, stride: int, pad: int,
               act: Optional[str], residual: Optional[torch.Tensor], res_mode: int = 0, out_f32: bool = False,
               w_frag: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fx_conv2d_nhwc_bf16 on NHWC bf16 tensors; ``w`` is a packed [Npad][KH][KW][C] bf16 image, ``w_frag`` its optional copy in MFMA
    fragment order (routes eligible layers to the halo / pointwise kernels).  ``res_mode``: 0 = add ``residual`` before the activation,
    2 = multiply by (residual > 0) - the ReLU backward of the layer that produced ``residual``."""
    B, H, W_, Cc = x.shape
    key = (w.data_ptr(), bias.data_ptr() if bias is not None else 0, B, H, W_, Cc, N, KH, KW, stride, pad, act, residual is not None, res_mode, out_f32,
           w_frag.data_ptr() if w_frag is not None else 0, mask is not None)
    ent = _DESC_CACHE.get(key)
    if ent is None:
        Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W_ + 2 * pad - KW) // stride + 1
        d = FxConvDesc()
        d.w = w.data_ptr()
        d.w_frag = w_frag.data_ptr() if w_frag is not None else None
        d.bias = bias.data_ptr() if bias is not None else None
        d.B, d.H, d.W, d.C, d.ldx = B, H, W_, Cc, Cc
        d.Ho, d.Wo, d.N, d.ldy, d.ldr = Ho, Wo, N, N, N if residual is not None else 0
        d.KH, d.KW, d.stride, d.pad = KH, KW, stride, pad
        d.pool2, d.act, d.out_f32, d.residual_after_act, d.y_batch_stride = 0, FX_ACT[act], int(out_f32), res_mode, 0
        ent = (d, C.byref(d), (B, Ho, Wo, N))
        if len(_DESC_CACHE) > 4096:
            _DESC_CACHE.clear()
        _DESC_CACHE[key] = ent
    d, ref, oshape = ent
    y = torch.empty(oshape, dtype=torch.float32 if out_f32 else _lib.act_dtype(), device=x.device)
    d.x, d.y = x.data_ptr(), y.data_ptr()
    d.residual = residual.data_ptr() if residual is not None else None
    d.mask, d.ldm = (mask.data_ptr(), N) if mask is not None else (None, 0)   # result *= (mask > 0): [B,Ho,Wo,N] bf16 (fx_conv_desc.mask)
    check(lib.fx_conv2d_nhwc_bf16(ref, _stream(x.device)), "fx_conv2d_nhwc_bf16")
    if VARIANT_CENSUS[0] is not None:
        buf = C.create_string_buffer(64)
        check(lib.fx_conv2d_variant(ref, buf, 64), "fx_conv2d_variant")
        _census("conv:" + buf.value.decode())
    return y


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _frag_eligible(rows: int, cin: int, k: int) -> bool:
    """Shapes fx_conv2d_nhwc_bf16 can run on the kernels of conv3x3_flat.hip, given the weight copy in fragment order (``rows`` output
    channels, ``cin`` input channels of the convolution that USES the image - swapped for the input-gradient convolution)."""
    if k == 3:
        return (rows in (64, 128) or rows % 256 == 0) and cin % 64 == 0
    return k == 1 and rows % 256 == 0 and cin % 256 == 0


class _ConvBnActFn(torch.autograd.Function):
    """y = act(conv(x, W * s) + (beta - mu * s) [+ residual]) with s = gamma / sqrt(var + eps) frozen."""

    @staticmethod
    def forward(ctx, x, weight, residual, layer: "ConvNormLayer"):
        lib = layer.lib
        layer.sync_packed()
        N, Cc, KH, KW = weight.shape
        fused = layer.act in (None, "relu")
        z = _conv_call(lib, x, layer.w_fwd, layer.shift, N, KH, KW, layer.stride, layer.pad, layer.act if fused else None, residual,
                       w_frag=layer.w_fwd_frag)
        y = z if fused else _act_fwd(lib, z, layer.act)  # SiLU / GELU: the backward needs the pre-activation
        ctx.layer = layer
        ctx.has_res = residual is not None
        ctx.save_for_backward(x, y if fused else z)
        return y

    @staticmethod
    def backward(ctx, dy):
        layer: ConvNormLayer = ctx.layer
        lib = layer.lib
        x, y = ctx.saved_tensors
        dev = x.device
        dy = dy.contiguous()
        B, Ho, Wo, N = y.shape
        _, H, W_, Cc = x.shape
        st = _stream(dev)
        if layer.act == "relu":
            dz = torch.empty_like(y)
            check(lib.fx_relu_bwd_bf16(dy.data_ptr(), N, None, 0, y.data_ptr(), N, dz.data_ptr(), N, B * Ho * Wo, N, 1, st), "fx_relu_bwd_bf16")
        elif layer.act is None:
            dz = dy
        else:
            dz = _act_bwd(lib, dy, y, layer.act)  # `y` holds the saved pre-activation here
        dx = _conv_input_grad(layer, dz, x.shape) if ctx.needs_input_grad[0] else None
        dw = _conv_param_grads(layer, x, dz, layer.scale) if ctx.needs_input_grad[1] else None
        return dx, dw, (dz if ctx.has_res else None), None


BN_MOMENTUM = 0.1


def _bn_sync_group(layer):
    """SyncBatchNorm semantics (norm "SyncBN"): statistics are summed over the data-parallel group (RCCL all-reduce of one
    [2][C] fp32 vector per layer and direction).  Per-rank row counts are equal by construction (fixed-shape batches)."""
    if layer.norm_mode != "SyncBN":
        return 1
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _bn_local_sums(layer, z: torch.Tensor, sums: torch.Tensor) -> None:
    """This rank's [sum z, sum z^2] per channel of the conv output z into the zeroed [2][N] fp32 view ``sums``."""
    N = z.shape[-1]
    check(layer.lib.fx_bn_stats_bf16(z.data_ptr(), N, int(z.dtype == torch.float32), sums.data_ptr(), z.numel() // N, N, _stream(z.device)), "fx_bn_stats_bf16")


def _bn_finish_forward(layer, z: torch.Tensor, residual: Optional[torch.Tensor], sums: torch.Tensor, n: float):
    """Statistics -> (mean, rstd, scale, shift) + running-statistics update, then y = act(scale z + shift [+ residual])."""
    lib, dev = layer.lib, z.device
    N = z.shape[-1]
    rows = z.numel() // N
    st = _stream(dev)
    zf = int(z.dtype == torch.float32)   # fp32 pre-normalisation tensor (conv epilogue out_f32): y depends on z - mean
    norm = layer._norm_h
    stats = torch.empty(4, N, dtype=torch.float32, device=dev)
    check(lib.fx_bn_finalize_f32(sums.data_ptr(), n, norm.weight.data_ptr(), norm.bias.data_ptr(), BN_EPS, BN_MOMENTUM,
                                 norm.running_mean.data_ptr(), norm.running_var.data_ptr(), norm.num_batches_tracked.data_ptr(),
                                 stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr(), stats[3].data_ptr(), N, st), "fx_bn_finalize_f32")
    layer._stats_epoch = getattr(layer, "_stats_epoch", 0) + 1   # the kernel moved the running statistics behind autograd's back
    y = torch.empty(z.shape, dtype=_lib.act_dtype(), device=dev)
    check(lib.fx_bn_apply_bf16(z.data_ptr(), N, zf, stats[2].data_ptr(), stats[3].data_ptr(), residual.data_ptr() if residual is not None else None, N,
                               FX_ACT[layer.act], y.data_ptr(), N, rows, N, st), "fx_bn_apply_bf16")
    return y, stats


SYNCBN_COLLECTIVES = [0]   # all-reduces issued by the BatchNorm nodes since the counter was last reset (tests / bench census)


def _bn_all_reduce(buf: torch.Tensor) -> None:
    import torch.distributed as dist

    SYNCBN_COLLECTIVES[0] += 1
    dist.all_reduce(buf)


def _bn_forward(layer, z: torch.Tensor, residual: Optional[torch.Tensor]):
    """Batch-statistics BatchNorm + activation on the conv output z [B,Ho,Wo,N]: returns (y, stats[4][N] = mean/rstd/scale/shift, n)."""
    N = z.shape[-1]
    sums = ARENA.zeros((2, N), z.device)
    _bn_local_sums(layer, z, sums)
    world = _bn_sync_group(layer)
    if world > 1:
        _bn_all_reduce(sums)
    n = float((z.numel() // N) * world)
    y, stats = _bn_finish_forward(layer, z, residual, sums, n)
    return y, stats, n


def _bn_bwd_local_sums(layer, dy: torch.Tensor, z: torch.Tensor, residual: Optional[torch.Tensor], stats: torch.Tensor, sums: torch.Tensor) -> None:
    """This rank's [sum da, sum da xhat] per channel (da = dy act'(.)) into the zeroed [2][N] fp32 view ``sums``."""
    N = z.shape[-1]
    rp = residual.data_ptr() if residual is not None else None
    check(layer.lib.fx_bn_bwd_stats_bf16(dy.data_ptr(), N, z.data_ptr(), N, int(z.dtype == torch.float32), rp, N, stats[2].data_ptr(), stats[3].data_ptr(),
                                         stats[0].data_ptr(), stats[1].data_ptr(), FX_ACT[layer.act], sums.data_ptr(), z.numel() // N, N, _stream(z.device)),
          "fx_bn_bwd_stats_bf16")


def _bn_finish_backward(layer, dy: torch.Tensor, z: torch.Tensor, residual: Optional[torch.Tensor], stats: torch.Tensor, n: float, want_affine: bool,
                        sums: torch.Tensor, local: torch.Tensor):
    """``sums``: the (all-reduced) backward sums, ``local``: this rank's own (the same tensor when nothing was reduced).  Returns
    (dz, da or None, dgamma, dbeta)."""
    lib, dev = layer.lib, z.device
    N = z.shape[-1]
    rows = z.numel() // N
    st = _stream(dev)
    rp = residual.data_ptr() if residual is not None else None
    zf = int(z.dtype == torch.float32)
    dz = torch.empty(z.shape, dtype=_lib.act_dtype(), device=dev)
    da = torch.empty(z.shape, dtype=_lib.act_dtype(), device=dev) if residual is not None else None
    norm = layer._norm_h
    direct = want_affine and DIRECT_GRAD[0] and norm.weight.grad is not None and norm.bias.grad is not None
    fused = direct and local is sums    # the kernel adds the (local) sums to the parameter gradients itself
    check(lib.fx_bn_bwd_apply_bf16(dy.data_ptr(), N, z.data_ptr(), N, zf, rp, N, stats[2].data_ptr(), stats[3].data_ptr(), stats[0].data_ptr(),
                                   stats[1].data_ptr(), FX_ACT[layer.act], sums.data_ptr(), 1.0 / n, da.data_ptr() if da is not None else None, N,
                                   dz.data_ptr(), N, rows, N, norm.weight.grad.data_ptr() if fused else None,
                                   norm.bias.grad.data_ptr() if fused else None, st), "fx_bn_bwd_apply_bf16")
    dgamma = dbeta = None
    if want_affine and not fused:
        if direct:    # SyncBN: `sums` was all-reduced, the parameter gradients take this rank's share
            norm.weight.grad.add_(local[1])
            norm.bias.grad.add_(local[0])
        else:
            dgamma, dbeta = local[1].clone(), local[0].clone()
    return dz, da, dgamma, dbeta


def _bn_backward(layer, dy: torch.Tensor, z: torch.Tensor, residual: Optional[torch.Tensor], stats: torch.Tensor, n: float, want_affine: bool):
    """Returns (dz, da or None): dz = gradient of the conv output, da = gradient of the residual branch.  The affine
    gradients (dgamma = sum da * xhat, dbeta = sum da; LOCAL sums, data parallelism averages them later) are added to
    ``norm.weight.grad`` / ``norm.bias.grad`` in place when those exist, else returned through ``layer._affine_grads``."""
    N = z.shape[-1]
    sums = ARENA.zeros((2, N), z.device)
    _bn_bwd_local_sums(layer, dy, z, residual, stats, sums)
    local = sums
    if _bn_sync_group(layer) > 1:
        local = sums.clone()
        _bn_all_reduce(sums)
    return _bn_finish_backward(layer, dy, z, residual, stats, n, want_affine, sums, local)


_WGRAD_WS: Dict[torch.device, torch.Tensor] = {}


_WGRAD_WS_RETIRED: List[torch.Tensor] = []


def _wgrad_workspace(numel: int, dev) -> torch.Tensor:
    """One reusable fp32 staging buffer for the per-pixel-range partial weight gradients (launches are serial on one stream and
    each unpack consumes the partials before the next layer overwrites them)."""
    ws = _WGRAD_WS.get(dev)
    if ws is None or ws.numel() < numel:
        if ws is not None:
            # launches queued earlier on the side stream may still be reading the old buffer: it is retired, not freed (the caching
            # allocator would hand its memory to the next allocation of the main stream); growth happens a handful of times per process
            _WGRAD_WS_RETIRED.append(ws)
        ws = torch.empty(max(numel, 16 << 20), dtype=torch.float32, device=dev)
        _WGRAD_WS[dev] = ws
    return ws


def _conv_param_grads(layer, x: torch.Tensor, dz: torch.Tensor, scale: Optional[torch.Tensor]):
    """Weight gradient of the layer's conv: MFMA wgrad of every pixel range into its own [N][k][k][C] partial slab (plain stores),
    then one pass that sums the slabs, applies the folded-BN scale and re-lays the result out for the master weight - straight
    into ``weight.grad`` when the optimizer's flat views are installed."""
    lib, dev = layer.lib, x.device
    B, H, W_, Cc = x.shape
    _, Ho, Wo, N = dz.shape
    wparam = layer._conv_h.weight
    direct = DIRECT_GRAD[0] and wparam.grad is not None
    side = _wgrad_fork(dev, x, dz) if direct else None
    st = C.c_void_p(side.cuda_stream) if side is not None else _stream(dev)
    k = layer.k
    key = (B, Ho, Wo, Cc, N, k)
    S = layer._wgrad_splits.get(key) if hasattr(layer, "_wgrad_splits") else None
    if S is None:
        S = int(lib.fx_conv2d_wgrad_splits(B, Ho, Wo, Cc, N, k, k))
        if not hasattr(layer, "_wgrad_splits"):
            object.__setattr__(layer, "_wgrad_splits", {})
        layer._wgrad_splits[key] = S
    slab = N * k * k * Cc
    ws = _wgrad_workspace(S * slab, dev)
    check(lib.fx_conv2d_wgrad_partial_nhwc_bf16(x.data_ptr(), Cc, dz.data_ptr(), N, ws.data_ptr(), slab, S, B, H, W_, Cc, Ho, Wo, N, k, k,
                                                layer.stride, layer.pad, st), "fx_conv2d_wgrad_partial_nhwc_bf16")
    if VARIANT_CENSUS[0] is not None:
        buf = C.create_string_buffer(64)
        check(lib.fx_conv2d_wgrad_variant(B, Ho, Wo, Cc, N, k, k, layer.stride, layer.pad, buf, 64), "fx_conv2d_wgrad_variant")
        _census("wgrad:" + buf.value.decode())
    dw = wparam.grad if direct else torch.empty(N, Cc, k, k, dtype=torch.float32, device=dev)
    check(lib.fx_unpack_conv_wgrad_sum_f32(ws.data_ptr(), slab, S, scale.data_ptr() if scale is not None else None, dw.data_ptr(), N, Cc, k, k, Cc,
                                           int(direct), st), "fx_unpack_conv_wgrad_sum_f32")
    return None if direct else dw


def _conv_input_grad(layer, dz: torch.Tensor, x_shape) -> torch.Tensor:
    lib, dev = layer.lib, dz.device
    B, H, W_, Cc = x_shape
    _, Ho, Wo, N = dz.shape
    if layer.stride == 1:
        src = dz
    else:  # stride 2: zero-insert, then the stride-1 transposed filter
        src = torch.empty(B, H, W_, N, dtype=_lib.act_dtype(), device=dev)
        check(lib.fx_zero_insert2_nhwc_bf16(dz.data_ptr(), N, src.data_ptr(), N, B, Ho, Wo, H, W_, N, _stream(dev)), "fx_zero_insert2_nhwc_bf16")
    return _conv_call(lib, src, layer.w_dgrad, None, Cc, layer.k, layer.k, 1, layer.pad, None, None, w_frag=layer.w_dgrad_frag)


class _ConvBnTrainFn(torch.autograd.Function):
    """ConvNormLayer under model.train() with a live BatchNorm (norm "BN" / "SyncBN"): z = conv(x, W);
    y = act(gamma * (z - mean_batch) * rstd_batch + beta [+ residual]); running statistics updated in place."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, residual, layer: "ConvNormLayer"):
        layer.sync_packed()
        N, Cc, KH, KW = weight.shape
        z = _conv_call(layer.lib, x, layer.w_fwd, None, N, KH, KW, layer.stride, layer.pad, None, None, out_f32=True)
        y, stats, n = _bn_forward(layer, z, residual)
        ctx.layer, ctx.n = layer, n
        ctx.save_for_backward(x, z, residual, stats)
        return y

    @staticmethod
    def backward(ctx, dy):
        layer: ConvNormLayer = ctx.layer
        x, z, residual, stats = ctx.saved_tensors
        dz, da, dgamma, dbeta = _bn_backward(layer, dy.contiguous(), z, residual, stats, ctx.n, ctx.needs_input_grad[2] or ctx.needs_input_grad[3])
        dx = _conv_input_grad(layer, dz, x.shape) if ctx.needs_input_grad[0] else None
        dw = _conv_param_grads(layer, x, dz, None) if ctx.needs_input_grad[1] else None
        return dx, dw, dgamma, dbeta, da, None


# SyncBN sibling fusion (round 6).  nn.SyncBatchNorm issues one small all-reduce per layer and direction: 97 BatchNorm layers
# = 194 serialised, latency-bound collectives per RT-DETR-L step (trainer/trainer.py:333-334).  A layer's statistics are a data
# dependency of its own output, so only layers whose conv outputs exist TOGETHER can share one: a bottleneck's shortcut conv with branch2a,
# the conv1 | conv2 halves of a CSP layer (independent outputs: one collective forward, one backward - the node waits for both output
# gradients), and the 3x3 | 1x1 branches of a RepVGG block (the 1x1's epilogue adds the 3x3's normalised output: one collective forward; in
# the backward the 3x3 branch's gradient comes out of the 1x1's post-collective pass, so those two stay separate).  The two [2][N] sum
# vectors lie back to back in ONE buffer that is all-reduced once; every other launch is the per-layer path's, on the same values - the
# results are bit-identical to two _ConvBnTrainFn nodes (tests/test_gpu_train_conv.py::test_syncbn_sibling_nodes_equal_per_layer_nodes).
# 20 pairs: 194 -> 166 collectives per step (forward 97 -> 77, backward 97 -> 89).  FX_BN_SIBLINGS=0: per-layer nodes.
BN_SIBLINGS = [os.environ.get("FX_BN_SIBLINGS", "1") != "0"]


def _siblings_on(layer) -> bool:
    return BN_SIBLINGS[0] and layer.batch_stats and layer.norm_mode == "SyncBN"


class _SiblingConvBnFn(torch.autograd.Function):
    """Two ConvNormLayers under batch statistics whose conv outputs exist together.  chain = False: (ya, yb) = (A(xa), B(xb)), independent.
    chain = True (RepVGG): returns yb = B(xb, residual = A(xa)) only."""

    @staticmethod
    def forward(ctx, xa, xb, wa, ga, ba, wb, gb, bb, la: "ConvNormLayer", lb: "ConvNormLayer", chain: bool):
        la.sync_packed()
        lb.sync_packed()
        dev = xa.device
        Na, Nb = wa.shape[0], wb.shape[0]
        za = _conv_call(la.lib, xa, la.w_fwd, None, Na, wa.shape[2], wa.shape[3], la.stride, la.pad, None, None, out_f32=True)
        zb = _conv_call(lb.lib, xb, lb.w_fwd, None, Nb, wb.shape[2], wb.shape[3], lb.stride, lb.pad, None, None, out_f32=True)
        flat = ARENA.zeros((2 * (Na + Nb),), dev)
        sa, sb = flat[:2 * Na].view(2, Na), flat[2 * Na:].view(2, Nb)
        _bn_local_sums(la, za, sa)
        _bn_local_sums(lb, zb, sb)
        world = _bn_sync_group(la)
        if world > 1:
            _bn_all_reduce(flat)                       # ONE collective for the pair
        na, nb = float((za.numel() // Na) * world), float((zb.numel() // Nb) * world)
        ya, stats_a = _bn_finish_forward(la, za, None, sa, na)
        yb, stats_b = _bn_finish_forward(lb, zb, ya if chain else None, sb, nb)
        ctx.la, ctx.lb, ctx.na, ctx.nb, ctx.chain = la, lb, na, nb, chain
        ctx.save_for_backward(xa, xb, za, zb, stats_a, stats_b, ya if chain else None)
        if chain:
            return yb
        return ya, yb

    @staticmethod
    def backward(ctx, *douts):
        la, lb = ctx.la, ctx.lb
        xa, xb, za, zb, stats_a, stats_b, ya = ctx.saved_tensors
        dev = za.device
        Na, Nb = za.shape[-1], zb.shape[-1]
        want_a = ctx.needs_input_grad[3] or ctx.needs_input_grad[4]
        want_b = ctx.needs_input_grad[6] or ctx.needs_input_grad[7]
        sync = _bn_sync_group(la) > 1
        if ctx.chain:
            (dyb,) = douts
            dzb, da, dgb, dbb = _bn_backward(lb, dyb.contiguous(), zb, ya, stats_b, ctx.nb, want_b)
            dza, _, dga, dba = _bn_backward(la, da, za, None, stats_a, ctx.na, want_a)
        else:
            dya, dyb = douts
            dya = dya.contiguous() if dya is not None else torch.zeros_like(za, dtype=_lib.act_dtype())
            dyb = dyb.contiguous() if dyb is not None else torch.zeros_like(zb, dtype=_lib.act_dtype())
            flat = ARENA.zeros((2 * (Na + Nb),), dev)
            sa, sb = flat[:2 * Na].view(2, Na), flat[2 * Na:].view(2, Nb)
            _bn_bwd_local_sums(la, dya, za, None, stats_a, sa)
            _bn_bwd_local_sums(lb, dyb, zb, None, stats_b, sb)
            la_, lb_ = sa, sb
            if sync:
                loc = flat.clone()
                la_, lb_ = loc[:2 * Na].view(2, Na), loc[2 * Na:].view(2, Nb)
                _bn_all_reduce(flat)                   # ONE collective for the pair
            dza, _, dga, dba = _bn_finish_backward(la, dya, za, None, stats_a, ctx.na, want_a, sa, la_)
            dzb, _, dgb, dbb = _bn_finish_backward(lb, dyb, zb, None, stats_b, ctx.nb, want_b, sb, lb_)
        dxa = _conv_input_grad(la, dza, xa.shape) if ctx.needs_input_grad[0] else None
        dxb = _conv_input_grad(lb, dzb, xb.shape) if ctx.needs_input_grad[1] else None
        # (one input tensor given twice: autograd sums the two slots' gradients exactly as it sums those of two separate nodes)
        dwa = _conv_param_grads(la, xa, dza, None) if ctx.needs_input_grad[2] else None
        dwb = _conv_param_grads(lb, xb, dzb, None) if ctx.needs_input_grad[5] else None
        return dxa, dxb, dwa, dga, dba, dwb, dgb, dbb, None, None, None


def sibling_conv_bn(la: "ConvNormLayer", lb: "ConvNormLayer", xa: torch.Tensor, xb: torch.Tensor, chain: bool = False):
    """(la(xa), lb(xb)) - or lb(xb, residual = la(xa)) with chain = True - through ONE autograd node whose two BatchNorm layers share their
    SyncBN collectives (see _SiblingConvBnFn); the same tensor may be given as xa and xb (autograd sums the two slots' input gradients)."""
    return _SiblingConvBnFn.apply(xa, xb, la._conv_h.weight, la._norm_h.weight, la._norm_h.bias, lb._conv_h.weight, lb._norm_h.weight, lb._norm_h.bias,
                                  la, lb, chain)


class _Holder(nn.Module):
    """Parameter container with the reference's attribute names (``conv.weight``; ``norm.weight/bias/running_*``)."""


class ConvNormLayer(nn.Module):
    """focoos/nn/layers/conv.py:78-98 — conv (no bias) + BatchNorm2d + activation.  ``norm_mode`` (set for the whole model by
    ``set_norm_mode``): "FrozenBN" = running statistics folded into the conv, affine fixed (the reference's freeze_bn /
    FrozenBatchNorm2d); "BN" / "SyncBN" = batch statistics under ``.train()`` with trainable affine and running-statistics
    updates (SyncBN: statistics all-reduced over the data-parallel group), running statistics under ``.eval()``."""

    norm_mode = "FrozenBN"

    def __init__(self, lib, cin: int, cout: int, k: int, stride: int = 1, act: Optional[str] = None, names=("conv", "norm")):
        super().__init__()
        self.lib, self.cin, self.cout, self.k, self.stride, self.act = lib, cin, cout, k, stride, act
        self.pad = (k - 1) // 2
        conv, norm = _Holder(), _Holder()
        self.add_module(names[0], conv)   # "conv"/"norm" (ConvNormLayer) or "0"/"1" (nn.Sequential(conv, bn) in the encoder)
        self.add_module(names[1], norm)
        object.__setattr__(self, "_conv_h", conv)
        object.__setattr__(self, "_norm_h", norm)
        conv.weight = nn.Parameter(torch.empty(cout, cin, k, k, dtype=torch.float32))
        norm.weight = nn.Parameter(torch.ones(cout), requires_grad=False)   # frozen BatchNorm
        norm.bias = nn.Parameter(torch.zeros(cout), requires_grad=False)
        norm.register_buffer("running_mean", torch.zeros(cout))
        norm.register_buffer("running_var", torch.ones(cout))
        norm.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self._packed_version = None
        self.w_fwd = self.w_dgrad = self.scale = self.shift = None
        self.w_fwd_frag = self.w_dgrad_frag = None   # MFMA-fragment-order copies for the halo / pointwise kernels (eligible shapes only)

    @property
    def batch_stats(self) -> bool:
        return self.training and self.norm_mode != "FrozenBN"

    def _fold_norm(self):
        """scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale; recomputed only when the norm tensors
        can have changed (they are constants for FrozenBN, so the per-step weight repack skips these small launches)."""
        norm = self._norm_h
        ver = (norm.weight._version, norm.bias._version, norm.running_var._version, norm.running_mean._version, norm.weight.device,
               WEIGHTS_EPOCH[0] if self.norm_mode != "FrozenBN" else -1, getattr(self, "_stats_epoch", 0))
        if ver == getattr(self, "_norm_version", None):
            return
        self.scale = (norm.weight.double() / torch.sqrt(norm.running_var.double() + BN_EPS)).float().contiguous()
        self._shift_n = (norm.bias.double() - norm.running_mean.double() * self.scale.double()).float().contiguous()
        self._norm_version = ver

    def _pack_version(self):
        w = self._conv_h.weight
        live = self.batch_stats
        return (w._version, self._norm_h.weight._version, self._norm_h.running_var._version, w.device, WEIGHTS_EPOCH[0], live,
                0 if live else getattr(self, "_stats_epoch", 0))

    def _prepare_images(self):
        """Allocate the images (first use / device change) and refresh the folded-BatchNorm scale and shift; returns the arguments of the
        pack kernels: (w, scale, bias, w_fwd, w_dgrad, w_fwd_frag, w_dgrad_frag, bias_out, N, C, KH, KW, ld_fwd, ld_dgrad) as integers."""
        w = self._conv_h.weight
        live = self.batch_stats
        dev = w.device
        N, Cc, k = self.cout, self.cin, self.k
        with torch.no_grad():
            Np, Cp = (N + 127) // 128 * 128, (Cc + 127) // 128 * 128
            if not live:   # fold the (running-statistics) BatchNorm into the conv: scaled weights + per-channel shift
                nver = getattr(self, "_norm_version", None)
                self._fold_norm()
                if self.shift is None or self.shift.device != dev or nver != self._norm_version:
                    self.shift = torch.zeros(Np, dtype=torch.float32, device=dev)
                    self.shift[:N] = self._shift_n
            if self.w_fwd is None or self.w_fwd.device != dev:
                self.w_fwd = torch.zeros(Np, k, k, Cc, dtype=_lib.act_dtype(), device=dev)
                self.w_dgrad = torch.zeros(Cp, k, k, N, dtype=_lib.act_dtype(), device=dev)
                self.w_fwd_frag = (torch.empty(N * k * k * Cc, dtype=_lib.act_dtype(), device=dev)
                                   if self.stride == 1 and _frag_eligible(N, Cc, k) else None)
                self.w_dgrad_frag = torch.empty(N * k * k * Cc, dtype=_lib.act_dtype(), device=dev) if _frag_eligible(Cc, N, k) else None
        return (w.data_ptr(), None if live else self.scale.data_ptr(), None, self.w_fwd.data_ptr(), self.w_dgrad.data_ptr(),
                _ptr(self.w_fwd_frag), _ptr(self.w_dgrad_frag), None, N, Cc, k, k, k * k * Cc, k * k * N)

    def sync_packed(self):
        """(Re)build the bf16 weight images when the master weight changed (optimizer step, load_state_dict).  With batch
        statistics the images hold the plain weights (the normalisation is a separate pass); otherwise BN is folded in."""
        ver = self._pack_version()
        if ver == self._packed_version:
            return
        f = self._prepare_images()
        check(self.lib.fx_pack_conv_weights_f32(f[0], f[1], f[3], f[4], f[5], f[6], f[8], f[9], f[10], f[11], _stream(self._conv_h.weight.device)),
              "fx_pack_conv_weights_f32")
        self._packed_version = ver

    def pack_fields(self, dev):
        """For pack_all: (version, pack-kernel arguments) when the images are stale and live on ``dev``, else None."""
        if self._conv_h.weight.device != dev:
            return None
        ver = self._pack_version()
        return None if ver == self._packed_version else (ver, self._prepare_images())

    def pack_stamp(self, ver):
        self._packed_version = ver

    def forward(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        if self.batch_stats:
            return _ConvBnTrainFn.apply(x, self._conv_h.weight, self._norm_h.weight, self._norm_h.bias, residual, self)
        return _ConvBnActFn.apply(x, self._conv_h.weight, residual, self)


def set_norm_mode(module: nn.Module, mode: str) -> nn.Module:
    """Model-wide BatchNorm behaviour: "FrozenBN" (freeze_bn), "BN", or "SyncBN" (the reference converts BN -> SyncBN for
    multi-GPU runs, focoos/trainer/trainer.py:175-178).  The affine parameters train only with live statistics."""
    if mode not in ("FrozenBN", "BN", "SyncBN"):
        raise ValueError(f"unknown norm mode {mode!r}")
    for m in module.modules():
        if isinstance(m, ConvNormLayer) or getattr(m, "has_batchnorm", False):   # train_bf.DwConvBn / VecBN carry the same attributes
            m.norm_mode = mode
            m._norm_h.weight.requires_grad_(mode != "FrozenBN")
            m._norm_h.bias.requires_grad_(mode != "FrozenBN")
    return module


class _StemFn(torch.autograd.Function):
    """conv1_1: normalise + 3x3 stride-2 conv 3->32 + frozen BN + ReLU straight from the uint8 / fp32 HWC image
    (fx_stem_conv3x3s2).  Backward: weight gradient only (the image needs none) through the generic wgrad kernel on the
    normalised image padded to 8 channels."""

    @staticmethod
    def forward(ctx, images, weight, layer: "StemConv"):
        lib = layer.lib
        layer.sync_packed()
        B, H, W_, _ = images.shape
        # ceil(H/2) x ceil(W/2): Conv2d(3, 32, 3, stride 2, padding 1) writes (H - 1) // 2 + 1 rows (pixel_ops.hip stem_launch) - odd sizes included
        y = torch.empty(B, (H - 1) // 2 + 1, (W_ - 1) // 2 + 1, 32, dtype=_lib.act_dtype(), device=images.device)
        check(lib.fx_stem_conv3x3s2(images.data_ptr(), int(images.dtype == torch.float32), layer.stem_w.data_ptr(), layer.stem_b.data_ptr(),
                                    layer.px_mean.data_ptr(), layer.px_inv_std.data_ptr(), y.data_ptr(), B, H, W_, 32, _stream(images.device)),
              "fx_stem_conv3x3s2")
        ctx.layer = layer
        ctx.save_for_backward(images, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        layer: StemConv = ctx.layer
        lib = layer.lib
        images, y = ctx.saved_tensors
        dev = images.device
        B, H, W_, _ = images.shape
        Ho, Wo = (H - 1) // 2 + 1, (W_ - 1) // 2 + 1
        st = _stream(dev)
        dy = dy.contiguous()
        dz = torch.empty_like(y)
        check(lib.fx_relu_bwd_bf16(dy.data_ptr(), 32, None, 0, y.data_ptr(), 32, dz.data_ptr(), 32, B * Ho * Wo, 32, 1, st), "fx_relu_bwd_bf16")
        return None, _stem_wgrad(layer, images, dz, layer.scale), None


class _StemTrainFn(torch.autograd.Function):
    """conv1_1 with batch statistics: the un-normalised conv through fx_stem_conv3x3s2_linear, then the BatchNorm passes."""

    @staticmethod
    def forward(ctx, images, weight, gamma, beta, layer: "StemConv"):
        lib = layer.lib
        layer.sync_packed()
        B, H, W_, _ = images.shape
        z = torch.empty(B, (H - 1) // 2 + 1, (W_ - 1) // 2 + 1, 32, dtype=_lib.act_dtype(), device=images.device)
        check(lib.fx_stem_conv3x3s2_linear(images.data_ptr(), int(images.dtype == torch.float32), layer.stem_w.data_ptr(), layer.stem_b.data_ptr(),
                                           layer.px_mean.data_ptr(), layer.px_inv_std.data_ptr(), z.data_ptr(), B, H, W_, 32,
                                           _stream(images.device)), "fx_stem_conv3x3s2_linear")
        y, stats, n = _bn_forward(layer, z, None)
        ctx.layer, ctx.n = layer, n
        ctx.save_for_backward(images, z, stats)
        return y

    @staticmethod
    def backward(ctx, dy):
        layer: StemConv = ctx.layer
        images, z, stats = ctx.saved_tensors
        dz, _, dgamma, dbeta = _bn_backward(layer, dy.contiguous(), z, None, stats, ctx.n, ctx.needs_input_grad[2] or ctx.needs_input_grad[3])
        return None, _stem_wgrad(layer, images, dz, None), dgamma, dbeta, None


def _stem_wgrad(layer, images, dz, scale):
    lib, dev = layer.lib, images.device
    B, H, W_, _ = images.shape
    st = _stream(dev)
    xn = torch.empty(B, H, W_, 8, dtype=_lib.act_dtype(), device=dev)
    check(lib.fx_normalize_pad8(images.data_ptr(), int(images.dtype == torch.float32), layer.px_mean.data_ptr(), layer.px_inv_std.data_ptr(),
                                xn.data_ptr(), B * H * W_, st), "fx_normalize_pad8")
    dw_eff = ARENA.zeros((32, 3, 3, 8), dev)
    check(lib.fx_conv2d_wgrad_nhwc_bf16(xn.data_ptr(), 8, dz.data_ptr(), 32, dw_eff.data_ptr(), B, H, W_, 8, (H - 1) // 2 + 1, (W_ - 1) // 2 + 1, 32, 3, 3, 2, 1, st),
          "fx_conv2d_wgrad_nhwc_bf16")
    dw = torch.empty(32, 3, 3, 3, dtype=torch.float32, device=dev)
    check(lib.fx_unpack_conv_wgrad_f32(dw_eff.data_ptr(), scale.data_ptr() if scale is not None else None, dw.data_ptr(), 32, 3, 3, 3, 8, 0, st),
          "fx_unpack_conv_wgrad_f32")
    return dw


class StemConv(ConvNormLayer):
    def __init__(self, lib, pixel_mean, pixel_std, names=("conv", "norm")):
        super().__init__(lib, 3, 32, 3, 2, "relu", names=names)
        self.register_buffer("px_mean", torch.tensor(pixel_mean, dtype=torch.float32), persistent=False)
        self.register_buffer("px_inv_std", 1.0 / torch.tensor(pixel_std, dtype=torch.float32), persistent=False)
        self.stem_w = self.stem_b = None

    def sync_packed(self):
        w = self._conv_h.weight
        live = self.batch_stats
        ver = (w._version, self._norm_h.weight._version, self._norm_h.running_var._version, w.device, WEIGHTS_EPOCH[0], live,
               0 if live else getattr(self, "_stats_epoch", 0))
        if ver == self._packed_version:
            return
        with torch.no_grad():
            if live:
                self.stem_b = torch.zeros(32, dtype=torch.float32, device=w.device)
                self.stem_w = w.permute(2, 3, 1, 0).contiguous()
            else:
                self._fold_norm()
                self.stem_b = self._shift_n
                self.stem_w = (w * self.scale.view(-1, 1, 1, 1)).permute(2, 3, 1, 0).contiguous()  # [kh][kw][c][n] fp32 (tiny: 864 values)
        self._packed_version = ver

    def pack_fields(self, dev):   # fp32 [kh][kw][c][n] weights through its own sync_packed: not part of the multi-tensor packing
        return None

    def forward(self, images: torch.Tensor) -> torch.Tensor:  # type: ignore[override]
        if self.batch_stats:
            return _StemTrainFn.apply(images, self._conv_h.weight, self._norm_h.weight, self._norm_h.bias, self)
        return _StemFn.apply(images, self._conv_h.weight, self)


class _PoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, lib, kind: str):
        B, H, W_, Cc = x.shape
        Ho, Wo = ((H + 2 - 3) // 2 + 1, (W_ + 2 - 3) // 2 + 1) if kind == "max" else ((H + 1) // 2, (W_ + 1) // 2)
        y = torch.empty(B, Ho, Wo, Cc, dtype=_lib.act_dtype(), device=x.device)
        fn = lib.fx_maxpool3x3s2_nhwc_bf16 if kind == "max" else lib.fx_avgpool2x2_nhwc_bf16
        check(fn(x.data_ptr(), Cc, y.data_ptr(), Cc, B, H, W_, Cc, _stream(x.device)), fn.__name__)
        ctx.lib, ctx.kind = lib, kind
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        lib = ctx.lib
        B, H, W_, Cc = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        if ctx.kind == "max":
            arg = torch.empty(dy.numel(), dtype=torch.uint8, device=x.device)   # arg-max tap per output element
            check(lib.fx_maxpool3x3s2_bwd_nhwc_bf16(x.data_ptr(), Cc, dy.data_ptr(), Cc, dx.data_ptr(), Cc, B, H, W_, Cc, arg.data_ptr(),
                                                    _stream(x.device)), "fx_maxpool3x3s2_bwd_nhwc_bf16")
        else:
            check(lib.fx_avgpool2x2_bwd_nhwc_bf16(dy.data_ptr(), Cc, dx.data_ptr(), Cc, B, H, W_, Cc, _stream(x.device)), "fx_avgpool2x2_bwd_nhwc_bf16")
        return dx, None, None


class _Short(nn.Module):
    """Variant-d shortcut: AvgPool2d(2,2,0,ceil_mode=True) + 1x1 ConvNormLayer (resnet.py:89-100); keys ``short.conv.*``."""

    def __init__(self, lib, cin, cout):
        super().__init__()
        self.lib = lib
        self.conv = ConvNormLayer(lib, cin, cout, 1, 1, None)

    def forward(self, x):
        return self.conv(_PoolFn.apply(x, self.lib, "avg"))


class _BottleneckFn(torch.autograd.Function):
    """A whole ResNet-vd bottleneck (frozen BatchNorm) as ONE autograd node.  Forward = the four fused conv launches of the
    per-layer path.  Backward, relative to chaining the per-layer nodes through autograd:
      * the ReLU backward of branch2a / branch2b is the epilogue of the dgrad convolution that produces their output gradient
        (conv epilogue mode 2: multiply by (saved activation > 0)) - two elementwise passes per block disappear;
      * the two gradients of the block input (main branch + shortcut) are summed in the epilogue of branch2a's dgrad convolution
        (its `residual` operand) instead of by an autograd add kernel over the largest tensors of the network;
      * one Python node instead of four to six."""

    @staticmethod
    def forward(ctx, x, wa, wb, wc, ws, blk: "BottleNeck"):
        a_l, b_l, c_l = blk.branch2a, blk.branch2b, blk.branch2c
        lib = a_l.lib
        for l in (a_l, b_l, c_l):
            l.sync_packed()
        a = _conv_call(lib, x, a_l.w_fwd, a_l.shift, a_l.cout, 1, 1, 1, 0, "relu", None, w_frag=a_l.w_fwd_frag)
        b = _conv_call(lib, a, b_l.w_fwd, b_l.shift, b_l.cout, 3, 3, b_l.stride, 1, "relu", None, w_frag=b_l.w_fwd_frag)
        pooled = None
        if blk.has_short:
            s_l = blk.short.conv if isinstance(blk.short, _Short) else blk.short
            s_l.sync_packed()
            sx = x
            if isinstance(blk.short, _Short):
                B, H, W_, Cc = x.shape
                pooled = torch.empty(B, (H + 1) // 2, (W_ + 1) // 2, Cc, dtype=_lib.act_dtype(), device=x.device)
                check(lib.fx_avgpool2x2_nhwc_bf16(x.data_ptr(), Cc, pooled.data_ptr(), Cc, B, H, W_, Cc, _stream(x.device)), "fx_avgpool2x2_nhwc_bf16")
                sx = pooled
            short = _conv_call(lib, sx, s_l.w_fwd, s_l.shift, s_l.cout, 1, 1, 1, 0, None, None, w_frag=s_l.w_fwd_frag)
        else:
            short = x
        y = _conv_call(lib, b, c_l.w_fwd, c_l.shift, c_l.cout, 1, 1, 1, 0, "relu", short, w_frag=c_l.w_fwd_frag)
        ctx.blk = blk
        ctx.save_for_backward(x, a, b, y, pooled)
        return y

    @staticmethod
    def backward(ctx, dy):
        blk: BottleNeck = ctx.blk
        x, a, b, y, pooled = ctx.saved_tensors
        a_l, b_l, c_l = blk.branch2a, blk.branch2b, blk.branch2c
        lib, dev = a_l.lib, x.device
        st = _stream(dev)
        ref = blk._premasked_ref
        premasked = ref is not None and ref() is dy   # the very tensor the next block's backward produced (not a copy, not a sum)
        blk._premasked_ref = None
        dy = dy.contiguous()
        B, Ho, Wo, N = y.shape
        if premasked and dy.dtype == y.dtype:
            dz_c = dy   # the next block's backward already applied relu'(y) in the epilogue of its branch2a input gradient (below)
            PREMASK_HITS[0] += 1
        else:
            dz_c = torch.empty_like(y)   # gradient of conv_c's output AND of the shortcut branch (pre-activation residual add)
            check(lib.fx_relu_bwd_bf16(dy.data_ptr(), N, None, 0, y.data_ptr(), N, dz_c.data_ptr(), N, B * Ho * Wo, N, 1, st), "fx_relu_bwd_bf16")
        need = ctx.needs_input_grad
        dwc = _conv_param_grads(c_l, b, dz_c, c_l.scale) if need[3] else None
        dz_b = _conv_call(lib, dz_c, c_l.w_dgrad, None, c_l.cin, 1, 1, 1, 0, None, b, res_mode=2, w_frag=c_l.w_dgrad_frag)       # dgrad_c * relu'(b)
        dwb = _conv_param_grads(b_l, a, dz_b, b_l.scale) if need[2] else None
        if b_l.stride == 1:
            src = dz_b
        else:
            Ba, Ha, Wa, Ca = a.shape
            src = torch.empty(Ba, Ha, Wa, b_l.cout, dtype=_lib.act_dtype(), device=dev)
            check(lib.fx_zero_insert2_nhwc_bf16(dz_b.data_ptr(), b_l.cout, src.data_ptr(), b_l.cout, Ba, dz_b.shape[1], dz_b.shape[2], Ha, Wa, b_l.cout, st),
                  "fx_zero_insert2_nhwc_bf16")
        dz_a = _conv_call(lib, src, b_l.w_dgrad, None, b_l.cin, 3, 3, 1, 1, None, a, res_mode=2, w_frag=b_l.w_dgrad_frag)       # dgrad_b * relu'(a)
        dwa = _conv_param_grads(a_l, x, dz_a, a_l.scale) if need[1] else None
        dws = None
        if blk.has_short:
            s_l = blk.short.conv if isinstance(blk.short, _Short) else blk.short
            sx = pooled if pooled is not None else x
            dws = _conv_param_grads(s_l, sx, dz_c, s_l.scale) if need[4] else None
            dshort = None
            if need[0]:
                dshort = _conv_call(lib, dz_c, s_l.w_dgrad, None, s_l.cin, 1, 1, 1, 0, None, None, w_frag=s_l.w_dgrad_frag)
                if pooled is not None:
                    dxs = torch.empty_like(x)
                    check(lib.fx_avgpool2x2_bwd_nhwc_bf16(dshort.data_ptr(), s_l.cin, dxs.data_ptr(), s_l.cin, x.shape[0], x.shape[1], x.shape[2], s_l.cin, st),
                          "fx_avgpool2x2_bwd_nhwc_bf16")
                    dshort = dxs
        else:
            dshort = dz_c
        # + shortcut gradient in the epilogue; and, when x is nothing but the previous block's ReLU output (premask_input), x that block's
        # relu'(.) as well - its backward then starts from dz_c directly (one pass over the widest tensors of the stage less per block)
        prev = getattr(blk, "_prev_block", None)
        pm = x if (prev is not None and PREMASK[0] and need[0]) else None
        dx = _conv_call(lib, dz_a, a_l.w_dgrad, None, a_l.cin, 1, 1, 1, 0, None, dshort, w_frag=None if pm is not None else a_l.w_dgrad_frag,
                        mask=pm) if need[0] else None
        if pm is not None:
            # tell the previous block which tensor arrives pre-masked - by object identity through a weak reference: if autograd hands it
            # anything else (a copy, a sum of several gradients) it applies its own ReLU backward, and masking twice is the identity
            prev._premasked_ref = weakref.ref(dx)
        return dx, dwa, dwb, dwc, dws, None


class BottleNeck(nn.Module):
    """focoos/nn/backbone/resnet.py:72-121 (variant d: stride on the 3x3)."""

    def __init__(self, lib, ch_in, width, stride, shortcut: bool, first_stage: bool):
        super().__init__()
        self.branch2a = ConvNormLayer(lib, ch_in, width, 1, 1, "relu")
        self.branch2b = ConvNormLayer(lib, width, width, 3, stride, "relu")
        self.branch2c = ConvNormLayer(lib, width, width * 4, 1, 1, "relu")  # ReLU applied AFTER the residual add (fused epilogue)
        self.has_short = not shortcut
        self._premasked_ref = None   # weak reference to an output gradient that arrives with relu'(output) already applied (see _BottleneckFn.backward)
        if self.has_short:
            self.short = ConvNormLayer(lib, ch_in, width * 4, 1, 1, None) if (first_stage or stride == 1) else _Short(lib, ch_in, width * 4)

    def forward(self, x):
        if not self.branch2a.batch_stats and FUSED_BLOCKS[0]:   # frozen / eval BatchNorm: the whole block is one autograd node
            s_l = (self.short.conv if isinstance(self.short, _Short) else self.short) if self.has_short else None
            return _BottleneckFn.apply(x, self.branch2a._conv_h.weight, self.branch2b._conv_h.weight, self.branch2c._conv_h.weight,
                                       s_l._conv_h.weight if s_l is not None else None, self)
        if self.has_short and _siblings_on(self.branch2a):   # SyncBN: branch2a and the shortcut conv share their collectives
            pooled = isinstance(self.short, _Short)
            s_l = self.short.conv if pooled else self.short
            a_out, short = sibling_conv_bn(self.branch2a, s_l, x, _PoolFn.apply(x, self.short.lib, "avg") if pooled else x)
            return self.branch2c(self.branch2b(a_out), residual=short)
        out = self.branch2b(self.branch2a(x))
        short = self.short(x) if self.has_short else x
        return self.branch2c(out, residual=short)  # relu(conv + bn + short)


# ReLU backward of a bottleneck's output fused into the NEXT block's branch2a input-gradient convolution (fx_conv_desc.mask), for the stages
# whose block-input gradient runs on the implicit-GEMM kernels (res2 / res3: branch width <= 128 - the widest tensors of the network; the
# deeper stages' pointwise layers run on the resident-tile kernel, which has no second epilogue operand).  FX_PREMASK=0: off.
PREMASK = [os.environ.get("FX_PREMASK", "1") != "0"]
PREMASK_HITS = [0]   # backward passes that started from a pre-masked gradient (tests assert the hand-over really happens)


class _Blocks(nn.Module):
    def __init__(self, blocks):
        super().__init__()
        self.blocks = nn.ModuleList(blocks)
        for prev, cur in zip(blocks[:-1], blocks[1:]):   # cur's input is prev's ReLU output and nothing else consumes it
            if isinstance(cur, BottleNeck) and isinstance(prev, BottleNeck) and cur.branch2a.cout <= 128:
                object.__setattr__(cur, "_prev_block", prev)   # plain reference: not a registered submodule

    def forward(self, x):
        for b in self.blocks:
            x = b(x)
        return x


class ResNetVd(nn.Module):
    """Trainable ResNet-vd (depth 50/101) on the HIP kernels; ``state_dict()`` keys = the reference's
    ``pixel_decoder.backbone.*`` names without the prefix.  Input: uint8 or fp32 HWC images [B,H,W,3] (0..255) on the GPU;
    output dict res2..res5 of NHWC bf16 tensors."""

    def __init__(self, depth: int = 50, pixel_mean=(123.675, 116.28, 103.53), pixel_std=(58.395, 57.12, 57.375)):
        super().__init__()
        if not torch.cuda.is_available():
            raise _lib.FocoosAmdError("focoos_amd needs a ROCm GPU (gfx950); no CPU fallback exists")
        self.lib = lib = _lib.load()
        self.conv1 = nn.Module()
        self.conv1.conv1_1 = StemConv(lib, pixel_mean, pixel_std)
        self.conv1.conv1_2 = ConvNormLayer(lib, 32, 32, 3, 1, "relu")
        self.conv1.conv1_3 = ConvNormLayer(lib, 32, 64, 3, 1, "relu")
        layers = []
        ch_in = 64
        for si, (nblk, width) in enumerate(zip(RESNET_BLOCKS[depth], [64, 128, 256, 512])):
            blocks = []
            for bi in range(nblk):
                stride = 2 if (bi == 0 and si != 0) else 1
                blocks.append(BottleNeck(lib, ch_in, width, stride, shortcut=bi != 0, first_stage=si == 0))
                ch_in = width * 4
            layers.append(_Blocks(blocks))
        self.res_layers = nn.ModuleList(layers)

    def forward(self, images: torch.Tensor) -> Dict[str, torch.Tensor]:
        x = self.conv1.conv1_1(images)
        x = self.conv1.conv1_2(x)
        x = self.conv1.conv1_3(x)
        x = _PoolFn.apply(x, self.lib, "max")
        outs = {}
        for si, layer in enumerate(self.res_layers):
            x = layer(x)
            outs[f"res{si + 2}"] = x
        return outs

    def trainable_parameters(self) -> List[nn.Parameter]:
        return [p for p in self.parameters() if p.requires_grad]


# ================================================================================================ token-space layers
def _rows(t: torch.Tensor) -> int:
    return t.numel() // t.shape[-1]


def _act_fwd(lib, z, act):
    y = torch.empty_like(z)
    Cc = z.shape[-1]
    check(lib.fx_act_fwd_bf16(z.data_ptr(), Cc, y.data_ptr(), Cc, _rows(z), Cc, FX_ACT[act], _stream(z.device)), "fx_act_fwd_bf16")
    return y


def _act_bwd(lib, dy, z, act):
    dz = torch.empty_like(z)
    Cc = z.shape[-1]
    check(lib.fx_act_bwd_bf16(dy.data_ptr(), Cc, z.data_ptr(), Cc, dz.data_ptr(), Cc, _rows(z), Cc, FX_ACT[act], _stream(z.device)), "fx_act_bwd_bf16")
    return dz


def _rup(v: int, m: int) -> int:
    return (v + m - 1) // m * m


class _PackedLinear:
    """bf16 images of a [N, K] fp32 weight (row range r0:r1 of a parameter): forward layout and its transpose.  The kernels
    want K % 32 == 0 and N % 8 == 0; other shapes (bbox heads N = 4, query-pos head K = 4, score heads N = 365) are
    zero-padded here and sliced by the caller."""

    def __init__(self):
        self.ver = None
        self.w_fwd = self.w_t = self.bias = self.w_fwd_frag = self.w_t_frag = self._src = None
        self.Np = self.Kp = 0

    def _version(self, weight, bias, r0, r1):
        return (weight._version, None if bias is None else bias._version, weight.device, r0, r1, WEIGHTS_EPOCH[0])

    def _prepare(self, weight, bias, r0, r1):
        dev = weight.device
        N, K = r1 - r0, weight.shape[1]
        Np, Kp = _rup(N, 32), _rup(K, 32)  # both serve as the reduction dim of one of the two GEMMs (C % 32 == 0)
        self.Np, self.Kp = Np, Kp
        with torch.no_grad():
            w = weight[r0:r1]   # rows of a contiguous [N_total, K(, 1, 1)] master ([N, K, 1, 1] conv weights of train_bf.Conv1x1 are GEMM weights too)
            assert w.is_contiguous()
            if self.w_fwd is None or self.w_fwd.device != dev:
                self.w_fwd = torch.zeros(_rup(Np, 128), 1, 1, Kp, dtype=_lib.act_dtype(), device=dev)
                self.w_t = torch.zeros(_rup(Kp, 128), 1, 1, Np, dtype=_lib.act_dtype(), device=dev)
                ok = _frag_eligible(Np, Kp, 1) and Np == N and Kp == K   # 256-multiples both ways: the flat pointwise kernel, forward and input gradient
                self.w_fwd_frag = torch.empty(Np * Kp, dtype=_lib.act_dtype(), device=dev) if ok else None
                self.w_t_frag = torch.empty(Np * Kp, dtype=_lib.act_dtype(), device=dev) if ok else None
            if self.bias is None or self.bias.device != dev or self.bias.numel() != _rup(Np, 128):
                self.bias = torch.zeros(_rup(Np, 128), dtype=torch.float32, device=dev)   # allocated (and its padding zeroed) once
            bsrc = bias[r0:r1] if bias is not None else None
        return (w.data_ptr(), None, _ptr(bsrc), self.w_fwd.data_ptr(), self.w_t.data_ptr(), _ptr(self.w_fwd_frag), _ptr(self.w_t_frag),
                self.bias.data_ptr() if bsrc is not None else None, N, K, 1, 1, Kp, Np)

    def sync(self, lib, weight, bias, r0, r1):
        ver = self._version(weight, bias, r0, r1)
        if ver == self.ver:
            return
        f = self._prepare(weight, bias, r0, r1)
        dev = weight.device
        check(lib.fx_pack_linear_weights_f32(f[0], f[2], f[3], f[4], f[7], f[8], f[9], f[13], f[12], _stream(dev)), "fx_pack_linear_weights_f32")
        if self.w_fwd_frag is not None:   # (the multi-tensor path writes the fragment copies itself)
            check(lib.fx_pack_frag_bf16(self.w_fwd.data_ptr(), self.w_fwd_frag.data_ptr(), self.Np, self.Kp, _stream(dev)), "fx_pack_frag_bf16")
            check(lib.fx_pack_frag_bf16(self.w_t.data_ptr(), self.w_t_frag.data_ptr(), self.Kp, self.Np, _stream(dev)), "fx_pack_frag_bf16")
        self.ver = ver
        self._src = (weight, bias, r0, r1)

    def pack_fields(self, dev):
        if self._src is None:    # not through its first (lazy) packing yet
            return None
        weight, bias, r0, r1 = self._src
        if weight.device != dev:
            return None
        ver = self._version(weight, bias, r0, r1)
        return None if ver == self.ver else (ver, self._prepare(weight, bias, r0, r1))

    def pack_stamp(self, ver):
        self.ver = ver


class _PackedLinearGroup:
    """bf16 images of G equally shaped [N, K] fp32 weights (+ biases) stacked along the output dimension: ONE forward GEMM with G*N outputs
    and ONE input-gradient GEMM whose reduction runs over all G*N channels - the six value projections of the decoder read the same
    ``memory`` (the inference plan's ``value_all``).  N, K multiples of 32."""

    def __init__(self):
        self.ver = None
        self.w_fwd = self.w_t = self.bias = self.w_fwd_frag = self.w_t_frag = self._src = None
        self.G = self.N = self.K = 0

    def _version(self, weights, biases):
        return (tuple(w._version for w in weights), tuple(b._version for b in biases), weights[0].device, WEIGHTS_EPOCH[0])

    def _prepare(self, weights, biases):
        dev = weights[0].device
        G, (N, K) = len(weights), weights[0].shape
        assert N % 32 == 0 and K % 32 == 0 and all(tuple(w.shape) == (N, K) and w.is_contiguous() for w in weights)
        self.G, self.N, self.K = G, N, K
        Nt = G * N
        if self.w_fwd is None or self.w_fwd.device != dev:
            self.w_fwd = torch.zeros(_rup(Nt, 128), 1, 1, K, dtype=_lib.act_dtype(), device=dev)
            self.w_t = torch.zeros(_rup(K, 128), 1, 1, Nt, dtype=_lib.act_dtype(), device=dev)
            self.bias = torch.zeros(_rup(Nt, 128), dtype=torch.float32, device=dev)
            ok = _frag_eligible(Nt, K, 1)
            self.w_fwd_frag = torch.empty(Nt * K, dtype=_lib.act_dtype(), device=dev) if ok else None
            self.w_t_frag = torch.empty(Nt * K, dtype=_lib.act_dtype(), device=dev) if ok else None
        return tuple((w.data_ptr(), None, b.data_ptr(), self.w_fwd.data_ptr(), self.w_t.data_ptr(), _ptr(self.w_fwd_frag), _ptr(self.w_t_frag),
                      self.bias.data_ptr(), N, K, 1, 1, K, Nt, g * N, Nt) for g, (w, b) in enumerate(zip(weights, biases)))

    def sync(self, lib, weights, biases):
        ver = self._version(weights, biases)
        if ver == self.ver:
            return
        dev = weights[0].device
        N, K, Nt = weights[0].shape[0], weights[0].shape[1], len(weights) * weights[0].shape[0]
        for g, f in enumerate(self._prepare(weights, biases)):   # lazy path: one launch per master into its rows / columns of the shared images
            check(lib.fx_pack_linear_weights_f32(f[0], f[2], f[3] + g * N * K * 2, f[4] + g * N * 2, f[7] + g * N * 4, N, K, Nt, K, _stream(dev)),
                  "fx_pack_linear_weights_f32")
        if self.w_fwd_frag is not None:
            check(lib.fx_pack_frag_bf16(self.w_fwd.data_ptr(), self.w_fwd_frag.data_ptr(), Nt, K, _stream(dev)), "fx_pack_frag_bf16")
            check(lib.fx_pack_frag_bf16(self.w_t.data_ptr(), self.w_t_frag.data_ptr(), K, Nt, _stream(dev)), "fx_pack_frag_bf16")
        self.ver = ver
        self._src = (tuple(weights), tuple(biases))

    def pack_fields(self, dev):
        if self._src is None:
            return None
        weights, biases = self._src
        if weights[0].device != dev:
            return None
        ver = self._version(weights, biases)
        return None if ver == self.ver else (ver, self._prepare(weights, biases))

    def pack_stamp(self, ver):
        self.ver = ver


def _pad_last(t: torch.Tensor, n: int) -> torch.Tensor:
    if t.shape[-1] == n:
        return t.contiguous()
    out = torch.zeros(*t.shape[:-1], n, dtype=t.dtype, device=t.device)
    out[..., : t.shape[-1]] = t
    return out


class _LinearFn(torch.autograd.Function):
    """y = act(x @ W[r0:r1]^T + b[r0:r1] [+ residual]) on [..., K] bf16 rows; W, b fp32 master parameters."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, pack: _PackedLinear, lib, r0, r1, act):
        pack.sync(lib, weight, bias, r0, r1)
        N, K = r1 - r0, weight.shape[1]
        Np, Kp = pack.Np, pack.Kp
        x2 = _pad_last(x.reshape(1, 1, -1, K), Kp)
        res2 = _pad_last(residual.reshape(1, 1, -1, N), Np) if residual is not None else None
        fused = act in (None, "relu")
        z = _conv_call(lib, x2, pack.w_fwd, pack.bias, Np, 1, 1, 1, 0, act if fused else None, res2, w_frag=pack.w_fwd_frag)
        y = z if fused else _act_fwd(lib, z, act)
        ctx.lib, ctx.pack, ctx.rng, ctx.act, ctx.has_res, ctx.has_bias = lib, pack, (r0, r1), act, residual is not None, bias is not None
        ctx.wshape, ctx.K = tuple(weight.shape), K
        ctx.wparam, ctx.bparam = weight, bias
        ctx.save_for_backward(x2, y if fused else z)
        out = y.reshape(*x.shape[:-1], Np)
        return out if Np == N else out[..., :N].contiguous()

    @staticmethod
    def _param_grads(ctx, x2, dz, want_w, want_b, direct, bdirect, st):
        """Weight (+ bias) gradient of the layer; returns (dw, db) for autograd, both None when the kernels wrote the flat views."""
        lib, pack, (r0, r1) = ctx.lib, ctx.pack, ctx.rng
        dev = x2.device
        N, K, Np, Kp = r1 - r0, ctx.K, pack.Np, pack.Kp
        R = x2.shape[2]
        dw = db = None
        same = Np == N and Kp == K
        if direct and not same and (bdirect or not want_b) and ctx.wparam.grad.is_contiguous():
            # padded operands of a narrow layer (N = 4 / 365, K = 4): the kernel accumulates the valid corner straight into the master
            # gradients (fx_linear_wgrad_bias_bf16) - no padded staging matrix, slice and add per layer
            wg = ctx.wparam.grad[r0:r1]
            bg = ctx.bparam.grad[r0:r1] if want_b else None
            check(lib.fx_linear_wgrad_bias_bf16(x2.data_ptr(), Kp, dz.data_ptr(), Np, wg.data_ptr(), K, bg.data_ptr() if bg is not None else None, R, Kp, Np,
                                                K, N, st), "fx_linear_wgrad_bias_bf16")
            return None, None
        if direct and same:
            wt = ctx.wparam.grad[r0:r1]
        else:
            dw = None if direct else ARENA.zeros(ctx.wshape, dev)
            wt = dw[r0:r1] if (same and not direct) else ARENA.zeros((Np, Kp), dev)
        wshape_rows = (r1 - r0,) + tuple(ctx.wshape[1:])   # [N, K] or [N, K, 1, 1]
        bt = None
        if want_b:
            if bdirect and Np == N:
                bt = ctx.bparam.grad[r0:r1]
            else:
                db = None if bdirect else ARENA.zeros((ctx.wshape[0],), dev)
                bt = db[r0:r1] if (Np == N and not bdirect) else ARENA.zeros((Np,), dev)
        # one launch: weight gradient + bias gradient (column sums of the dZ tiles it stages anyway)
        check(lib.fx_conv2d_wgrad_bias_nhwc_bf16(x2.data_ptr(), Kp, dz.data_ptr(), Np, wt.data_ptr(), bt.data_ptr() if bt is not None else None,
                                                 1, 1, R, Kp, 1, R, Np, 1, 1, 1, 0, st), "fx_conv2d_wgrad_bias_nhwc_bf16")
        if not same:
            if direct:
                ctx.wparam.grad[r0:r1] += wt[:N, :K].reshape(wshape_rows)
            else:
                dw[r0:r1] = wt[:N, :K].reshape(wshape_rows)
        if want_b and Np != N:
            if bdirect:
                ctx.bparam.grad[r0:r1] += bt[:N]
            else:
                db[r0:r1] = bt[:N]
        if not want_w:
            dw = None
        return dw, db

    @staticmethod
    def backward(ctx, dy):
        lib, pack, (r0, r1), act = ctx.lib, ctx.pack, ctx.rng, ctx.act
        x2, saved = ctx.saved_tensors
        dev = x2.device
        N, K, Np, Kp = r1 - r0, ctx.K, pack.Np, pack.Kp
        R = x2.shape[2]
        st = _stream(dev)
        dy2 = _pad_last(dy.reshape(1, 1, R, N), Np)
        if act == "relu":
            dz = torch.empty_like(saved)
            check(lib.fx_relu_bwd_bf16(dy2.data_ptr(), Np, None, 0, saved.data_ptr(), Np, dz.data_ptr(), Np, R, Np, 1, st), "fx_relu_bwd_bf16")
        elif act is None:
            dz = dy2
        else:
            dz = _act_bwd(lib, dy2, saved, act)
        dx = None
        if ctx.needs_input_grad[0]:
            dxp = _conv_call(lib, dz, pack.w_t, None, Kp, 1, 1, 1, 0, None, None, w_frag=pack.w_t_frag).reshape(dy.shape[:-1] + (Kp,))
            dx = dxp if Kp == K else dxp[..., :K].contiguous()
        dw = db = None
        want_w, want_b = ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        if want_w or want_b:
            direct = DIRECT_GRAD[0] and ctx.wparam.grad is not None  # kernels accumulate into the pre-zeroed flat gradients
            bdirect = want_b and DIRECT_GRAD[0] and ctx.bparam.grad is not None
            side = _wgrad_fork(dev, x2, dz) if (direct and (bdirect or not want_b)) else None   # nothing handed back to autograd: off the critical path
            if side is not None:
                with torch.cuda.stream(side):
                    _LinearFn._param_grads(ctx, x2, dz, want_w, want_b, direct, bdirect, C.c_void_p(side.cuda_stream))
            else:
                dw, db = _LinearFn._param_grads(ctx, x2, dz, want_w, want_b, direct, bdirect, st)
        dres = None
        if ctx.has_res:
            dres = (dz if Np == N else dz[..., :N].contiguous()).reshape(dy.shape)
        return dx, dw, db, dres, None, None, None, None, None


class _LinearGroupFn(torch.autograd.Function):
    """y[..., g*N:(g+1)*N] = x @ W_g^T + b_g for G equally shaped Linear layers reading the same input: one forward GEMM with G*N
    outputs, one input-gradient GEMM over all G*N channels (the sum over the layers happens inside the reduction), G weight-gradient
    launches on column slices of dy.  args: x [..., K] bf16, then G weights, then G biases."""

    @staticmethod
    def forward(ctx, x, group: _PackedLinearGroup, lib, *wb):
        G = len(wb) // 2
        weights, biases = wb[:G], wb[G:]
        group.sync(lib, weights, biases)
        N, K = weights[0].shape
        x2 = x.reshape(1, 1, -1, K).contiguous()
        y = _conv_call(lib, x2, group.w_fwd, group.bias, G * N, 1, 1, 1, 0, None, None, w_frag=group.w_fwd_frag)
        ctx.group, ctx.lib, ctx.G, ctx.N, ctx.K = group, lib, G, N, K
        ctx.params = (weights, biases)
        ctx.save_for_backward(x2)
        return y.reshape(*x.shape[:-1], G * N)

    @staticmethod
    def backward(ctx, dy):
        (x2,) = ctx.saved_tensors
        group, lib, G, N, K = ctx.group, ctx.lib, ctx.G, ctx.N, ctx.K
        weights, biases = ctx.params
        dev = x2.device
        R = x2.shape[2]
        dz = dy.reshape(1, 1, R, G * N).contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _conv_call(lib, dz, group.w_t, None, K, 1, 1, 1, 0, None, None, w_frag=group.w_t_frag).reshape(dy.shape[:-1] + (K,))
        direct = DIRECT_GRAD[0] and all(w.grad is not None for w in weights) and all(b.grad is not None for b in biases)
        side = _wgrad_fork(dev, x2, dz) if direct else None
        st = C.c_void_p(side.cuda_stream) if side is not None else _stream(dev)
        dws, dbs = [], []
        for g in range(G):
            wt = weights[g].grad if direct else ARENA.zeros((N, K), dev)
            bt = biases[g].grad if direct else ARENA.zeros((N,), dev)
            check(lib.fx_conv2d_wgrad_bias_nhwc_bf16(x2.data_ptr(), K, dz.data_ptr() + g * N * 2, G * N, wt.data_ptr(), bt.data_ptr(),
                                                     1, 1, R, K, 1, R, N, 1, 1, 1, 0, st), "fx_conv2d_wgrad_bias_nhwc_bf16")
            dws.append(None if direct else wt)
            dbs.append(None if direct else bt)
        return (dx, None, None) + tuple(dws) + tuple(dbs)


class Linear(nn.Module):
    """nn.Linear with the reference's parameter names; forward/backward on the MFMA conv + wgrad kernels."""

    def __init__(self, lib, cin, cout, act: Optional[str] = None, bias: bool = True):
        super().__init__()
        self.lib, self.act = lib, act
        self.weight = nn.Parameter(torch.empty(cout, cin))
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None
        self._pack = _PackedLinear()

    def forward(self, x, residual=None, act="__default__"):
        a = self.act if act == "__default__" else act
        return _LinearFn.apply(x, self.weight, self.bias, residual, self._pack, self.lib, 0, self.weight.shape[0], a)


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, lib):
        R, c = _rows(x), x.shape[-1]     # c = 256, or 128 (the narrow pixel-decoder encoders of fai-mf-{m,s}-coco-ins)
        y = torch.empty_like(x)
        check(lib.fx_layernorm_bf16(x.data_ptr(), c, None, 0, gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), c, R, c, _stream(x.device)),
              "fx_layernorm_bf16")
        ctx.lib = lib
        ctx.gparam, ctx.bparam = gamma, beta
        ctx.save_for_backward(x, gamma)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma = ctx.saved_tensors
        lib = ctx.lib
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        direct = DIRECT_GRAD[0] and ctx.gparam.grad is not None and ctx.bparam.grad is not None
        c = x.shape[-1]
        dg = ctx.gparam.grad if direct else ARENA.zeros((c,), x.device)
        db = ctx.bparam.grad if direct else ARENA.zeros((c,), x.device)
        check(lib.fx_layernorm_bwd_bf16(dy.data_ptr(), c, x.data_ptr(), c, gamma.data_ptr(), dx.data_ptr(), c, dg.data_ptr(), db.data_ptr(),
                                        _rows(x), c, _stream(x.device)), "fx_layernorm_bwd_bf16")
        return dx, (None if direct else dg), (None if direct else db), None


class LayerNorm(nn.Module):
    def __init__(self, lib, c=256):
        super().__init__()
        assert c in (128, 256)
        self.lib = lib
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))

    def forward(self, x):
        return _LayerNormFn.apply(x.contiguous(), self.weight, self.bias, self.lib)


class _MHACoreFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(32) [masked]) v per head on projected [B, L, 256] bf16 tensors: fx_mha_masked_bf16 forward, the MFMA
    backward fx_mha_masked_bwd_bf16 (attn_bwd.hip).  ``mask_bits``: int32 [B*Lq, ceil(Lk/32)] bitmap (bit set = key not allowed; rows
    that forbid every key attend everywhere), or None."""

    @staticmethod
    def forward(ctx, q, k, v, lib, mask_bits=None):
        B, Lq, Cc = q.shape
        Lk = k.shape[1]
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        o = torch.empty_like(q)
        words = mask_bits.shape[1] if mask_bits is not None else 0
        ws = None
        nws = lib.fx_mha_workspace_bytes(B, Lq, Lk, Cc // 32, int(mask_bits is not None))
        if nws > 0:   # few queries x many keys: key-sliced forward (flash-decoding) needs a scratch buffer
            ws = torch.empty(nws, dtype=torch.uint8, device=q.device)
        check(lib.fx_mha_masked_bf16(q.data_ptr(), Cc, k.data_ptr(), Cc, v.data_ptr(), Cc, o.data_ptr(), Cc, B, Lq, Lk, Cc // 32,
                                     mask_bits.data_ptr() if mask_bits is not None else None, words, ws.data_ptr() if ws is not None else None, nws,
                                     _stream(q.device)), "fx_mha_masked_bf16")
        ctx.lib, ctx.mask_bits = lib, mask_bits
        ctx.save_for_backward(q, k, v)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v = ctx.saved_tensors
        lib, bits = ctx.lib, ctx.mask_bits
        B, Lq, Cc = q.shape
        Lk, H = k.shape[1], Cc // 32
        do = do.contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        nb = lib.fx_mha_bwd_workspace_bytes(B, Lq, Lk, H)
        ws = torch.empty(nb, dtype=torch.uint8, device=q.device)
        check(lib.fx_mha_masked_bwd_bf16(q.data_ptr(), Cc, k.data_ptr(), Cc, v.data_ptr(), Cc, do.data_ptr(), Cc, dq.data_ptr(), Cc, dk.data_ptr(), Cc,
                                         dv.data_ptr(), Cc, B, Lq, Lk, H, bits.data_ptr() if bits is not None else None,
                                         bits.shape[1] if bits is not None else 0, ws.data_ptr(), nb, _stream(q.device)), "fx_mha_masked_bwd_bf16")
        return dq, dk, dv, None, None


class MultiheadAttention(nn.Module):
    """nn.MultiheadAttention(c, 8, batch_first=True), c = 256 (or 128, see forward) with the reference's parameter names (in_proj_weight,
    in_proj_bias, out_proj.weight/bias); q = k inputs share one projection GEMM."""

    def __init__(self, lib, c=256):
        super().__init__()
        self.lib, self.c = lib, c
        self.in_proj_weight = nn.Parameter(torch.empty(3 * c, c))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * c))
        self.out_proj = Linear(lib, c, c)
        self._pq, self._pk, self._pv, self._pqk = _PackedLinear(), _PackedLinear(), _PackedLinear(), _PackedLinear()

    def forward(self, q_in, k_in, v_in, residual=None, mask_bits=None):
        c, lib, W, b = self.c, self.lib, self.in_proj_weight, self.in_proj_bias
        if q_in is k_in:
            qk = _LinearFn.apply(q_in, W, b, None, self._pqk, lib, 0, 2 * c, None)
            q, k = qk[..., :c], qk[..., c:]
        else:
            q = _LinearFn.apply(q_in, W, b, None, self._pq, lib, 0, c, None)
            k = _LinearFn.apply(k_in, W, b, None, self._pk, lib, c, 2 * c, None)
        v = _LinearFn.apply(v_in, W, b, None, self._pv, lib, 2 * c, 3 * c, None)
        if c == 128:
            # 8 heads of 16 channels on the head-dim-32 attention kernels: every head zero-padded to 32 channels (scores and outputs
            # unchanged), sqrt(2) on q because the kernels scale by 1/sqrt(32) where the reference scales by 1/sqrt(16).  Glue on
            # [B, L, 256] tensors at the stride-32 level (<= 1024 tokens per image).
            def pad(t, scale=None):
                t = t.reshape(*t.shape[:-1], 8, 16)
                if scale is not None:
                    t = t * scale
                return torch.nn.functional.pad(t, (0, 16)).reshape(*t.shape[:-2], 256)

            o = _MHACoreFn.apply(pad(q, math.sqrt(2.0)), pad(k), pad(v), lib, mask_bits)
            o = o.reshape(*o.shape[:-1], 8, 32)[..., :16].reshape(*o.shape[:-1], 128)
        else:
            o = _MHACoreFn.apply(q, k, v, lib, mask_bits)
        return self.out_proj(o, residual=residual)


class _ResizeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Ho, Wo, lib):
        B, H, W_, Cc = x.shape
        x = x.contiguous()
        y = torch.empty(B, Ho, Wo, Cc, dtype=_lib.act_dtype(), device=x.device)
        check(lib.fx_resize_bilinear_nhwc_bf16(x.data_ptr(), Cc, y.data_ptr(), Cc, B, H, W_, Cc, Ho, Wo, _stream(x.device)), "fx_resize_bilinear_nhwc_bf16")
        ctx.lib, ctx.shape = lib, (B, H, W_, Cc)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = ctx.lib
        B, H, W_, Cc = ctx.shape
        dy = dy.contiguous()
        dx = torch.empty(B, H, W_, Cc, dtype=_lib.act_dtype(), device=dy.device)
        check(lib.fx_resize_bilinear_bwd_nhwc_bf16(dy.data_ptr(), Cc, dx.data_ptr(), Cc, B, H, W_, Cc, dy.shape[1], dy.shape[2], _stream(dy.device)),
              "fx_resize_bilinear_bwd_nhwc_bf16")
        return dx, None, None, None


class _AddFn(torch.autograd.Function):
    """x + y (same shape, or y broadcast over the leading rows: positional embeddings) through fx_add_rows_bf16."""

    @staticmethod
    def forward(ctx, x, y, lib):
        x, y = x.contiguous(), y.contiguous()
        Cc = x.shape[-1]
        out = torch.empty_like(x)
        check(lib.fx_add_rows_bf16(x.data_ptr(), Cc, y.data_ptr(), Cc, _rows(y), out.data_ptr(), Cc, _rows(x), Cc, _stream(x.device)), "fx_add_rows_bf16")
        ctx.same = y.shape == x.shape
        ctx.yshape = tuple(y.shape)
        return out

    @staticmethod
    def backward(ctx, d):
        if ctx.same:
            return d, d, None
        dy = None
        if ctx.needs_input_grad[1]:   # a learned embedding broadcast over the batch (query_embed): sum over the repeats ([B, Q, 256]: glue)
            dy = d.reshape(-1, *ctx.yshape).float().sum(0).to(d.dtype)
        return d, dy, None


# ================================================================================================ hybrid encoder (RT-DETR)
def _pos_embed_sine(h: int, w: int, npf: int, temperature: float = 10000.0) -> torch.Tensor:
    """AIFI sine position embedding [h*w, 2*npf] = [y_sin | y_cos | x_sin | x_cos] (fai_detr/modelling.py:148-179)."""
    ys = torch.arange(h, dtype=torch.float32).view(h, 1).expand(h, w)
    xs = torch.arange(w, dtype=torch.float32).view(1, w).expand(h, w)
    i = torch.arange(npf, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / npf)
    px, py = xs[..., None] / dim_t, ys[..., None] / dim_t
    return torch.cat([py[..., 0::2].sin(), py[..., 1::2].cos(), px[..., 0::2].sin(), px[..., 1::2].cos()], dim=-1).reshape(h * w, 2 * npf)


class TransformerEncoderLayer(nn.Module):
    """Post-norm encoder layer with GELU FFN (focoos/nn/layers/transformer.py:553-601)."""

    def __init__(self, lib, c=256, ffn=1024):
        super().__init__()
        self.lib = lib
        self.self_attn = MultiheadAttention(lib, c)
        self.linear1 = Linear(lib, c, ffn, act="gelu")
        self.linear2 = Linear(lib, ffn, c)
        self.norm1 = LayerNorm(lib, c)
        self.norm2 = LayerNorm(lib, c)

    def forward(self, src, pos):
        qk = _AddFn.apply(src, pos, self.lib)
        src = self.norm1(self.self_attn(qk, qk, src, residual=src))
        return self.norm2(self.linear2(self.linear1(src), residual=src))


class RepVggBlock(nn.Module):
    """Unfused training form: silu(conv3x3+bn(x) + conv1x1+bn(x)) (fai_detr/modelling.py:30-45)."""

    def __init__(self, lib, c):
        super().__init__()
        self.conv1 = ConvNormLayer(lib, c, c, 3, 1, None)
        self.conv2 = ConvNormLayer(lib, c, c, 1, 1, "silu")  # its epilogue adds conv1's branch, then SiLU

    def forward(self, x):
        if _siblings_on(self.conv1):     # SyncBN: one forward collective for the two branches
            return sibling_conv_bn(self.conv1, self.conv2, x, x, chain=True)
        return self.conv2(x, residual=self.conv1(x))


class CSPRepLayer(nn.Module):
    """fai_detr/modelling.py:84-107 (expansion 1.0: conv3 = Identity)."""

    def __init__(self, lib, cin, cout, n=3):
        super().__init__()
        self.lib = lib
        self.conv1 = ConvNormLayer(lib, cin, cout, 1, 1, "silu")
        self.conv2 = ConvNormLayer(lib, cin, cout, 1, 1, "silu")
        self.bottlenecks = nn.Sequential(*[RepVggBlock(lib, cout) for _ in range(n)])

    def forward(self, x):
        if _siblings_on(self.conv1):     # SyncBN: conv1 | conv2 share their collectives
            c1, c2 = sibling_conv_bn(self.conv1, self.conv2, x, x)
            return _AddFn.apply(self.bottlenecks(c1), c2, self.lib)
        return _AddFn.apply(self.bottlenecks(self.conv1(x)), self.conv2(x), self.lib)


class HybridEncoder(nn.Module):
    """Encoder.forward (fai_detr/modelling.py:297-347) as HIP autograd nodes; parameter names = ``pixel_decoder.*`` minus the
    backbone.  Channel concats use torch.cat (a copy; the inference engine writes channel slices in place instead)."""

    def __init__(self, lib, in_channels=(512, 1024, 2048), c=256, ffn=1024, n_enc=1):
        super().__init__()
        self.lib, self.c = lib, c
        self.input_proj = nn.ModuleList([ConvNormLayer(lib, ci, c, 1, 1, None, names=("0", "1")) for ci in in_channels])
        self.encoder = nn.ModuleList([_Layers([TransformerEncoderLayer(lib, c, ffn) for _ in range(n_enc)])])
        self.lateral_convs = nn.ModuleList([ConvNormLayer(lib, c, c, 1, 1, "silu") for _ in range(2)])
        self.fpn_blocks = nn.ModuleList([CSPRepLayer(lib, 2 * c, c) for _ in range(2)])
        self.downsample_convs = nn.ModuleList([ConvNormLayer(lib, c, c, 3, 1, "silu") for _ in range(2)])
        self.pan_blocks = nn.ModuleList([CSPRepLayer(lib, 2 * c, c) for _ in range(2)])
        self.mask_features = _Holder()  # dead in RT-DETR's forward (fai_detr/modelling.py:347); kept for checkpoint compatibility
        self.mask_features.weight = nn.Parameter(torch.zeros(c, c, 3, 3), requires_grad=False)
        self.mask_features.bias = nn.Parameter(torch.zeros(c), requires_grad=False)
        self._pos = {}

    def _pos_for(self, h, w, dev):
        key = (h, w, dev)
        if key not in self._pos:
            self._pos[key] = _pos_embed_sine(h, w, self.c // 2).to(device=dev, dtype=_lib.act_dtype()).contiguous()
        return self._pos[key]

    def forward(self, feats: List[torch.Tensor]) -> List[torch.Tensor]:
        lib = self.lib
        proj = [p(f) for p, f in zip(self.input_proj, feats)]
        B, h, w, c = proj[2].shape
        src = proj[2].reshape(B, h * w, c)
        pos = self._pos_for(h, w, src.device)
        for layer in self.encoder[0].layers:
            src = layer(src, pos)
        proj[2] = src.reshape(B, h, w, c)
        inner = [proj[2]]
        for idx in (2, 1):
            high = self.lateral_convs[2 - idx](inner[0])
            inner[0] = high
            low = proj[idx - 1]
            up = _ResizeFn.apply(high, low.shape[1], low.shape[2], lib)
            inner.insert(0, self.fpn_blocks[2 - idx](torch.cat([up, low], dim=-1)))
        outs = [inner[0]]
        for idx in range(2):
            nxt = inner[idx + 1]
            down = self.downsample_convs[idx](_ResizeFn.apply(outs[-1], nxt.shape[1], nxt.shape[2], lib))
            outs.append(self.pan_blocks[idx](torch.cat([down, nxt], dim=-1)))
        return outs[::-1]  # [stride 32, 16, 8]


class _Layers(nn.Module):
    def __init__(self, layers):
        super().__init__()
        self.layers = nn.ModuleList(layers)
