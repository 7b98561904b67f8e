"""BiSeNetFormer as a trainable HIP autograd graph (BASELINE config 5): what
``BisenetFormer.forward(images, targets)`` computes under ``model.train()`` (focoos/models/bisenetformer/modelling.py:594-609) -
STDC backbone (focoos/nn/backbone/stdc.py:108-166, 282-320), ContextPath / AttentionRefinementModule / FeatureFusionModule
(modelling.py:149-237), the masked-attention ``TransformerDecoder`` with its 7 supervised prediction heads (:375-447, :68-113) and the
point-sampled Hungarian ``SetCriterion`` (bisenetformer/loss.py == fai_mf/loss.py:345-607, 626-723) - with parameter names equal to
the reference's, so checkpoints travel both ways.

Every conv / GEMM / attention / normalisation / pooling / gate / resize / einsum / loss, forward and backward, is a libfocoos_amd.so
kernel (train_nn.py nodes + the ones below).  PyTorch supplies the tape and *glue* on small tensors, listed so nothing hides:
``torch.cat`` of the four CatBottleneck branches, the 1x1 convolutions / BatchNorm / sigmoid on pooled ``[B, C]`` vectors (ARM and FFM
gates, conv_avg: a few KFLOP), the bf16 casts / batch repeat of the two query embeddings, the per-image transposed copy of the mask
embedding (``[128, 128]``) in the einsum backward, and the fp32 view of the ``[B, Q, K+1]`` class logits handed to the criterion.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import torch
from torch import nn
import torch.nn.functional as F

from . import _lib
from ._lib import check
from .engine_maskdec import pack_mask_bits, pos_embed_sine_normalized
from .mask_criterion import MaskHungarianMatcher, SetCriterion
from .train_detr import MLP
from .train_nn import (ARENA, BN_EPS, BN_MOMENTUM, DIRECT_GRAD, WEIGHTS_EPOCH, ConvNormLayer, LayerNorm, Linear, MultiheadAttention, StemConv, _AddFn,
                       _bn_backward, _bn_forward, _bn_sync_group, _conv_call, _Holder, _LinearFn, _PackedLinear, _ResizeFn, _rup, _stream,
                       set_norm_mode)


# ================================================================================================ STDC pieces
class _DwConvFn(torch.autograd.Function):
    """Depthwise 3x3 stride-2 conv + BatchNorm (CatBottleneck.avd_layer, stdc.py:120-127): fx_dwconv3x3s2_nhwc_bf16 with the frozen
    BatchNorm folded into the taps, or the plain conv followed by the batch-statistics passes."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, layer: "DwConvBn"):
        lib, dev = layer.lib, x.device
        B, H, W_, Cc = x.shape
        x = x.contiguous()
        Ho, Wo = (H - 1) // 2 + 1, (W_ - 1) // 2 + 1
        live = layer.batch_stats
        w9, shift = layer.taps(live)
        z = torch.empty(B, Ho, Wo, Cc, dtype=torch.float32 if live else _lib.act_dtype(), device=dev)
        fn = lib.fx_dwconv3x3s2_nhwc_f32out if live else lib.fx_dwconv3x3s2_nhwc_bf16   # fp32 in front of a batch-statistics BatchNorm
        check(fn(x.data_ptr(), Cc, w9.data_ptr(), None if live else shift.data_ptr(), z.data_ptr(), Cc, B, H, W_, Cc, _stream(dev)), fn.__name__)
        ctx.layer, ctx.live = layer, live
        if live:
            y, stats, n = _bn_forward(layer, z, None)
            ctx.n = n
            ctx.save_for_backward(x, w9, z, stats)
            return y
        ctx.save_for_backward(x, w9)
        return z

    @staticmethod
    def backward(ctx, dy):
        layer: DwConvBn = ctx.layer
        lib = layer.lib
        dy = dy.contiguous()
        dgamma = dbeta = None
        if ctx.live:
            x, w9, z, stats = ctx.saved_tensors
            dz, _, dgamma, dbeta = _bn_backward(layer, dy, z, None, stats, ctx.n, ctx.needs_input_grad[2] or ctx.needs_input_grad[3])
        else:
            x, w9 = ctx.saved_tensors
            dz = dy
        B, H, W_, Cc = x.shape
        dev = x.device
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw9 = ARENA.zeros((9, Cc), dev) if ctx.needs_input_grad[1] else None
        check(lib.fx_dwconv3x3s2_bwd_nhwc_bf16(dz.data_ptr(), Cc, x.data_ptr(), Cc, w9.data_ptr(), dx.data_ptr() if dx is not None else None, Cc,
                                               dw9.data_ptr() if dw9 is not None else None, B, H, W_, Cc, _stream(dev)), "fx_dwconv3x3s2_bwd_nhwc_bf16")
        dw = None
        if dw9 is not None:
            if not ctx.live:
                dw9 = dw9 * layer.scale            # folded BatchNorm: w_eff = w * scale
            dw = dw9.t().reshape(Cc, 1, 3, 3)
        return dx, dw, dgamma, dbeta, None


class DwConvBn(nn.Module):
    """nn.Sequential(Conv2d(C, C, 3, 2, 1, groups=C, bias=False), BatchNorm2d(C)) with keys ``0.weight`` / ``1.*``."""

    has_batchnorm = True
    norm_mode = "FrozenBN"
    act = None

    def __init__(self, lib, c: int):
        super().__init__()
        self.lib, self.c = lib, c
        conv, norm = _Holder(), _Holder()
        self.add_module("0", conv)
        self.add_module("1", norm)
        object.__setattr__(self, "_conv_h", conv)
        object.__setattr__(self, "_norm_h", norm)
        conv.weight = nn.Parameter(torch.empty(c, 1, 3, 3))
        norm.weight = nn.Parameter(torch.ones(c), requires_grad=False)
        norm.bias = nn.Parameter(torch.zeros(c), requires_grad=False)
        norm.register_buffer("running_mean", torch.zeros(c))
        norm.register_buffer("running_var", torch.ones(c))
        norm.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self._ver = None
        self.scale = None

    @property
    def batch_stats(self) -> bool:
        return self.training and self.norm_mode != "FrozenBN"

    def taps(self, live: bool):
        w, n = self._conv_h.weight, self._norm_h
        ver = (w._version, n.weight._version, n.bias._version, n.running_var._version, n.running_mean._version, w.device, WEIGHTS_EPOCH[0], live,
               getattr(self, "_stats_epoch", 0))
        if ver != self._ver:
            with torch.no_grad():
                w9 = w[:, 0].permute(1, 2, 0).reshape(9, self.c)
                if live:
                    self._w9, self._shift = w9.contiguous(), None
                else:
                    self.scale = (n.weight.double() / torch.sqrt(n.running_var.double() + BN_EPS)).float()
                    self._shift = (n.bias.double() - n.running_mean.double() * self.scale.double()).float().contiguous()
                    self._w9 = (w9 * self.scale).contiguous()
            self._ver = ver
        return self._w9, self._shift

    def forward(self, x):
        return _DwConvFn.apply(x, self._conv_h.weight, self._norm_h.weight, self._norm_h.bias, self)


class _AvgPool3x3s2Fn(torch.autograd.Function):
    """AvgPool2d(3, 2, 1) (count_include_pad=True): the depthwise kernel with w = 1/9, backward = its transpose."""

    @staticmethod
    def forward(ctx, x, lib):
        B, H, W_, Cc = x.shape
        x = x.contiguous()
        w = _const_ninth(Cc, x.device)
        y = torch.empty(B, (H - 1) // 2 + 1, (W_ - 1) // 2 + 1, Cc, dtype=_lib.act_dtype(), device=x.device)
        check(lib.fx_dwconv3x3s2_nhwc_bf16(x.data_ptr(), Cc, w.data_ptr(), None, y.data_ptr(), Cc, B, H, W_, Cc, _stream(x.device)), "fx_dwconv3x3s2_nhwc_bf16")
        ctx.lib, ctx.shape = lib, tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, H, W_, Cc = ctx.shape
        dy = dy.contiguous()
        dx = torch.empty(B, H, W_, Cc, dtype=_lib.act_dtype(), device=dy.device)
        w = _const_ninth(Cc, dy.device)
        check(ctx.lib.fx_dwconv3x3s2_bwd_nhwc_bf16(dy.data_ptr(), Cc, None, 0, w.data_ptr(), dx.data_ptr(), Cc, None, B, H, W_, Cc, _stream(dy.device)),
              "fx_dwconv3x3s2_bwd_nhwc_bf16")
        return dx, None


_NINTH: Dict[tuple, torch.Tensor] = {}


def _const_ninth(c: int, dev) -> torch.Tensor:
    key = (c, dev)
    if key not in _NINTH:
        _NINTH[key] = torch.full((9, c), 1.0 / 9.0, dtype=torch.float32, device=dev)
    return _NINTH[key]


def _convx(lib, cin, cout, k, stride=1):
    """ConvX / ConvBNReLU: conv (no bias, padding k//2) + BatchNorm2d + ReLU with keys ``conv.weight`` / ``bn.*`` (stdc.py:15-31)."""
    return ConvNormLayer(lib, cin, cout, k, stride, "relu", names=("conv", "bn"))


class CatBottleneck(nn.Module):
    """stdc.py:108-166 (block_num 4): [skip(out1) | conv1 | conv2 | conv3] concatenated."""

    def __init__(self, lib, cin, cout, stride):
        super().__init__()
        self.lib, self.stride = lib, stride
        self.conv_list = nn.ModuleList([_convx(lib, cin, cout // 2, 1), _convx(lib, cout // 2, cout // 4, 3), _convx(lib, cout // 4, cout // 8, 3),
                                        _convx(lib, cout // 8, cout // 8, 3)])
        if stride == 2:
            self.avd_layer = DwConvBn(lib, cout // 2)

    def forward(self, x):
        out1 = self.conv_list[0](x)
        cur = out1
        if self.stride == 2:
            cur = self.avd_layer(out1)
            out1 = _AvgPool3x3s2Fn.apply(out1, self.lib)
        o1 = self.conv_list[1](cur)
        o2 = self.conv_list[2](o1)
        o3 = self.conv_list[3](o2)
        return torch.cat([out1, o1, o2, o3], dim=-1)


class STDC(nn.Module):
    """STDC.forward (stdc.py:313-320); input uint8 / fp32 HWC images, outputs res2..res5 NHWC bf16."""

    def __init__(self, lib, base=64, layers=(4, 5, 3), pixel_mean=(123.675, 116.28, 103.53), pixel_std=(58.395, 57.12, 57.375)):
        super().__init__()
        assert base == 64, "the stem kernel produces 32 channels (STDC base 64)"
        feats: List[nn.Module] = [StemConv(lib, pixel_mean, pixel_std, names=("conv", "bn")), _convx(lib, base // 2, base, 3, 2)]
        cin = base
        self.stage_ends = []
        for i, n in enumerate(layers):
            cout = base * 2 ** (i + 2)
            for j in range(n):
                feats.append(CatBottleneck(lib, cin, cout, 2 if j == 0 else 1))
                cin = cout
            self.stage_ends.append(len(feats) - 1)
        self.features = nn.ModuleList(feats)

    def forward(self, images):
        x = self.features[0](images)
        x = self.features[1](x)
        outs = {"res2": x}
        si = 0
        for idx in range(2, len(self.features)):
            x = self.features[idx](x)
            if idx == self.stage_ends[si]:
                outs[f"res{si + 3}"] = x
                si += 1
        return outs


# ================================================================================================ BiseNet pixel decoder
class _GlobalMeanFn(torch.autograd.Function):
    """feat.mean(dim=(2,3)) -> f32 [B, C]; backward broadcasts d / (H W)."""

    @staticmethod
    def forward(ctx, x, lib):
        B, H, W_, Cc = x.shape
        x = x.contiguous()
        out = torch.empty(B, Cc, dtype=torch.float32, device=x.device)
        check(lib.fx_global_mean_nhwc_bf16(x.data_ptr(), Cc, out.data_ptr(), Cc, B, H * W_, Cc, _stream(x.device)), "fx_global_mean_nhwc_bf16")
        ctx.lib, ctx.shape = lib, tuple(x.shape)
        return out

    @staticmethod
    def backward(ctx, d):
        B, H, W_, Cc = ctx.shape
        d = d.float().contiguous()
        dx = torch.empty(B, H, W_, Cc, dtype=_lib.act_dtype(), device=d.device)
        check(ctx.lib.fx_bcast_vec_nhwc_bf16(d.data_ptr(), Cc, 1.0 / (H * W_), dx.data_ptr(), Cc, B, H * W_, Cc, _stream(d.device)), "fx_bcast_vec_nhwc_bf16")
        return dx, None


class _GateFn(torch.autograd.Function):
    """y = x * gate[b, c] (+ x) (+ add_vec[b, c]) (+ add_map): fx_channel_gate_nhwc_bf16.  Backward: dx through the same kernel on dy,
    d gate = sum_p dy * x and d add_vec = sum_p dy through fx_rowdot_nhwc_bf16, d add_map = dy."""

    @staticmethod
    def forward(ctx, x, gate, add_vec, add_map, self_add: bool, lib):
        B, H, W_, Cc = x.shape
        x, gate = x.contiguous(), gate.float().contiguous()
        if add_vec is not None:
            add_vec = add_vec.float().contiguous()
        if add_map is not None:
            add_map = add_map.contiguous()
        y = torch.empty_like(x)
        check(lib.fx_channel_gate_nhwc_bf16(x.data_ptr(), Cc, gate.data_ptr(), Cc, int(self_add), add_vec.data_ptr() if add_vec is not None else None, Cc,
                                            add_map.data_ptr() if add_map is not None else None, Cc, y.data_ptr(), Cc, B, H * W_, Cc, _stream(x.device)),
              "fx_channel_gate_nhwc_bf16")
        ctx.lib, ctx.self_add, ctx.has_vec, ctx.has_map = lib, self_add, add_vec is not None, add_map is not None
        ctx.save_for_backward(x, gate)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gate = ctx.saved_tensors
        lib = ctx.lib
        B, H, W_, Cc = x.shape
        P = H * W_
        dy = dy.contiguous()
        st = _stream(x.device)
        dx = torch.empty_like(x)
        check(lib.fx_channel_gate_nhwc_bf16(dy.data_ptr(), Cc, gate.data_ptr(), Cc, int(ctx.self_add), None, Cc, None, Cc, dx.data_ptr(), Cc, B, P, Cc, st),
              "fx_channel_gate_nhwc_bf16")
        splits = max(1, min(64, P // 2048))
        dg = ARENA.zeros((B, Cc), x.device)
        check(lib.fx_rowdot_nhwc_bf16(dy.data_ptr(), Cc, x.data_ptr(), Cc, 1.0, dg.data_ptr(), Cc, B, P, Cc, splits, st), "fx_rowdot_nhwc_bf16")
        dv = None
        if ctx.has_vec:
            dv = ARENA.zeros((B, Cc), x.device)
            check(lib.fx_rowdot_nhwc_bf16(dy.data_ptr(), Cc, None, 0, 1.0, dv.data_ptr(), Cc, B, P, Cc, splits, st), "fx_rowdot_nhwc_bf16")
        return dx, dg, dv, (dy if ctx.has_map else None), None, None


class VecBN(nn.Module):
    """BatchNorm2d applied to a pooled [B, C, 1, 1] tensor (bn_atten, conv_avg.bn) on its [B, C] fp32 vector form - torch glue on a few
    hundred values.  Follows the model-wide norm mode: running statistics ("FrozenBN" / eval), batch statistics over the B samples
    ("BN"), or batch statistics summed over the data-parallel group ("SyncBN")."""

    has_batchnorm = True
    norm_mode = "FrozenBN"

    def __init__(self, c: int):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(c), requires_grad=False)
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        object.__setattr__(self, "_norm_h", self)

    def forward(self, v: torch.Tensor) -> torch.Tensor:
        if not (self.training and self.norm_mode != "FrozenBN"):
            return F.batch_norm(v, self.running_mean, self.running_var, self.weight, self.bias, False, 0.0, BN_EPS)
        world = _bn_sync_group(self)
        if world == 1:
            return F.batch_norm(v, self.running_mean, self.running_var, self.weight, self.bias, True, BN_MOMENTUM, BN_EPS)
        import torch.distributed as dist
        import torch.distributed.nn.functional as distF

        n = float(v.shape[0] * world)
        sums = distF.all_reduce(torch.stack([v.sum(0), (v * v).sum(0)]))   # differentiable all-reduce (SyncBatchNorm semantics)
        mean = sums[0] / n
        var = sums[1] / n - mean * mean
        with torch.no_grad():
            self.running_mean.mul_(1 - BN_MOMENTUM).add_(mean, alpha=BN_MOMENTUM)
            self.running_var.mul_(1 - BN_MOMENTUM).add_(var * (n / max(n - 1.0, 1.0)), alpha=BN_MOMENTUM)
            self.num_batches_tracked += 1
        return (v - mean) * torch.rsqrt(var + BN_EPS) * self.weight + self.bias


class Conv1x1(nn.Module):
    """Conv2d(cin, cout, 1) without normalisation on an NHWC map, parameters in the reference's [N, C, 1, 1] / [N] shapes; the GEMM and
    its gradients run on the MFMA conv / wgrad kernels (train_nn._LinearFn)."""

    def __init__(self, lib, cin, cout, bias: bool):
        super().__init__()
        self.lib = lib
        self.weight = nn.Parameter(torch.empty(cout, cin, 1, 1))
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None
        self._pack = _PackedLinear()

    def forward(self, x, residual=None):
        return _LinearFn.apply(x, self.weight, self.bias, residual, self._pack, self.lib, 0, self.weight.shape[0], None)


class _PooledConv(nn.Module):
    """A bias-free 1x1 Conv2d applied to a pooled vector (conv_atten, conv_avg.conv, ffm.conv1 / conv2): ``weight`` [N, C, 1, 1]."""

    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, 1, 1))

    def forward(self, v):
        return F.linear(v, self.weight.flatten(1))


class AttentionRefinementModule(nn.Module):
    """modelling.py:149-167: returns (feat, sigmoid attention [B, fd]); the product is fused with the following sum by the caller."""

    def __init__(self, lib, cin, fd):
        super().__init__()
        self.lib = lib
        self.proj = Conv1x1(lib, cin, fd, bias=False)
        self.conv = _convx(lib, fd, fd, 3)
        self.conv_atten = _PooledConv(fd, fd)
        self.bn_atten = VecBN(fd)

    def forward(self, x):
        feat = self.conv(self.proj(x))
        att = torch.sigmoid(self.bn_atten(self.conv_atten(_GlobalMeanFn.apply(feat, self.lib))))
        return feat, att


class _ConvAvg(nn.Module):
    """ConvBNReLU(c5, fd, 1) on the global mean of feat32 (modelling.py:187-188), keys ``conv.weight`` / ``bn.*``."""

    def __init__(self, cin, fd):
        super().__init__()
        self.conv = _PooledConv(cin, fd)
        self.bn = VecBN(fd)

    def forward(self, v):
        return F.relu(self.bn(self.conv(v)))


class ContextPath(nn.Module):
    """modelling.py:170-212 (out4=False).  Returns (cp8, cp16, cp32)."""

    def __init__(self, lib, chans, fd):
        super().__init__()
        self.lib = lib
        self.arm32 = AttentionRefinementModule(lib, chans[3], fd)
        self.conv_avg = _ConvAvg(chans[3], fd)
        self.conv_head32 = _convx(lib, fd, fd, 3)
        self.arm16 = AttentionRefinementModule(lib, chans[2], fd)
        self.conv_head16 = _convx(lib, fd, fd, 3)

    def forward(self, feat8, feat16, feat32):
        lib = self.lib
        avg = self.conv_avg(_GlobalMeanFn.apply(feat32, lib))
        f32, a32 = self.arm32(feat32)
        cp32 = _GateFn.apply(f32, a32, avg, None, False, lib)
        up32 = self.conv_head32(_ResizeFn.apply(cp32, feat16.shape[1], feat16.shape[2], lib))
        f16, a16 = self.arm16(feat16)
        cp16 = _GateFn.apply(f16, a16, None, up32, False, lib)
        cp8 = self.conv_head16(_ResizeFn.apply(cp16, feat8.shape[1], feat8.shape[2], lib))
        return cp8, cp16, cp32


class FeatureFusionModule(nn.Module):
    """modelling.py:213-237."""

    def __init__(self, lib, c_sp, fd):
        super().__init__()
        self.lib = lib
        self.proj1 = Conv1x1(lib, c_sp, fd, bias=True)
        self.proj2 = Conv1x1(lib, fd, fd, bias=True)
        self.convblk = _convx(lib, fd, fd, 1)
        self.conv1 = _PooledConv(fd, fd // 4)
        self.conv2 = _PooledConv(fd // 4, fd)

    def forward(self, fsp, fcp):
        feat = self.convblk(self.proj2(fcp, residual=self.proj1(fsp)))
        att = torch.sigmoid(self.conv2(F.relu(self.conv1(_GlobalMeanFn.apply(feat, self.lib)))))
        return _GateFn.apply(feat, att, None, None, True, self.lib)


class BiseNet(nn.Module):
    """modelling.py:238-282: backbone + ContextPath + FFM + conv_out.  Returns (mask_features, (cp32, cp16, cp8))."""

    def __init__(self, lib, config: Dict):
        super().__init__()
        bb = config["backbone_config"]
        base = int(bb.get("base", 64))
        fd = int(config.get("pixel_decoder_feat_dim", 128))
        od = int(config.get("pixel_decoder_out_dim", 128))
        self.backbone = STDC(lib, base, tuple(bb.get("layers", (4, 5, 3))), config.get("pixel_mean", (123.675, 116.28, 103.53)),
                             config.get("pixel_std", (58.395, 57.12, 57.375)))
        chans = [base, base * 4, base * 8, base * 16]
        self.cp = ContextPath(lib, chans, fd)
        self.ffm = FeatureFusionModule(lib, chans[1], fd)
        self.conv_out = _convx(lib, fd, od, 3)

    def decode(self, f: Dict[str, torch.Tensor]):
        cp8, cp16, cp32 = self.cp(f["res3"], f["res4"], f["res5"])
        return self.conv_out(self.ffm(f["res3"], cp8)), (cp32, cp16, cp8)

    def forward(self, images):
        return self.decode(self.backbone(images))


# ================================================================================================ masked-attention decoder
def _einsum_operands(emb, feat):
    """The mask einsum kernel takes 128 or 256 channels; a narrower mask dimension (bisenetformer-m-ade: 96) runs on zero-padded copies
    of its two operands (glue: one pad of the mask embeddings and of the stride-8 mask features per prediction head)."""
    Cc = emb.shape[-1]
    if Cc in (128, 256):
        return emb.contiguous(), feat.contiguous(), Cc
    if Cc > 128:
        raise _lib.FocoosAmdError(f"mask dimension {Cc}: the einsum kernel covers up to 128 (zero-padded) or exactly 256 channels")
    return F.pad(emb, (0, 128 - Cc)).contiguous(), F.pad(feat, (0, 128 - Cc)).contiguous(), 128


class _MaskEinsumFn(torch.autograd.Function):
    """masks[b, q, p] = sum_c embed[b, q, c] * feat[b, p, c] (einsum "bqc,bchw->bqhw", modelling.py:84) -> f32 [B, Q, h, w]:
    fx_query_pixel_logits_bf16 mode 0.  Backward: the f32 gradient planes (the criterion scatters into them) are transposed to
    pixel-major bf16 rows once (fx_planes_to_rows_bf16); per image d feat = rows x embed on the 1x1 conv kernel and
    d embed = rows^T x feat on the weight-gradient kernel (the embeddings are per-image weights)."""

    @staticmethod
    def forward(ctx, emb, feat, lib):
        B, Q, Cc = emb.shape
        _, h, w, _ = feat.shape
        emb, feat = emb.contiguous(), feat.contiguous()
        out = torch.empty(B, Q, h, w, dtype=torch.float32, device=emb.device)
        ep, fp, cp = _einsum_operands(emb, feat)
        check(lib.fx_query_pixel_logits_bf16(ep.data_ptr(), cp, fp.data_ptr(), cp, 0, out.data_ptr(), h * w, None, 0, B, Q, h * w, cp, _stream(emb.device)),
              "fx_query_pixel_logits_bf16")
        ctx.lib = lib
        ctx.save_for_backward(emb, feat)
        return out

    @staticmethod
    def backward(ctx, dm):
        emb, feat = ctx.saved_tensors
        lib, dev = ctx.lib, emb.device
        B, Q, Cc = emb.shape
        _, h, w, _ = feat.shape
        P = h * w
        Qp = _rup(Q, 32)
        st = _stream(dev)
        dm = dm.float().contiguous()
        rows = torch.empty(B, 1, P, Qp, dtype=_lib.act_dtype(), device=dev)
        check(lib.fx_planes_to_rows_bf16(dm.data_ptr(), Q, P, rows.data_ptr(), Qp, Qp, B, st), "fx_planes_to_rows_bf16")
        dfeat = demb = None
        if ctx.needs_input_grad[1]:
            et = torch.zeros(B, _rup(Cc, 128), 1, 1, Qp, dtype=_lib.act_dtype(), device=dev)   # per-image [N = C][K = Qp] weight images
            et[:, :Cc, 0, 0, :Q] = emb.transpose(1, 2)
            dfeat = torch.cat([_conv_call(lib, rows[b:b + 1], et[b], None, Cc, 1, 1, 1, 0, None, None) for b in range(B)], 0).view(B, h, w, Cc)
        if ctx.needs_input_grad[0]:
            dw = ARENA.zeros((B, Qp, Cc), dev)
            for b in range(B):
                check(lib.fx_conv2d_wgrad_bias_nhwc_bf16(feat[b].data_ptr(), Cc, rows[b].data_ptr(), Qp, dw[b].data_ptr(), None, 1, 1, P, Cc, 1, P, Qp, 1, 1,
                                                         1, 0, st), "fx_conv2d_wgrad_bias_nhwc_bf16")
            demb = dw[:, :Q].to(_lib.act_dtype())
        return demb, dfeat, None


class SelfAttentionLayer(nn.Module):
    """focoos/nn/layers/transformer.py:17-106 (pre-norm): tgt + self_attn(norm(tgt) + qpos, ., norm(tgt))."""

    def __init__(self, lib, c=256):
        super().__init__()
        self.lib = lib
        self.self_attn = MultiheadAttention(lib, c)
        self.norm = LayerNorm(lib, c)

    def forward(self, tgt, qe):
        t2 = self.norm(tgt)
        qk = _AddFn.apply(t2, qe, self.lib)
        return self.self_attn(qk, qk, t2, residual=tgt)


class CrossAttentionLayer(nn.Module):
    """transformer.py:109-238 (pre-norm) with the boolean memory mask as a bitmap."""

    def __init__(self, lib, c=256):
        super().__init__()
        self.lib = lib
        self.multihead_attn = MultiheadAttention(lib, c)
        self.norm = LayerNorm(lib, c)

    def forward(self, tgt, mem_k, mem_v, qe, mask_bits):
        t2 = self.norm(tgt)
        return self.multihead_attn(_AddFn.apply(t2, qe, self.lib), mem_k, mem_v, residual=tgt, mask_bits=mask_bits)


class FFNLayer(nn.Module):
    """transformer.py:241-409 (pre-norm, ReLU)."""

    def __init__(self, lib, c=256, ffn=1024):
        super().__init__()
        self.linear1 = Linear(lib, c, ffn, act="relu")
        self.linear2 = Linear(lib, ffn, c)
        self.norm = LayerNorm(lib, c)

    def forward(self, tgt):
        return self.linear2(self.linear1(self.norm(tgt)), residual=tgt)


class PredictionHeads(nn.Module):
    """modelling.py:26-113: decoder_norm -> (class logits, mask logits = mask_embed x mask_features, attention bitmap)."""

    def __init__(self, lib, c, nc, md):
        super().__init__()
        self.lib = lib
        self.decoder_norm = LayerNorm(lib, c)
        self.classifier = Linear(lib, c, nc + 1)
        self.mask_classifier = MLP(lib, c, c, md, 3)

    def forward(self, x, mask_features, mf_level: Optional[torch.Tensor]):
        lib = self.lib
        dn = self.decoder_norm(x)
        cls = self.classifier(dn)
        emb = self.mask_classifier(dn)
        masks = _MaskEinsumFn.apply(emb, mask_features, lib)
        bits = None
        if mf_level is not None:   # `interpolate(masks) < 0` = `mask_embed x interpolate(mask_features) < 0`, detached (modelling.py:92-107)
            B, Q, Cc = emb.shape
            L = mf_level.shape[1] * mf_level.shape[2]
            words = (L + 31) // 32
            bits = torch.zeros(B * Q, words, dtype=torch.int32, device=emb.device)
            e, ml, cp = _einsum_operands(emb.detach(), mf_level)
            check(lib.fx_query_pixel_logits_bf16(e.data_ptr(), cp, ml.data_ptr(), cp, 2, None, 0, bits.data_ptr(), words, B, Q, L, cp, _stream(e.device)),
                  "fx_query_pixel_logits_bf16")
        return cls, masks, bits


class TransformerDecoder(nn.Module):
    """modelling.py:285-461 over the first ``nlev`` levels (2 for BiSeNetFormer)."""

    def __init__(self, lib, nc, od, md, c=256, nq=100, ffn=1024, nl=6, nlev=2):
        super().__init__()
        self.lib, self.c, self.nq, self.nl, self.nlev = lib, c, nq, nl, min(nlev, nl)
        self.transformer_self_attention_layers = nn.ModuleList([SelfAttentionLayer(lib, c) for _ in range(nl)])
        self.transformer_cross_attention_layers = nn.ModuleList([CrossAttentionLayer(lib, c) for _ in range(nl)])
        self.transformer_ffn_layers = nn.ModuleList([FFNLayer(lib, c, ffn) for _ in range(nl)])
        self.query_feat = _Holder()
        self.query_feat.weight = nn.Parameter(torch.empty(nq, c))
        self.query_embed = _Holder()
        self.query_embed.weight = nn.Parameter(torch.empty(nq, c))
        self.input_proj = nn.ModuleList([Conv1x1(lib, od, c, bias=True) for _ in range(self.nlev)])
        self.forward_prediction_heads = PredictionHeads(lib, c, nc, md)
        self._pos: Dict[tuple, torch.Tensor] = {}

    def _pos_for(self, h, w, dev):
        key = (h, w, dev)
        if key not in self._pos:
            self._pos[key] = pos_embed_sine_normalized(h, w, self.c // 2).to(device=dev, dtype=_lib.act_dtype()).contiguous()
        return self._pos[key]

    def forward(self, msf: Sequence[torch.Tensor], mask_features: torch.Tensor, forced_attn: Optional[Sequence[torch.Tensor]] = None):
        lib = self.lib
        B = mask_features.shape[0]
        dev = mask_features.device
        mem_k, mem_v, mf_lvl = [], [], []
        for i in range(self.nlev):
            f = msf[i]
            h, w = f.shape[1], f.shape[2]
            src = self.input_proj[i](f).reshape(B, h * w, self.c)
            mem_v.append(src)
            mem_k.append(_AddFn.apply(src, self._pos_for(h, w, dev), lib))
            with torch.no_grad():
                mf_lvl.append(_ResizeFn.apply(mask_features.detach(), h, w, lib))
        qe = self.query_embed.weight.to(_lib.act_dtype())
        out = self.query_feat.weight.to(_lib.act_dtype()).unsqueeze(0).expand(B, -1, -1).contiguous()
        heads = self.forward_prediction_heads
        pc, pm = [], []
        cls, masks, bits = heads(out, mask_features, mf_lvl[0])
        pc.append(cls)
        pm.append(masks)
        self.attn_bits = []
        for i in range(self.nl):
            lvl = i % self.nlev
            if forced_attn is not None:
                bits = pack_mask_bits(forced_attn[i].reshape(B * self.nq, -1), bits.shape[1]).to(dev)
            self.attn_bits.append(bits)
            out = self.transformer_cross_attention_layers[i](out, mem_k[lvl], mem_v[lvl], qe, bits)
            out = self.transformer_self_attention_layers[i](out, qe)
            out = self.transformer_ffn_layers[i](out)
            cls, masks, bits = heads(out, mask_features, mf_lvl[(i + 1) % self.nlev] if i < self.nl - 1 else None)
            pc.append(cls)
            pm.append(masks)
        return {"pred_logits": pc[-1], "pred_masks": pm[-1], "aux_outputs": [{"pred_logits": a, "pred_masks": b} for a, b in zip(pc[:-1], pm[:-1])]}


class _CriterionHolder(nn.Module):
    """``head.criterion``: the checkpoint buffer ``empty_weight`` + the differentiable SetCriterion mirror."""

    def __init__(self, config: Dict, rand=None):
        super().__init__()
        nc = int(config["num_classes"])
        eos = float(config.get("criterion_eos_coef", 0.1))
        w = torch.ones(nc + 1)
        w[-1] = eos
        self.register_buffer("empty_weight", w)
        P = int(config.get("criterion_num_points", 12544))
        matcher = MaskHungarianMatcher(float(config.get("matcher_cost_class", 2)), float(config.get("matcher_cost_mask", 5)),
                                       float(config.get("matcher_cost_dice", 5)), P, bool(config.get("cls_sigmoid", False)), rand=rand)
        self.fn = SetCriterion(nc, matcher, {"loss_ce": float(config.get("weight_dict_loss_ce", 2)), "loss_mask": float(config.get("weight_dict_loss_mask", 5)),
                                             "loss_dice": float(config.get("weight_dict_loss_dice", 5))},
                               eos_coef=eos, num_points=P, oversample_ratio=3.0, importance_sample_ratio=0.75,
                               deep_supervision=bool(config.get("criterion_deep_supervision", True)), rand=rand)

    def forward(self, outputs, targets, fixed_matches=None):
        return self.fn(outputs, targets, fixed_matches)


class BisenetFormerTrainable(nn.Module):
    """Reference-compatible parameter tree (``pixel_decoder.*``, ``head.predictor.*``, ``head.criterion.empty_weight``) whose
    ``forward(images, targets)`` returns the dict of 3 x (dec_layers + 1) weighted losses (loss_ce / loss_mask / loss_dice, ``_i`` suffixes)."""

    family = "bisenetformer"

    def __init__(self, config: Dict, norm: str = "BN", rand=None):
        super().__init__()
        if not torch.cuda.is_available():
            raise _lib.FocoosAmdError("focoos_amd needs a ROCm GPU (gfx950); no CPU fallback exists")
        lib = _lib.load()
        self.config = dict(config)
        bb = config["backbone_config"]
        if bb.get("model_type") != "stdc" or bb.get("block_type", "cat") != "cat" or int(bb.get("block_num", 4)) != 4:
            raise _lib.FocoosAmdError("trainable graph covers the STDC backbone with CatBottleneck blocks of 4 convs (bisenetformer-*)")
        if bool(config.get("cls_sigmoid", False)):
            raise NotImplementedError("cls_sigmoid / bce class loss is not part of the engine criterion (registry models use softmax + CE)")
        md = int(config.get("transformer_predictor_out_dim", 128))
        self.pixel_decoder = BiseNet(lib, config)
        self.head = nn.Module()
        self.head.criterion = _CriterionHolder(config, rand)
        self.head.predictor = TransformerDecoder(lib, int(config["num_classes"]), int(config.get("pixel_decoder_out_dim", 128)), md,
                                                 c=int(config.get("transformer_predictor_hidden_dim", 256)), nq=int(config.get("num_queries", 100)),
                                                 ffn=int(config.get("transformer_predictor_dim_feedforward", 1024)),
                                                 nl=int(config.get("transformer_predictor_dec_layers", 6)), nlev=2)
        set_norm_mode(self, norm)

    grad_ready = None   # callable(segment_name) set by TrainStep (overlapped gradient all-reduce)

    def forward_outputs(self, images: torch.Tensor, forced_attn=None):
        """Everything before the criterion: the prediction sets (static shapes - what a captured training step replays, TrainStep graphs)."""
        f = dict(self.pixel_decoder.backbone(images))
        from .train import notify_when_all_grads

        # segment boundaries as aliases (see train_detr.FAIDetrTrainable.forward_outputs: the raw tensors are not an antichain of the graph)
        for k in ("res3", "res4", "res5"):
            f[k] = f[k].view_as(f[k])
        if self.grad_ready is not None:
            notify_when_all_grads([f["res3"], f["res4"], f["res5"]], self.grad_ready, "encoder")
        mask_features, msf = self.pixel_decoder.decode(f)
        mask_features, msf = mask_features.view_as(mask_features), [m.view_as(m) for m in msf]
        if self.grad_ready is not None:
            notify_when_all_grads([mask_features, msf[0], msf[1]], self.grad_ready, "head")
        self.segment_boundaries = {"head": [mask_features, msf[0], msf[1]], "encoder": [f["res3"], f["res4"], f["res5"]]}   # TrainStep._staged_backward
        out = self.head.predictor(msf[:-1], mask_features, forced_attn)
        self.last_outputs = out
        return out

    def forward(self, images: torch.Tensor, targets: Sequence, forced_attn=None, fixed_matches=None):
        return self.head.criterion(self.forward_outputs(images, forced_attn), targets, fixed_matches)
