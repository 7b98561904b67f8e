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
from typing import Dict, List, Optional

import torch
from torch import nn

from . import _lib
from ._lib import FX_ACT, FxConvDesc, check
from .state_spec import RESNET_BLOCKS

BN_EPS = 1e-5


def _stream(dev) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _conv_call(lib, x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], N: int, KH: int, KW: int, stride: int, pad: int,
               act: Optional[str], residual: Optional[torch.Tensor]) -> torch.Tensor:
    """fx_conv2d_nhwc_bf16 on NHWC bf16 tensors; ``w`` is a packed [Npad][KH][KW][C] bf16 image."""
    B, H, W_, Cc = x.shape
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W_ + 2 * pad - KW) // stride + 1
    y = torch.empty(B, Ho, Wo, N, dtype=torch.bfloat16, device=x.device)
    d = FxConvDesc()
    d.x, d.w, d.y = x.data_ptr(), w.data_ptr(), y.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.residual = residual.data_ptr() if residual is not None else None
    d.B, d.H, d.W, d.C, d.ldx = B, H, W_, Cc, Cc
    d.Ho, d.Wo, d.N, d.ldy, d.ldr = Ho, Wo, N, N, N if residual is not None else 0
    d.KH, d.KW, d.stride, d.pad = KH, KW, stride, pad
    d.pool2, d.act, d.out_f32, d.residual_after_act, d.y_batch_stride = 0, FX_ACT[act], 0, 0, 0
    check(lib.fx_conv2d_nhwc_bf16(C.byref(d), _stream(x.device)), "fx_conv2d_nhwc_bf16")
    return y


class _ConvBnActFn(torch.autograd.Function):
    """y = act(conv(x, W * s) + (beta - mu * s) [+ residual]) with s = gamma / sqrt(var + eps) frozen."""

    @staticmethod
    def forward(ctx, x, weight, residual, layer: "ConvNormLayer"):
        lib = layer.lib
        layer.sync_packed()
        N, Cc, KH, KW = weight.shape
        y = _conv_call(lib, x, layer.w_fwd, layer.shift, N, KH, KW, layer.stride, layer.pad, layer.act, residual)
        ctx.layer = layer
        ctx.has_res = residual is not None
        ctx.save_for_backward(x, y)
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
        else:  # pragma: no cover
            raise NotImplementedError(f"backward of activation {layer.act}")
        dx = None
        if ctx.needs_input_grad[0]:
            if layer.stride == 1:
                src = dz
            else:  # stride 2: zero-insert, then the stride-1 transposed filter
                src = torch.empty(B, H, W_, N, dtype=torch.bfloat16, device=dev)
                check(lib.fx_zero_insert2_nhwc_bf16(dz.data_ptr(), N, src.data_ptr(), N, B, Ho, Wo, H, W_, N, st), "fx_zero_insert2_nhwc_bf16")
            dx = _conv_call(lib, src, layer.w_dgrad, None, Cc, layer.k, layer.k, 1, layer.pad, None, None)
        dw = None
        if ctx.needs_input_grad[1]:
            dw_eff = torch.zeros(N, layer.k, layer.k, Cc, dtype=torch.float32, device=dev)
            check(lib.fx_conv2d_wgrad_nhwc_bf16(x.data_ptr(), Cc, dz.data_ptr(), N, dw_eff.data_ptr(), B, H, W_, Cc, Ho, Wo, N, layer.k, layer.k,
                                                layer.stride, layer.pad, st), "fx_conv2d_wgrad_nhwc_bf16")
            dw = torch.empty(N, Cc, layer.k, layer.k, dtype=torch.float32, device=dev)
            check(lib.fx_unpack_conv_wgrad_f32(dw_eff.data_ptr(), layer.scale.data_ptr(), dw.data_ptr(), N, Cc, layer.k, layer.k, Cc, 0, st),
                  "fx_unpack_conv_wgrad_f32")
        return dx, dw, (dz if ctx.has_res else None), None


class _Holder(nn.Module):
    """Parameter container with the reference's attribute names (``conv.weight``; ``norm.weight/bias/running_*``)."""


class ConvNormLayer(nn.Module):
    """focoos/nn/layers/conv.py:78-98 — conv (no bias) + BatchNorm2d (frozen here) + activation."""

    def __init__(self, lib, cin: int, cout: int, k: int, stride: int = 1, act: Optional[str] = None):
        super().__init__()
        self.lib, self.cin, self.cout, self.k, self.stride, self.act = lib, cin, cout, k, stride, act
        self.pad = (k - 1) // 2
        self.conv = _Holder()
        self.conv.weight = nn.Parameter(torch.empty(cout, cin, k, k, dtype=torch.float32))
        self.norm = _Holder()
        self.norm.weight = nn.Parameter(torch.ones(cout), requires_grad=False)   # frozen BatchNorm
        self.norm.bias = nn.Parameter(torch.zeros(cout), requires_grad=False)
        self.norm.register_buffer("running_mean", torch.zeros(cout))
        self.norm.register_buffer("running_var", torch.ones(cout))
        self.norm.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self._packed_version = None
        self.w_fwd = self.w_dgrad = self.scale = self.shift = None

    def sync_packed(self):
        """(Re)build the bf16 weight images when the master weight changed (optimizer step, load_state_dict)."""
        w = self.conv.weight
        ver = (w._version, self.norm.weight._version, self.norm.running_var._version, w.device)
        if ver == self._packed_version:
            return
        dev = w.device
        N, Cc, k = self.cout, self.cin, self.k
        with torch.no_grad():
            self.scale = (self.norm.weight.double() / torch.sqrt(self.norm.running_var.double() + BN_EPS)).float().contiguous()
            shift = (self.norm.bias.double() - self.norm.running_mean.double() * self.scale.double()).float()
            Np, Cp = (N + 127) // 128 * 128, (Cc + 127) // 128 * 128
            self.shift = torch.zeros(Np, dtype=torch.float32, device=dev)
            self.shift[:N] = shift
            if self.w_fwd is None or self.w_fwd.device != dev:
                self.w_fwd = torch.zeros(Np, k, k, Cc, dtype=torch.bfloat16, device=dev)
                self.w_dgrad = torch.zeros(Cp, k, k, N, dtype=torch.bfloat16, device=dev)
            check(self.lib.fx_pack_conv_weights_f32(w.data_ptr(), self.scale.data_ptr(), self.w_fwd.data_ptr(), self.w_dgrad.data_ptr(), N, Cc, k, k,
                                                    _stream(dev)), "fx_pack_conv_weights_f32")
        self._packed_version = ver

    def forward(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        return _ConvBnActFn.apply(x, self.conv.weight, residual, self)


class _StemFn(torch.autograd.Function):
    """conv1_1: normalise + 3x3 stride-2 conv 3->32 + frozen BN + ReLU straight from the uint8 / fp32 HWC image
    (fx_stem_conv3x3s2).  Backward: weight gradient only (the image needs none) through the generic wgrad kernel on the
    normalised image padded to 8 channels."""

    @staticmethod
    def forward(ctx, images, weight, layer: "StemConv"):
        lib = layer.lib
        layer.sync_packed()
        B, H, W_, _ = images.shape
        y = torch.empty(B, H // 2, W_ // 2, 32, dtype=torch.bfloat16, device=images.device)
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
        Ho, Wo = H // 2, W_ // 2
        st = _stream(dev)
        dy = dy.contiguous()
        dz = torch.empty_like(y)
        check(lib.fx_relu_bwd_bf16(dy.data_ptr(), 32, None, 0, y.data_ptr(), 32, dz.data_ptr(), 32, B * Ho * Wo, 32, 1, st), "fx_relu_bwd_bf16")
        xn = torch.empty(B, H, W_, 8, dtype=torch.bfloat16, device=dev)
        check(lib.fx_normalize_pad8(images.data_ptr(), int(images.dtype == torch.float32), layer.px_mean.data_ptr(), layer.px_inv_std.data_ptr(),
                                    xn.data_ptr(), B * H * W_, st), "fx_normalize_pad8")
        dw_eff = torch.zeros(32, 3, 3, 8, dtype=torch.float32, device=dev)
        check(lib.fx_conv2d_wgrad_nhwc_bf16(xn.data_ptr(), 8, dz.data_ptr(), 32, dw_eff.data_ptr(), B, H, W_, 8, Ho, Wo, 32, 3, 3, 2, 1, st),
              "fx_conv2d_wgrad_nhwc_bf16")
        dw = torch.empty(32, 3, 3, 3, dtype=torch.float32, device=dev)
        check(lib.fx_unpack_conv_wgrad_f32(dw_eff.data_ptr(), layer.scale.data_ptr(), dw.data_ptr(), 32, 3, 3, 3, 8, 0, st), "fx_unpack_conv_wgrad_f32")
        return None, dw, None


class StemConv(ConvNormLayer):
    def __init__(self, lib, pixel_mean, pixel_std):
        super().__init__(lib, 3, 32, 3, 2, "relu")
        self.register_buffer("px_mean", torch.tensor(pixel_mean, dtype=torch.float32), persistent=False)
        self.register_buffer("px_inv_std", 1.0 / torch.tensor(pixel_std, dtype=torch.float32), persistent=False)
        self.stem_w = self.stem_b = None

    def sync_packed(self):
        w = self.conv.weight
        ver = (w._version, w.device)
        if ver == self._packed_version:
            return
        with torch.no_grad():
            self.scale = (self.norm.weight.double() / torch.sqrt(self.norm.running_var.double() + BN_EPS)).float().contiguous()
            self.stem_b = (self.norm.bias.double() - self.norm.running_mean.double() * self.scale.double()).float().contiguous()
            self.stem_w = (w * self.scale.view(-1, 1, 1, 1)).permute(2, 3, 1, 0).contiguous()  # [kh][kw][c][n] fp32 (tiny: 864 values)
        self._packed_version = ver

    def forward(self, images: torch.Tensor) -> torch.Tensor:  # type: ignore[override]
        return _StemFn.apply(images, self.conv.weight, self)


class _PoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, lib, kind: str):
        B, H, W_, Cc = x.shape
        Ho, Wo = ((H + 2 - 3) // 2 + 1, (W_ + 2 - 3) // 2 + 1) if kind == "max" else ((H + 1) // 2, (W_ + 1) // 2)
        y = torch.empty(B, Ho, Wo, Cc, dtype=torch.bfloat16, device=x.device)
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
            check(lib.fx_maxpool3x3s2_bwd_nhwc_bf16(x.data_ptr(), Cc, dy.data_ptr(), Cc, dx.data_ptr(), Cc, B, H, W_, Cc, _stream(x.device)),
                  "fx_maxpool3x3s2_bwd_nhwc_bf16")
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


class BottleNeck(nn.Module):
    """focoos/nn/backbone/resnet.py:72-121 (variant d: stride on the 3x3)."""

    def __init__(self, lib, ch_in, width, stride, shortcut: bool, first_stage: bool):
        super().__init__()
        self.branch2a = ConvNormLayer(lib, ch_in, width, 1, 1, "relu")
        self.branch2b = ConvNormLayer(lib, width, width, 3, stride, "relu")
        self.branch2c = ConvNormLayer(lib, width, width * 4, 1, 1, "relu")  # ReLU applied AFTER the residual add (fused epilogue)
        self.has_short = not shortcut
        if self.has_short:
            self.short = ConvNormLayer(lib, ch_in, width * 4, 1, 1, None) if (first_stage or stride == 1) else _Short(lib, ch_in, width * 4)

    def forward(self, x):
        out = self.branch2b(self.branch2a(x))
        short = self.short(x) if self.has_short else x
        return self.branch2c(out, residual=short)  # relu(conv + bn + short)


class _Blocks(nn.Module):
    def __init__(self, blocks):
        super().__init__()
        self.blocks = nn.ModuleList(blocks)

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
