"""MaskFormer (``fai-mf-*``) as a trainable HIP autograd graph: what ``FAIMaskFormer.forward(images,
targets)`` computes under ``model.train()`` (focoos/models/fai_mf/modelling.py:712-725) - ResNet-vd (train_nn.ResNetVd) or STDC (train_bf.STDC) backbone, the
``TransformerFPN`` pixel decoder (:201-369: 1x1 input projection, pre-norm transformer encoder on res5 with the normalised sine position
embedding :143-198, lateral 1x1 / output 3x3 convs with BatchNorm, nearest-neighbour top-down additions, biased 3x3 ``mask_features``), the
``MultiScaleMaskedTransformerDecoder`` over three levels with every prediction head supervised (:453-549; the decoder modules are
train_bf's, the reference's two decoder files differ only in the number of levels and the mask dimension) and the point-sampled Hungarian
``SetCriterion`` (fai_mf/loss.py:345-607, 626-723 = mask_criterion.py).  Parameter names are the reference's.

New nodes here: ``ConvNormFlat`` (detectron-style ``Conv2d`` whose weight lives on the module itself next to ``norm.*``), ``ConvBias``
(biased convolution without normalisation), ``_UpAddFn`` (nearest upsample + add; backward = sums over each source's pre-image), the pre-norm
encoder layer.  Glue on small tensors as listed in train_bf.py."""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch
from torch import nn

from . import _lib
from ._lib import check
from .engine_maskdec import pos_embed_sine_normalized
from .train_bf import STDC, Conv1x1, TransformerDecoder, _CriterionHolder
from .train_nn import (ARENA, DIRECT_GRAD, WEIGHTS_EPOCH, ConvNormLayer, LayerNorm, Linear, MultiheadAttention, ResNetVd, _AddFn, _conv_call,
                       _conv_input_grad, _conv_param_grads, _frag_eligible, _ptr, _stream, set_norm_mode)


class ConvNormFlat(ConvNormLayer):
    """focoos.nn.layers.conv.Conv2d(bias=False, norm=BatchNorm2d, activation=...) (conv.py:33-69): keys ``weight`` / ``norm.*``."""

    def __init__(self, lib, cin, cout, k, act):
        super().__init__(lib, cin, cout, k, 1, act, names=("_conv", "norm"))
        w = self._conv_h.weight
        del self._modules["_conv"]
        self.weight = w                               # the parameter is registered on this module
        object.__setattr__(self, "_conv_h", self)


class _ConvBiasFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, layer: "ConvBias"):
        layer.sync_packed()
        N, Cc, KH, KW = weight.shape
        y = _conv_call(layer.lib, x, layer.w_fwd, layer.shift, N, KH, KW, 1, layer.pad, None, None, w_frag=layer.w_fwd_frag)
        ctx.layer = layer
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        layer: ConvBias = ctx.layer
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = _conv_input_grad(layer, dy, x.shape) if ctx.needs_input_grad[0] else None
        dw = _conv_param_grads(layer, x, dy, None) if ctx.needs_input_grad[1] else None
        db = None
        if ctx.needs_input_grad[2]:
            N = dy.shape[-1]
            direct = DIRECT_GRAD[0] and layer.bias.grad is not None
            db = layer.bias.grad if direct else ARENA.zeros((N,), dy.device)
            check(layer.lib.fx_colsum_bf16(dy.data_ptr(), N, db.data_ptr(), dy.numel() // N, N, _stream(dy.device)), "fx_colsum_bf16")
            if direct:
                db = None
        return dx, dw, db, None


class ConvBias(nn.Module):
    """Conv2d(cin, cout, k, padding=k//2) with bias, no normalisation (``mask_features``): keys ``weight`` / ``bias``."""

    norm_mode = "FrozenBN"
    batch_stats = False
    scale = None

    def __init__(self, lib, cin, cout, k):
        super().__init__()
        self.lib, self.cin, self.cout, self.k, self.stride, self.pad = lib, cin, cout, k, 1, (k - 1) // 2
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.zeros(cout))
        object.__setattr__(self, "_conv_h", self)
        self._ver = None
        self.w_fwd = self.w_dgrad = self.shift = self.w_fwd_frag = self.w_dgrad_frag = None

    def sync_packed(self):
        w, b = self.weight, self.bias
        ver = (w._version, b._version, w.device, WEIGHTS_EPOCH[0])
        if ver == self._ver:
            return
        dev = w.device
        N, Cc, k = self.cout, self.cin, self.k
        with torch.no_grad():
            Np, Cp = (N + 127) // 128 * 128, (Cc + 127) // 128 * 128
            if self.w_fwd is None or self.w_fwd.device != dev:
                self.w_fwd = torch.zeros(Np, k, k, Cc, dtype=_lib.act_dtype(), device=dev)
                self.w_dgrad = torch.zeros(Cp, k, k, N, dtype=_lib.act_dtype(), device=dev)
                self.shift = torch.zeros(Np, dtype=torch.float32, device=dev)
                self.w_fwd_frag = torch.empty(N * k * k * Cc, dtype=_lib.act_dtype(), device=dev) if _frag_eligible(N, Cc, k) else None
                self.w_dgrad_frag = torch.empty(N * k * k * Cc, dtype=_lib.act_dtype(), device=dev) if _frag_eligible(Cc, N, k) else None
            self.shift[:N] = b
            check(self.lib.fx_pack_conv_weights_f32(w.data_ptr(), None, self.w_fwd.data_ptr(), self.w_dgrad.data_ptr(), _ptr(self.w_fwd_frag),
                                                    _ptr(self.w_dgrad_frag), N, Cc, k, k, _stream(dev)), "fx_pack_conv_weights_f32")
        self._ver = ver

    def forward(self, x):
        return _ConvBiasFn.apply(x, self.weight, self.bias, self)


class _UpAddFn(torch.autograd.Function):
    """cur + interpolate(y, size=cur.shape, mode="nearest") (TransformerFPN top-down path, modelling.py:361-364): fx_upsample_nearest_add_nhwc_bf16;
    backward: d cur = d, d y = the sums of d over each source pixel's pre-image (fx_upsample_nearest_bwd_nhwc_bf16; 2x2 blocks for the exact
    x2 of inputs that are multiples of 32, 1-2 pixels per axis at the ceil(H/2) levels of other sizes)."""

    @staticmethod
    def forward(ctx, cur, y, lib):
        B, H, W_, Cc = cur.shape
        if y.shape[0] != B or y.shape[3] != Cc:
            raise _lib.FocoosAmdError(f"nearest top-down addition: batch / channel mismatch ({tuple(y.shape)} -> {tuple(cur.shape)})")
        cur, y = cur.contiguous(), y.contiguous()
        out = torch.empty_like(cur)
        check(lib.fx_upsample_nearest_add_nhwc_bf16(cur.data_ptr(), Cc, y.data_ptr(), Cc, out.data_ptr(), Cc, B, H, W_, y.shape[1], y.shape[2], Cc,
                                                    _stream(cur.device)), "fx_upsample_nearest_add_nhwc_bf16")
        ctx.lib, ctx.src_hw = lib, (y.shape[1], y.shape[2])
        return out

    @staticmethod
    def backward(ctx, d):
        d = d.contiguous()
        B, H, W_, Cc = d.shape
        hs, ws = ctx.src_hw
        dy = torch.empty(B, hs, ws, Cc, dtype=_lib.act_dtype(), device=d.device)
        check(ctx.lib.fx_upsample_nearest_bwd_nhwc_bf16(d.data_ptr(), Cc, dy.data_ptr(), Cc, B, H, W_, hs, ws, Cc, _stream(d.device)),
              "fx_upsample_nearest_bwd_nhwc_bf16")
        return d, dy, None


class PreNormEncoderLayer(nn.Module):
    """focoos/nn/layers/transformer.py:553-601 with normalize_before=True and ReLU: src + attn(norm1(src) + pos, ., norm1(src)), then the FFN."""

    def __init__(self, lib, c, ffn):
        super().__init__()
        self.lib = lib
        self.self_attn = MultiheadAttention(lib, c)
        self.linear1 = Linear(lib, c, ffn, act="relu")
        self.linear2 = Linear(lib, ffn, c)
        self.norm1 = LayerNorm(lib, c)
        self.norm2 = LayerNorm(lib, c)

    def forward(self, src, pos):
        s2 = self.norm1(src)
        qk = _AddFn.apply(s2, pos, self.lib)
        src = self.self_attn(qk, qk, s2, residual=src)
        return self.linear2(self.linear1(self.norm2(src)), residual=src)


class _Encoder(nn.Module):
    def __init__(self, lib, c, ffn, n):
        super().__init__()
        self.layers = nn.ModuleList([PreNormEncoderLayer(lib, c, ffn) for _ in range(n)])
        self.norm = LayerNorm(lib, c)


class _TransformerEncoderOnly(nn.Module):
    """TransformerEncoderOnly (modelling.py:143-198): keys ``encoder.layers.i.*`` / ``encoder.norm.*``."""

    def __init__(self, lib, c, ffn, n):
        super().__init__()
        self.encoder = _Encoder(lib, c, ffn, n)

    def forward(self, src, pos):
        for layer in self.encoder.layers:
            src = layer(src, pos)
        return self.encoder.norm(src)


class TransformerFPN(nn.Module):
    """modelling.py:201-369.  Returns (mask_features [B,H/4,W/4,od], [stride 32, 16, 8 maps])."""

    def __init__(self, lib, config: Dict):
        super().__init__()
        self.lib = lib
        fd = int(config.get("pixel_decoder_feat_dim", 256))
        od = int(config.get("pixel_decoder_out_dim", 256))
        n_enc = int(config.get("pixel_decoder_transformer_layers", 0))
        if fd not in (128, 256):
            raise _lib.FocoosAmdError("pixel_decoder_feat_dim 256, or 128 (fai-mf-{l,m}-ade; fai-mf-{m,s}-coco-ins: the encoder's 8 heads of 16 channels run "
                                      "zero-padded on the head-dim-32 attention kernels, train_nn.MultiheadAttention)")
        if int(config.get("pixel_decoder_transformer_nheads", 8)) != 8:
            raise _lib.FocoosAmdError("attention kernels: 8 heads of 32 channels")
        self.fd, self.n_enc = fd, n_enc
        bb = config["backbone_config"]
        mean, std = config.get("pixel_mean", (123.675, 116.28, 103.53)), config.get("pixel_std", (58.395, 57.12, 57.375))
        if bb.get("model_type") == "stdc":      # fai-mf-m-ade: the BiSeNetFormer backbone under the MaskFormer FPN
            base = int(bb.get("base", 64))
            self.backbone = STDC(lib, base, tuple(bb.get("layers", (4, 5, 3))), mean, std)
            chans = [base, base * 4, base * 8, base * 16]
        else:
            self.backbone = ResNetVd(int(bb.get("depth", 50)), mean, std)
            chans = [256, 512, 1024, 2048]
        if n_enc > 0:
            self.input_proj = Conv1x1(lib, chans[-1], fd, bias=True)
            self.transformer = _TransformerEncoderOnly(lib, fd, int(config.get("pixel_decoder_transformer_dim_feedforward", 1024)), n_enc)
        for idx, c in enumerate(chans):
            if idx < len(chans) - 1:
                setattr(self, f"adapter_{idx + 1}", ConvNormFlat(lib, c, fd, 1, None))
            setattr(self, f"layer_{idx + 1}", ConvNormFlat(lib, fd if (idx < len(chans) - 1 or n_enc > 0) else c, fd, 3, "relu"))
        self.mask_features = ConvBias(lib, fd, od, 3)
        self._pos: Dict[tuple, torch.Tensor] = {}

    def decode(self, feats: Dict[str, torch.Tensor]):
        lib = self.lib
        x = feats["res5"]
        if self.n_enc > 0:
            x = self.input_proj(x)
            B, h, w, c = x.shape
            key = (h, w, x.device)
            if key not in self._pos:
                self._pos[key] = pos_embed_sine_normalized(h, w, c // 2).to(device=x.device, dtype=_lib.act_dtype()).contiguous()
            x = self.transformer(x.reshape(B, h * w, c), self._pos[key]).reshape(B, h, w, c)
        y = self.layer_4(x)
        msf = [y]
        for idx, name in ((3, "res4"), (2, "res3"), (1, "res2")):
            cur = getattr(self, f"adapter_{idx}")(feats[name])
            y = getattr(self, f"layer_{idx}")(_UpAddFn.apply(cur, y, lib))
            if len(msf) < 3:
                msf.append(y)
        return self.mask_features(y), msf

    def forward(self, images):
        return self.decode(self.backbone(images))


class FAIMaskFormerTrainable(nn.Module):
    """Reference-compatible parameter tree of ``FAIMaskFormer`` whose ``forward(images, targets)`` returns the dict of
    3 x (dec_layers + 1) weighted losses."""

    family = "fai_mf"

    def __init__(self, config: Dict, norm: str = "BN", rand=None):
        super().__init__()
        if not torch.cuda.is_available():
            raise _lib.FocoosAmdError("focoos_amd needs a ROCm GPU (gfx950); no CPU fallback exists")
        lib = _lib.load()
        self.config = dict(config)
        if bool(config.get("cls_sigmoid", False)):
            raise NotImplementedError("cls_sigmoid / bce class loss is not part of the engine criterion (registry models use softmax + CE)")
        self.pixel_decoder = TransformerFPN(lib, config)
        self.head = nn.Module()
        self.head.criterion = _CriterionHolder(config, rand)
        self.head.predictor = TransformerDecoder(lib, int(config["num_classes"]), int(config.get("pixel_decoder_out_dim", 256)),
                                                 int(config.get("transformer_predictor_out_dim", 256)), c=int(config.get("transformer_predictor_hidden_dim", 256)),
                                                 nq=int(config.get("num_queries", 100)), ffn=int(config.get("transformer_predictor_dim_feedforward", 1024)),
                                                 nl=int(config.get("transformer_predictor_dec_layers", 6)), nlev=3)
        set_norm_mode(self, norm)

    grad_ready = None   # callable(segment_name) set by TrainStep (overlapped gradient all-reduce)

    def forward_outputs(self, images: torch.Tensor, forced_attn=None):
        """Everything before the criterion: the prediction sets (static shapes - what a captured training step replays, TrainStep graphs)."""
        f = dict(self.pixel_decoder.backbone(images))
        from .train import notify_when_all_grads

        # segment boundaries as aliases (see train_detr.FAIDetrTrainable.forward_outputs: the raw tensors are not an antichain of the graph)
        for k in ("res2", "res3", "res4", "res5"):
            f[k] = f[k].view_as(f[k])
        if self.grad_ready is not None:
            notify_when_all_grads([f["res2"], f["res3"], f["res4"], f["res5"]], self.grad_ready, "encoder")
        mask_features, msf = self.pixel_decoder.decode(f)
        mask_features, msf = mask_features.view_as(mask_features), [m.view_as(m) for m in msf]
        if self.grad_ready is not None:
            notify_when_all_grads([mask_features] + list(msf), self.grad_ready, "head")
        self.segment_boundaries = {"head": [mask_features] + list(msf), "encoder": [f["res2"], f["res3"], f["res4"], f["res5"]]}   # TrainStep._staged_backward
        out = self.head.predictor(msf, mask_features, forced_attn)
        self.last_outputs = out
        return out

    def forward(self, images: torch.Tensor, targets: Sequence, forced_attn=None, fixed_matches=None):
        return self.head.criterion(self.forward_outputs(images, forced_attn), targets, fixed_matches)
