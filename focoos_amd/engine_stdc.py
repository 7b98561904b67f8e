"""STDC backbone (focoos/nn/backbone/stdc.py:108-166 CatBottleneck, :282-320 STDC) for the inference engines that can sit on it:
BiSeNetFormer (bisenetformer-*), MaskFormer (fai-mf-m-ade) - shared weight packing and launch sequence (round 4: moved out of engine_bf.py).

Restructured relative to the reference (same arithmetic): eval BatchNorm folded into every conv; the four branches of a CatBottleneck are
written straight into channel slices of the block output (no torch.cat copy); the depthwise stride-2 conv (``avd_layer``) and the
``AvgPool2d(3, 2, 1)`` skip are one bandwidth kernel each (the pool = the same kernel with w = 1/9)."""
from __future__ import annotations

from typing import Dict

import torch

from . import _lib

# (engine.py imports this module - the RT-DETR engine takes the STDC backbone too (fai-detr-m-coco) - so its helpers are imported at call time)

BN_EPS = 1e-5


def _bn_scale_shift(sd, prefix: str):
    g, bta = sd[f"{prefix}.weight"].double(), sd[f"{prefix}.bias"].double()
    mu, var = sd[f"{prefix}.running_mean"].double(), sd[f"{prefix}.running_var"].double()
    s = g / torch.sqrt(var + BN_EPS)
    return s, bta - mu * s


class StdcEngineMixin:
    """Engine side: ``_init_stdc(backbone_config)`` validates / records the geometry, ``_pack_stdc(sd, P)`` packs the weights."""

    def _init_stdc(self, bb: Dict) -> None:
        if bb.get("model_type") != "stdc" or bb.get("block_type", "cat") != "cat" or int(bb.get("block_num", 4)) != 4:
            raise _lib.FocoosAmdError("engine covers the STDC backbone with CatBottleneck blocks of 4 convs (bisenetformer-*, fai-mf-m-ade)")
        self.layers = tuple(int(v) for v in bb.get("layers", (4, 5, 3)))
        self.base = int(bb.get("base", 64))
        if self.base != 64:
            raise _lib.FocoosAmdError("engine conv kernels need channel counts that are multiples of 32: STDC base must be 64")
        if not hasattr(self, "vec"):
            self.vec: Dict[str, torch.Tensor] = {}

    def _pack_stdc(self, sd, P: Dict[str, PackedConv]) -> None:
        from .engine import _fold_bn

        bb = "pixel_decoder.backbone"
        if not hasattr(self, "vec"):
            self.vec = {}

        def cbn(name):   # ConvX: conv + BatchNorm folded
            P[name] = self._pack(*_fold_bn(sd, f"{name}.conv.weight", f"{name}.bn"))

        w, b = _fold_bn(sd, f"{bb}.features.0.conv.weight", f"{bb}.features.0.bn")
        self.stem_w = self._dev(w.permute(2, 3, 1, 0).contiguous())  # [kh][kw][c][n] fp32 (direct-conv stem kernel)
        self.stem_b = self._dev(b)
        mean = torch.tensor(self.cfg.get("pixel_mean", [123.675, 116.28, 103.53]), dtype=torch.float32)
        std = torch.tensor(self.cfg.get("pixel_std", [58.395, 57.12, 57.375]), dtype=torch.float32)
        self.px_mean, self.px_inv_std = self._dev(mean), self._dev(1.0 / std)
        cbn(f"{bb}.features.1")
        idx = 2
        for n in self.layers:
            for j in range(n):
                p = f"{bb}.features.{idx}"
                for k in range(4):
                    cbn(f"{p}.conv_list.{k}")
                if j == 0:
                    s, sh = _bn_scale_shift(sd, f"{p}.avd_layer.1")
                    wd = sd[f"{p}.avd_layer.0.weight"].double()[:, 0] * s.view(-1, 1, 1)          # [C,3,3]
                    self.vec[f"{p}.avd.w"] = self._dev(wd.permute(1, 2, 0).reshape(9, -1).float())  # [9][C]
                    self.vec[f"{p}.avd.b"] = self._dev(sh.float())
                    self.vec[f"{p}.pool.w"] = self._dev(torch.full((9, wd.shape[0]), 1.0 / 9.0, dtype=torch.float32))
                idx += 1


class StdcPlanMixin:
    """Plan side (a subclass of engine._PlanBase whose engine is a StdcEngineMixin)."""

    def build_stdc(self) -> Dict[int, NT]:
        """STDC.forward (stdc.py:313-320) -> {2: res2 (stride 4), 3: res3, 4: res4, 5: res5} (NHWC bf16)."""
        e, P, B, lib = self.eng, self.eng.P, self.B, self.lib
        H, W = self.H, self.W
        bb = "pixel_decoder.backbone"
        self.input = self._io("input", (B, H, W, 3), torch.float32 if self.f32_input else torch.uint8)
        self.sizes = self._io("sizes", (B, 2), torch.int32)
        c1 = self._new("features.0", B, (H + 1) // 2, (W + 1) // 2, 32)
        self._op(lib.fx_stem_conv3x3s2, self.input.data_ptr(), int(self.f32_input), e.stem_w.data_ptr(), e.stem_b.data_ptr(), e.px_mean.data_ptr(),
                 e.px_inv_std.data_ptr(), c1.ptr, B, H, W, 32)
        x = self.conv(c1, P[f"{bb}.features.1"], name="features.1", stride=2, act="relu")
        feats = {2: x}
        idx = 2
        for i, n in enumerate(e.layers):
            cout = e.base * 2 ** (i + 2)
            for j in range(n):
                p = f"{bb}.features.{idx}"
                stride = 2 if j == 0 else 1
                Ho, Wo = (x.H + 1) // 2 if stride == 2 else x.H, (x.W + 1) // 2 if stride == 2 else x.W
                blk = self._new(f"features.{idx}", B, Ho, Wo, cout)
                if stride == 1:
                    out1 = self.conv(x, P[f"{p}.conv_list.0"], out=blk.slice(0, cout // 2), act="relu")
                    cur = out1
                else:
                    out1 = self.conv(x, P[f"{p}.conv_list.0"], name=f"features.{idx}.out1", act="relu")
                    cur = self._new(f"features.{idx}.avd", B, Ho, Wo, cout // 2)
                    self._op(lib.fx_dwconv3x3s2_nhwc_bf16, out1.ptr, out1.ld, e.vec[f"{p}.avd.w"].data_ptr(), e.vec[f"{p}.avd.b"].data_ptr(), cur.ptr,
                             cur.ld, B, out1.H, out1.W, cout // 2)
                    skip = blk.slice(0, cout // 2)
                    self._op(lib.fx_dwconv3x3s2_nhwc_bf16, out1.ptr, out1.ld, e.vec[f"{p}.pool.w"].data_ptr(), None, skip.ptr, skip.ld, B, out1.H,
                             out1.W, cout // 2)
                o1 = self.conv(cur, P[f"{p}.conv_list.1"], out=blk.slice(cout // 2, cout // 4), act="relu")
                o2 = self.conv(o1, P[f"{p}.conv_list.2"], out=blk.slice(3 * cout // 4, cout // 8), act="relu")
                self.conv(o2, P[f"{p}.conv_list.3"], out=blk.slice(7 * cout // 8, cout // 8), act="relu")
                x = blk
                idx += 1
            feats[i + 3] = x
        for k, v in feats.items():
            self.bufs[f"res{k}"] = v
        return feats

