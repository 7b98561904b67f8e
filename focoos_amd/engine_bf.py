"""BiSeNetFormer inference engine (bisenetformer-*, STDC backbone): packs a reference-layout state_dict for the gfx950
kernels and runs BisenetFormer.forward (eval) + the device side of BisenetFormerProcessor.postprocess as one hipGraph of
C-ABI calls.

Reference path being replaced (file:line in FocoosAI/focoos):
  BisenetFormer.forward                   focoos/models/bisenetformer/modelling.py:594-609
  STDC.forward / CatBottleneck            focoos/nn/backbone/stdc.py:313-320, 108-166
  BiseNet.forward_features                modelling.py:272-279 (ContextPath :186-212, AttentionRefinementModule :159-167,
                                          FeatureFusionModule :226-237)
  TransformerDecoder / PredictionHeads / head tail / processor        see focoos_amd/engine_maskdec.py

Restructured relative to the reference (same arithmetic, fewer bytes moved): eval BatchNorm folded into every conv; the four
branches of a CatBottleneck are written straight into channel slices of the block output (no torch.cat copy); the depthwise
stride-2 conv and the AvgPool2d skip are one bandwidth kernel each; the attention gates (global mean -> 1x1 conv -> sigmoid)
run on [B, C] vectors in fp32 and are applied together with the following addition in one pass; FFM's proj1(fsp) + proj2(fcp)
uses the conv kernel's residual epilogue; the [B,Q,H,W] probability / boolean tensors of the post-process are never written.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib
from .engine import DEFAULT_STREAMS, MIN_PART_BATCH, NT, PackedConv, _EngineBase, _MultiPlan, _PlanBase, _fold_bn
from .engine_maskdec import MaskDecoderPlanMixin, pack_masked_decoder, pos_embed_sine_normalized
from .engine_stdc import StdcEngineMixin, StdcPlanMixin, _bn_scale_shift

BN_EPS = 1e-5
FX_ACT_SIGMOID = 4


class BfEngine(StdcEngineMixin, _EngineBase):
    def __init__(self, config: Dict, state_dict: Dict[str, torch.Tensor], device: str = "cuda:0", full_masks: bool = False):
        super().__init__(config, device)
        bb = config["backbone_config"]
        self._init_stdc(bb)
        self.nc = int(config["num_classes"])
        self.nq = int(config.get("num_queries", 100))
        self.hd = int(config.get("transformer_predictor_hidden_dim", 256))
        self.nl = int(config.get("transformer_predictor_dec_layers", 6))
        self.nlev = min(2, self.nl)
        self.fd = int(config.get("pixel_decoder_feat_dim", 128))
        self.md = int(config.get("transformer_predictor_out_dim", 128))
        if self.hd != 256 or self.md not in (96, 128, 256) or int(config.get("pixel_decoder_out_dim", 128)) != self.md or self.fd % 32:
            raise _lib.FocoosAmdError("engine kernels are specialised for hidden 256 / mask dim 96, 128 or 256 / 8 heads (bisenetformer-*)")
        # the mask einsum kernel (fx_query_pixel_logits_bf16) takes 128 or 256 channels: a 96-wide mask dimension (bisenetformer-m-ade) runs
        # zero-padded to 128 - mask features and mask embeddings live in 128-wide buffers whose upper 32 channels are written once with zeros
        self.md_eff = 128 if self.md < 128 else self.md
        if self.nq > 128 or self.nc + 1 > 256:
            raise _lib.FocoosAmdError("engine kernels cover num_queries <= 128 and num_classes <= 255")
        self.predict_all_pixels = bool(config.get("predict_all_pixels", False))
        self.mask_threshold = float(config.get("mask_threshold", 0.5))
        self.threshold = float(config.get("threshold", 0.5))
        self.use_mask_score = bool(config.get("use_mask_score", False))
        self.cls_sigmoid = bool(config.get("cls_sigmoid", False))
        self.full_masks = bool(full_masks)
        self.masks_dtype = os.environ.get("FX_MF_MASKS_DTYPE", "fp32")   # "bf16": the [B,Q,H,W] masks tensor in 16 bits (set before the first plan is built)
        self.load_state_dict(state_dict)

    _pos_embed_sine_normalized = staticmethod(pos_embed_sine_normalized)

    # ------------------------------------------------------------------ weight packing
    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        sd = {k: v.detach().cpu() for k, v in sd.items()}
        P: Dict[str, PackedConv] = {}
        self.ln = {}
        self.vec: Dict[str, torch.Tensor] = {}   # fp32 side tables: depthwise filters, pooled 1x1 convs
        bb = "pixel_decoder.backbone"

        def cbn(name):   # ConvX / ConvBNReLU: conv + BatchNorm folded
            P[name] = self._pack(*_fold_bn(sd, f"{name}.conv.weight", f"{name}.bn"))

        self._pack_stdc(sd, P)
        pd = "pixel_decoder"
        for arm in ("arm32", "arm16"):
            p = f"{pd}.cp.{arm}"
            P[f"{p}.proj"] = self._pack(sd[f"{p}.proj.weight"].float(), None)
            cbn(f"{p}.conv")
            s, sh = _bn_scale_shift(sd, f"{p}.bn_atten")
            self.vec[f"{p}.att.w"] = self._dev((sd[f"{p}.conv_atten.weight"].double()[:, :, 0, 0] * s.view(-1, 1)).float())
            self.vec[f"{p}.att.b"] = self._dev(sh.float())
        s, sh = _bn_scale_shift(sd, f"{pd}.cp.conv_avg.bn")
        self.vec["conv_avg.w"] = self._dev((sd[f"{pd}.cp.conv_avg.conv.weight"].double()[:, :, 0, 0] * s.view(-1, 1)).float())
        self.vec["conv_avg.b"] = self._dev(sh.float())
        cbn(f"{pd}.cp.conv_head32")
        cbn(f"{pd}.cp.conv_head16")
        for name in ("proj1", "proj2"):
            P[f"{pd}.ffm.{name}"] = self._pack(sd[f"{pd}.ffm.{name}.weight"].float(), sd[f"{pd}.ffm.{name}.bias"].float())
        cbn(f"{pd}.ffm.convblk")
        self.vec["ffm.conv1.w"] = self._dev(sd[f"{pd}.ffm.conv1.weight"].float()[:, :, 0, 0])
        self.vec["ffm.conv2.w"] = self._dev(sd[f"{pd}.ffm.conv2.weight"].float()[:, :, 0, 0])
        cbn(f"{pd}.conv_out")
        pack_masked_decoder(self, sd, P, self.nlev)
        self.P = P
        self.plans.clear()

    # ------------------------------------------------------------------ run
    def plan(self, B: int, H: int, W: int, f32_input: bool = False, full_masks: Optional[bool] = None, nsplit: Optional[int] = None):
        full = self.full_masks if full_masks is None else bool(full_masks)
        if nsplit is None:
            nsplit = int(os.environ.get("FX_STREAMS", str(DEFAULT_STREAMS)))
        if nsplit > 1 and not _lib.two_queue_safe():
            nsplit = 1
        while nsplit > 1 and (B % nsplit or B // nsplit < MIN_PART_BATCH):
            nsplit -= 1
        key = (B, H, W, f32_input, full, nsplit)
        if key not in self.plans:
            self.plans[key] = (_BfPlan(self, B, H, W, f32_input, full) if nsplit <= 1
                               else _MultiPlan(self, _BfPlan, B, H, W, f32_input, nsplit, full_masks=full))
        return self.plans[key]

    def pipeline(self, B: int, H: int, W: int, depth: Optional[int] = None, f32_input: bool = False, nsplit: int = 1, full_masks: Optional[bool] = None):
        """Throughput mode: `depth` batches in flight, each on its own whole-batch plan and stream (engine._Pipeline; see DetrEngine.pipeline)."""
        from .engine import DEFAULT_PIPELINE_DEPTH, _Pipeline

        full = self.full_masks if full_masks is None else bool(full_masks)
        if depth is None:
            depth = int(os.environ.get("FX_PIPELINE_DEPTH", str(DEFAULT_PIPELINE_DEPTH)))
        if not _lib.two_queue_safe():
            depth, nsplit = 1, 1
        key = ("pipeline", B, H, W, f32_input, full, nsplit, depth)
        if key not in self.plans:
            self.plans[key] = _Pipeline(self, _BfPlan, B, H, W, f32_input, depth, nsplit, full_masks=full)
        return self.plans[key]

    def forward(self, images: torch.Tensor, threshold: Optional[float] = None, forced_attn: Optional[Sequence[torch.Tensor]] = None,
                use_graph: bool = True, full_masks: Optional[bool] = None) -> "_BfPlan":
        """images: uint8 [B,H,W,3] (fused normalise path) or float32 [B,H,W,3] (0..255 scale) on the engine device; the model runs
        at the image size (BisenetFormerProcessor.preprocess does not resize, bisenetformer/processor.py:91-93).  Returns the plan
        whose output buffers (probs, mask_probs, [masks], winner, det_*) hold the results until the next call."""
        assert images.dim() == 4 and images.shape[-1] == 3 and images.is_contiguous() and images.device == self.dev
        f32 = images.dtype == torch.float32
        assert f32 or images.dtype == torch.uint8
        B, H, W, _ = images.shape
        pl = self.plan(B, H, W, f32, full_masks, 1 if (forced_attn is not None or not use_graph) else None)
        cur = torch.cuda.current_stream(self.dev)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            pl.input.copy_(images, non_blocking=True)
            pl.run(self.stream.cuda_stream, threshold if threshold is not None else self.threshold, forced_attn, use_graph)
        cur.wait_stream(self.stream)
        return pl


class _BfPlan(StdcPlanMixin, MaskDecoderPlanMixin, _PlanBase):
    """BiSeNetFormer launch sequence for one (batch, height, width)."""

    size_multiple = 1   # the processor hands the image over at its own size (bisenetformer/processor.py:96)

    def __init__(self, eng: "BfEngine", B: int, H: int, W: int, f32_input: bool, full_masks: bool = False, parent=None, index: int = 0):
        self.full_masks = bool(full_masks)
        super().__init__(eng, B, H, W, f32_input, parent, index)

    # ---- small launch helpers
    def _fvec(self, name: str, B: int, n: int) -> torch.Tensor:
        t = torch.empty(B, n, dtype=torch.float32, device=self.dev)
        self.bufs_f32 = getattr(self, "bufs_f32", {})
        self.bufs_f32[name] = t
        return t

    def global_mean(self, x: NT, name: str) -> torch.Tensor:
        out = self._fvec(name, x.B, x.C)
        self._op(self.lib.fx_global_mean_nhwc_bf16, x.ptr, x.ld, out.data_ptr(), x.C, x.B, x.H * x.W, x.C)
        return out

    def pooled_linear(self, v: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], act: int, name: str) -> torch.Tensor:
        N, Cc = w.shape
        assert v.shape[1] == Cc
        out = self._fvec(name, v.shape[0], N)
        self._op(self.lib.fx_pooled_linear_f32, v.data_ptr(), Cc, w.data_ptr(), b.data_ptr() if b is not None else None, act, out.data_ptr(), N,
                 v.shape[0], Cc, N)
        return out

    def gate(self, x: NT, g: torch.Tensor, name: str, self_add: bool = False, add_vec: Optional[torch.Tensor] = None,
             add_map: Optional[NT] = None) -> NT:
        out = self._new(name, x.B, x.H, x.W, x.C)
        self._op(self.lib.fx_channel_gate_nhwc_bf16, x.ptr, x.ld, g.data_ptr(), x.C, int(self_add),
                 add_vec.data_ptr() if add_vec is not None else None, x.C, add_map.ptr if add_map is not None else None,
                 add_map.ld if add_map is not None else 0, out.ptr, out.ld, x.B, x.H * x.W, x.C)
        return out

    def _arm(self, x: NT, p: str, tag: str):
        """AttentionRefinementModule up to the gate: returns (feat, sigmoid attention [B, fd])."""
        e, P = self.eng, self.eng.P
        pr = self.conv(x, P[f"{p}.proj"], name=f"{tag}.proj")
        feat = self.conv(pr, P[f"{p}.conv"], name=f"{tag}.feat", act="relu")
        att = self.pooled_linear(self.global_mean(feat, f"{tag}.mean"), e.vec[f"{p}.att.w"], e.vec[f"{p}.att.b"], 4, f"{tag}.att")
        return feat, att

    def _build(self):
        e, P, B = self.eng, self.eng.P, self.B
        feats = self.build_stdc()
        pd = "pixel_decoder"
        fd = e.fd
        # ---- ContextPath (modelling.py:186-212)
        avg = self.pooled_linear(self.global_mean(feats[5], "cp.mean5"), e.vec["conv_avg.w"], e.vec["conv_avg.b"], 1, "cp.avg")
        f32, a32 = self._arm(feats[5], f"{pd}.cp.arm32", "arm32")
        cp32 = self.gate(f32, a32, "cp32", add_vec=avg)
        up = self._new("cp.up32", B, feats[4].H, feats[4].W, fd)
        self.resize(cp32, up)
        up32 = self.conv(up, P[f"{pd}.cp.conv_head32"], name="cp.head32", act="relu")
        f16, a16 = self._arm(feats[4], f"{pd}.cp.arm16", "arm16")
        cp16 = self.gate(f16, a16, "cp16", add_map=up32)
        up = self._new("cp.up16", B, feats[3].H, feats[3].W, fd)
        self.resize(cp16, up)
        cp8 = self.conv(up, P[f"{pd}.cp.conv_head16"], name="cp8", act="relu")
        # ---- FeatureFusionModule (:226-237) + conv_out (:277)
        s1 = self.conv(feats[3], P[f"{pd}.ffm.proj1"], name="ffm.proj1")
        s2 = self.conv(cp8, P[f"{pd}.ffm.proj2"], name="ffm.sum", residual=s1)
        feat = self.conv(s2, P[f"{pd}.ffm.convblk"], name="ffm.feat", act="relu")
        a1 = self.pooled_linear(self.global_mean(feat, "ffm.mean"), e.vec["ffm.conv1.w"], None, 1, "ffm.a1")
        a2 = self.pooled_linear(a1, e.vec["ffm.conv2.w"], None, 4, "ffm.a2")
        fuse = self.gate(feat, a2, "ffm", self_add=True)
        if e.md_eff == e.md:
            mf = self.conv(fuse, P[f"{pd}.conv_out"], name="mask_features", act="relu")
        else:
            mf = self._new("mask_features.padded", B, fuse.H, fuse.W, e.md_eff)
            mf.t.zero_()
            self.conv(fuse, P[f"{pd}.conv_out"], out=mf.slice(0, e.md), act="relu")
            self.bufs["mask_features"] = mf.slice(0, e.md)
        # ---- masked-attention decoder over (cp32, cp16), heads, outputs and post-process (engine_maskdec.py)
        dn, emb = self.build_masked_decoder([cp32, cp16], mf, e.md_eff)
        self.build_mask_outputs(dn, emb, mf, e.md_eff, self.full_masks, predict_all_pixels=e.predict_all_pixels)
