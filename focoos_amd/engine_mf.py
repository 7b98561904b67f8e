"""MaskFormer inference engine (fai-mf-*, ResNet-vd backbone): packs a reference-layout state_dict for the gfx950
kernels and runs FAIMaskFormer.forward (eval) + the device side of MaskFormerProcessor.postprocess as one hipGraph of
C-ABI calls.

Reference path being replaced (file:line in FocoosAI/focoos):
  FAIMaskFormer.forward                   focoos/models/fai_mf/modelling.py:712-725
  TransformerFPN.forward_features         focoos/models/fai_mf/modelling.py:347-369 (+ TransformerEncoderOnly :177-198,
                                          pre-norm TransformerEncoderLayer focoos/nn/layers/transformer.py:583-601)
  MultiScaleMaskedTransformerDecoder      focoos/models/fai_mf/modelling.py:453-549 (layers transformer.py:83-106,206-238,365-380)
  PredictionHeads.forward                 focoos/models/fai_mf/modelling.py:71-113
  MaskFormerHead.forward tail             focoos/models/fai_mf/modelling.py:599-617
  MaskFormerProcessor.postprocess         focoos/models/fai_mf/processor.py:168-306 (device part: :212-262)

What is restructured relative to the reference (same arithmetic, fewer bytes moved):
  * eval BatchNorm folded into the FPN convs; K/V projections of the three decoder layers that attend the same level are
    one N=768 GEMM each;
  * the boolean attention mask `interpolate(mask_embed x mask_features) < 0` is computed as
    `mask_embed x interpolate(mask_features)` (the bilinear resize is linear and per-channel, so it commutes with the
    einsum) straight into a bitmap: the [B,Q,H/4,W/4] logits of the 9 intermediate prediction heads are never written;
  * the [B,Q,H,W] fp32 `masks` output (256 MB/img at 800^2) is optional: the post-process consumes the quarter-resolution
    probabilities and upsamples on the fly.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import check
from .engine import DEFAULT_STREAMS, MIN_PART_BATCH, NT, PackedConv, _EngineBase, _MultiPlan, _PlanBase, _fold_bn
from .engine_stdc import StdcEngineMixin, StdcPlanMixin
from .engine_maskdec import MaskDecoderPlanMixin, pack_mask_bits, pack_masked_decoder, pos_embed_sine_normalized  # noqa: F401  (pack_mask_bits re-exported)


class MfEngine(StdcEngineMixin, _EngineBase):
    def __init__(self, config: Dict, state_dict: Dict[str, torch.Tensor], device: str = "cuda:0", full_masks: bool = False):
        super().__init__(config, device)
        # backbone: ResNet-vd (fai-mf-l-*) or STDC (fai-mf-m-ade; engine_stdc.py)
        self.stdc = config["backbone_config"].get("model_type") == "stdc"
        if self.stdc:
            self._init_stdc(config["backbone_config"])
        self.nc = int(config["num_classes"])
        self.nq = int(config.get("num_queries", 100))
        self.hd = int(config.get("transformer_predictor_hidden_dim", 256))
        self.nl = int(config.get("transformer_predictor_dec_layers", 6))
        self.n_enc = int(config.get("pixel_decoder_transformer_layers", 0))
        self.nlev = min(3, self.nl)
        # pixel-decoder width fd = mask-feature / mask-embedding width md: 256 (fai-mf-l-coco-ins) or 128 (fai-mf-{l,m}-ade,
        # fai-mf-{m,s}-coco-ins); the decoder's hidden width is 256 in every registry model.  The pixel decoder's own transformer encoder
        # runs at fd channels with 8 heads: head dim 32 at fd = 256; at fd = 128 (head dim 16) the heads are ZERO-PADDED to 32 channels
        # in the packed projection weights (load_state_dict) and run on the same attention kernel.
        self.fd = int(config.get("pixel_decoder_feat_dim", 256))
        self.md = int(config.get("transformer_predictor_out_dim", 256))
        if (self.hd != 256 or self.fd not in (128, 256) or self.md != self.fd or int(config.get("pixel_decoder_out_dim", 256)) != self.fd
                or int(config.get("pixel_decoder_transformer_nheads", 8)) != 8):
            raise _lib.FocoosAmdError("engine kernels cover hidden 256 / 8 heads with pixel-decoder = mask width 256 or 128 (fai-mf-*)")
        if self.nq > 128 or self.nc + 1 > 256:
            raise _lib.FocoosAmdError("engine kernels cover num_queries <= 128 and num_classes <= 255")
        # post-processing: threshold branch (instance, fx_mf_postprocess) or per-pixel argmax (predict_all_pixels, fx_seg_postprocess)
        self.predict_all_pixels = bool(config.get("predict_all_pixels", False))
        self.mask_threshold = float(config.get("mask_threshold", 0.5))
        self.threshold = float(config.get("threshold", 0.5))
        self.use_mask_score = bool(config.get("use_mask_score", False))
        self.cls_sigmoid = bool(config.get("cls_sigmoid", False))
        self.full_masks = bool(full_masks)
        self.masks_dtype = os.environ.get("FX_MF_MASKS_DTYPE", "fp32")   # "bf16": the [B,Q,H,W] masks tensor in 16 bits (set before the first plan is built)
        self.load_state_dict(state_dict)

    # ------------------------------------------------------------------ weight packing
    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        sd = {k: v.detach().cpu() for k, v in sd.items()}
        P: Dict[str, PackedConv] = {}
        self.ln = {}
        if self.stdc:
            self._pack_stdc(sd, P)
        else:
            self._pack_backbone(sd, P)
        pd = "pixel_decoder"

        def lin(key, wkey, rows=None):
            W, b = sd[f"{wkey}.weight"], sd[f"{wkey}.bias"]
            if rows is not None:
                W, b = W[rows], b[rows]
            P[key] = self._pack_linear(W, b)

        def attn_padded_heads(prefix, name):
            """nn.MultiheadAttention at 128 channels / 8 heads (head dim 16) on the head-dim-32 attention kernel: every head's q / k / v get 16
            zero channels (zero weight rows, zero bias) - scores and outputs are unchanged, the extra output channels are zero and meet
            zero columns of out_proj.  The kernel scales by 1/sqrt(32); the reference by 1/sqrt(16): the q rows carry the sqrt(2)."""
            Wi, bi = sd[f"{prefix}.{name}.in_proj_weight"].float(), sd[f"{prefix}.{name}.in_proj_bias"].float()
            c = Wi.shape[1]

            def pad_rows(W, b, scale=1.0):
                Wp, bp = torch.zeros(256, c), torch.zeros(256)
                for h in range(8):
                    Wp[32 * h:32 * h + 16] = W[16 * h:16 * h + 16] * scale
                    bp[32 * h:32 * h + 16] = b[16 * h:16 * h + 16] * scale
                return Wp, bp

            wq, bq = pad_rows(Wi[:c], bi[:c], math.sqrt(2.0))
            wk, bk = pad_rows(Wi[c:2 * c], bi[c:2 * c])
            wv, bv = pad_rows(Wi[2 * c:], bi[2 * c:])
            P[f"{prefix}.qk"] = self._pack_linear(torch.cat([wq, wk], 0), torch.cat([bq, bk], 0))
            P[f"{prefix}.v"] = self._pack_linear(wv, bv)
            Wo, bo = sd[f"{prefix}.{name}.out_proj.weight"].float(), sd[f"{prefix}.{name}.out_proj.bias"].float()
            Wop = torch.zeros(c, 256)
            for h in range(8):
                Wop[:, 32 * h:32 * h + 16] = Wo[:, 16 * h:16 * h + 16]
            P[f"{prefix}.out_proj"] = self._pack_linear(Wop, bo)

        def attn(prefix, name, split_q: bool):
            Wi, bi = sd[f"{prefix}.{name}.in_proj_weight"], sd[f"{prefix}.{name}.in_proj_bias"]
            if split_q:
                P[f"{prefix}.q"] = self._pack_linear(Wi[:256], bi[:256])
            else:
                P[f"{prefix}.qk"] = self._pack_linear(Wi[:512], bi[:512])
                P[f"{prefix}.v"] = self._pack_linear(Wi[512:], bi[512:])
            lin(f"{prefix}.out_proj", f"{prefix}.{name}.out_proj")
            return Wi, bi

        if self.n_enc > 0:
            P[f"{pd}.input_proj"] = self._pack(sd[f"{pd}.input_proj.weight"].float(), sd[f"{pd}.input_proj.bias"].float())
            for li in range(self.n_enc):
                p = f"{pd}.transformer.encoder.layers.{li}"
                if self.fd == 256:
                    attn(p, "self_attn", False)
                else:
                    attn_padded_heads(p, "self_attn")
                lin(f"{p}.linear1", f"{p}.linear1")
                lin(f"{p}.linear2", f"{p}.linear2")
                self._pack_ln(sd, f"{p}.norm1")
                self._pack_ln(sd, f"{p}.norm2")
            self._pack_ln(sd, f"{pd}.transformer.encoder.norm")
        for idx in (1, 2, 3, 4):
            if idx < 4:
                P[f"{pd}.adapter_{idx}"] = self._pack(*_fold_bn(sd, f"{pd}.adapter_{idx}.weight", f"{pd}.adapter_{idx}.norm"))
            P[f"{pd}.layer_{idx}"] = self._pack(*_fold_bn(sd, f"{pd}.layer_{idx}.weight", f"{pd}.layer_{idx}.norm"))
        P[f"{pd}.mask_features"] = self._pack(sd[f"{pd}.mask_features.weight"].float(), sd[f"{pd}.mask_features.bias"].float())
        pack_masked_decoder(self, sd, P, self.nlev)
        self.P = P
        self.plans.clear()

    _pos_embed_sine_normalized = staticmethod(pos_embed_sine_normalized)

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
            self.plans[key] = (_MfPlan(self, B, H, W, f32_input, full) if nsplit <= 1
                               else _MultiPlan(self, _MfPlan, B, H, W, f32_input, nsplit, full_masks=full))
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
            self.plans[key] = _Pipeline(self, _MfPlan, B, H, W, f32_input, depth, nsplit, full_masks=full)
        return self.plans[key]

    def forward(self, images: torch.Tensor, threshold: Optional[float] = None, forced_attn: Optional[Sequence[torch.Tensor]] = None,
                use_graph: bool = True, full_masks: Optional[bool] = None) -> "_MfPlan":
        """images: uint8 [B,H,W,3] (fused normalise path) or float32 [B,H,W,3] (0..255 scale) on the engine device; the
        model runs at the image size (MaskFormerProcessor.preprocess does not resize, fai_mf/processor.py:96).  Returns the
        plan whose output buffers (probs, mask_probs, [masks], det_*) hold the results until the next call."""
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


class _MfPlan(StdcPlanMixin, MaskDecoderPlanMixin, _PlanBase):
    """MaskFormer launch sequence for one (batch, height, width)."""

    size_multiple = 1   # MaskFormerProcessor.preprocess hands the image over at its own size (fai_mf/processor.py:96)

    def __init__(self, eng: "MfEngine", B: int, H: int, W: int, f32_input: bool, full_masks: bool = False, parent=None, index: int = 0):
        self.full_masks = bool(full_masks)
        super().__init__(eng, B, H, W, f32_input, parent, index)

    def _build(self):
        e, P, B, lib = self.eng, self.eng.P, self.B, self.lib
        H, W = self.H, self.W
        Q, K = e.nq, e.nc
        feats = self.build_stdc() if e.stdc else self.build_backbone()
        pd = "pixel_decoder"
        h32, w32 = feats[5].H, feats[5].W
        # ---- pixel decoder (fai_mf/modelling.py:347-369)
        x5 = feats[5]
        if e.n_enc > 0:
            src = self.conv(feats[5], P[f"{pd}.input_proj"], name="pd.proj5")
            L5 = h32 * w32
            pos = e._pos_embed_sine_normalized(h32, w32, e.fd // 2).to(device=self.dev, dtype=torch.bfloat16).contiguous()
            self.pos5 = NT(pos, L5, 1, 1, e.fd, e.fd)
            s = src.as_rows()
            for li in range(e.n_enc):
                p = f"{pd}.transformer.encoder.layers.{li}"
                s2 = self.layernorm(s, f"{p}.norm1", f"enc{li}.n1")
                qk_in = self.add_rows(s2, self.pos5, L5, f"enc{li}.qk_in")
                qkv = self._new(f"enc{li}.qkv", B * L5, 1, 1, 768)
                self.linear(qk_in, P[f"{p}.qk"], out=qkv.slice(0, 512))
                self.linear(s2, P[f"{p}.v"], out=qkv.slice(512, 256))
                att = self.mha(qkv, B, L5, f"enc{li}.att")
                o = self.linear(att, P[f"{p}.out_proj"], name=f"enc{li}.o", residual=s)
                s2 = self.layernorm(o, f"{p}.norm2", f"enc{li}.n2")
                f1 = self.linear(s2, P[f"{p}.linear1"], name=f"enc{li}.f1", act="relu")
                s = self.linear(f1, P[f"{p}.linear2"], name=f"enc{li}.f2", residual=o)
            s = self.layernorm(s, f"{pd}.transformer.encoder.norm", "enc_tokens")
            x5 = NT(s.t, B, h32, w32, e.fd, e.fd, s.off)
        y = self.conv(x5, P[f"{pd}.layer_4"], name="msf0", act="relu")
        msf = [y]
        for idx, f in ((3, feats[4]), (2, feats[3]), (1, feats[2])):
            cur = self.conv(f, P[f"{pd}.adapter_{idx}"], name=f"fpn.lat{idx}")
            ysum = self._new(f"fpn.sum{idx}", B, f.H, f.W, e.fd)
            self._op(lib.fx_upsample_nearest_add_nhwc_bf16, cur.ptr, cur.ld, y.ptr, y.ld, ysum.ptr, ysum.ld, B, f.H, f.W, y.H, y.W, e.fd)
            y = self.conv(ysum, P[f"{pd}.layer_{idx}"], name=f"msf{4 - idx}" if len(msf) < 3 else "fpn_s4", act="relu")
            if len(msf) < 3:
                msf.append(y)
        mf = self.conv(y, P[f"{pd}.mask_features"], name="mask_features")
        # ---- masked-attention decoder, heads, outputs and post-process (engine_maskdec.py)
        dn, emb = self.build_masked_decoder(msf, mf, e.md)
        self.build_mask_outputs(dn, emb, mf, e.md, self.full_masks, predict_all_pixels=e.predict_all_pixels)

