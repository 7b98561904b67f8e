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


class MfEngine(_EngineBase):
    def __init__(self, config: Dict, state_dict: Dict[str, torch.Tensor], device: str = "cuda:0", full_masks: bool = False):
        super().__init__(config, device)
        self.nc = int(config["num_classes"])
        self.nq = int(config.get("num_queries", 100))
        self.hd = int(config.get("transformer_predictor_hidden_dim", 256))
        self.nl = int(config.get("transformer_predictor_dec_layers", 6))
        self.n_enc = int(config.get("pixel_decoder_transformer_layers", 0))
        self.nlev = min(3, self.nl)
        dims = [self.hd, int(config.get("pixel_decoder_feat_dim", 256)), int(config.get("pixel_decoder_out_dim", 256)),
                int(config.get("transformer_predictor_out_dim", 256))]
        if any(d != 256 for d in dims) or int(config.get("pixel_decoder_transformer_nheads", 8)) != 8:
            raise _lib.FocoosAmdError("engine kernels are specialised for 256 channels / 8 heads (fai-mf-l)")
        if self.nq > 128 or self.nc + 1 > 256:
            raise _lib.FocoosAmdError("engine kernels cover num_queries <= 128 and num_classes <= 255")
        if config.get("postprocessing_type", "instance") != "instance" or config.get("predict_all_pixels", False):
            raise _lib.FocoosAmdError("engine covers the instance post-processing branch (predict_all_pixels=False)")
        self.mask_threshold = float(config.get("mask_threshold", 0.5))
        self.threshold = float(config.get("threshold", 0.5))
        self.use_mask_score = bool(config.get("use_mask_score", False))
        self.cls_sigmoid = bool(config.get("cls_sigmoid", False))
        self.full_masks = bool(full_masks)
        self.load_state_dict(state_dict)

    # ------------------------------------------------------------------ weight packing
    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        sd = {k: v.detach().cpu() for k, v in sd.items()}
        P: Dict[str, PackedConv] = {}
        self.ln = {}
        self._pack_backbone(sd, P)
        pd = "pixel_decoder"

        def lin(key, wkey, rows=None):
            W, b = sd[f"{wkey}.weight"], sd[f"{wkey}.bias"]
            if rows is not None:
                W, b = W[rows], b[rows]
            P[key] = self._pack_linear(W, b)

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
                attn(p, "self_attn", False)
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
        hp = "head.predictor"
        kw: List[List[torch.Tensor]] = [[] for _ in range(self.nlev)]
        kb: List[List[torch.Tensor]] = [[] for _ in range(self.nlev)]
        vw: List[List[torch.Tensor]] = [[] for _ in range(self.nlev)]
        vb: List[List[torch.Tensor]] = [[] for _ in range(self.nlev)]
        for li in range(self.nl):
            p = f"{hp}.transformer_cross_attention_layers.{li}"
            Wi, bi = attn(p, "multihead_attn", True)
            lvl = li % self.nlev
            kw[lvl].append(Wi[256:512]); kb[lvl].append(bi[256:512])
            vw[lvl].append(Wi[512:]); vb[lvl].append(bi[512:])
            self._pack_ln(sd, f"{p}.norm")
            p = f"{hp}.transformer_self_attention_layers.{li}"
            attn(p, "self_attn", False)
            self._pack_ln(sd, f"{p}.norm")
            p = f"{hp}.transformer_ffn_layers.{li}"
            lin(f"{p}.linear1", f"{p}.linear1")
            lin(f"{p}.linear2", f"{p}.linear2")
            self._pack_ln(sd, f"{p}.norm")
        for lvl in range(self.nlev):
            # the layers attending level lvl share their memory: all their key (value) projections as ONE GEMM
            P[f"{hp}.k_all.{lvl}"] = self._pack_linear(torch.cat(kw[lvl], 0), torch.cat(kb[lvl], 0))
            P[f"{hp}.v_all.{lvl}"] = self._pack_linear(torch.cat(vw[lvl], 0), torch.cat(vb[lvl], 0))
            P[f"{hp}.input_proj.{lvl}"] = self._pack(sd[f"{hp}.input_proj.{lvl}.weight"].float(), sd[f"{hp}.input_proj.{lvl}.bias"].float())
        self.query_feat = self._dev(sd[f"{hp}.query_feat.weight"].float(), torch.bfloat16)
        self.query_embed = self._dev(sd[f"{hp}.query_embed.weight"].float(), torch.bfloat16)
        ph = f"{hp}.forward_prediction_heads"
        self._pack_ln(sd, f"{ph}.decoder_norm")
        lin(f"{ph}.classifier", f"{ph}.classifier")
        for j in range(3):
            lin(f"{ph}.mask_classifier.{j}", f"{ph}.mask_classifier.layers.{j}")
        self.P = P
        self.plans.clear()

    @staticmethod
    def _pos_embed_sine_normalized(h: int, w: int, npf: int, temperature: float = 10000.0, scale: float = 2 * math.pi,
                                   eps: float = 1e-6) -> torch.Tensor:
        """PositionEmbeddingSine(normalize=True) (nn/layers/position_encoding.py:52-81), token-major [h*w, 2*npf]:
        embed = (index+1)/(size+eps)*2pi, sin/cos interleaved per channel pair, [y half | x half]."""
        ys = (torch.arange(1, h + 1, dtype=torch.float32) / (h + eps) * scale).view(h, 1).expand(h, w)
        xs = (torch.arange(1, w + 1, dtype=torch.float32) / (w + eps) * scale).view(1, w).expand(h, w)
        i = torch.arange(npf, dtype=torch.float32)
        dim_t = temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / npf)
        px, py = xs[..., None] / dim_t, ys[..., None] / dim_t
        px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=-1).flatten(-2)
        py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=-1).flatten(-2)
        return torch.cat([py, px], dim=-1).reshape(h * w, 2 * npf)

    # ------------------------------------------------------------------ run
    def plan(self, B: int, H: int, W: int, f32_input: bool = False, full_masks: Optional[bool] = None, nsplit: Optional[int] = None):
        full = self.full_masks if full_masks is None else bool(full_masks)
        if nsplit is None:
            nsplit = int(os.environ.get("FX_STREAMS", str(DEFAULT_STREAMS)))
        while nsplit > 1 and (B % nsplit or B // nsplit < MIN_PART_BATCH):
            nsplit -= 1
        key = (B, H, W, f32_input, full, nsplit)
        if key not in self.plans:
            self.plans[key] = (_MfPlan(self, B, H, W, f32_input, full) if nsplit <= 1
                               else _MultiPlan(self, _MfPlan, B, H, W, f32_input, nsplit, full_masks=full))
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


def pack_mask_bits(mask: torch.Tensor, words: int) -> torch.Tensor:
    """bool [R, L] (True = key not allowed) -> int32 [R, words], bit (key & 31) of word key/32; padding keys masked."""
    R, L = mask.shape
    m = np.ones((R, words * 32), dtype=np.uint8)
    m[:, :L] = mask.cpu().numpy().astype(np.uint8)
    packed = np.packbits(m, axis=-1, bitorder="little")  # [R, words*4] bytes, little-endian words
    return torch.from_numpy(np.ascontiguousarray(packed).view("<u4").view(np.int32).reshape(R, words).copy())


class _MfPlan(_PlanBase):
    """MaskFormer launch sequence for one (batch, height, width)."""

    def __init__(self, eng: "MfEngine", B: int, H: int, W: int, f32_input: bool, full_masks: bool = False, parent=None, index: int = 0):
        self.full_masks = bool(full_masks)
        super().__init__(eng, B, H, W, f32_input, parent, index)

    def _build(self):
        e, P, B, lib = self.eng, self.eng.P, self.B, self.lib
        H, W = self.H, self.W
        Q, K = e.nq, e.nc
        feats = self.build_backbone()
        pd = "pixel_decoder"
        h32, w32 = H // 32, W // 32
        # ---- pixel decoder (fai_mf/modelling.py:347-369)
        x5 = feats[5]
        if e.n_enc > 0:
            src = self.conv(feats[5], P[f"{pd}.input_proj"], name="pd.proj5")
            L5 = h32 * w32
            pos = e._pos_embed_sine_normalized(h32, w32, 128).to(device=self.dev, dtype=torch.bfloat16).contiguous()
            self.pos5 = NT(pos, L5, 1, 1, 256, 256)
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
            x5 = NT(s.t, B, h32, w32, 256, 256, s.off)
        y = self.conv(x5, P[f"{pd}.layer_4"], name="msf0", act="relu")
        msf = [y]
        for idx, f in ((3, feats[4]), (2, feats[3]), (1, feats[2])):
            cur = self.conv(f, P[f"{pd}.adapter_{idx}"], name=f"fpn.lat{idx}")
            ysum = self._new(f"fpn.sum{idx}", B, f.H, f.W, 256)
            self._op(lib.fx_upsample_nearest_add_nhwc_bf16, cur.ptr, cur.ld, y.ptr, y.ld, ysum.ptr, ysum.ld, B, f.H, f.W, y.H, y.W, 256)
            y = self.conv(ysum, P[f"{pd}.layer_{idx}"], name=f"msf{4 - idx}" if len(msf) < 3 else "fpn_s4", act="relu")
            if len(msf) < 3:
                msf.append(y)
        mf = self.conv(y, P[f"{pd}.mask_features"], name="mask_features")
        h4, w4 = mf.H, mf.W
        # ---- masked-attention decoder (fai_mf/modelling.py:453-549)
        hp = "head.predictor"
        nlev = e.nlev
        Ls, k_all, v_all, mfp, W32 = [], [], [], [], []
        for l in range(nlev):
            f = msf[l]
            L = f.H * f.W
            Ls.append(L)
            W32.append((L + 31) // 32)
            src_l = self.conv(f, P[f"{hp}.input_proj.{l}"], name=f"dec.src{l}").as_rows()
            pos = e._pos_embed_sine_normalized(f.H, f.W, 128).to(device=self.dev, dtype=torch.bfloat16).contiguous()
            pos_nt = NT(pos, L, 1, 1, 256, 256)
            self.keep.append(pos)
            srcpos = self.add_rows(src_l, pos_nt, L, f"dec.srcpos{l}")
            k_all.append(self.linear(srcpos, P[f"{hp}.k_all.{l}"], name=f"dec.k_all{l}"))
            v_all.append(self.linear(src_l, P[f"{hp}.v_all.{l}"], name=f"dec.v_all{l}"))
            # attention-mask source: the mask features bilinearly resized to this level (commutes with the mask einsum)
            m = self._new(f"dec.mfp{l}", B, f.H, f.W, 256)
            self.resize(mf, m)
            mfp.append(m)
        R = B * Q
        qe = NT(e.query_embed, Q, 1, 1, 256, 256)
        out0 = e.query_feat.repeat(B, 1).contiguous()
        self.keep.append(out0)
        out = NT(out0, R, 1, 1, 256, 256)
        ph = f"{hp}.forward_prediction_heads"
        self.attn_bits: List[torch.Tensor] = []
        self.force_points: List[int] = []

        def heads(x: NT, idx: int, level: Optional[int]):
            dn = self.layernorm(x, f"{ph}.decoder_norm", f"ph{idx}.dn")
            m1 = self.linear(dn, P[f"{ph}.mask_classifier.0"], name=f"ph{idx}.m1", act="relu")
            m2 = self.linear(m1, P[f"{ph}.mask_classifier.1"], name=f"ph{idx}.m2", act="relu")
            emb = self.linear(m2, P[f"{ph}.mask_classifier.2"], name=f"ph{idx}.emb")
            if level is not None:
                bits = torch.zeros(R, W32[level], dtype=torch.int32, device=self.dev)
                self._op(lib.fx_query_pixel_logits_bf16, emb.ptr, emb.ld, mfp[level].ptr, mfp[level].ld, 2, None, 0, bits.data_ptr(),
                         W32[level], B, Q, Ls[level], 256)
                self.attn_bits.append(bits)
                self.force_points.append(len(self.ops))
            return dn, emb

        heads(out, 0, 0)
        dn = emb = None
        # one workspace for the key-sliced cross attention (launches are serial on one stream)
        mha_ws = torch.empty(max(8, max(lib.fx_mha_workspace_bytes(B, Q, L, 8, 1) for L in Ls)), dtype=torch.uint8, device=self.dev)
        self.keep.append(mha_ws)
        for i in range(e.nl):
            lvl, j = i % nlev, i // nlev
            p = f"{hp}.transformer_cross_attention_layers.{i}"
            t2 = self.layernorm(out, f"{p}.norm", f"dec{i}.c_n")
            qin = self.add_rows(t2, qe, Q, f"dec{i}.c_qin")
            qc = self.linear(qin, P[f"{p}.q"], name=f"dec{i}.c_q")
            att = self._new(f"dec{i}.c_att", R, 1, 1, 256)
            ks, vs = k_all[lvl].slice(j * 256, 256), v_all[lvl].slice(j * 256, 256)
            self._op(lib.fx_mha_masked_bf16, qc.ptr, qc.ld, ks.ptr, ks.ld, vs.ptr, vs.ld, att.ptr, att.ld, B, Q, Ls[lvl], 8,
                     self.attn_bits[i].data_ptr(), W32[lvl], mha_ws.data_ptr(), C.c_size_t(mha_ws.numel()))
            out = self.linear(att, P[f"{p}.out_proj"], name=f"dec{i}.c_o", residual=out)
            p = f"{hp}.transformer_self_attention_layers.{i}"
            t2 = self.layernorm(out, f"{p}.norm", f"dec{i}.s_n")
            qk_in = self.add_rows(t2, qe, Q, f"dec{i}.s_qk_in")
            qkv = self._new(f"dec{i}.s_qkv", R, 1, 1, 768)
            self.linear(qk_in, P[f"{p}.qk"], out=qkv.slice(0, 512))
            self.linear(t2, P[f"{p}.v"], out=qkv.slice(512, 256))
            att = self.mha(qkv, B, Q, f"dec{i}.s_att")
            out = self.linear(att, P[f"{p}.out_proj"], name=f"dec{i}.s_o", residual=out)
            p = f"{hp}.transformer_ffn_layers.{i}"
            t2 = self.layernorm(out, f"{p}.norm", f"dec{i}.f_n")
            f1 = self.linear(t2, P[f"{p}.linear1"], name=f"dec{i}.f1", act="relu")
            out = self.linear(f1, P[f"{p}.linear2"], name=f"dec{i}.out", residual=out)
            dn, emb = heads(out, i + 1, (i + 1) % nlev if i < e.nl - 1 else None)
        # ---- outputs (MaskFormerHead.forward :599-617, FAIMaskFormer.forward :720-725)
        cls_logits = self.linear(dn, P[f"{ph}.classifier"], name="cls_logits", out_f32=True)
        self.probs = self._io("probs", (B, Q, K), torch.float32)
        self.cls_score = self._io("cls_score", (B, Q), torch.float32)
        self.cls_label = self._io("cls_label", (B, Q), torch.int32)
        self._op(lib.fx_mf_class_head, cls_logits.ptr, cls_logits.ld, self.probs.data_ptr(), self.cls_score.data_ptr(), self.cls_label.data_ptr(),
                 R, K, int(e.cls_sigmoid))
        P4 = h4 * w4
        self.mask_probs = self._io("mask_probs", (B, Q, h4, w4), torch.float32)  # sigmoid(mask logits) at 1/4 resolution
        mf_rows = mf.as_rows()
        self._op(lib.fx_query_pixel_logits_bf16, emb.ptr, emb.ld, mf_rows.ptr, mf_rows.ld, 1, self.mask_probs.data_ptr(), P4, None, 0, B, Q, P4, 256)
        self.masks = None
        if self.full_masks:
            self.masks = self._io("masks", (B, Q, H, W), torch.float32)
            self._op(lib.fx_mf_upsample_probs_f32, self.mask_probs.data_ptr(), h4, w4, self.masks.data_ptr(), H, W, R)
        # ---- device side of MaskFormerProcessor.postprocess (processor.py:212-262)
        ws_bytes = lib.fx_mf_postprocess_workspace_bytes(B, Q, H)
        self.post_ws = torch.empty(max(ws_bytes, 8), dtype=torch.uint8, device=self.dev)
        self.det_count = self._io("det_count", (B,), torch.int32).zero_()
        self.det_query = self._io("det_query", (B, Q), torch.int32).zero_()
        self.det_scores = self._io("det_scores", (B, Q), torch.float32).zero_()
        self.det_labels = self._io("det_labels", (B, Q), torch.int32).zero_()
        self.det_boxes = self._io("det_boxes", (B, Q, 4), torch.int32).zero_()
        self.det_area = self._io("det_area", (B, Q), torch.int32).zero_()
        self.mask_words = self._io("mask_words", (B, Q, H, W // 32), torch.int32).zero_()
        self.post_index = len(self.ops)
        self._op(lib.fx_mf_postprocess, self.mask_probs.data_ptr(), h4, w4, H, W, self.cls_score.data_ptr(), self.cls_label.data_ptr(), B, Q,
                 C.c_float(e.mask_threshold), None, int(e.use_mask_score), self.post_ws.data_ptr(), C.c_size_t(self.post_ws.numel()),
                 self.det_count.data_ptr(), self.det_query.data_ptr(), self.det_scores.data_ptr(), self.det_labels.data_ptr(),
                 self.det_boxes.data_ptr(), self.det_area.data_ptr(), self.mask_words.data_ptr())
        self.levels = Ls
        self.W32 = W32

    # -------------------------------------------------------------- execution
    def patch_args(self, fn, args, thr: float):
        if fn is self.lib.fx_mf_postprocess:
            return args[:10] + (C.c_float(thr),) + args[11:]
        return args

    def run(self, stream: int, thr: float, forced_attn: Optional[Sequence[torch.Tensor]] = None, use_graph: bool = True):
        if forced_attn is not None:
            # teacher-forced attention masks (parity tests): overwrite each layer's bitmap right after it is produced
            assert len(forced_attn) == len(self.force_points)
            prev = 0
            for i, (pt, m) in enumerate(zip(self.force_points, forced_attn)):
                self._launch(self.ops[prev:pt], stream, thr)
                bits = pack_mask_bits(m.reshape(self.B * self.eng.nq, -1), self.attn_bits[i].shape[1])
                self.attn_bits[i].copy_(bits.to(self.dev))
                prev = pt
            self._launch(self.ops[prev:], stream, thr)
            return
        if not use_graph:
            self._launch(self.ops, stream, thr)
            return
        self.capture_and_launch(stream, thr)
