"""The masked-attention query decoder shared by the MaskFormer (fai-mf-*) and BiSeNetFormer (bisenetformer-*) engines:
weight packing and the launch sequence of

  MultiScaleMaskedTransformerDecoder.forward     focoos/models/fai_mf/modelling.py:453-549          (3 memory levels)
  TransformerDecoder.forward                     focoos/models/bisenetformer/modelling.py:375-447   (2 memory levels)
  PredictionHeads.forward                        fai_mf/modelling.py:71-113 == bisenetformer/modelling.py:68-113
  MaskFormerHead.forward tail                    fai_mf/modelling.py:599-617 == bisenetformer/modelling.py:497-510
  {MaskFormer,BisenetFormer}Processor.postprocess (device part)  fai_mf/processor.py:212-262 == bisenetformer/processor.py:212-262

(the two reference files are the same code up to the number of levels and the channel count of the mask embedding).
Layer order: cross-attention (masked), self-attention, FFN, all pre-norm; layers cycle over the levels; K/V projections of the
layers that attend the same level are one GEMM; the boolean attention mask `interpolate(mask_embed x mask_features) < 0` is
computed as `mask_embed x interpolate(mask_features)` straight into a bitmap."""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from ._lib import FX_ACT, FxRcStage
from .engine import NT, PackedConv

HP = "head.predictor"
PH = f"{HP}.forward_prediction_heads"


def pack_mask_bits(mask: torch.Tensor, words: int) -> torch.Tensor:
    """bool [R, L] (True = key not allowed) -> int32 [R, words], bit (key & 31) of word key/32; padding keys masked."""
    R, L = mask.shape
    m = np.ones((R, words * 32), dtype=np.uint8)
    m[:, :L] = mask.cpu().numpy().astype(np.uint8)
    packed = np.packbits(m, axis=-1, bitorder="little")  # [R, words*4] bytes, little-endian words
    return torch.from_numpy(np.ascontiguousarray(packed).view("<u4").view(np.int32).reshape(R, words).copy())


def pos_embed_sine_normalized(h: int, w: int, npf: int, temperature: float = 10000.0, scale: float = 2 * math.pi, eps: float = 1e-6) -> torch.Tensor:
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


def pack_masked_decoder(eng, sd: Dict[str, torch.Tensor], P: Dict[str, PackedConv], nlev: int) -> None:
    """Decoder + prediction-head weights of ``eng`` (attributes nl, hd=256) into P / eng.ln / eng.query_*."""

    def lin(key, wkey):
        P[key] = eng._pack_linear(sd[f"{wkey}.weight"], sd[f"{wkey}.bias"])

    kw: List[List[torch.Tensor]] = [[] for _ in range(nlev)]
    kb: List[List[torch.Tensor]] = [[] for _ in range(nlev)]
    vw: List[List[torch.Tensor]] = [[] for _ in range(nlev)]
    vb: List[List[torch.Tensor]] = [[] for _ in range(nlev)]
    for li in range(eng.nl):
        p = f"{HP}.transformer_cross_attention_layers.{li}"
        Wi, bi = sd[f"{p}.multihead_attn.in_proj_weight"], sd[f"{p}.multihead_attn.in_proj_bias"]
        P[f"{p}.q"] = eng._pack_linear(Wi[:256], bi[:256])
        lin(f"{p}.out_proj", f"{p}.multihead_attn.out_proj")
        lvl = li % nlev
        kw[lvl].append(Wi[256:512]); kb[lvl].append(bi[256:512])
        vw[lvl].append(Wi[512:]); vb[lvl].append(bi[512:])
        eng._pack_ln(sd, f"{p}.norm")
        p = f"{HP}.transformer_self_attention_layers.{li}"
        Wi, bi = sd[f"{p}.self_attn.in_proj_weight"], sd[f"{p}.self_attn.in_proj_bias"]
        P[f"{p}.qk"] = eng._pack_linear(Wi[:512], bi[:512])
        P[f"{p}.v"] = eng._pack_linear(Wi[512:], bi[512:])
        lin(f"{p}.out_proj", f"{p}.self_attn.out_proj")
        eng._pack_ln(sd, f"{p}.norm")
        p = f"{HP}.transformer_ffn_layers.{li}"
        lin(f"{p}.linear1", f"{p}.linear1")
        lin(f"{p}.linear2", f"{p}.linear2")
        eng._pack_ln(sd, f"{p}.norm")
    for lvl in range(nlev):
        # the layers attending level lvl share their memory: all their key (value) projections as ONE GEMM
        P[f"{HP}.k_all.{lvl}"] = eng._pack_linear(torch.cat(kw[lvl], 0), torch.cat(kb[lvl], 0))
        P[f"{HP}.v_all.{lvl}"] = eng._pack_linear(torch.cat(vw[lvl], 0), torch.cat(vb[lvl], 0))
        P[f"{HP}.input_proj.{lvl}"] = eng._pack(sd[f"{HP}.input_proj.{lvl}.weight"].float(), sd[f"{HP}.input_proj.{lvl}.bias"].float())
    # fragment-ordered copies of the row-local linears for fx_row_chain (two launches per decoder layer instead of ~17; build_masked_decoder)
    RC: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}

    def rc(key, W, b, pad_n=None):
        W, b = W.float(), (b.float() if b is not None else torch.zeros(W.shape[0]))
        if pad_n is not None and W.shape[0] < pad_n:
            W = torch.cat([W, torch.zeros(pad_n - W.shape[0], W.shape[1])], 0)
            b = torch.cat([b, torch.zeros(pad_n - b.shape[0])], 0)
        RC[key] = (eng._pack_frag(W), eng._dev(b))

    for li in range(eng.nl):
        p = f"{HP}.transformer_cross_attention_layers.{li}"
        Wi, bi = sd[f"{p}.multihead_attn.in_proj_weight"], sd[f"{p}.multihead_attn.in_proj_bias"]
        rc(f"{li}.cq", Wi[:256], bi[:256])
        rc(f"{li}.co", sd[f"{p}.multihead_attn.out_proj.weight"], sd[f"{p}.multihead_attn.out_proj.bias"])
        p = f"{HP}.transformer_self_attention_layers.{li}"
        Wi, bi = sd[f"{p}.self_attn.in_proj_weight"], sd[f"{p}.self_attn.in_proj_bias"]
        rc(f"{li}.sqk", Wi[:512], bi[:512])
        rc(f"{li}.sv", Wi[512:], bi[512:])
        rc(f"{li}.so", sd[f"{p}.self_attn.out_proj.weight"], sd[f"{p}.self_attn.out_proj.bias"])
        p = f"{HP}.transformer_ffn_layers.{li}"
        W1, b1, W2, b2 = sd[f"{p}.linear1.weight"], sd[f"{p}.linear1.bias"], sd[f"{p}.linear2.weight"], sd[f"{p}.linear2.bias"]
        ffn = W1.shape[0]
        for h0 in range(0, ffn, 1024):      # the [32][1024] LDS slot holds 1024 hidden channels at a time: wider FFNs run as halves whose outputs add
            h1 = min(ffn, h0 + 1024)
            rc(f"{li}.f1.{h0 // 1024}", W1[h0:h1], b1[h0:h1])
            rc(f"{li}.f2.{h0 // 1024}", W2[:, h0:h1], b2 if h0 == 0 else None)
    eng.rc_ffn_parts = [(h0 // 1024, min(1024, ffn - h0)) for h0 in range(0, ffn, 1024)]
    for j in range(2):
        rc(f"mc{j}", sd[f"{PH}.mask_classifier.layers.{j}.weight"], sd[f"{PH}.mask_classifier.layers.{j}.bias"])
    md_w = sd[f"{PH}.mask_classifier.layers.2.weight"]
    rc("mc2", md_w, sd[f"{PH}.mask_classifier.layers.2.bias"], pad_n=(128 if md_w.shape[0] < 128 else None))
    eng.RCM = RC
    eng.query_feat = eng._dev(sd[f"{HP}.query_feat.weight"].float(), torch.bfloat16)
    eng.query_embed = eng._dev(sd[f"{HP}.query_embed.weight"].float(), torch.bfloat16)
    eng._pack_ln(sd, f"{PH}.decoder_norm")
    lin(f"{PH}.classifier", f"{PH}.classifier")
    for j in range(3):
        lin(f"{PH}.mask_classifier.{j}", f"{PH}.mask_classifier.layers.{j}")


class MaskDecoderPlanMixin:
    """Launch-sequence builders for plans (subclasses of engine._PlanBase) whose engine has nq, nc, nl, nlev, P, ln, query_*."""

    def build_masked_decoder(self, msf: Sequence[NT], mf: NT, md: int):
        """``msf``: the nlev memory levels (coarsest first), ``mf``: mask features [B,h,w,md] (md = 128 or 256: a narrower mask dimension
        arrives zero-padded, see _BfPlan._build).  Returns (decoder_norm output of the last layer [B*Q,256], its mask embedding [B*Q,md])."""
        e, P, B, lib = self.eng, self.eng.P, self.B, self.lib
        Q = e.nq
        nlev = e.nlev
        Ls, k_all, v_all, mfp, W32 = [], [], [], [], []
        for l in range(nlev):
            f = msf[l]
            L = f.H * f.W
            Ls.append(L)
            W32.append((L + 31) // 32)
            src_l = self.conv(f, P[f"{HP}.input_proj.{l}"], name=f"dec.src{l}").as_rows()
            pos = pos_embed_sine_normalized(f.H, f.W, 128).to(device=self.dev, dtype=torch.bfloat16).contiguous()
            pos_nt = NT(pos, L, 1, 1, 256, 256)
            self.keep.append(pos)
            srcpos = self.add_rows(src_l, pos_nt, L, f"dec.srcpos{l}")
            k_all.append(self.linear(srcpos, P[f"{HP}.k_all.{l}"], name=f"dec.k_all{l}"))
            v_all.append(self.linear(src_l, P[f"{HP}.v_all.{l}"], name=f"dec.v_all{l}"))
            # attention-mask source: the mask features bilinearly resized to this level (commutes with the mask einsum)
            m = self._new(f"dec.mfp{l}", B, f.H, f.W, md)
            self.resize(mf, m)
            mfp.append(m)
        R = B * Q
        qe = NT(e.query_embed, Q, 1, 1, 256, 256)
        out0 = e.query_feat.repeat(B, 1).contiguous()
        self.keep.append(out0)
        out = NT(out0, R, 1, 1, 256, 256)
        self.attn_bits: List[torch.Tensor] = []
        self.force_points: List[int] = []

        def heads(x: NT, idx: int, level: Optional[int]):
            dn = self.layernorm(x, f"{PH}.decoder_norm", f"ph{idx}.dn")
            m1 = self.linear(dn, P[f"{PH}.mask_classifier.0"], name=f"ph{idx}.m1", act="relu")
            m2 = self.linear(m1, P[f"{PH}.mask_classifier.1"], name=f"ph{idx}.m2", act="relu")
            pcm = P[f"{PH}.mask_classifier.2"]
            if pcm.N == md:
                emb = self.linear(m2, pcm, name=f"ph{idx}.emb")
            else:   # mask dimension below the einsum kernel's 128 channels (bisenetformer-m-ade: 96): zero channels up to md, written once here
                emb = self._new(f"ph{idx}.emb", R, 1, 1, md)
                emb.t.zero_()
                self.linear(m2, pcm, out=emb.slice(0, pcm.N))
            if level is not None:
                bits = torch.zeros(R, W32[level], dtype=torch.int32, device=self.dev)
                self._op(lib.fx_query_pixel_logits_bf16, emb.ptr, emb.ld, mfp[level].ptr, mfp[level].ld, 2, None, 0, bits.data_ptr(),
                         W32[level], B, Q, Ls[level], md)
                self.attn_bits.append(bits)
                self.force_points.append(len(self.ops))
            return dn, emb

        # row chains from 1 000 rows on (FX_MASKDEC_ROW_CHAIN_MIN_ROWS; 0 = never): a 32-row workgroup walks its ~10 GEMM stages serially, each
        # streaming the stage's whole weight matrix - 25 workgroups (MaskFormer bs=16: two parts of 8 x 100 rows) are no faster than the
        # 50-workgroup launches per layer they replace (1441 vs 1448 img/s), 50+ are (BiSeNetFormer bs=32: 7 406 -> 7 767-8 373 img/s)
        min_rows = int(os.environ.get("FX_MASKDEC_ROW_CHAIN_MIN_ROWS", "1000"))
        if min_rows > 0 and R >= min_rows and hasattr(e, "RCM"):
            dn, emb = self._masked_decoder_row_chains(out, qe, k_all, v_all, mfp, Ls, W32, md, R, B, Q)
            self.levels = Ls
            self.W32 = W32
            return dn, emb
        heads(out, 0, 0)
        dn = emb = None
        # one workspace for the key-sliced cross attention (launches are serial on one stream)
        mha_ws = torch.empty(max(8, max(lib.fx_mha_workspace_bytes(B, Q, L, 8, 1) for L in Ls)), dtype=torch.uint8, device=self.dev)
        self.keep.append(mha_ws)
        for i in range(e.nl):
            lvl, j = i % nlev, i // nlev
            p = f"{HP}.transformer_cross_attention_layers.{i}"
            t2 = self.layernorm(out, f"{p}.norm", f"dec{i}.c_n")
            qin = self.add_rows(t2, qe, Q, f"dec{i}.c_qin")
            qc = self.linear(qin, P[f"{p}.q"], name=f"dec{i}.c_q")
            att = self._new(f"dec{i}.c_att", R, 1, 1, 256)
            ks, vs = k_all[lvl].slice(j * 256, 256), v_all[lvl].slice(j * 256, 256)
            self._op(lib.fx_mha_masked_bf16, qc.ptr, qc.ld, ks.ptr, ks.ld, vs.ptr, vs.ld, att.ptr, att.ld, B, Q, Ls[lvl], 8,
                     self.attn_bits[i].data_ptr(), W32[lvl], mha_ws.data_ptr(), C.c_size_t(mha_ws.numel()))
            out = self.linear(att, P[f"{p}.out_proj"], name=f"dec{i}.c_o", residual=out)
            p = f"{HP}.transformer_self_attention_layers.{i}"
            t2 = self.layernorm(out, f"{p}.norm", f"dec{i}.s_n")
            qk_in = self.add_rows(t2, qe, Q, f"dec{i}.s_qk_in")
            qkv = self._new(f"dec{i}.s_qkv", R, 1, 1, 768)
            self.linear(qk_in, P[f"{p}.qk"], out=qkv.slice(0, 512))
            self.linear(t2, P[f"{p}.v"], out=qkv.slice(512, 256))
            att = self.mha(qkv, B, Q, f"dec{i}.s_att")
            out = self.linear(att, P[f"{p}.out_proj"], name=f"dec{i}.s_o", residual=out)
            p = f"{HP}.transformer_ffn_layers.{i}"
            t2 = self.layernorm(out, f"{p}.norm", f"dec{i}.f_n")
            f1 = self.linear(t2, P[f"{p}.linear1"], name=f"dec{i}.f1", act="relu")
            out = self.linear(f1, P[f"{p}.linear2"], name=f"dec{i}.out", residual=out)
            dn, emb = heads(out, i + 1, (i + 1) % nlev if i < e.nl - 1 else None)
        self.levels = Ls
        self.W32 = W32
        return dn, emb

    def _masked_decoder_row_chains(self, out: NT, qe: NT, k_all, v_all, mfp, Ls, W32, md: int, R: int, B: int, Q: int):
        """The decoder layers of build_masked_decoder with every row-local run of layers as ONE fx_row_chain launch (round 4): per layer
        [masked cross-attention core] [out_proj + residual, LayerNorm, + query embedding, q = k and v projections of the self-attention]
        [self-attention core] [out_proj + residual, LayerNorm, FFN + residual, decoder_norm + the 3-layer mask-embedding MLP, the NEXT layer's
        LayerNorm + query embedding + q projection] [attention-mask bits] - 5 launches instead of ~22 (the pre-norm form of
        fai_mf/modelling.py:453-549: norm -> attention / FFN -> residual add)."""
        e, lib = self.eng, self.lib
        RC = e.RCM
        S0, S1, S2, S3, BIG = self.RC_S0, self.RC_S1, self.RC_S2, self.RC_S3, self.RC_BIG
        relu = FX_ACT["relu"]
        nlev = e.nlev

        def st(type_, K=0, N=0, act=0, src=-1, dst=-1, aux=-1, ld=0, ld2=0, flags=0, w=None, bias=None, gamma=None, beta=None, g0=None, g1=None):
            s_ = FxRcStage()
            s_.type, s_.K, s_.N, s_.act, s_.src, s_.dst, s_.aux, s_.ld, s_.ld2, s_.flags = type_, K, N, act, src, dst, aux, ld, ld2, flags
            s_.w, s_.bias, s_.gamma, s_.beta, s_.g0, s_.g1 = w, bias, gamma, beta, g0, g1
            return s_

        def load(nt: NT, dst):
            return st(0, K=nt.C, dst=dst, g0=nt.ptr, ld=nt.ld)

        def gemm(key, src, K, N, dst=-1, act=0, out: Optional[NT] = None):
            w, b = RC[key]
            return st(1, K=K, N=N, act=act, src=src, dst=dst, w=w.data_ptr(), bias=b.data_ptr(), g0=out.ptr if out is not None else None,
                      ld=out.ld if out is not None else 0)

        def ln(name, src, dst, out: Optional[NT] = None):
            g_, b_ = e.ln[name]
            return st(6, K=256, src=src, dst=dst, gamma=g_.data_ptr(), beta=b_.data_ptr(), g0=out.ptr if out is not None else None,
                      ld=out.ld if out is not None else 0)

        def add(a, b_, dst, out: Optional[NT] = None):
            return st(3, K=256, src=a, aux=b_, dst=dst, g0=out.ptr if out is not None else None, ld=out.ld if out is not None else 0)

        qe_rep = e.query_embed.repeat(B, 1).contiguous()     # [R, 256]: a LOAD stage reads row m of a global matrix
        self.keep.append(qe_rep)
        qe_r = NT(qe_rep, R, 1, 1, 256, 256)
        mdn = RC["mc2"][0].shape[0] * 32                      # mask-embedding width as packed (zero-padded to 128 below that)
        assert mdn == md, (mdn, md)
        n_ffn = sum(n for _, n in e.rc_ffn_parts)
        fl_heads = 2.0 * R * 256 * (256 + 256 + md)
        fl_cq = 2.0 * R * 256 * 256

        def heads_stages(src_out, idx):
            """decoder_norm + mask MLP on the LDS slot `src_out` (kept intact); emb -> global.  Uses S0, S2, S3."""
            dn = self._new(f"ph{idx}.dn", R, 1, 1, 256)
            emb = self._new(f"ph{idx}.emb", R, 1, 1, md)
            return [ln(f"{PH}.decoder_norm", src_out, S2, out=dn), gemm("mc0", S2, 256, 256, dst=S0, act=relu), gemm("mc1", S0, 256, 256, dst=S2, act=relu),
                    gemm("mc2", S2, 256, md, out=emb)], dn, emb

        def cross_pre_stages(src_out, li, qc: NT):
            """LayerNorm + query embedding + q projection of layer li's cross attention from the LDS slot `src_out`.  Uses S0, S2, S3."""
            p = f"{HP}.transformer_cross_attention_layers.{li}"
            return [ln(f"{p}.norm", src_out, S2), load(qe_r, S3), add(S2, S3, S0), gemm(f"{li}.cq", S0, 256, 256, out=qc)]

        def bits_op(emb: NT, level: int):
            bits = torch.zeros(R, W32[level], dtype=torch.int32, device=self.dev)
            self._op(lib.fx_query_pixel_logits_bf16, emb.ptr, emb.ld, mfp[level].ptr, mfp[level].ld, 2, None, 0, bits.data_ptr(), W32[level], B, Q,
                     Ls[level], md)
            self.attn_bits.append(bits)
            self.force_points.append(len(self.ops))

        qc = self._new("dec0.c_q", R, 1, 1, 256)
        hs, dn, emb = heads_stages(S1, 0)
        self._rc_program([load(out, S1)] + hs + cross_pre_stages(S1, 0, qc), R, "maskdec.pre", fl_heads + fl_cq)
        bits_op(emb, 0)
        mha_ws = torch.empty(max(8, max(lib.fx_mha_workspace_bytes(B, Q, L, 8, 1) for L in Ls)), dtype=torch.uint8, device=self.dev)
        self.keep.append(mha_ws)
        for i in range(e.nl):
            lvl, j = i % nlev, i // nlev
            last = i == e.nl - 1
            att = self._new(f"dec{i}.c_att", R, 1, 1, 256)
            ks, vs = k_all[lvl].slice(j * 256, 256), v_all[lvl].slice(j * 256, 256)
            self._op(lib.fx_mha_masked_bf16, qc.ptr, qc.ld, ks.ptr, ks.ld, vs.ptr, vs.ld, att.ptr, att.ld, B, Q, Ls[lvl], 8,
                     self.attn_bits[i].data_ptr(), W32[lvl], mha_ws.data_ptr(), C.c_size_t(mha_ws.numel()))
            # ---- after the cross attention: residual, self-attention projections
            p = f"{HP}.transformer_self_attention_layers.{i}"
            out1 = self._new(f"dec{i}.c_o", R, 1, 1, 256)
            qkv = self._new(f"dec{i}.s_qkv", R, 1, 1, 768)
            prog = [load(att, S0), load(out, S1), gemm(f"{i}.co", S0, 256, 256, dst=S2), add(S2, S1, S1, out=out1),
                    ln(f"{p}.norm", S1, S2), load(qe_r, S3), add(S2, S3, S0),
                    gemm(f"{i}.sqk", S0, 256, 512, out=qkv.slice(0, 512)), gemm(f"{i}.sv", S2, 256, 256, out=qkv.slice(512, 256))]
            self._rc_program(prog, R, f"dec{i}.post_cross", 2.0 * R * 256 * (256 + 768))
            att = self.mha(qkv, B, Q, f"dec{i}.s_att")
            # ---- after the self attention: residual, FFN, heads, the next layer's query projection
            p = f"{HP}.transformer_ffn_layers.{i}"
            out3 = self._new(f"dec{i}.out", R, 1, 1, 256)
            prog = [load(att, S0), load(out1, S1), gemm(f"{i}.so", S0, 256, 256, dst=S2), add(S2, S1, S1), ln(f"{p}.norm", S1, S2)]
            acc_slot = None
            for part, n in e.rc_ffn_parts:       # hidden channels in blocks of <= 1024 (the [32][1024] slot); partial outputs add
                dst = S0 if acc_slot is None else S3
                prog += [gemm(f"{i}.f1.{part}", S2, 256, n, dst=BIG, act=relu), gemm(f"{i}.f2.{part}", BIG, n, 256, dst=dst)]
                if acc_slot is None:
                    acc_slot = S0
                else:
                    prog.append(add(S0, S3, S0))
            prog.append(add(S0, S1, S1, out=out3))
            hs, dn, emb = heads_stages(S1, i + 1)
            prog += hs
            fl = 2.0 * R * (256 * 256 + 2 * 256 * n_ffn) + fl_heads
            if not last:
                qc = self._new(f"dec{i + 1}.c_q", R, 1, 1, 256)
                prog += cross_pre_stages(S1, i + 1, qc)
                fl += fl_cq
            self._rc_program(prog, R, f"dec{i}.post_self", fl)
            out = out3
            if not last:
                bits_op(emb, (i + 1) % nlev)
        return dn, emb

    def build_mask_outputs(self, dn: NT, emb: NT, mf: NT, md: int, full_masks: bool, predict_all_pixels: bool):
        """Class probabilities, low-resolution mask probabilities, optional full-resolution ``masks`` and the device side of the
        processor's postprocess (threshold branch: fx_mf_postprocess; predict_all_pixels branch: fx_seg_postprocess)."""
        e, P, B, lib = self.eng, self.eng.P, self.B, self.lib
        H, W = self.H, self.W
        Q, K = e.nq, e.nc
        R = B * Q
        hl, wl = mf.H, mf.W
        cls_logits = self.linear(dn, P[f"{PH}.classifier"], name="cls_logits", out_f32=True)
        self.probs = self._io("probs", (B, Q, K), torch.float32)
        self.cls_score = self._io("cls_score", (B, Q), torch.float32)
        self.cls_label = self._io("cls_label", (B, Q), torch.int32)
        self._op(lib.fx_mf_class_head, cls_logits.ptr, cls_logits.ld, self.probs.data_ptr(), self.cls_score.data_ptr(), self.cls_label.data_ptr(),
                 R, K, int(e.cls_sigmoid))
        PL = hl * wl
        self.mask_probs = self._io("mask_probs", (B, Q, hl, wl), torch.float32)  # sigmoid(mask logits) at the mask-feature resolution
        mf_rows = mf.as_rows()
        self._op(lib.fx_query_pixel_logits_bf16, emb.ptr, emb.ld, mf_rows.ptr, mf_rows.ld, 1, self.mask_probs.data_ptr(), PL, None, 0, B, Q, PL, md)
        self.masks = None
        if full_masks:
            # engine.masks_dtype = "bf16" (FX_MF_MASKS_DTYPE / bench.py --mf-masks-dtype): the [B,Q,H,W] tensor in 16 bits - half the bytes of
            # the reference's fp32 tensor (4.1 GB per bs = 16 800^2 step), same taps, one rounding at the store; "fp32" (default) = the reference's
            if getattr(e, "masks_dtype", "fp32") == "bf16":
                self.masks = self._io("masks", (B, Q, H, W), torch.bfloat16)
                self._op(lib.fx_mf_upsample_probs_bf16, self.mask_probs.data_ptr(), hl, wl, self.masks.data_ptr(), H, W, R)
            else:
                self.masks = self._io("masks", (B, Q, H, W), torch.float32)
                self._op(lib.fx_mf_upsample_probs_f32, self.mask_probs.data_ptr(), hl, wl, self.masks.data_ptr(), H, W, R)
        self.det_count = self._io("det_count", (B,), torch.int32).zero_()
        self.det_query = self._io("det_query", (B, Q), torch.int32).zero_()
        self.det_scores = self._io("det_scores", (B, Q), torch.float32).zero_()
        self.det_labels = self._io("det_labels", (B, Q), torch.int32).zero_()
        self.det_boxes = self._io("det_boxes", (B, Q, 4), torch.int32).zero_()
        self.det_area = self._io("det_area", (B, Q), torch.int32).zero_()
        self.mask_width = W
        self.mask_words = self._io("mask_words", (B, Q, H, (W + 31) // 32), torch.int32).zero_()   # rows padded to whole words (bits >= W are 0)
        self.winner = None
        self.post_index = len(self.ops)
        if predict_all_pixels:
            ws_bytes = lib.fx_seg_postprocess_workspace_bytes(B, Q, hl, wl, H, W)
            self.post_ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=self.dev)
            self.winner = self._io("winner", (B, H, W), torch.uint8)   # per-pixel query index (semantic map = cls_label[winner])
            self._op(lib.fx_seg_postprocess, self.mask_probs.data_ptr(), hl, wl, H, W, self.cls_score.data_ptr(), self.cls_label.data_ptr(), B, Q,
                     None, int(e.use_mask_score), self.post_ws.data_ptr(), C.c_size_t(self.post_ws.numel()), self.det_count.data_ptr(),
                     self.det_query.data_ptr(), self.det_scores.data_ptr(), self.det_labels.data_ptr(), self.det_boxes.data_ptr(),
                     self.det_area.data_ptr(), self.mask_words.data_ptr(), self.winner.data_ptr())
        else:
            # (the larger workspace: the statistics pass also leaves one bit plane per evaluated query, the kept detections' masks are plane copies)
            ws_bytes = lib.fx_mf_postprocess_workspace_bytes_fused(B, Q, hl, wl, H, W) if int(os.environ.get("FX_MF_POST_FUSED", "1")) else lib.fx_mf_postprocess_workspace_bytes(B, Q, H)
            self.post_ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=self.dev)
            self._op(lib.fx_mf_postprocess, self.mask_probs.data_ptr(), hl, wl, H, W, self.cls_score.data_ptr(), self.cls_label.data_ptr(), B, Q,
                     C.c_float(e.mask_threshold), None, int(e.use_mask_score), self.post_ws.data_ptr(), C.c_size_t(self.post_ws.numel()),
                     self.det_count.data_ptr(), self.det_query.data_ptr(), self.det_scores.data_ptr(), self.det_labels.data_ptr(),
                     self.det_boxes.data_ptr(), self.det_area.data_ptr(), self.mask_words.data_ptr())

    # -------------------------------------------------------------- execution
    def patch_args(self, fn, args, thr: float):
        if fn is self.lib.fx_mf_postprocess:
            return args[:10] + (C.c_float(thr),) + args[11:]
        if fn is self.lib.fx_seg_postprocess:
            return args[:9] + (C.c_float(thr),) + args[10:]
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
