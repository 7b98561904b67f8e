"""RT-DETR inference engine: packs a reference-layout state_dict for the gfx950 kernels and runs
FAIDetr.forward (eval) + the device side of DETRProcessor.postprocess as one hipGraph of C-ABI calls.

Reference path being replaced (file:line in FocoosAI/focoos):
  FAIDetr.forward            focoos/models/fai_detr/modelling.py:1344-1358
  ResNet.forward             focoos/nn/backbone/resnet.py:252-266
  Encoder.forward            focoos/models/fai_detr/modelling.py:297-347
  TransformerPredictor       focoos/models/fai_detr/modelling.py:1145-1263
  TransformerDecoder(+Layer) focoos/models/fai_detr/modelling.py:924-1020
  DETRHead.forward tail      focoos/models/fai_detr/modelling.py:392-401
  DETRProcessor.postprocess  focoos/models/fai_detr/processor.py:146-197

PyTorch is used for device memory (torch.empty), stream handles and host-side weight packing
only; every FLOP of the forward goes through libfocoos_amd.so.

Data layout in HBM: activations NHWC bf16, one buffer per layer output (288 GB HBM: no arena
juggling; channel-concats are free because producers write channel slices of the concat buffer);
weights [Npad][KH][KW][C] bf16 with eval-BatchNorm folded in (and RepVGG 3x3+1x1 branches
re-parameterised into one 3x3), biases fp32; scores / boxes / sampling offsets fp32.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import FX_ACT, FxConvDesc, FxPwChainDesc, FxRcStage, check
from .engine_stdc import StdcEngineMixin, StdcPlanMixin
from .state_spec import RESNET_BLOCKS

BN_EPS = 1e-5
DEFAULT_STREAMS = 2   # batch parts per step (two half-batch parts replayed concurrently: +9 %; see _MultiPlan).  FX_STREAMS=1: one part
MIN_PART_BATCH = 4
DEFAULT_PIPELINE_DEPTH = 3   # batches in flight of the throughput mode (_Pipeline); 2 / 3 / 4 measured 5 002-5 115 / 5 084-5 112 / 5 095-5 133 img/s in one call


class NT:
    """View of an NHWC (or [rows, C]) device buffer for the C ABI: base tensor + channel offset + pixel stride."""

    __slots__ = ("t", "B", "H", "W", "C", "ld", "off")

    def __init__(self, t: torch.Tensor, B: int, H: int, W: int, Cc: int, ld: int, off: int = 0):
        self.t, self.B, self.H, self.W, self.C, self.ld, self.off = t, B, H, W, Cc, ld, off

    @property
    def ptr(self) -> int:
        return self.t.data_ptr() + self.off * self.t.element_size()

    @property
    def rows(self) -> int:
        return self.B * self.H * self.W

    def slice(self, off: int, Cc: int) -> "NT":
        return NT(self.t, self.B, self.H, self.W, Cc, self.ld, self.off + off)

    def as_rows(self) -> "NT":
        return NT(self.t, self.rows, 1, 1, self.C, self.ld, self.off)

    def torch_view(self) -> torch.Tensor:
        """[B,H,W,C] strided view (tests / debugging)."""
        base = self.t.reshape(-1)
        return torch.as_strided(base, (self.B, self.H, self.W, self.C), (self.H * self.W * self.ld, self.W * self.ld, self.ld, 1), self.off)


class PackedConv:
    __slots__ = ("w", "b", "N", "C", "KH", "KW", "wf")

    def __init__(self, w, b, N, Cc, KH, KW, wf=None):
        self.w, self.b, self.N, self.C, self.KH, self.KW, self.wf = w, b, N, Cc, KH, KW, wf


def _fold_bn(sd, conv_w_key: str, bn_prefix: str) -> Tuple[torch.Tensor, torch.Tensor]:
    """conv weight * gamma/sqrt(var+eps), bias = beta - mean*gamma/sqrt(var+eps)  (F.batch_norm eval, norm.py:49-57)."""
    W = sd[conv_w_key].double()
    g, bta = sd[f"{bn_prefix}.weight"].double(), sd[f"{bn_prefix}.bias"].double()
    mu, var = sd[f"{bn_prefix}.running_mean"].double(), sd[f"{bn_prefix}.running_var"].double()
    s = g / torch.sqrt(var + BN_EPS)
    return (W * s.view(-1, 1, 1, 1)).float(), (bta - mu * s).float()


_STREAMS: Dict[Tuple[int, int], "torch.cuda.Stream"] = {}


def _device_stream(dev: torch.device, i: int) -> "torch.cuda.Stream":
    """Stream `i` of the per-device pool (0 = the engines' main stream, 1.. = the side streams of concurrent batch parts).  Every engine
    and plan on a device shares these few streams: HIP maps streams onto a small set of hardware queues in creation order, and a second
    engine creating its own pair could land both of its streams on ONE queue - its two batch parts then run back to back (measured: the
    MaskFormer leg 12.5 ms as the first engine of a process, 16.8 ms as the second)."""
    key = (dev.index or 0, i)
    st = _STREAMS.get(key)
    if st is None:
        st = _STREAMS[key] = torch.cuda.Stream(dev)
    return st


_LANE_STREAMS: Dict[Tuple[int, int], List["torch.cuda.Stream"]] = {}


def _concurrent_streams(dev: torch.device, n: int) -> List["torch.cuda.Stream"]:
    """`n` streams of the device pool that were MEASURED to run side by side (throughput mode: one batch per stream).  HIP multiplexes
    the streams of a process onto GPU_MAX_HW_QUEUES hardware queues (4 by default) and two streams on one queue serialise; which streams
    share a queue is fixed once they were used, but not a simple function of the creation order (scripts/dev/stream_queue_probe.py: of eight
    streams (0,5) (1,4) (2,3) (2,7) (3,7) collide; 'first 4 together' take twice the time of one).  So: a ~0.4 ms spin kernel on a candidate
    and on every stream already chosen, at once - a pair that takes as long as two spins shares a queue and the candidate is dropped.
    One-time cost ~20 ms per (device, n); the pool's first stream (the engines' main stream) is always lane 0."""
    key = (dev.index or 0, n)
    if key in _LANE_STREAMS:
        return _LANE_STREAMS[key]
    import time

    ticks = int(1e6)

    def spin(sts):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for s_ in sts:
            with torch.cuda.stream(s_):
                torch.cuda._sleep(ticks)
        for s_ in sts:
            s_.synchronize()
        return time.perf_counter() - t0

    chosen = [_device_stream(dev, 0)]
    spin(chosen)
    one = min(spin(chosen) for _ in range(3))
    cand = 1
    while len(chosen) < n and cand < 12:
        s_ = _device_stream(dev, cand)
        cand += 1
        spin([s_])
        if all(min(spin([s_, c]) for _ in range(2)) < 1.5 * one for c in chosen):
            chosen.append(s_)
    cand = 1
    while len(chosen) < n:      # fewer concurrent streams than lanes (GPU_MAX_HW_QUEUES too small): the remaining lanes share queues
        s_ = _device_stream(dev, cand)
        cand += 1
        if s_ not in chosen:
            chosen.append(s_)
    _LANE_STREAMS[key] = chosen
    return chosen


class _EngineBase:
    """Device handle, weight-packing helpers and the ResNet-vd backbone weights shared by the model families."""

    def __init__(self, config: Dict, device: str = "cuda:0"):
        if not torch.cuda.is_available():
            raise _lib.FocoosAmdError("focoos_amd needs a ROCm GPU (gfx950); no CPU fallback exists")
        if _lib.compute_dtype() != "bf16":
            raise _lib.FocoosAmdError("the inference engines compute in bfloat16 (north_star: bf16 MFMA); the fp16 element type "
                                      "(_lib.set_compute_dtype('fp16')) is the training step's - build the engine under 'bf16'")
        self.lib = _lib.load()
        self.cfg = dict(config)
        self.dev = torch.device(device)
        cu, arch = C.c_int(0), C.create_string_buffer(64)
        check(self.lib.fx_device_info(self.dev.index or 0, C.byref(cu), arch, 64), "fx_device_info (gfx950 required)")
        self.cu_count, self.arch = cu.value, arch.value.decode()
        self.depth = int(config["backbone_config"].get("depth", 50))
        self.stream = _device_stream(self.dev, 0)
        self.plans: Dict[Tuple, "_PlanBase"] = {}
        self.ln: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}

    def _dev(self, t: torch.Tensor, dtype=None) -> torch.Tensor:
        return t.to(device=self.dev, dtype=dtype or t.dtype).contiguous()

    def _pack(self, W4: torch.Tensor, bias: Optional[torch.Tensor]) -> PackedConv:
        N, Cc, KH, KW = W4.shape
        Np = (N + 127) // 128 * 128
        w = torch.zeros(Np, KH, KW, Cc, dtype=torch.float32)
        w[:N] = W4.permute(0, 2, 3, 1)
        b = torch.zeros(Np, dtype=torch.float32)
        if bias is not None:
            b[:N] = bias
        wf = None
        if ((KH == 3 and KW == 3 and (N in (64, 128) or N % 256 == 0) and Cc % 64 == 0) or (KH == 1 and KW == 1 and N % 256 == 0 and Cc % 256 == 0)
                or (KH == 3 and KW == 3 and Cc == 32 and N in (32, 64))):   # the stem's conv1_2 / conv1_3 (conv3x3_c32.hip)
            # second copy in MFMA fragment order (fx_conv_desc.w_frag) for the halo / pointwise kernels of conv3x3_flat.hip:
            # k = (kh*KW + kw)*C + c
            wf = self._pack_frag(W4.permute(0, 2, 3, 1).reshape(N, KH * KW * Cc))
        return PackedConv(self._dev(w, torch.bfloat16), self._dev(b), N, Cc, KH, KW, wf)

    def _pack_frag(self, W2: torch.Tensor) -> torch.Tensor:
        """[N, K] weights in MFMA fragment order [N/32][K/16][lane][8] (include/focoos_amd.h, fx_pw_chain_desc): lane l of
        fragment (nb, ks) holds W[nb*32 + l%32][ks*16 + (l//32)*8 : +8] - one contiguous 1 KiB read per wave and fragment."""
        N, K = W2.shape
        assert N % 32 == 0 and K % 16 == 0, (N, K)
        w = W2.float().reshape(N // 32, 32, K // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous()  # [nb][ks][half][row][8]
        return self._dev(w.reshape(N // 32, K // 16, 64, 8), torch.bfloat16)

    def _pack_linear(self, W: torch.Tensor, b: Optional[torch.Tensor]) -> PackedConv:
        return self._pack(W.float().view(W.shape[0], W.shape[1], 1, 1), None if b is None else b.float())

    def _pack_ln(self, sd, name: str) -> None:
        self.ln[name] = (self._dev(sd[f"{name}.weight"].float()), self._dev(sd[f"{name}.bias"].float()))

    def _pack_backbone(self, sd, P: Dict[str, PackedConv]) -> None:
        """ResNet-vd: eval BatchNorm folded into every conv (resnet.py:164-250)."""
        bb = "pixel_decoder.backbone"

        def cbn(name):
            P[name] = self._pack(*_fold_bn(sd, f"{name}.conv.weight", f"{name}.norm"))

        # stem conv1_1 stays fp32 [kh][kw][c][n] (direct-conv kernel, weights via LDS)
        w, b = _fold_bn(sd, f"{bb}.conv1.conv1_1.conv.weight", f"{bb}.conv1.conv1_1.norm")
        self.stem_w = self._dev(w.permute(2, 3, 1, 0).contiguous())  # [kh][kw][c][n]
        self.stem_b = self._dev(b)
        mean = torch.tensor(self.cfg.get("pixel_mean", [123.675, 116.28, 103.53]), dtype=torch.float32)
        std = torch.tensor(self.cfg.get("pixel_std", [58.395, 57.12, 57.375]), dtype=torch.float32)
        self.px_mean, self.px_inv_std = self._dev(mean), self._dev(1.0 / std)
        cbn(f"{bb}.conv1.conv1_2")
        cbn(f"{bb}.conv1.conv1_3")
        # fragment-ordered copies of the 1x1 layers around the residual add for fx_pw_chain_bf16 (branch2c [+ shortcut conv as a
        # second K segment] -> next block's branch2a in one launch): chain_c[(si,bi)] = (W1, bias1, K1a, K1b, N1), chain_a = (W2, bias2, N2)
        self.chain_c: Dict[Tuple[int, int], Tuple] = {}
        self.chain_a: Dict[Tuple[int, int], Tuple] = {}
        for si, n in enumerate(RESNET_BLOCKS[self.depth]):
            for bi in range(n):
                p = f"{bb}.res_layers.{si}.blocks.{bi}"
                for br in ("branch2a", "branch2b", "branch2c"):
                    cbn(f"{p}.{br}")
                wa, ba = _fold_bn(sd, f"{p}.branch2a.conv.weight", f"{p}.branch2a.norm")
                wc, bc = _fold_bn(sd, f"{p}.branch2c.conv.weight", f"{p}.branch2c.norm")
                w1, b1, k1b = wc[:, :, 0, 0], bc, 0
                if bi == 0:
                    sk = f"{p}.short" if si == 0 else f"{p}.short.conv"
                    cbn(sk)
                    ws, bs = _fold_bn(sd, f"{sk}.conv.weight", f"{sk}.norm")
                    w1, b1, k1b = torch.cat([w1, ws[:, :, 0, 0]], 1), bc + bs, ws.shape[1]
                if w1.shape[0] % 256 == 0 and wc.shape[1] % 64 == 0 and k1b % 64 == 0:
                    self.chain_c[(si, bi)] = (self._pack_frag(w1), self._dev(b1), wc.shape[1], k1b, w1.shape[0])
                if wa.shape[0] % 64 == 0 and wa.shape[1] % 256 == 0:
                    self.chain_a[(si, bi)] = (self._pack_frag(wa[:, :, 0, 0]), self._dev(ba), wa.shape[0])


class DetrEngine(StdcEngineMixin, _EngineBase):
    def __init__(self, config: Dict, state_dict: Dict[str, torch.Tensor], device: str = "cuda:0"):
        super().__init__(config, device)
        # backbone: ResNet-vd (fai-detr-l-*) or STDC (fai-detr-m-coco; engine_stdc.py)
        self.stdc = config["backbone_config"].get("model_type") == "stdc"
        if self.stdc:
            self._init_stdc(config["backbone_config"])
        self.nc = int(config["num_classes"])
        self.nq = int(config.get("num_queries", 300))
        self.hd = int(config.get("transformer_predictor_hidden_dim", 256))
        self.nl = int(config.get("transformer_predictor_dec_layers", 6))
        self.nhead = int(config.get("transformer_predictor_nhead", 8))
        self.n_enc = int(config.get("pixel_decoder_num_encoder_layers", 1))
        # hybrid-encoder width: 256 (fai-detr-l-*) or 128 (fai-detr-m-coco); the AIFI layer's attention / LayerNorm kernels exist at 256 only
        # (8 heads of 32 channels), so the narrow encoder needs pixel_decoder_num_encoder_layers = 0 - as fai-detr-m-coco has
        self.fd = int(config.get("pixel_decoder_feat_dim", 256))
        if (self.hd != 256 or self.fd not in (128, 256) or int(config.get("pixel_decoder_out_dim", self.fd)) != self.fd or self.nhead != 8
                or (self.n_enc > 0 and self.fd != 256)):
            raise _lib.FocoosAmdError("engine kernels cover decoder width 256 / 8 heads with a 256-channel hybrid encoder, or a 128-channel one "
                                      "without the AIFI layer (fai-detr-l-*, fai-detr-m-coco)")
        self.top_k = int(config.get("top_k", 300))
        self.threshold = float(config.get("threshold", 0.5))
        self.load_state_dict(state_dict)

    # ------------------------------------------------------------------ weight packing
    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        sd = {k: v.detach().cpu() for k, v in sd.items()}
        P: Dict[str, PackedConv] = {}

        def cbn(name, conv="conv", norm="norm"):
            P[name] = self._pack(*_fold_bn(sd, f"{name}.{conv}.weight", f"{name}.{norm}"))

        if self.stdc:
            self._pack_stdc(sd, P)
        else:
            self._pack_backbone(sd, P)
        pd = "pixel_decoder"
        for i in range(3):
            P[f"{pd}.input_proj.{i}"] = self._pack(*_fold_bn(sd, f"{pd}.input_proj.{i}.0.weight", f"{pd}.input_proj.{i}.1"))
        for li in range(self.n_enc):
            p = f"{pd}.encoder.0.layers.{li}"
            Wi, bi_ = sd[f"{p}.self_attn.in_proj_weight"], sd[f"{p}.self_attn.in_proj_bias"]
            P[f"{p}.qk"] = self._pack_linear(Wi[:512], bi_[:512])
            P[f"{p}.v"] = self._pack_linear(Wi[512:], bi_[512:])
            P[f"{p}.out_proj"] = self._pack_linear(sd[f"{p}.self_attn.out_proj.weight"], sd[f"{p}.self_attn.out_proj.bias"])
            for l in ("linear1", "linear2"):
                P[f"{p}.{l}"] = self._pack_linear(sd[f"{p}.{l}.weight"], sd[f"{p}.{l}.bias"])
            for nrm in ("norm1", "norm2"):
                setattr(self, f"_ln_{p}.{nrm}", None)
        self.ln = {}

        def ln(name):
            self._pack_ln(sd, name)

        for li in range(self.n_enc):
            ln(f"{pd}.encoder.0.layers.{li}.norm1")
            ln(f"{pd}.encoder.0.layers.{li}.norm2")
        for i in range(2):
            cbn(f"{pd}.lateral_convs.{i}")
            cbn(f"{pd}.downsample_convs.{i}")
        for blk in ("fpn_blocks", "pan_blocks"):
            for i in range(2):
                p = f"{pd}.{blk}.{i}"
                # conv1 and conv2 read the same input: one N=512 GEMM ([conv1 | conv2])
                w1, b1 = _fold_bn(sd, f"{p}.conv1.conv.weight", f"{p}.conv1.norm")
                w2, b2 = _fold_bn(sd, f"{p}.conv2.conv.weight", f"{p}.conv2.norm")
                P[f"{p}.conv12"] = self._pack(torch.cat([w1, w2], 0), torch.cat([b1, b2], 0))
                for j in range(3):
                    q = f"{p}.bottlenecks.{j}"
                    # RepVGG re-parameterisation (modelling.py:57-81): 3x3 + zero-padded 1x1, biases add
                    w3, b3 = _fold_bn(sd, f"{q}.conv1.conv.weight", f"{q}.conv1.norm")
                    w1_, b1_ = _fold_bn(sd, f"{q}.conv2.conv.weight", f"{q}.conv2.norm")
                    w3 = w3.clone()
                    w3[:, :, 1:2, 1:2] += w1_
                    P[f"{q}.rep"] = self._pack(w3, b3 + b1_)
        hp = "head.predictor"
        for i in range(3):
            cbn(f"{hp}.input_proj.{i}")
        P[f"{hp}.enc_output.0"] = self._pack_linear(sd[f"{hp}.enc_output.0.weight"], sd[f"{hp}.enc_output.0.bias"])
        ln(f"{hp}.enc_output.1")
        P[f"{hp}.enc_score"] = self._pack_linear(sd[f"{hp}.enc_score_classifier.weight"], sd[f"{hp}.enc_score_classifier.bias"])
        # fused score head (fx_enc_score_head_bf16): fragment-ordered enc_output / enc_score weights, classes padded to 128
        self.score_head = None
        ncp = (self.nc + 127) // 128 * 128
        if ncp <= 384:
            w2 = torch.zeros(ncp, self.hd)
            w2[: self.nc] = sd[f"{hp}.enc_score_classifier.weight"].float()
            b2 = torch.full((ncp,), -3e38)
            b2[: self.nc] = sd[f"{hp}.enc_score_classifier.bias"].float()
            self.score_head = (self._pack_frag(sd[f"{hp}.enc_output.0.weight"].float()), self._dev(sd[f"{hp}.enc_output.0.bias"].float()),
                               self._pack_frag(w2), self._dev(b2), ncp)
        # constant output_memory row of masked (invalid-anchor) tokens: LN(Linear(0)) = LN(bias)
        b0 = sd[f"{hp}.enc_output.0.bias"].float()
        row = torch.nn.functional.layer_norm(b0.to(torch.bfloat16).float(), (self.hd,), sd[f"{hp}.enc_output.1.weight"].float(),
                                             sd[f"{hp}.enc_output.1.bias"].float(), 1e-5)
        self.invalid_row = self._dev(row, torch.bfloat16)

        def mlp3(prefix, key):
            P[f"{key}.0"] = self._pack_linear(sd[f"{prefix}.layers.0.weight"], sd[f"{prefix}.layers.0.bias"])
            P[f"{key}.1"] = self._pack_linear(sd[f"{prefix}.layers.1.weight"], sd[f"{prefix}.layers.1.bias"])
            return self._dev(sd[f"{prefix}.layers.2.weight"].float()), self._dev(sd[f"{prefix}.layers.2.bias"].float())

        self.bbox_last: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}
        self.bbox_last["enc"] = mlp3(f"{hp}.enc_bbox_classifier", f"{hp}.enc_bbox")
        self.qpos0 = (self._dev(sd[f"{hp}.query_pos_head.layers.0.weight"].float()), self._dev(sd[f"{hp}.query_pos_head.layers.0.bias"].float()))
        P[f"{hp}.qpos1"] = self._pack_linear(sd[f"{hp}.query_pos_head.layers.1.weight"], sd[f"{hp}.query_pos_head.layers.1.bias"])
        vw, vb = [], []
        for li in range(self.nl):
            p = f"{hp}.decoder.layers.{li}"
            Wi, bi_ = sd[f"{p}.self_attn.in_proj_weight"], sd[f"{p}.self_attn.in_proj_bias"]
            P[f"{p}.qk"] = self._pack_linear(Wi[:512], bi_[:512])
            P[f"{p}.v"] = self._pack_linear(Wi[512:], bi_[512:])
            P[f"{p}.out_proj"] = self._pack_linear(sd[f"{p}.self_attn.out_proj.weight"], sd[f"{p}.self_attn.out_proj.bias"])
            ca = f"{p}.cross_attn"
            # sampling_offsets (192) and attention_weights (96) read the same query: one N=288 GEMM, fp32 out
            P[f"{ca}.offaw"] = self._pack_linear(torch.cat([sd[f"{ca}.sampling_offsets.weight"], sd[f"{ca}.attention_weights.weight"]], 0),
                                                 torch.cat([sd[f"{ca}.sampling_offsets.bias"], sd[f"{ca}.attention_weights.bias"]], 0))
            vw.append(sd[f"{ca}.value_proj.weight"])
            vb.append(sd[f"{ca}.value_proj.bias"])
            P[f"{ca}.output_proj"] = self._pack_linear(sd[f"{ca}.output_proj.weight"], sd[f"{ca}.output_proj.bias"])
            for l in ("linear1", "linear2"):
                P[f"{p}.{l}"] = self._pack_linear(sd[f"{p}.{l}.weight"], sd[f"{p}.{l}.bias"])
            for nrm in ("norm1", "norm2", "norm3"):
                ln(f"{p}.{nrm}")
            self.bbox_last[f"dec{li}"] = mlp3(f"{hp}.dec_bbox_classifier.{li}", f"{hp}.dec_bbox.{li}")
        # the decoder memory is the same for all layers: all value_proj's as ONE N = 6*256 GEMM
        P[f"{hp}.value_all"] = self._pack_linear(torch.cat(vw, 0), torch.cat(vb, 0))
        last = self.nl - 1
        P[f"{hp}.dec_score"] = self._pack_linear(sd[f"{hp}.dec_score_classifier.{last}.weight"], sd[f"{hp}.dec_score_classifier.{last}.bias"])
        # fragment-ordered copies of the decoder's row-local linears for fx_row_chain (one launch per chain of layers)
        RC: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}

        def rc(key, W, b, pad_to=None):
            W, b = W.float(), b.float()
            if pad_to is not None and W.shape[0] < pad_to:
                W = torch.cat([W, torch.zeros(pad_to - W.shape[0], W.shape[1])], 0)
                b = torch.cat([b, torch.zeros(pad_to - b.shape[0])], 0)
            RC[key] = (self._pack_frag(W), self._dev(b))

        rc("qpos1", sd[f"{hp}.query_pos_head.layers.1.weight"], sd[f"{hp}.query_pos_head.layers.1.bias"])
        for li in range(self.nl):
            p = f"{hp}.decoder.layers.{li}"
            Wi, bi_ = sd[f"{p}.self_attn.in_proj_weight"], sd[f"{p}.self_attn.in_proj_bias"]
            rc(f"{li}.qk", Wi[:512], bi_[:512])
            rc(f"{li}.v", Wi[512:], bi_[512:])
            rc(f"{li}.o", sd[f"{p}.self_attn.out_proj.weight"], sd[f"{p}.self_attn.out_proj.bias"])
            ca = f"{p}.cross_attn"
            rc(f"{li}.offaw", torch.cat([sd[f"{ca}.sampling_offsets.weight"], sd[f"{ca}.attention_weights.weight"]], 0),
               torch.cat([sd[f"{ca}.sampling_offsets.bias"], sd[f"{ca}.attention_weights.bias"]], 0))
            rc(f"{li}.o2", sd[f"{ca}.output_proj.weight"], sd[f"{ca}.output_proj.bias"])
            rc(f"{li}.f1", sd[f"{p}.linear1.weight"], sd[f"{p}.linear1.bias"])
            rc(f"{li}.f2", sd[f"{p}.linear2.weight"], sd[f"{p}.linear2.bias"])
            # RC_FFN_LN (round 5): both matrices in fragment order + the two bias vectors as one [b1 | b2] array
            RC[f"{li}.ffn_b"] = (None, self._dev(torch.cat([sd[f"{p}.linear1.bias"].float(), sd[f"{p}.linear2.bias"].float()])))
            self.dec_ffn = int(sd[f"{p}.linear1.weight"].shape[0])
            bb = f"{hp}.dec_bbox_classifier.{li}"
            rc(f"{li}.bb0", sd[f"{bb}.layers.0.weight"], sd[f"{bb}.layers.0.bias"])
            rc(f"{li}.bb1", sd[f"{bb}.layers.1.weight"], sd[f"{bb}.layers.1.bias"])
        self.nc_pad32 = (self.nc + 31) // 32 * 32
        rc("score", sd[f"{hp}.dec_score_classifier.{last}.weight"], sd[f"{hp}.dec_score_classifier.{last}.bias"], pad_to=self.nc_pad32)
        self.RC = RC
        self.P = P
        self.plans.clear()

    # ------------------------------------------------------------------ constants that depend on the input size
    @staticmethod
    def _pos_embed_sine(h: int, w: int, npf: int, temperature: float = 10000.0) -> torch.Tensor:
        """Sine position embedding of the AIFI layer, [h*w, 2*npf] = [y_sin | y_cos | x_sin | x_cos] (modelling.py:148-179)."""
        ys = torch.arange(h, dtype=torch.float32).view(h, 1).expand(h, w)
        xs = torch.arange(w, dtype=torch.float32).view(1, w).expand(h, w)
        i = torch.arange(npf, dtype=torch.float32)
        dim_t = temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / npf)
        px, py = xs[..., None] / dim_t, ys[..., None] / dim_t
        out = torch.cat([py[..., 0::2].sin(), py[..., 1::2].cos(), px[..., 0::2].sin(), px[..., 1::2].cos()], dim=-1)
        return out.reshape(h * w, 2 * npf)

    @staticmethod
    def _anchors(shapes: Sequence[Tuple[int, int]], grid_size: float = 0.05, eps: float = 1e-2):
        """Anchors (logit space) + invalid-token list (modelling.py:1169-1189)."""
        out = []
        for lvl, (h, w) in enumerate(shapes):
            gy, gx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
            xy = (torch.stack([gx, gy], -1) + 0.5) / torch.tensor([w, h], dtype=torch.float32)
            wh = torch.ones_like(xy) * grid_size * (2.0 ** (2 - lvl))
            out.append(torch.cat([xy, wh], -1).reshape(h * w, 4))
        a = torch.cat(out, 0)
        valid = ((a > eps) & (a < 1 - eps)).all(-1)
        a = torch.log(a / (1 - a))
        a = torch.where(valid[:, None], a, torch.zeros_like(a))
        return a, torch.nonzero(~valid).flatten().to(torch.int32)

    # ------------------------------------------------------------------ run
    def plan(self, B: int, H: int, W: int, f32_input: bool = False, nsplit: Optional[int] = None):
        """nsplit > 1: the batch is cut into nsplit parts whose launch sequences are captured as concurrent branches of
        one hipGraph (default from FX_STREAMS, see _MultiPlan)."""
        if nsplit is None:
            nsplit = int(os.environ.get("FX_STREAMS", str(DEFAULT_STREAMS)))
        if nsplit > 1 and not _lib.two_queue_safe():
            nsplit = 1
        while nsplit > 1 and (B % nsplit or B // nsplit < MIN_PART_BATCH):
            nsplit -= 1
        key = (B, H, W, f32_input, nsplit)
        if key not in self.plans:
            self.plans[key] = _Plan(self, B, H, W, f32_input) if nsplit <= 1 else _MultiPlan(self, _Plan, B, H, W, f32_input, nsplit)
        return self.plans[key]

    def pipeline(self, B: int, H: int, W: int, depth: Optional[int] = None, f32_input: bool = False, nsplit: int = 1) -> "_Pipeline":
        """Throughput mode (round 6): `depth` batches in flight, each on its OWN plan (own activation / output buffers) and its own stream
        of the device pool - batch i+1's backbone runs beside batch i's decoder tail, and a whole-batch plan (nsplit = 1) launches every
        layer once at twice the M of a half-batch part.  See _Pipeline.  depth: FX_PIPELINE_DEPTH, default 3."""
        if depth is None:
            depth = int(os.environ.get("FX_PIPELINE_DEPTH", str(DEFAULT_PIPELINE_DEPTH)))
        if not _lib.two_queue_safe():     # a packed-fp32 build must stay on one hardware queue (two-queue hazard)
            depth, nsplit = 1, 1
        key = ("pipeline", B, H, W, f32_input, nsplit, depth)
        if key not in self.plans:
            self.plans[key] = _Pipeline(self, _Plan, B, H, W, f32_input, depth, nsplit)
        return self.plans[key]

    def forward(self, images: torch.Tensor, sizes: Optional[torch.Tensor] = None, threshold: Optional[float] = None,
                forced_topk: Optional[torch.Tensor] = None, use_graph: bool = True) -> "_Plan":
        """images: uint8 [B,H,W,3] (fused normalise path) or float32 [B,H,W,3] (0..255 scale), on the engine device.
        Returns the plan whose output buffers (probs, boxes, det_*) hold the results (valid until the next call)."""
        assert images.dim() == 4 and images.shape[-1] == 3 and images.is_contiguous() and images.device == self.dev
        f32 = images.dtype == torch.float32
        assert f32 or images.dtype == torch.uint8
        B, H, W, _ = images.shape
        pl = self.plan(B, H, W, f32, 1 if (forced_topk is not None or not use_graph) else None)
        cur = torch.cuda.current_stream(self.dev)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            pl.input.copy_(images, non_blocking=True)
            if sizes is None:
                pl.sizes.copy_(torch.tensor([[H, W]] * B, dtype=torch.int32), non_blocking=False)
            else:
                pl.sizes.copy_(sizes.to(torch.int32), non_blocking=True)
            pl.run(self.stream.cuda_stream, threshold if threshold is not None else self.threshold, forced_topk, use_graph)
        cur.wait_stream(self.stream)
        return pl


class _PlanBase:
    """Buffers + the static launch sequence for one (batch, height, width); helpers shared by the model families."""

    size_multiple = 32
    needed_res = (2, 3, 4, 5)   # backbone stage outputs the plan reads as such (build_backbone drops the store of the others where a fused launch allows it)

    def __init__(self, eng: _EngineBase, B: int, H: int, W: int, f32_input: bool, parent: Optional["_MultiPlan"] = None, index: int = 0):
        # Every family runs at the image's own size with ceil(H/2) at every stride-2 layer (size_multiple = 1 in all three plans; the class
        # default stays 32 for plans that do not say otherwise).
        if H % self.size_multiple or W % self.size_multiple or H < 32 or W < 32:
            raise _lib.FocoosAmdError(f"input height/width must be multiples of {self.size_multiple} (and at least 32)")
        self.eng, self.B, self.H, self.W, self.f32_input = eng, B, H, W, f32_input
        self.parent, self.index = parent, index
        self.lib = eng.lib
        self.dev = eng.dev
        self.ops: List[Tuple] = []        # (fn, args) ; stream appended at call time
        self.split_at: Optional[int] = None  # ops[:split_at] end with the encoder top-k
        self.keep: List = []              # ctypes structs kept alive
        self.meta: Dict[int, dict] = {}   # op index -> {kind, variant, flops} for bench.py
        self.bufs: Dict[str, NT] = {}
        self._win = None
        self.graph = None
        self.graph_thr = None
        self._thr_cell = None
        self._build()

    # -------------------------------------------------------------- helpers
    def _new(self, name: str, B: int, H: int, W: int, Cc: int, dtype=torch.bfloat16) -> NT:
        if self._win is not None:
            # batch-window mode (front of the network run chunk by chunk): the buffer is allocated once for the full batch,
            # the op sees the contiguous NHWC slice of images [b0, b0+bs)
            b0, bs = self._win
            assert B == bs, (name, B, bs)
            full = self.bufs.get(name)
            if full is None:
                full = NT(torch.empty(self.B * H * W * Cc, dtype=dtype, device=self.dev), self.B, H, W, Cc, Cc, 0)
                self.bufs[name] = full
            return NT(full.t, bs, H, W, Cc, Cc, b0 * H * W * Cc)
        guard = int(os.environ.get("FX_GUARD", "0"))  # debugging aid: sentinel elements around every activation buffer
        if guard:
            t = torch.empty(B * H * W * Cc + 2 * guard, dtype=dtype, device=self.dev)
            t.view(torch.uint8).fill_(0x5A)
            nt = NT(t, B, H, W, Cc, Cc, guard)
            self.guards = getattr(self, "guards", [])
            self.guards.append((name, t, guard))
        else:
            t = torch.empty(B * H * W * Cc, dtype=dtype, device=self.dev)
            nt = NT(t, B, H, W, Cc, Cc, 0)
        self.bufs[name] = nt
        return nt

    def check_guards(self):
        """FX_GUARD=n: report buffers whose sentinel elements before / after the payload were overwritten."""
        bad = []
        for name, t, g in getattr(self, "guards", []):
            u = t.view(torch.uint8)
            es = t.element_size()
            head, tail = u[: g * es], u[-g * es:]
            if not bool((head == 0x5A).all()) or not bool((tail == 0x5A).all()):
                bad.append((name, int((head != 0x5A).sum()), int((tail != 0x5A).sum())))
        return bad

    def _op(self, fn, *args):
        self.ops.append((fn, args))

    def _io(self, name: str, shape: Tuple[int, ...], dtype) -> torch.Tensor:
        """Input / output tensor with a leading batch dimension.  Under a _MultiPlan it is this part's contiguous batch slice
        of one full-batch tensor, so callers see a single result buffer whichever way the step is split."""
        if self.parent is not None:
            return self.parent.io_slice(name, shape, dtype, self.index)
        return torch.empty(*shape, dtype=dtype, device=self.dev)

    def conv(self, x: NT, pc: PackedConv, out: Optional[NT] = None, name: Optional[str] = None, stride: int = 1, act=None,
             residual: Optional[NT] = None, res_after: bool = False, pool2: bool = False, out_f32: bool = False,
             y_batch_stride: int = 0, extra_flops_per_pixel: float = 0.0) -> NT:
        assert x.C == pc.C, (name, x.C, pc.C)
        pad = (pc.KH - 1) // 2
        if pool2:
            Ho, Wo = (x.H + 1) // 2, (x.W + 1) // 2
        else:
            Ho, Wo = (x.H + 2 * pad - pc.KH) // stride + 1, (x.W + 2 * pad - pc.KW) // stride + 1
        if out is None:
            ncols = (pc.N + 7) // 8 * 8
            out = self._new(name, x.B, Ho, Wo, ncols, torch.float32 if out_f32 else torch.bfloat16)
            out.C = pc.N
        assert out.H == Ho and out.W == Wo and out.C == pc.N, (name, out.H, Ho, out.C, pc.N)
        d = FxConvDesc()
        d.x, d.w, d.bias = x.ptr, pc.w.data_ptr(), pc.b.data_ptr()
        d.residual = residual.ptr if residual is not None else None
        d.y = out.ptr
        d.B, d.H, d.W, d.C, d.ldx = x.B, x.H, x.W, x.C, x.ld
        d.Ho, d.Wo, d.N, d.ldy = Ho, Wo, pc.N, out.ld
        d.ldr = residual.ld if residual is not None else 0
        d.KH, d.KW, d.stride, d.pad = pc.KH, pc.KW, stride, pad
        d.pool2, d.act, d.out_f32 = int(pool2), FX_ACT[act], int(out_f32)
        d.residual_after_act = int(res_after)
        d.y_batch_stride = y_batch_stride
        d.w_frag = pc.wf.data_ptr() if pc.wf is not None else None
        self.keep.append(d)
        # bookkeeping for bench.py: which template instance runs and the ALGORITHMIC flops of the reference layer(s)
        # this launch replaces (RepVGG 1x1 branch counted although it is re-parameterised away).
        M = x.B * Ho * Wo
        flops = 2.0 * M * pc.N * pc.KH * pc.KW * pc.C + extra_flops_per_pixel * M
        label = C.create_string_buffer(64)   # the library's own routing decision for this descriptor (fx_conv2d_variant), not a Python mirror of it
        check(self.lib.fx_conv2d_variant(C.byref(d), label, 64), "fx_conv2d_variant")
        variant = label.value.decode()
        # algorithmic (compulsory) HBM bytes of this launch: input read once, output written once, residual, weights
        alg_bytes = 2.0 * x.B * x.H * x.W * pc.C + (4.0 if out_f32 else 2.0) * M * pc.N + (2.0 * M * pc.N if residual is not None else 0.0) \
            + 2.0 * pc.N * pc.KH * pc.KW * pc.C
        self.meta[len(self.ops)] = {"kind": "conv", "variant": variant, "flops": flops, "flops_executed": flops - extra_flops_per_pixel * M, "bytes": alg_bytes,
                                    "name": name or "slice", "M": M, "N": pc.N, "K": pc.KH * pc.KW * pc.C}
        self._op(self.lib.fx_conv2d_nhwc_bf16, C.byref(d))
        return out

    def pw_chain(self, x1: NT, x2: Optional[NT], residual: Optional[NT], cc: Tuple, ca: Optional[Tuple], name1: str, name2: Optional[str],
                 pool_name: Optional[str] = None, keep_y1: bool = True):
        """y1 = relu([x1 | x2] W1^T + b1 (+ residual)), y2 = relu(y1 W2^T + b2) in one launch (include/focoos_amd.h, fx_pw_chain_desc):
        BottleNeck branch2c (+ shortcut conv) + add + ReLU and the next block's branch2a + ReLU (resnet.py:107-121)."""
        w1, b1, k1a, k1b, n1 = cc
        assert x1.C == k1a and (x2 is None) == (k1b == 0) and (x2 is None or x2.C == k1b), (name1, x1.C, k1a, k1b)
        M = x1.rows
        # keep_y1=False (with pool_name and ca): the block output is consumed only as the next block's branch2a input (y2) and as the pooled
        # shortcut input - y1 itself is neither allocated nor written (RT-DETR's res2: 105 MB per 16-image part at 640 x 640)
        assert keep_y1 or (pool_name is not None and ca is not None)
        y1 = self._new(name1, x1.B, x1.H, x1.W, n1) if keep_y1 else None
        d = FxPwChainDesc()
        d.x1, d.ldx1, d.K1a = x1.ptr, x1.ld, k1a
        d.x2, d.ldx2, d.K1b = (x2.ptr, x2.ld, k1b) if x2 is not None else (None, 0, 0)
        d.residual, d.ldr = (residual.ptr, residual.ld) if residual is not None else (None, 0)
        d.w1, d.bias1, d.N1, d.M = w1.data_ptr(), b1.data_ptr(), n1, M
        d.y1, d.ldy1 = (y1.ptr, y1.ld) if y1 is not None else (None, 0)
        d.act1 = d.act2 = FX_ACT["relu"]
        y2, n2 = None, 0
        if ca is not None:
            w2, b2, n2 = ca
            y2 = self._new(name2, x1.B, x1.H, x1.W, n2)
            d.w2, d.bias2, d.y2, d.ldy2 = w2.data_ptr(), b2.data_ptr(), y2.ptr, y2.ld
        d.N2 = n2
        pooled = None
        if pool_name is not None:   # AvgPool2d(2, 2) of y1 as a third output (the next stage's variant-d shortcut input): quad-ordered tiles
            pooled = self._new(pool_name, x1.B, x1.H // 2, x1.W // 2, n1)
            d.pool, d.ldp, d.img_h, d.img_w = pooled.ptr, pooled.ld, x1.H, x1.W
        self.keep.append(d)
        self.meta[len(self.ops)] = {"kind": "conv", "variant": f"pw_chain<{k1a},{k1b},{n2}>" + ("+pool" if pooled is not None else "") + ("-y1" if y1 is None else ""),
                                    "flops": 2.0 * M * n1 * (k1a + k1b + n2),
                                    "name": name1 + ("+" + name2.rsplit("res_layers.", 1)[-1] if name2 else ""), "M": M, "N": n1, "K": k1a + k1b,
                                    "bytes": 2.0 * M * (k1a + k1b + (n1 if residual is not None else 0) + (n1 if y1 is not None else 0) + n2
                                                        + (n1 / 4 if pooled is not None else 0))}
        self._op(self.lib.fx_pw_chain_bf16, C.byref(d))
        if pool_name is not None:
            return y1, y2, pooled
        return y1, y2

    def linear(self, x: NT, pc: PackedConv, **kw) -> NT:
        return self.conv(x.as_rows() if (x.H != 1 or x.W != 1) else x, pc, **kw)

    def layernorm(self, x: NT, name_ln: str, out_name: str, residual: Optional[NT] = None) -> NT:
        g, b = self.eng.ln[name_ln]
        cols = int(g.numel())     # 256, or 128 (the narrow pixel-decoder encoders of fai-mf-{m,s}-coco-ins)
        out = self._new(out_name, x.rows, 1, 1, cols)
        self._op(self.lib.fx_layernorm_bf16, x.ptr, x.ld, residual.ptr if residual is not None else None,
                 residual.ld if residual is not None else 0, g.data_ptr(), b.data_ptr(), out.ptr, out.ld, x.rows, cols)
        return out

    def add_rows(self, x: NT, y: NT, y_rows: int, out_name: str) -> NT:
        out = self._new(out_name, x.rows, 1, 1, x.C)
        self._op(self.lib.fx_add_rows_bf16, x.ptr, x.ld, y.ptr, y.ld, y_rows, out.ptr, out.ld, x.rows, x.C)
        return out

    # LDS map of the decoder's row chains (bytes): four [32][256] slots, one [32][1024] slot, the refined-box hand-over, LN scratch
    RC_S0, RC_S1, RC_S2, RC_S3, RC_BIG, RC_REF, RC_RED, RC_LDS = 0, 16384, 32768, 49152, 65536, 131072, 131584, 132608
    # Round 5 ("lean" programs, FX_RC_LEAN=1): no wide slot - the FFN is ONE stage whose hidden layer lives in two ping-pong [32][256] slots
    # (RC_FFN_LN), the query-position MLP's 512-wide hidden in two slots (flags bit 2) - 65 536 + 1 536 bytes + the kernel's 12 KiB scratch
    # = 79 360 <= 80 KiB: TWO workgroups per CU (the kernel is compiled for 128 registers per wave)
    RC2_REF, RC2_RED, RC2_LDS = 65536, 66048, 67072

    def _rc_program(self, stages: List[FxRcStage], rows: int, label: str, flops: float, lds: Optional[int] = None):
        """Upload a stage list and append the fx_row_chain launch."""
        arr = (FxRcStage * len(stages))(*stages)
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        dev = host.to(self.dev)
        self.keep.append(dev)
        self.meta[len(self.ops)] = {"kind": "conv", "variant": "row_chain", "flops": flops, "name": label, "M": rows, "N": 0, "K": 0}
        self._op(self.lib.fx_row_chain, dev.data_ptr(), len(stages), rows, self.RC_LDS if lds is None else lds)

    def resize(self, x: NT, out: NT):
        self._op(self.lib.fx_resize_bilinear_nhwc_bf16, x.ptr, x.ld, out.ptr, out.ld, x.B, x.H, x.W, x.C, out.H, out.W)

    def mha(self, qkv: NT, B: int, L: int, out_name: str) -> NT:
        out = self._new(out_name, B * L, 1, 1, 256)
        q, k, v = qkv.slice(0, 256), qkv.slice(256, 256), qkv.slice(512, 256)
        self._op(self.lib.fx_mha_bf16, q.ptr, q.ld, k.ptr, k.ld, v.ptr, v.ld, out.ptr, out.ld, B, L, L, 8)
        return out

    # -------------------------------------------------------------- shared front: input buffers + ResNet-vd
    def build_backbone(self) -> Dict[int, NT]:
        """ResNet.forward (resnet.py:252-266) -> {2: res2, 3: res3, 4: res4, 5: res5} (NHWC bf16)."""
        e, P, B, lib = self.eng, self.eng.P, self.B, self.lib
        H, W = self.H, self.W
        bb = "pixel_decoder.backbone"
        self.input = self._io("input", (B, H, W, 3), torch.float32 if self.f32_input else torch.uint8)
        self.sizes = self._io("sizes", (B, 2), torch.int32)
        # (A batch-chunked front of the network - stem .. res3 run chunk by chunk for Infinity-Cache residency - was measured
        # negative on MI355X at bs=32: 1/2/4/8 chunks = 2777/2732/2655/2449 img/s, and is gone.)
        blocks = RESNET_BLOCKS[e.depth]
        feats = {}
        # branch2c (+ shortcut conv) -> next block's branch2a as ONE launch (fx_pw_chain_bf16): the block output is written once
        # and consumed from LDS; FX_PW_CHAIN=0 restores one launch per layer, FX_PW_CHAIN_MAX_STAGE limits the stages covered.
        use_chain = int(os.environ.get("FX_PW_CHAIN", "1")) != 0
        max_stage = int(os.environ.get("FX_PW_CHAIN_MAX_STAGE", "1"))  # res2 + res3: HBM-bound seams; res4 measured neutral (profiles/r02a)
        pc2 = P[f"{bb}.conv1.conv1_2"]
        if os.environ.get("FX_STEM12_FUSE", "1") != "0" and not self.f32_input and pc2.wf is not None and pc2.N == 32:
            # normalise + conv1_1 + conv1_2 in one launch from the uint8 batch (csrc/stem12.hip, round 5): the conv1_1 activation is never written
            x = self._new("conv1_2", B, (H + 1) // 2, (W + 1) // 2, 32)
            M2 = x.rows
            self.meta[len(self.ops)] = {"kind": "conv", "variant": "stem_c1+c2", "flops": 2.0 * M2 * 32 * (27 + 288), "flops_executed": 2.0 * M2 * 32 * (32 * 1.29 + 288),
                                        "bytes": 3.0 * B * H * W + 2.0 * M2 * 32 + 2.0 * 32 * 288, "name": "conv1_1+conv1_2", "M": M2, "N": 32, "K": 288}
            self._op(lib.fx_stem_conv12_u8_bf16, self.input.data_ptr(), e.stem_w.data_ptr(), e.stem_b.data_ptr(), e.px_mean.data_ptr(), e.px_inv_std.data_ptr(),
                     pc2.wf.data_ptr(), pc2.b.data_ptr(), x.ptr, x.ld, B, H, W)
        else:
            c1 = self._new("conv1_1", B, (H + 1) // 2, (W + 1) // 2, 32)   # 3x3 / s2 / p1: ceil(H/2)
            self._op(lib.fx_stem_conv3x3s2, self.input.data_ptr(), int(self.f32_input), e.stem_w.data_ptr(), e.stem_b.data_ptr(),
                     e.px_mean.data_ptr(), e.px_inv_std.data_ptr(), c1.ptr, B, H, W, 32)
            x = self.conv(c1, pc2, name="conv1_2", act="relu")
        pc3 = P[f"{bb}.conv1.conv1_3"]
        if os.environ.get("FX_STEM_FUSE", "1") != "0" and pc3.wf is not None and lib.fx_stem_conv_pool_supported(x.C, pc3.N, x.H, x.W) == 1:
            # conv1_3 + ReLU + max-pool in one launch (csrc/stem_pool.hip, round 5): the [B,H/2,W/2,64] conv1_3 activation is never written
            mp = self._new("maxpool", B, (x.H + 1) // 2, (x.W + 1) // 2, 64)
            M3 = B * x.H * x.W
            self.meta[len(self.ops)] = {"kind": "conv", "variant": "stem_c3+pool", "flops": 2.0 * M3 * 64 * 288, "flops_executed": 2.0 * M3 * 64 * 288 * 1.37,
                                        "bytes": 2.0 * M3 * 32 + 2.0 * mp.rows * 64 + 2.0 * 64 * 288, "name": "conv1_3+maxpool", "M": M3, "N": 64, "K": 288}
            self._op(lib.fx_stem_conv3x3_relu_maxpool_bf16, x.ptr, x.ld, pc3.wf.data_ptr(), pc3.b.data_ptr(), mp.ptr, mp.ld, B, x.H, x.W)
        else:
            x = self.conv(x, pc3, name="conv1_3", act="relu")
            mp = self._new("maxpool", B, (x.H + 1) // 2, (x.W + 1) // 2, 64)
            self._op(lib.fx_maxpool3x3s2_nhwc_bf16, x.ptr, x.ld, mp.ptr, mp.ld, B, x.H, x.W, 64)
        x = mp
        seq = [(si, bi) for si in range(4) for bi in range(blocks[si])]
        a_next: Optional[NT] = None
        pooled_next: Optional[NT] = None   # AvgPool2d(2,2) of the previous stage's output, written by its last pw_chain launch (FX_PW_CHAIN_POOL=0: the stand-alone pass)
        use_pool = int(os.environ.get("FX_PW_CHAIN_POOL", "1")) != 0
        for idx, (si, bi) in enumerate(seq):
            p = f"{bb}.res_layers.{si}.blocks.{bi}"
            stride = 2 if (bi == 0 and si != 0) else 1
            a = a_next if a_next is not None else self.conv(x, P[f"{p}.branch2a"], name=f"{p}.a", act="relu")
            a_next = None
            bmid = self.conv(a, P[f"{p}.branch2b"], name=f"{p}.b", stride=stride, act="relu")
            short_in = None
            if bi == 0:
                short_in = x
                if stride == 2 and pooled_next is not None:
                    short_in, pooled_next = pooled_next, None
                elif stride == 2:
                    short_in = self._new(f"{p}.pool", x.B, (x.H + 1) // 2, (x.W + 1) // 2, x.C)
                    self._op(lib.fx_avgpool2x2_nhwc_bf16, x.ptr, x.ld, short_in.ptr, short_in.ld, x.B, x.H, x.W, x.C)
            nxt = seq[idx + 1] if idx + 1 < len(seq) else None
            cc = e.chain_c.get((si, bi)) if (use_chain and si <= max_stage) else None
            n2 = 0
            if cc is not None:
                ca = e.chain_a.get(nxt) if nxt is not None else None
                if ca is not None and lib.fx_pw_chain_supported(cc[2], cc[3], cc[4], ca[2]) == 1:
                    n2 = ca[2]
                elif lib.fx_pw_chain_supported(cc[2], cc[3], cc[4], 0) != 1:
                    cc = None
            if cc is not None:
                nxt_name = f"{bb}.res_layers.{nxt[0]}.blocks.{nxt[1]}.a" if n2 else None
                # last block of a stage whose successor starts with a stride-2 block: the pooled shortcut input comes out of this launch
                want_pool = (use_pool and nxt is not None and nxt[0] == si + 1 and bi != 0 and bmid.H % 2 == 0 and bmid.W % 2 == 0
                             and lib.fx_pw_chain_pool_supported(cc[2], cc[3], cc[4], n2) == 1)
                if want_pool:
                    # a stage output nobody reads as such (RT-DETR's res2: the encoder takes res3..res5) is not written: FX_SKIP_UNUSED_RES=0 keeps it
                    keep = bool(n2 == 0 or (si + 2) in self.needed_res or os.environ.get("FX_SKIP_UNUSED_RES", "1") == "0")
                    x, a_next, pooled_next = self.pw_chain(bmid, short_in, x, cc, e.chain_a[nxt] if n2 else None, f"{p}.c", nxt_name,
                                                           pool_name=f"{bb}.res_layers.{nxt[0]}.blocks.0.pool", keep_y1=keep)
                else:
                    x, a_next = self.pw_chain(bmid, short_in, None if bi == 0 else x, cc, e.chain_a[nxt] if n2 else None, f"{p}.c", nxt_name)
            else:
                short = x if bi != 0 else self.conv(short_in, P[f"{p}.short" if si == 0 else f"{p}.short.conv"], name=f"{p}.s")
                x = self.conv(bmid, P[f"{p}.branch2c"], name=f"{p}.c", residual=short, act="relu")
            if bi == blocks[si] - 1:
                feats[si + 2] = x
        for k, v in feats.items():
            if v is not None:
                self.bufs[f"res{k}"] = v
        return feats

    # -------------------------------------------------------------- execution
    def patch_args(self, fn, args, thr: float):
        """Per-call arguments that are not baked into the op list (the score threshold of the post-process op)."""
        return args

    def _launch(self, ops, stream: int, thr: float):
        for fn, args in ops:
            check(fn(*self.patch_args(fn, args, thr), C.c_void_p(stream)), fn.__name__)

    def capture_and_launch(self, stream: int, thr: float):
        if self.graph is None or self.graph_thr != thr:
            if self.graph is not None:
                check(self.lib.fx_graph_destroy(self.graph), "fx_graph_destroy")
                self.graph = None
            self._launch(self.ops, stream, thr)  # warm-up (first-use initialisation must not happen under capture)
            torch.cuda.current_stream(self.dev).synchronize()
            check(self.lib.fx_graph_begin(C.c_void_p(stream)), "fx_graph_begin")
            try:
                self._launch(self.ops, stream, thr)
            finally:
                g = C.c_void_p()
                rc = self.lib.fx_graph_end(C.c_void_p(stream), C.byref(g))
            check(rc, "fx_graph_end")
            self.graph, self.graph_thr = g, thr
        check(self.lib.fx_graph_launch(self.graph, C.c_void_p(stream)), "fx_graph_launch")

    def time_graph(self, stream: int, iters: int) -> float:
        ms = C.c_float(0)
        check(self.lib.fx_graph_time(self.graph, C.c_void_p(stream), iters, C.byref(ms)), "fx_graph_time")
        return ms.value

    def __del__(self):
        try:
            if self.graph is not None:
                self.lib.fx_graph_destroy(self.graph)
        except Exception:
            pass


class _Plan(StdcPlanMixin, _PlanBase):
    """RT-DETR launch sequence."""

    needed_res = (3, 4, 5)   # the hybrid encoder reads res3..res5 (modelling.py:297-347): res2 is only the input of res3's first block
    size_multiple = 1   # round 5: any size >= 32 with enough tokens for the query selection (ragged training batches are padded to the batch maximum)

    # -------------------------------------------------------------- the network
    def _build(self):
        e, P, B, lib = self.eng, self.eng.P, self.B, self.lib
        H, W = self.H, self.W
        feats = self.build_stdc() if e.stdc else self.build_backbone()
        fd = e.fd
        # ---- hybrid encoder (modelling.py:297-347)
        pd = "pixel_decoder"
        # level sizes = what the backbone produced (ceil(H/2) at every stride-2 layer, like the reference's: any H, W >= 32 - the reference's
        # encoder resizes with F.interpolate(size=...) in both directions, modelling.py:334,342, so nothing here needs multiples of 32)
        (h8, w8), (h16, w16), (h32, w32) = ((feats[k].H, feats[k].W) for k in (3, 4, 5))
        if h8 * w8 + h16 * w16 + h32 * w32 < e.nq:   # the reference's torch.topk raises "selected index k out of range" (modelling.py:1219)
            raise _lib.FocoosAmdError(f"input {H}x{W} gives {h8 * w8 + h16 * w16 + h32 * w32} encoder tokens, fewer than the {e.nq} queries to select")
        cat80 = self._new("cat80", B, h8, w8, 2 * fd)     # [up(lat1) | proj(res3)]
        cat40a = self._new("cat40a", B, h16, w16, 2 * fd)  # [up(lat0) | proj(res4)]
        cat40b = self._new("cat40b", B, h16, w16, 2 * fd)  # [downconv0 | lat1]
        cat20 = self._new("cat20", B, h32, w32, 2 * fd)    # [downconv1 | lat0]
        self.conv(feats[3], P[f"{pd}.input_proj.0"], out=cat80.slice(fd, fd))
        self.conv(feats[4], P[f"{pd}.input_proj.1"], out=cat40a.slice(fd, fd))
        src = self.conv(feats[5], P[f"{pd}.input_proj.2"], name="proj5")
        L = h32 * w32
        if e.n_enc > 0:
            pos = e._pos_embed_sine(h32, w32, 128).to(device=self.dev, dtype=torch.bfloat16).contiguous()
            self.pos = NT(pos, L, 1, 1, 256, 256)
            s = src.as_rows()
            for li in range(e.n_enc):
                p = f"{pd}.encoder.0.layers.{li}"
                qk_in = self.add_rows(s, self.pos, L, f"aifi{li}.qk_in")
                qkv = self._new(f"aifi{li}.qkv", B * L, 1, 1, 768)
                self.linear(qk_in, P[f"{p}.qk"], out=qkv.slice(0, 512))
                self.linear(s, P[f"{p}.v"], out=qkv.slice(512, 256))
                att = self.mha(qkv, B, L, f"aifi{li}.att")
                o = self.linear(att, P[f"{p}.out_proj"], name=f"aifi{li}.o", residual=s)
                s1 = self.layernorm(o, f"{p}.norm1", f"aifi{li}.s1")
                f1 = self.linear(s1, P[f"{p}.linear1"], name=f"aifi{li}.f1", act="gelu")
                f2 = self.linear(f1, P[f"{p}.linear2"], name=f"aifi{li}.f2", residual=s1)
                s = self.layernorm(f2, f"{p}.norm2", f"aifi{li}.s2")
            self.bufs["aifi"] = s
            src = NT(s.t, B, h32, w32, 256, 256, s.off)

        def csp(xin: NT, p: str, out_name: str) -> NT:
            c12 = self.conv(xin, P[f"{p}.conv12"], name=f"{p}.c12", act="silu")  # [x_1 | x_2]
            x1 = c12.slice(0, fd)
            for j in range(3):
                last = j == 2
                x1 = self.conv(x1, P[f"{p}.bottlenecks.{j}.rep"], name=out_name if last else f"{p}.rep{j}", act="silu",
                               residual=c12.slice(fd, fd) if last else None, res_after=True,
                               extra_flops_per_pixel=2.0 * fd * fd)  # the reference's separate 1x1 RepVGG branch
            return x1

        lat0 = self.conv(src, P[f"{pd}.lateral_convs.0"], out=cat20.slice(fd, fd), act="silu")
        self.resize(lat0, cat40a.slice(0, fd))
        fpn0 = csp(cat40a, f"{pd}.fpn_blocks.0", "fpn0")
        lat1 = self.conv(fpn0, P[f"{pd}.lateral_convs.1"], out=cat40b.slice(fd, fd), act="silu")
        self.resize(lat1, cat80.slice(0, fd))
        out80 = csp(cat80, f"{pd}.fpn_blocks.1", "enc_s8")
        d0 = self._new("down0", B, h16, w16, fd)
        self.resize(out80, d0)
        self.conv(d0, P[f"{pd}.downsample_convs.0"], out=cat40b.slice(0, fd), act="silu")
        out40 = csp(cat40b, f"{pd}.pan_blocks.0", "enc_s16")
        d1 = self._new("down1", B, h32, w32, fd)
        self.resize(out40, d1)
        self.conv(d1, P[f"{pd}.downsample_convs.1"], out=cat20.slice(0, fd), act="silu")
        out20 = csp(cat20, f"{pd}.pan_blocks.1", "enc_s32")
        # ---- predictor (modelling.py:1145-1263); memory rows in the reference order [s32 | s16 | s8]
        hp = "head.predictor"
        shapes = [(h32, w32), (h16, w16), (h8, w8)]
        S = sum(a * b for a, b in shapes)
        self.S = S
        starts = [0, shapes[0][0] * shapes[0][1], shapes[0][0] * shapes[0][1] + shapes[1][0] * shapes[1][1]]
        memory = self._new("memory", B, S, 1, 256)
        for i, (f, st) in enumerate(zip((out20, out40, out80), starts)):
            lvl = NT(memory.t, B, f.H, f.W, 256, 256, memory.off + st * 256)
            self.conv(f, P[f"{hp}.input_proj.{i}"], out=lvl, y_batch_stride=S * 256)
        mem_rows = memory.as_rows()
        anchors, invalid = e._anchors(shapes)
        self.anchors = anchors.to(self.dev).contiguous()
        self.invalid = invalid.to(self.dev).contiguous()
        self.shapes_t = torch.tensor(shapes, dtype=torch.int32, device=self.dev)
        self.starts_t = torch.tensor(starts, dtype=torch.int32, device=self.dev)
        Q, K = e.nq, e.nc
        self.enc_scores = self._io("enc_scores", (B, S), torch.float32)
        if e.score_head is not None and int(os.environ.get("FX_SCORE_HEAD", "1")) != 0:
            # enc_output (Linear + LayerNorm) + enc_score_classifier + max over classes in one launch; invalid anchors enter as zero rows
            w1, b1, w2, b2, ncp = e.score_head
            valid = torch.ones(S, dtype=torch.uint8)
            valid[invalid.long()] = 0
            self.valid_u8 = valid.to(self.dev)
            om = self._new("output_memory", B * S, 1, 1, 256)
            g_, b_ = e.ln[f"{hp}.enc_output.1"]
            self.meta[len(self.ops)] = {"kind": "conv", "variant": f"score_head<{ncp}>", "flops": 2.0 * B * S * 256 * (256 + K), "name": "enc_output+enc_score+max",
                                        "M": B * S, "N": 256 + K, "K": 256, "bytes": 2.0 * B * S * 256 * 2 + 4.0 * B * S}
            self._op(lib.fx_enc_score_head_bf16, mem_rows.ptr, mem_rows.ld, self.valid_u8.data_ptr(), S, w1.data_ptr(), b1.data_ptr(), g_.data_ptr(),
                     b_.data_ptr(), C.c_float(1e-5), w2.data_ptr(), b2.data_ptr(), ncp, om.ptr, om.ld, self.enc_scores.data_ptr(), B * S)
        else:
            om_lin = self.linear(mem_rows, P[f"{hp}.enc_output.0"], name="om_lin")
            om = self.layernorm(om_lin, f"{hp}.enc_output.1", "output_memory")
            self._op(lib.fx_fill_rows_bf16, om.ptr, om.ld, S, self.invalid.data_ptr(), int(self.invalid.numel()), e.invalid_row.data_ptr(), B, 256)
            enc_logits = self.linear(om, P[f"{hp}.enc_score"], name="enc_logits", out_f32=True)
            self._op(lib.fx_rowmax_f32, enc_logits.ptr, enc_logits.ld, self.enc_scores.data_ptr(), B * S, K)
        self.enc_topk_val = self._io("enc_topk_val", (B, Q), torch.float32)
        self.enc_topk = self._io("enc_topk", (B, Q), torch.int32)
        self._op(lib.fx_topk_rows_f32, self.enc_scores.data_ptr(), S, B, S, Q, self.enc_topk_val.data_ptr(), self.enc_topk.data_ptr())
        self.split_at = len(self.ops)
        R = B * Q
        tgt = self._new("target", R, 1, 1, 256)
        self._op(lib.fx_gather_rows_bf16, om.ptr, om.ld, S, self.enc_topk.data_ptr(), Q, tgt.ptr, tgt.ld, B, 256)
        hb = self.linear(tgt, P[f"{hp}.enc_bbox.0"], name="encbb.0", act="relu")
        hb = self.linear(hb, P[f"{hp}.enc_bbox.1"], name="encbb.1", act="relu")
        self.ref_unact = torch.empty(R, 4, dtype=torch.float32, device=self.dev)
        refs = [torch.empty(R, 4, dtype=torch.float32, device=self.dev) for _ in range(e.nl + 1)]
        wl, bl = e.bbox_last["enc"]
        self._op(lib.fx_bbox_head, hb.ptr, hb.ld, wl.data_ptr(), bl.data_ptr(), None, self.anchors.data_ptr(), self.enc_topk.data_ptr(), Q, 1,
                 refs[0].data_ptr(), self.ref_unact.data_ptr(), R, 256)
        value = self.linear(mem_rows, P[f"{hp}.value_all"], name="value_all")
        if int(os.environ.get("FX_ROW_CHAIN", "1")) != 0 and e.hd == 256:
            logits = self._build_decoder_row_chain(tgt, refs, value, R, B, Q, S)
        else:
            logits = self._build_decoder_per_op(tgt, refs, value, R, B, Q, S)
        self.refs = refs
        self.probs = self._io("probs", (B, Q, K), torch.float32)
        self.boxes = self._io("boxes", (B, Q, 4), torch.float32)
        self._op(lib.fx_detr_head_out, logits.ptr, logits.ld, refs[e.nl].data_ptr(), self.probs.data_ptr(), self.boxes.data_ptr(), R, K)
        # ---- device side of DETRProcessor.postprocess (processor.py:146-151,183-197)
        tk = min(e.top_k, Q * K)
        self.top_k = tk
        self.det_scores = self._io("det_scores", (B, tk), torch.float32)
        self.det_flat = self._io("det_flat", (B, tk), torch.int32)
        self.det_labels = self._io("det_labels", (B, tk), torch.int32)
        self.det_queries = self._io("det_queries", (B, tk), torch.int32)
        self.det_boxes = self._io("det_boxes", (B, tk, 4), torch.int32)
        self.det_count = self._io("det_count", (B,), torch.int32)
        nws = int(lib.fx_topk_rows_workspace_bytes(B, Q * K, tk))   # two-level exact top-k: chunk candidates first (B * 14 workgroups instead of B)
        self.topk_ws = torch.empty(max(nws, 8), dtype=torch.uint8, device=self.dev)
        self.keep.append(self.topk_ws)
        self._op(lib.fx_topk_rows_ws_f32, self.probs.data_ptr(), Q * K, B, Q * K, tk, self.det_scores.data_ptr(), self.det_flat.data_ptr(),
                 self.topk_ws.data_ptr(), C.c_size_t(self.topk_ws.numel()))
        self.post_index = len(self.ops)
        self._op(lib.fx_detr_postprocess, self.det_scores.data_ptr(), self.det_flat.data_ptr(), self.boxes.data_ptr(), self.sizes.data_ptr(), B, Q,
                 K, tk, None, self.det_labels.data_ptr(), self.det_queries.data_ptr(), self.det_boxes.data_ptr(), self.det_count.data_ptr())

    def _build_decoder_per_op(self, tgt: NT, refs, value: NT, R: int, B: int, Q: int, S: int) -> NT:
        """TransformerDecoder.forward (modelling.py:969-1020) with one launch per operation (FX_ROW_CHAIN=0)."""
        e, P, lib = self.eng, self.eng.P, self.lib
        hp = "head.predictor"
        qp1 = self._new("qpos_h", R, 1, 1, 512)
        for li in range(e.nl):
            p = f"{hp}.decoder.layers.{li}"
            ref = refs[li]
            self._op(lib.fx_linear_k4_relu, ref.data_ptr(), e.qpos0[0].data_ptr(), e.qpos0[1].data_ptr(), qp1.ptr, qp1.ld, R, 512)
            qpos = self.linear(qp1, P[f"{hp}.qpos1"], name=f"dec{li}.qpos")
            qk_in = self.add_rows(tgt, qpos, R, f"dec{li}.qk_in")
            qkv = self._new(f"dec{li}.qkv", R, 1, 1, 768)
            self.linear(qk_in, P[f"{p}.qk"], out=qkv.slice(0, 512))
            self.linear(tgt, P[f"{p}.v"], out=qkv.slice(512, 256))
            att = self.mha(qkv, B, Q, f"dec{li}.att")
            o = self.linear(att, P[f"{p}.out_proj"], name=f"dec{li}.o", residual=tgt)
            t1 = self.layernorm(o, f"{p}.norm1", f"dec{li}.t1")
            q2 = self.add_rows(t1, qpos, R, f"dec{li}.q2")
            offaw = self.linear(q2, P[f"{p}.cross_attn.offaw"], name=f"dec{li}.offaw", out_f32=True)
            ms = self._new(f"dec{li}.msda", R, 1, 1, 256)
            vsl = value.slice(li * 256, 256)
            self._op(lib.fx_msda_bf16, vsl.ptr, vsl.ld, self.shapes_t.data_ptr(), self.starts_t.data_ptr(), 3, 4, offaw.ptr, offaw.ld,
                     offaw.ptr + 192 * 4, offaw.ld, ref.data_ptr(), 1, ms.ptr, ms.ld, B, S, Q, 8)
            o2 = self.linear(ms, P[f"{p}.cross_attn.output_proj"], name=f"dec{li}.o2", residual=t1)
            t2 = self.layernorm(o2, f"{p}.norm2", f"dec{li}.t2")
            f1 = self.linear(t2, P[f"{p}.linear1"], name=f"dec{li}.f1", act="relu")
            f2 = self.linear(f1, P[f"{p}.linear2"], name=f"dec{li}.f2", residual=t2)
            tgt = self.layernorm(f2, f"{p}.norm3", f"dec{li}.out")
            hb = self.linear(tgt, P[f"{hp}.dec_bbox.{li}.0"], name=f"dec{li}.bb0", act="relu")
            hb = self.linear(hb, P[f"{hp}.dec_bbox.{li}.1"], name=f"dec{li}.bb1", act="relu")
            wl, bl = e.bbox_last[f"dec{li}"]
            self._op(lib.fx_bbox_head, hb.ptr, hb.ld, wl.data_ptr(), bl.data_ptr(), ref.data_ptr(), None, None, Q, 0, refs[li + 1].data_ptr(),
                     None, R, 256)
        logits = self.linear(tgt, P[f"{hp}.dec_score"], name="logits", out_f32=True)
        return logits

    def _build_decoder_row_chain(self, tgt: NT, refs, value: NT, R: int, B: int, Q: int, S: int) -> NT:
        """TransformerDecoder.forward (modelling.py:969-1020) as row chains: per layer  [self-attention core]  [out_proj + LN1 + offsets]
        [deformable sampling]  [output_proj + LN2 + FFN + LN3 + bbox refinement + the NEXT layer's query_pos / q,k,v projections]."""
        e, lib = self.eng, self.lib
        RC = e.RC
        S0, S1, S2, S3, BIG, REF, RED = self.RC_S0, self.RC_S1, self.RC_S2, self.RC_S3, self.RC_BIG, self.RC_REF, self.RC_RED
        relu = FX_ACT["relu"]

        def st(type_, K=0, N=0, act=0, src=-1, dst=-1, aux=-1, ld=0, ld2=0, flags=0, w=None, bias=None, gamma=None, beta=None, g0=None, g1=None):
            s_ = FxRcStage()
            s_.type, s_.K, s_.N, s_.act, s_.src, s_.dst, s_.aux, s_.ld, s_.ld2, s_.flags = type_, K, N, act, src, dst, aux, ld, ld2, flags
            s_.w, s_.bias, s_.gamma, s_.beta, s_.g0, s_.g1 = w, bias, gamma, beta, g0, g1
            return s_

        def load(nt: NT, dst):
            return st(0, K=nt.C, dst=dst, g0=nt.ptr, ld=nt.ld)

        def gemm(key, src, K, N, dst=-1, act=0, out: Optional[NT] = None, f32=False):
            w, b = RC[key]
            return st(1, K=K, N=N, act=act, src=src, dst=dst, w=w.data_ptr(), bias=b.data_ptr(), g0=out.ptr if out is not None else None,
                      ld=out.ld if out is not None else 0, flags=int(f32))

        def gemm_ln(key, src, K, aux, ln_name, dst, out: Optional[NT] = None):
            w, b = RC[key]
            g_, b_ = e.ln[ln_name]
            return st(2, K=K, N=256, src=src, dst=dst, aux=aux, w=w.data_ptr(), bias=b.data_ptr(), gamma=g_.data_ptr(), beta=b_.data_ptr(),
                      g0=out.ptr if out is not None else None, ld=out.ld if out is not None else 0, ld2=RED)

        hp = "head.predictor"
        qkv = self._new("dec.qkv", R, 1, 1, 768)
        qpos = self._new("dec.qpos", R, 1, 1, 256)
        t1 = self._new("dec.t1", R, 1, 1, 256)
        offaw = self._new("dec.offaw", R, 1, 1, 288, torch.float32)
        ncp = e.nc_pad32
        logits = self._new("logits", R, 1, 1, ncp, torch.float32)
        logits.C = e.nc

        hidden = int(e.dec_ffn)
        lean = os.environ.get("FX_RC_LEAN", "1") != "0" and hidden % 256 == 0 and hidden <= 2048
        if lean:
            REF, RED = self.RC2_REF, self.RC2_RED
        lds = self.RC2_LDS if lean else None

        def gemm_ln(key, src, K, aux, ln_name, dst, out: Optional[NT] = None):   # (re-defined: RED depends on the LDS map)
            w, b = RC[key]
            g_, b_ = e.ln[ln_name]
            return st(2, K=K, N=256, src=src, dst=dst, aux=aux, w=w.data_ptr(), bias=b.data_ptr(), gamma=g_.data_ptr(), beta=b_.data_ptr(),
                      g0=out.ptr if out is not None else None, ld=out.ld if out is not None else 0, ld2=RED)

        def ffn_ln(li, src, res, slot_a, slot_b, ln_name, dst, out: Optional[NT] = None):
            """RC_FFN_LN: LayerNorm(linear2(relu(linear1(x))) + residual) in one stage, hidden layer in two ping-pong [32][256] slots."""
            g_, b_ = e.ln[ln_name]
            return st(7, K=256, N=hidden, act=slot_a, flags=slot_b, src=src, aux=res, dst=dst, w=RC[f"{li}.f1"][0].data_ptr(), g1=RC[f"{li}.f2"][0].data_ptr(),
                      bias=RC[f"{li}.ffn_b"][1].data_ptr(), gamma=g_.data_ptr(), beta=b_.data_ptr(), g0=out.ptr if out is not None else None,
                      ld=out.ld if out is not None else 0, ld2=RED)

        def pre_attention(li, tgt_slot, ref_global=None):
            """query_pos_head(ref) -> qpos; q = k = (tgt + qpos) W_qk, v = tgt W_v  (modelling.py:996, transformer MHA in_proj)."""
            if lean:   # tgt in S3; the 512-wide hidden of the query-position MLP in (S0, S1), qpos -> S2, tgt + qpos -> S0
                assert tgt_slot == S3
                k4 = st(4, N=512, dst=S0, ld2=S1, flags=4, aux=(-1 if ref_global is not None else REF), w=e.qpos0[0].data_ptr(), bias=e.qpos0[1].data_ptr(),
                        g0=ref_global)
                w, b = RC["qpos1"]
                q1 = st(1, K=512, N=256, src=S0, aux=S1, dst=S2, flags=4, w=w.data_ptr(), bias=b.data_ptr(), g0=qpos.ptr, ld=qpos.ld)
                return [k4, q1, st(3, K=256, src=S3, aux=S2, dst=S0), gemm(f"{li}.qk", S0, 256, 512, out=qkv.slice(0, 512)),
                        gemm(f"{li}.v", S3, 256, 256, out=qkv.slice(512, 256))]
            k4 = st(4, N=512, dst=BIG, aux=(-1 if ref_global is not None else REF), w=e.qpos0[0].data_ptr(), bias=e.qpos0[1].data_ptr(), g0=ref_global)
            return [k4, gemm("qpos1", BIG, 512, 256, dst=S1, out=qpos), st(3, K=256, src=tgt_slot, aux=S1, dst=S2),
                    gemm(f"{li}.qk", S2, 256, 512, out=qkv.slice(0, 512)), gemm(f"{li}.v", tgt_slot, 256, 256, out=qkv.slice(512, 256))]

        fl_pre = 2.0 * R * (512 * 256 + 256 * 768)
        first_slot = S3 if lean else S0
        self._rc_program([load(tgt, first_slot)] + pre_attention(0, first_slot, refs[0].data_ptr()), R, "dec0.pre", fl_pre, lds)
        for li in range(e.nl):
            p = f"{hp}.decoder.layers.{li}"
            att = self.mha(qkv, B, Q, f"dec{li}.att")
            prog = [load(att, S0), load(tgt, S1), load(qpos, S2), gemm_ln(f"{li}.o", S0, 256, S1, f"{p}.norm1", S3, out=t1),
                    st(3, K=256, src=S3, aux=S2, dst=S0), gemm(f"{li}.offaw", S0, 256, 288, out=offaw, f32=True)]
            self._rc_program(prog, R, f"dec{li}.post_attn", 2.0 * R * 256 * (256 + 288), lds)
            ms = self._new(f"dec{li}.msda", R, 1, 1, 256)
            vsl = value.slice(li * 256, 256)
            self._op(lib.fx_msda_bf16, vsl.ptr, vsl.ld, self.shapes_t.data_ptr(), self.starts_t.data_ptr(), 3, 4, offaw.ptr, offaw.ld,
                     offaw.ptr + 192 * 4, offaw.ld, refs[li].data_ptr(), 1, ms.ptr, ms.ld, B, S, Q, 8)
            out = self._new(f"dec{li}.out", R, 1, 1, 256)
            wl, bl = e.bbox_last[f"dec{li}"]
            last = li == e.nl - 1
            if lean:   # ms S0, t1 S1 -> LN2 S2 -> FFN (chunks in S0 / S1) -> LN3 S3 (+ global) -> bbox MLP through S0, S1
                ffn = [ffn_ln(li, S2, S2, S0, S1, f"{p}.norm3", S3, out=out)]
            else:
                ffn = [gemm(f"{li}.f1", S2, 256, 1024, dst=BIG, act=relu), gemm_ln(f"{li}.f2", BIG, 1024, S2, f"{p}.norm3", S3, out=out)]
            prog = [load(ms, S0), load(t1, S1), gemm_ln(f"{li}.o2", S0, 256, S1, f"{p}.norm2", S2)] + ffn + [
                    gemm(f"{li}.bb0", S3, 256, 256, dst=S0, act=relu), gemm(f"{li}.bb1", S0, 256, 256, dst=S1, act=relu),
                    st(5, K=256, src=S1, aux=(-1 if last else REF), w=wl.data_ptr(), bias=bl.data_ptr(), g0=refs[li].data_ptr(), g1=refs[li + 1].data_ptr())]
            fl = 2.0 * R * (256 * 256 * 3 + 2 * 256 * 1024)
            if last:
                prog.append(gemm("score", S3, 256, ncp, out=logits, f32=True))
                fl += 2.0 * R * 256 * e.nc
            else:
                prog += pre_attention(li + 1, S3)
                fl += fl_pre
            self._rc_program(prog, R, f"dec{li}.post_msda", fl, lds)
            tgt = out
        return logits

    # -------------------------------------------------------------- execution
    def patch_args(self, fn, args, thr: float):
        if fn is self.lib.fx_detr_postprocess:
            return args[:8] + (C.c_float(thr),) + args[9:]
        return args

    def run(self, stream: int, thr: float, forced_topk: Optional[torch.Tensor] = None, use_graph: bool = True):
        if forced_topk is not None:
            self._launch(self.ops[: self.split_at], stream, thr)
            self.enc_topk.copy_(forced_topk.to(device=self.dev, dtype=torch.int32))
            self._launch(self.ops[self.split_at:], stream, thr)
            return
        if not use_graph:
            self._launch(self.ops, stream, thr)
            return
        self.capture_and_launch(stream, thr)


class _Pipeline:
    """`depth` batches in flight (throughput mode; engine.pipeline()).

    ``forward()`` is the latency form: ONE batch at a time, cut into two half-batch parts that run concurrently (_MultiPlan) - its decoder
    tail (300 row tiles per part, launches of 15-60 us that cannot fill 256 CUs) ends before the next batch may start.  A serving loop
    that has the next batch ready does not need that: here lane j is a complete plan of the WHOLE batch with its own buffers, replayed on
    stream j of the device pool; ``submit`` hands batch i to lane i % depth and returns at once, so the backbone of batch i+1 runs beside
    the encoder / decoder of batch i on the same CUs, and every layer is ONE launch at M = B x H x W instead of two at half of it (sum of
    kernel time of a bs = 32 RT-DETR step 8.0 ms instead of 9.0 for the two parts).  Measured in one call on one box (profiles/r06_pipeline_ab.txt):
    forward()-style steps 4 803-4 809 img/s, depth 2 / 3 / 4 with whole-batch plans 5 002-5 133, two-part plans with depth 3 4 940-4 990.
    The lanes are independent (nothing is shared but the read-only weights), so results are those of a plain plan, bit for bit
    (tests/test_gpu_e2e.py::test_pipeline_lanes_equal_single_plan).

    A lane's outputs stay valid until the lane is submitted to again (depth submits later); ``host`` buffers (pinned) receive the packed
    detections on the lane's stream, ``wait(ticket)`` blocks until that batch is complete."""

    def __init__(self, eng: _EngineBase, plan_cls, B: int, H: int, W: int, f32_input: bool, depth: int, nsplit: int = 1, **kw):
        self.eng, self.B, self.H, self.W, self.depth, self.nsplit = eng, B, H, W, max(1, depth), nsplit
        self.dev = eng.dev
        self.lanes = []
        lane_streams = _concurrent_streams(self.dev, self.depth) if nsplit <= 1 else None
        for j in range(self.depth):
            if nsplit > 1:
                pl = _MultiPlan(eng, plan_cls, B, H, W, f32_input, nsplit, stream_offset=nsplit * j, **kw)
                st = eng.stream if j == 0 else _device_stream(self.dev, nsplit * j)
            else:
                pl = plan_cls(eng, B, H, W, f32_input, **kw)
                st = lane_streams[j]
            self.lanes.append((pl, st))
        self.tickets = 0
        self._events = [None] * self.depth

    def lane(self, ticket: int):
        return self.lanes[ticket % self.depth]

    def submit(self, images: torch.Tensor, sizes: Optional[torch.Tensor] = None, threshold: Optional[float] = None, host: Optional[dict] = None) -> int:
        """Enqueue one batch (device tensors; `images` must stay untouched until the lane's copy ran - it is read on the lane's stream after
        the caller's current stream reached this point).  Returns the ticket; the plan is ``lane(ticket)[0]``."""
        t = self.tickets
        self.tickets += 1
        pl, st = self.lanes[t % self.depth]
        st.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(st):
            pl.input.copy_(images, non_blocking=True)
            if sizes is None:
                pl.sizes.copy_(torch.tensor([[self.H, self.W]] * self.B, dtype=torch.int32), non_blocking=False)
            else:
                pl.sizes.copy_(sizes, non_blocking=True)
            pl.run(st.cuda_stream, threshold if threshold is not None else self.eng.threshold, None, True)
            if host is not None:
                for k, h in host.items():
                    h.copy_(getattr(pl, k), non_blocking=True)
            ev = self._events[t % self.depth]
            if ev is None:
                ev = self._events[t % self.depth] = torch.cuda.Event()
            ev.record(st)
        return t

    def wait(self, ticket: int):
        ev = self._events[ticket % self.depth]
        if ev is not None:
            ev.synchronize()
        return self.lanes[ticket % self.depth][0]

    def synchronize(self):
        for _, st in self.lanes:
            st.synchronize()


class _MultiPlan:
    """One step = `n` batch parts (default 2, FX_STREAMS), each a complete plan of B/n images with its own activation buffers, replayed
    concurrently on `n` streams so that an HBM-bound layer of one part overlaps an MFMA-bound layer of another and fills its tail
    wave: RT-DETR bs=32 3281 -> 3580 img/s.  Round 1 measured this and REJECTED it for correctness (30 of 40 replays had wrong decoder
    rows); round 2 root-caused it: kernels containing packed-fp32 VALU instructions (v_pk_*_f32 / v_pk_mov_b32) compute wrong values in
    lanes 48-63 when waves of a second hardware queue share their CU (first victim: fx_bbox_head; nothing is shared between the parts,
    inputs bit-identical, per-lane dumps in DESIGN.md §5).  The library is compiled without that instruction class (build.py): 0 of 60
    concurrent replays differ from the serial result (tests/test_gpu_two_streams.py asserts it).
    Inputs and outputs are single full-batch tensors (each part reads / writes its contiguous batch slice)."""

    def __init__(self, eng: _EngineBase, plan_cls, B: int, H: int, W: int, f32_input: bool, n: int, stream_offset: int = 0, **kw):
        self.eng, self.B, self.H, self.W, self.n = eng, B, H, W, n
        self.lib, self.dev = eng.lib, eng.dev
        self._io_full: Dict[str, torch.Tensor] = {}
        self.parts = [plan_cls(eng, B // n, H, W, f32_input, parent=self, index=i, **kw) for i in range(n)]
        # stream_offset: a second plan of the same shape in flight beside the first one (bench.py --pipeline: batch i+1's backbone beside batch
        # i's decoder tail) takes its own streams of the pool
        self.side = [_device_stream(self.dev, 1 + i + stream_offset) for i in range(n - 1)]
        if os.environ.get("FX_PARTS_SERIAL") == "1":   # profiling aid: the same parts back to back on ONE stream (per-kernel durations without overlap)
            self.side = [eng.stream] * (n - 1)
        self.graph = None
        self.graph_thr = None
        # bench.py's per-op view: the parts' launch lists back to back
        self.ops = [op for p in self.parts for op in p.ops]
        self.meta = {}
        off = 0
        for p in self.parts:
            for i, m in p.meta.items():
                self.meta[off + i] = m
            off += len(p.ops)

    def io_slice(self, name: str, shape, dtype, index: int) -> torch.Tensor:
        bp = shape[0]
        full = self._io_full.get(name)
        if full is None:
            full = torch.empty(bp * self.n, *shape[1:], dtype=dtype, device=self.dev)
            self._io_full[name] = full
        return full[index * bp:(index + 1) * bp]

    def __getattr__(self, name):
        io = self.__dict__.get("_io_full", {})
        if name in io:
            return io[name]
        parts = self.__dict__.get("parts")
        if parts and name in ("top_k", "S", "post_index", "levels", "W32", "full_masks"):
            return getattr(parts[0], name)
        raise AttributeError(name)

    def patch_args(self, fn, args, thr: float):
        return self.parts[0].patch_args(fn, args, thr)

    def run(self, stream: int, thr: float, forced=None, use_graph: bool = True):
        if forced is not None:
            raise _lib.FocoosAmdError("teacher forcing runs on a single-part plan (plan(..., nsplit=1))")
        if not use_graph:
            for p in self.parts:
                p._launch(p.ops, stream, thr)
            return
        mode = os.environ.get("FX_MULTI_MODE", "graphs")
        if mode == "branches":
            return self._run_branches(stream, thr)
        # One linear graph per part, each replayed on its own stream (part 0 on the caller's stream), joined by events.
        # (A single graph with the parts as concurrent BRANCHES - FX_MULTI_MODE=branches - is ~1 % faster but on ROCm 7.2 it
        # occasionally lets a small decoder kernel read stale data: 1 image in 32 differed in 1 of 4 replays.  Per-stream
        # linear graphs use the ordinary in-queue ordering and were bit-stable over hundreds of replays.)
        if self.graph is None or self.graph_thr != thr:
            for g in (self.graph or []):
                check(self.lib.fx_graph_destroy(g), "fx_graph_destroy")
            self.graph = None
            graphs = []
            streams = [stream] + [s.cuda_stream for s in self.side]
            for p in self.parts:  # warm-up outside capture
                p._launch(p.ops, stream, thr)
            torch.cuda.current_stream(self.dev).synchronize()
            for p, st in zip(self.parts, streams):
                check(self.lib.fx_graph_begin(C.c_void_p(st)), "fx_graph_begin")
                try:
                    p._launch(p.ops, st, thr)
                finally:
                    g = C.c_void_p()
                    rc = self.lib.fx_graph_end(C.c_void_p(st), C.byref(g))
                check(rc, "fx_graph_end")
                graphs.append(g)
            self.graph, self.graph_thr = graphs, thr
        for s in self.side:
            check(self.lib.fx_stream_fork(C.c_void_p(stream), C.c_void_p(s.cuda_stream)), "fx_stream_fork")
        check(self.lib.fx_graph_launch(self.graph[0], C.c_void_p(stream)), "fx_graph_launch")
        for g, s in zip(self.graph[1:], self.side):
            check(self.lib.fx_graph_launch(g, C.c_void_p(s.cuda_stream)), "fx_graph_launch")
        for s in self.side:
            check(self.lib.fx_stream_join(C.c_void_p(stream), C.c_void_p(s.cuda_stream)), "fx_stream_join")

    def _run_branches(self, stream: int, thr: float):
        if self.graph is None or self.graph_thr != thr:
            if self.graph is not None:
                check(self.lib.fx_graph_destroy(self.graph), "fx_graph_destroy")
                self.graph = None
            for p in self.parts:  # warm-up (first-use initialisation must not happen under capture)
                p._launch(p.ops, stream, thr)
            torch.cuda.current_stream(self.dev).synchronize()
            check(self.lib.fx_graph_begin(C.c_void_p(stream)), "fx_graph_begin")
            try:
                for s in self.side:
                    check(self.lib.fx_stream_fork(C.c_void_p(stream), C.c_void_p(s.cuda_stream)), "fx_stream_fork")
                self.parts[0]._launch(self.parts[0].ops, stream, thr)
                for p, s in zip(self.parts[1:], self.side):
                    p._launch(p.ops, s.cuda_stream, thr)
                for s in self.side:
                    check(self.lib.fx_stream_join(C.c_void_p(stream), C.c_void_p(s.cuda_stream)), "fx_stream_join")
            finally:
                g = C.c_void_p()
                rc = self.lib.fx_graph_end(C.c_void_p(stream), C.byref(g))
            check(rc, "fx_graph_end")
            self.graph, self.graph_thr = g, thr
        check(self.lib.fx_graph_launch(self.graph, C.c_void_p(stream)), "fx_graph_launch")

    def __del__(self):
        try:
            for g in (self.graph if isinstance(self.graph, list) else [self.graph]):
                if g is not None:
                    self.lib.fx_graph_destroy(g)
        except Exception:
            pass
