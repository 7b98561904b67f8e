"""Per-kernel parity of the HIP C-ABI (libfocoos_amd.so) on a real MI355X against CPU fp32 references:
the oracle's functions where the reference has a named function for the op (ms_deform_attn_core,
postprocess, inverse_sigmoid ...) and the plain torch op the reference calls otherwise (F.conv2d,
F.interpolate, F.max_pool2d, F.layer_norm, torch.topk).  All calls go through ctypes -> extern "C"."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from focoos_amd import _lib  # noqa: E402
from focoos_amd._lib import FX_ACT, FxConvDesc, check  # noqa: E402
from oracle import detr_oracle as O  # noqa: E402
from tests._cases import MSDA_SHAPES, msda_case_inputs  # noqa: E402
from tests.helpers import load_golden  # noqa: E402

DEV = "cuda:0"


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return _lib.load()


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def bf(t):
    return t.to(torch.bfloat16)


def to_dev(t, dtype=None):
    return t.to(device=DEV, dtype=dtype or t.dtype).contiguous()


def pack_w(W4, bias):
    N, Cc, KH, KW = W4.shape
    Np = (N + 127) // 128 * 128
    w = torch.zeros(Np, KH, KW, Cc)
    w[:N] = W4.permute(0, 2, 3, 1)
    b = torch.zeros(Np)
    if bias is not None:
        b[:N] = bias
    return to_dev(w, torch.bfloat16), to_dev(b)


def run_conv(lib, x_nhwc, W4, bias, stride=1, act=None, residual=None, res_after=False, pool2=False, out_f32=False, ldx=None, ldy=None,
             ybs=0, y_buf=None, y_off=0, frag=False):
    """x_nhwc: [B,H,W,C] float (will be rounded to bf16). Returns y [B,Ho,Wo,N] float32 (cpu)."""
    B, H, W, Cc = x_nhwc.shape
    N, _, KH, KW = W4.shape
    pad = (KH - 1) // 2
    Ho, Wo = ((H + 1) // 2, (W + 1) // 2) if pool2 else ((H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1)
    ldx = ldx or Cc
    xb = torch.zeros(B, H, W, ldx, dtype=torch.bfloat16)
    xb[..., :Cc] = bf(x_nhwc)
    xd = to_dev(xb)
    wd, bd = pack_w(W4, bias)
    N8 = (N + 7) // 8 * 8
    ldy = ldy or N8
    yd = torch.full((B, Ho, Wo, ldy), float("nan"), dtype=torch.float32 if out_f32 else torch.bfloat16, device=DEV) if y_buf is None else y_buf
    d = FxConvDesc()
    d.x, d.w, d.bias, d.y = xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), yd.data_ptr() + y_off * yd.element_size()
    rd = None
    if residual is not None:
        rd = to_dev(bf(residual))
        d.residual, d.ldr = rd.data_ptr(), residual.shape[-1]
    d.B, d.H, d.W, d.C, d.ldx = B, H, W, Cc, ldx
    d.Ho, d.Wo, d.N, d.ldy = Ho, Wo, N, ldy
    d.KH, d.KW, d.stride, d.pad = KH, KW, stride, pad
    d.pool2, d.act, d.out_f32, d.residual_after_act, d.y_batch_stride = int(pool2), FX_ACT[act], int(out_f32), int(res_after), ybs
    wf = None
    if frag:  # second weight copy in MFMA fragment order -> 3x3/s1 layers run on the halo kernel (conv3x3_flat.hip)
        wf = frag_pack(W4.permute(0, 2, 3, 1).reshape(N, KH * KW * Cc))
        d.w_frag = wf.data_ptr()
    check(lib.fx_conv2d_nhwc_bf16(C.byref(d), stream()), "conv")
    torch.cuda.synchronize()
    return yd.float().cpu()


def ref_conv(x_nhwc, W4, bias, stride=1, act=None, residual=None, res_after=False, pool2=False):
    x = bf(x_nhwc).float().permute(0, 3, 1, 2)
    Wq = bf(W4).float()
    if pool2:
        x = bf(F.avg_pool2d(x, 2, 2, 0, ceil_mode=True)).float()
    y = F.conv2d(x, Wq, bias, stride=stride, padding=(W4.shape[-1] - 1) // 2)
    r = bf(residual).float().permute(0, 3, 1, 2) if residual is not None else None
    if r is not None and not res_after:
        y = y + r
    y = O.apply_act(y, act)
    if r is not None and res_after:
        y = y + r
    return y.permute(0, 2, 3, 1)


CONV_CASES = [
    # B,H,W,C,N,k,stride,act,residual,res_after,pool2,out_f32
    (2, 20, 20, 64, 256, 1, 1, "relu", True, False, False, False),
    (1, 24, 40, 128, 128, 3, 1, "relu", False, False, False, False),
    (2, 16, 16, 128, 128, 3, 2, "relu", False, False, False, False),
    (1, 13, 9, 256, 512, 1, 1, None, False, False, True, False),     # odd dims: ceil-mode pooled shortcut, M tail
    (2, 16, 16, 256, 512, 1, 1, None, False, False, True, False),
    (1, 30, 30, 32, 32, 3, 1, "relu", False, False, False, False),    # stem conv1_2 shape class (BK=32, BN=32)
    (1, 30, 30, 32, 64, 3, 1, "relu", False, False, False, False),    # stem conv1_3 (BK=32, BN=64)
    (1, 10, 10, 64, 64, 3, 1, "relu", False, False, False, False),    # BN=64
    (1, 8, 8, 64, 32, 1, 1, "silu", False, False, False, False),      # BN=32, BK=64
    (3, 10, 10, 256, 256, 3, 1, "silu", True, True, False, False),    # RepVGG/CSP: act then add
    (1, 1, 300, 256, 365, 1, 1, None, False, False, False, True),     # score head: N tail, fp32 out
    (1, 1, 77, 256, 1024, 1, 1, "gelu", False, False, False, False),
    (1, 1, 200, 1024, 256, 1, 1, None, True, False, False, False),
    (1, 1, 50, 32, 160, 1, 1, None, False, False, False, False),      # BK=32, BN=128
    # large-M deep-K shapes -> conv_igemm_dma (buffer_load..lds, 256-row tiles); M tails, padding, residual-after-act
    (5, 91, 90, 128, 256, 3, 1, "silu", True, True, False, False),    # 256x256 tile, M = 40950 (tail), 3x3 zero padding
    (2, 145, 142, 128, 128, 3, 1, "relu", False, False, False, False),  # 256x128 tile
    (1, 1, 40100, 1024, 384, 1, 1, None, True, False, False, False),  # N = 3 x 128, residual before act
    (11, 121, 123, 256, 256, 3, 2, "relu", False, False, False, False),  # stride 2, 3x3, M = 41602
    # large-M short-K pointwise layers -> conv_pw_stream (register-resident W, LDS-free, permlane32_swap epilogue)
    (2, 97, 95, 64, 256, 1, 1, "relu", True, False, False, False),     # res2 "c": M = 18430 (tail of 30 rows), residual
    (1, 130, 131, 128, 512, 1, 1, "relu", True, False, False, False),  # res3 "c"
    (1, 129, 130, 256, 1024, 1, 1, "silu", True, True, False, False),  # K = 256, residual after act
    (1, 1, 16500, 256, 365, 1, 1, None, False, False, False, True),    # N tail (368 stored), fp32 out
    (1, 1, 17000, 256, 288, 1, 1, None, False, False, False, True),    # 4.5 channel blocks
    (3, 80, 80, 256, 64, 1, 1, "relu", False, False, False, False),    # N = one channel block
    (1, 128, 130, 64, 72, 1, 1, "gelu", False, False, False, False),   # N = 72: second block partly stored
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_igemm(lib, case):
    B, H, W, Cc, N, k, stride, act, has_res, res_after, pool2, out_f32 = case
    g = torch.Generator().manual_seed(100 + CONV_CASES.index(case))
    x = torch.randn(B, H, W, Cc, generator=g)
    W4 = torch.randn(N, Cc, k, k, generator=g) / math.sqrt(Cc * k * k)
    bias = torch.randn(N, generator=g) * 0.5
    ref0 = ref_conv(x, W4, bias, stride, None, None, False, pool2)
    res = torch.randn(ref0.shape, generator=g) if has_res else None
    ref = ref_conv(x, W4, bias, stride, act, res, res_after, pool2)
    got = run_conv(lib, x, W4, bias, stride, act, res, res_after, pool2, out_f32)[..., :N]
    tol = 2e-4 if out_f32 else 1.2e-2  # bf16 output rounding is 2^-8 relative
    err = (got - ref).abs().max() / ref.abs().max()
    assert err < tol, f"rel err {err}"
    assert not torch.isnan(got).any()


def test_conv_asymmetric_identity(lib):
    """Transpose-detecting check (guide rule 16): identity activations, asymmetric weights."""
    Cc, N = 64, 128
    x = torch.eye(Cc).view(1, 1, Cc, Cc)  # pixel m has a one at channel m
    W4 = (torch.arange(N * Cc).float().view(N, Cc, 1, 1) % 251) / 64.0
    got = run_conv(lib, x, W4, None)[0, 0, :, :N]  # [m, n] = W[n, m]
    assert torch.allclose(got, bf(W4).float().view(N, Cc).t(), atol=0, rtol=2e-2)


def test_conv_slices_and_batch_stride(lib):
    """Reading a channel slice (ldx > C), writing a channel slice of a concat buffer (ldy > N) and a
    per-image row offset into the [B, S, C] decoder memory (y_batch_stride)."""
    g = torch.Generator().manual_seed(3)
    B, H, W, Cc, N = 2, 6, 5, 64, 256
    x = torch.randn(B, H, W, Cc, generator=g)
    W4 = torch.randn(N, Cc, 1, 1, generator=g) / 8
    bias = torch.randn(N, generator=g)
    ref = ref_conv(x, W4, bias)
    got = run_conv(lib, x, W4, bias, ldx=96, ldy=512)
    assert torch.isnan(got[..., N:]).all(), "columns outside the slice must stay untouched"
    assert (got[..., :N] - ref).abs().max() / ref.abs().max() < 1.2e-2
    S, start = 100, 17
    mem = torch.full((B, S, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    run_conv(lib, x, W4, bias, ybs=S * N, y_buf=mem, y_off=start * N)
    m = mem.float().cpu()
    assert torch.isnan(m[:, :start]).all() and torch.isnan(m[:, start + H * W:]).all()
    assert (m[:, start:start + H * W].reshape(B, H, W, N) - ref).abs().max() / ref.abs().max() < 1.2e-2


def test_conv_pw_stream_slices_and_batch_stride(lib):
    """Same view semantics on the large-M pointwise path (conv_pw_stream): channel-slice input (ldx > C), channel-slice output
    of a concat buffer (ldy > N) and the per-image row offset into the decoder memory (y_batch_stride)."""
    g = torch.Generator().manual_seed(5)
    B, H, W, Cc, N = 2, 100, 90, 256, 256
    x = torch.randn(B, H, W, Cc, generator=g)
    W4 = torch.randn(N, Cc, 1, 1, generator=g) / 16
    bias = torch.randn(N, generator=g)
    ref = ref_conv(x, W4, bias)
    got = run_conv(lib, x, W4, bias, ldx=320, ldy=512)
    assert torch.isnan(got[..., N:]).all(), "columns outside the slice must stay untouched"
    assert (got[..., :N] - ref).abs().max() / ref.abs().max() < 1.2e-2
    S, start = H * W + 300, 123
    mem = torch.full((B, S, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    run_conv(lib, x, W4, bias, ybs=S * N, y_buf=mem, y_off=start * N)
    m = mem.float().cpu()
    assert torch.isnan(m[:, :start]).all() and torch.isnan(m[:, start + H * W:]).all()
    assert (m[:, start:start + H * W].reshape(B, H, W, N) - ref).abs().max() / ref.abs().max() < 1.2e-2


def test_conv_rejects_bad_arguments(lib):
    d = FxConvDesc()
    assert lib.fx_conv2d_nhwc_bf16(C.byref(d), stream()) == -1
    x = torch.zeros(1, 4, 4, 48)
    with pytest.raises(_lib.FocoosAmdError):
        run_conv(lib, x, torch.zeros(8, 48, 1, 1), None)  # C % 32 != 0


@pytest.mark.parametrize("hw", [(38, 50), (37, 49), (32, 1024)])
@pytest.mark.parametrize("in_f32", [0, 1])
def test_stem(lib, in_f32, hw):
    """uint8 images run the MFMA kernel (bf16 inputs / weights: one rounding each), fp32 images the VALU kernel; odd sizes exercise
    the bottom / right zero padding, the first pixel the clamped run start."""
    g = torch.Generator().manual_seed(11)
    B, (H, W) = 2, hw
    img = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
    Wt = torch.randn(32, 3, 3, 3, generator=g) * 0.2
    bias = torch.randn(32, generator=g) * 0.1
    mean, std = torch.tensor([123.675, 116.28, 103.53]), torch.tensor([58.395, 57.12, 57.375])
    xin = to_dev(img.float() if in_f32 else img)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty(B, Ho, Wo, 32, dtype=torch.bfloat16, device=DEV)
    wd, bd, md, sd_ = to_dev(Wt.permute(2, 3, 1, 0).contiguous()), to_dev(bias), to_dev(mean), to_dev(1.0 / std)
    check(lib.fx_stem_conv3x3s2(xin.data_ptr(), in_f32, wd.data_ptr(), bd.data_ptr(), md.data_ptr(), sd_.data_ptr(), y.data_ptr(), B, H, W, 32, stream()))
    torch.cuda.synchronize()
    xn = (img.float().permute(0, 3, 1, 2) - mean.view(-1, 1, 1)) / std.view(-1, 1, 1)
    ref = F.relu(F.conv2d(xn, Wt, bias, stride=2, padding=1)).permute(0, 2, 3, 1)
    assert (y.float().cpu() - ref).abs().max() / ref.abs().max() < 6e-3


@pytest.mark.parametrize("hw", [(480, 600, 640, 640), (50, 37, 64, 96), (64, 64, 64, 64), (130, 70, 32, 32)])
def test_resize_u8(lib, hw):
    H, W, Ho, Wo = hw
    g = torch.Generator().manual_seed(5)
    img = torch.randint(0, 256, (H, W, 3), generator=g, dtype=torch.uint8)
    xd = to_dev(img)
    y = torch.empty(Ho, Wo, 3, dtype=torch.float32, device=DEV)
    check(lib.fx_resize_bilinear_u8(xd.data_ptr(), H, W, y.data_ptr(), Ho, Wo, stream()))
    torch.cuda.synchronize()
    ref = O.get_torch_batch([img.numpy()], (Ho, Wo))[0].permute(1, 2, 0)
    assert (y.cpu() - ref).abs().max() < 2e-3  # values 0..255, fp32 arithmetic in both


@pytest.mark.parametrize("cfg", [(2, 20, 20, 256, 40, 40), (2, 40, 40, 256, 20, 20), (1, 6, 10, 64, 3, 5), (1, 5, 7, 64, 10, 14), (3, 41, 33, 96, 80, 67)])
def test_resize_nhwc_and_maxpool(lib, cfg):
    B, H, W, Cc, Ho, Wo = cfg
    g = torch.Generator().manual_seed(2)
    x = bf(torch.randn(B, H, W, Cc, generator=g))
    xd = to_dev(x)
    y = torch.full((B, Ho, Wo, Cc + 64), float("nan"), dtype=torch.bfloat16, device=DEV)
    check(lib.fx_resize_bilinear_nhwc_bf16(xd.data_ptr(), Cc, y.data_ptr() + 64 * 2, Cc + 64, B, H, W, Cc, Ho, Wo, stream()))
    torch.cuda.synchronize()
    ref = F.interpolate(x.float().permute(0, 3, 1, 2), size=(Ho, Wo), mode="bilinear").permute(0, 2, 3, 1)
    yc = y.float().cpu()
    assert torch.isnan(yc[..., :64]).all()
    assert (yc[..., 64:] - ref).abs().max() < 2e-2
    Hp, Wp = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    yp = torch.empty(B, Hp, Wp, Cc, dtype=torch.bfloat16, device=DEV)
    check(lib.fx_maxpool3x3s2_nhwc_bf16(xd.data_ptr(), Cc, yp.data_ptr(), Cc, B, H, W, Cc, stream()))
    torch.cuda.synchronize()
    refp = F.max_pool2d(x.float().permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)
    assert torch.equal(yp.float().cpu(), refp)  # max of bf16 values is exact
    Ha, Wa = (H + 1) // 2, (W + 1) // 2
    ya = torch.empty(B, Ha, Wa, Cc, dtype=torch.bfloat16, device=DEV)
    check(lib.fx_avgpool2x2_nhwc_bf16(xd.data_ptr(), Cc, ya.data_ptr(), Cc, B, H, W, Cc, stream()))
    torch.cuda.synchronize()
    refa = F.avg_pool2d(x.float().permute(0, 3, 1, 2), 2, 2, 0, ceil_mode=True).permute(0, 2, 3, 1)
    assert (ya.float().cpu() - refa).abs().max() < 2e-2


def test_layernorm_and_add(lib):
    g = torch.Generator().manual_seed(9)
    rows = 301
    x, r = bf(torch.randn(rows, 256, generator=g) * 3), bf(torch.randn(rows, 256, generator=g))
    gam, bet = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g) * 0.1
    xd, rd, gd, bd = to_dev(x), to_dev(r), to_dev(gam), to_dev(bet)
    out = torch.empty(rows, 256, dtype=torch.bfloat16, device=DEV)
    for res in (rd, None):
        check(lib.fx_layernorm_bf16(xd.data_ptr(), 256, res.data_ptr() if res is not None else None, 256, gd.data_ptr(), bd.data_ptr(), out.data_ptr(),
                                    256, rows, 256, stream()))
        torch.cuda.synchronize()
        ref = F.layer_norm(x.float() + (r.float() if res is not None else 0), (256,), gam, bet, 1e-5)
        assert (out.float().cpu() - ref).abs().max() < 3e-2
    pos = bf(torch.randn(43, 256, generator=g))
    pd_ = to_dev(pos)
    check(lib.fx_add_rows_bf16(xd.data_ptr(), 256, pd_.data_ptr(), 256, 43, out.data_ptr(), 256, rows, 256, stream()))
    torch.cuda.synchronize()
    ref = bf(x.float() + pos.float()[torch.arange(rows) % 43]).float()
    assert torch.equal(out.float().cpu(), ref)


@pytest.mark.parametrize("BL", [(2, 300), (1, 400), (3, 77)])
def test_mha(lib, BL):
    B, L = BL
    g = torch.Generator().manual_seed(4)
    qkv = bf(torch.randn(B, L, 768, generator=g) * 1.5)
    d = to_dev(qkv)
    out = torch.empty(B, L, 256, dtype=torch.bfloat16, device=DEV)
    check(lib.fx_mha_bf16(d.data_ptr(), 768, d.data_ptr() + 512, 768, d.data_ptr() + 1024, 768, out.data_ptr(), 256, B, L, L, 8, stream()))
    torch.cuda.synchronize()
    q, k, v = (t.float().view(B, L, 8, 32).transpose(1, 2) for t in qkv.split(256, -1))
    att = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(32), -1)
    ref = (att @ v).transpose(1, 2).reshape(B, L, 256)
    assert (out.float().cpu() - ref).abs().max() < 2e-2


def test_msda_core_golden_and_fused(lib):
    """B4 seam: kernel vs the reference's ms_deform_attn_core_pytorch output committed as a golden vector
    (mode 0), and the fused mode (softmax + location arithmetic of modelling.py:860-874 in-kernel) vs the oracle."""
    gold = load_golden("msda_core.npz")["out"]
    value, loc, w = (torch.from_numpy(a) for a in msda_case_inputs())
    N, S, M, D = value.shape
    Lq = loc.shape[1]
    shapes = torch.tensor(MSDA_SHAPES, dtype=torch.int32)
    starts = torch.tensor([0] + list(np.cumsum([h * w_ for h, w_ in MSDA_SHAPES])[:-1]), dtype=torch.int32)
    vd = to_dev(bf(value.reshape(N, S, M * D)))
    out = torch.empty(N, Lq, 256, dtype=torch.bfloat16, device=DEV)
    ld, wd, sd_, st = to_dev(loc), to_dev(w), to_dev(shapes), to_dev(starts)
    check(lib.fx_msda_bf16(vd.data_ptr(), 256, sd_.data_ptr(), st.data_ptr(), 3, 4, ld.data_ptr(), 8 * 3 * 4 * 2, wd.data_ptr(), 8 * 3 * 4, None, 0,
                           out.data_ptr(), 256, N, S, Lq, 8, stream()))
    torch.cuda.synchronize()
    ref_bf = O.ms_deform_attn_core(bf(value).float(), MSDA_SHAPES, loc, w)
    got = out.float().cpu()
    assert (got - ref_bf).abs().max() < 1.5e-2          # same bf16-rounded values, fp32 math
    assert (got - torch.from_numpy(gold)).abs().max() < 4e-2  # vs reference golden (fp32 values)
    # fused mode
    g = torch.Generator().manual_seed(8)
    off = torch.randn(N, Lq, M, 3, 4, 2, generator=g) * 2
    logit = torch.randn(N, Lq, M, 12, generator=g)
    ref4 = torch.rand(N, Lq, 4, generator=g) * torch.tensor([1.0, 1.0, 0.4, 0.4]) + torch.tensor([0.0, 0.0, 0.02, 0.02])
    aw = torch.softmax(logit, -1).view(N, Lq, M, 3, 4)
    locs = ref4[:, :, None, None, None, :2] + off / 4 * ref4[:, :, None, None, None, 2:] * 0.5
    ref_f = O.ms_deform_attn_core(bf(value).float(), MSDA_SHAPES, locs, aw)
    cat = to_dev(torch.cat([off.reshape(N, Lq, 192), logit.reshape(N, Lq, 96)], -1))
    rd = to_dev(ref4)
    check(lib.fx_msda_bf16(vd.data_ptr(), 256, sd_.data_ptr(), st.data_ptr(), 3, 4, cat.data_ptr(), 288, cat.data_ptr() + 192 * 4, 288, rd.data_ptr(), 1,
                           out.data_ptr(), 256, N, S, Lq, 8, stream()))
    torch.cuda.synchronize()
    assert (out.float().cpu() - ref_f).abs().max() < 1.5e-2


@pytest.mark.parametrize("cfg", [(4, 8400, 300), (2, 109500, 300), (3, 1000, 1), (2, 700, 700), (1, 64, 5),
                                 # round 5: the register-resident forms (5 / 9 / 12 keys per thread) at their boundaries, and the streaming form behind them
                                 (4, 5120, 300), (4, 5121, 300), (4, 9216, 1024), (4, 9217, 300), (4, 11000, 300), (4, 12288, 256), (4, 12289, 300), (4, 21504, 300)])
def test_topk_exact(lib, cfg):
    B, n, k = cfg
    g = torch.Generator().manual_seed(n + k)
    s = torch.randn(B, n, generator=g)
    if B == 4:
        s[1] = torch.sigmoid(s[1])                       # positive, clustered exponents
        s[2, ::3] = s[2, 0]                               # massive ties straddling the cut
        s[3] = 0.25                                       # all equal: lowest indices win
    sd_ = to_dev(s)
    val = torch.empty(B, k, dtype=torch.float32, device=DEV)
    idx = torch.empty(B, k, dtype=torch.int32, device=DEV)
    check(lib.fx_topk_rows_f32(sd_.data_ptr(), n, B, n, k, val.data_ptr(), idx.data_ptr(), stream()))
    torch.cuda.synchronize()
    # reference order: value descending, index ascending among equals (stable sort == that order)
    order = torch.sort(s, dim=1, descending=True, stable=True).indices[:, :k]
    assert torch.equal(idx.cpu().long(), order), "indices must be bit-exact"
    assert torch.equal(val.cpu(), s.gather(1, order))
    tv = torch.topk(s, k, dim=1).values
    assert torch.equal(val.cpu(), tv)  # same multiset of values as torch.topk (modelling.py:1214)


@pytest.mark.parametrize("cfg", [(2, 109500, 300), (3, 30000, 300), (2, 8193, 300), (2, 20000, 1000), (1, 8500, 400)])
def test_topk_two_level_exact(lib, cfg):
    """fx_topk_rows_ws_f32 (chunk top-ks + top-k of the candidates) == one stable descending sort, incl. ties that straddle the cut,
    span chunk boundaries (chunks of 8192) and a last chunk shorter than k."""
    B, n, k = cfg
    g = torch.Generator().manual_seed(n * 3 + k)
    s = torch.sigmoid(torch.randn(B, n, generator=g) * 2)
    s[0, ::7] = 0.9375                                   # thousands of equal values above most others: the cut falls inside the tie run
    if B > 1:
        s[1] = torch.round(s[1] * 64) / 64               # 65 distinct values: ties everywhere, in every chunk
    if B > 2:
        s[2] = 0.5                                       # all equal: the k lowest indices, all from chunk 0
    sd_ = to_dev(s)
    val = torch.empty(B, k, dtype=torch.float32, device=DEV)
    idx = torch.empty(B, k, dtype=torch.int32, device=DEV)
    nws = lib.fx_topk_rows_workspace_bytes(B, n, k)
    assert nws > 0
    ws = torch.empty(nws, dtype=torch.uint8, device=DEV)
    check(lib.fx_topk_rows_ws_f32(sd_.data_ptr(), n, B, n, k, val.data_ptr(), idx.data_ptr(), ws.data_ptr(), nws, stream()))
    torch.cuda.synchronize()
    order = torch.sort(s, dim=1, descending=True, stable=True).indices[:, :k]
    assert torch.equal(idx.cpu().long(), order), "indices must be bit-exact"
    assert torch.equal(val.cpu(), s.gather(1, order))
    # and identical to the one-level kernel
    v1, i1 = torch.empty_like(val), torch.empty_like(idx)
    check(lib.fx_topk_rows_f32(sd_.data_ptr(), n, B, n, k, v1.data_ptr(), i1.data_ptr(), stream()))
    assert torch.equal(i1, idx) and torch.equal(v1, val)


def test_rowmax_gather_fill(lib):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1000, 368, generator=g)
    xd = to_dev(x)
    out = torch.empty(1000, dtype=torch.float32, device=DEV)
    check(lib.fx_rowmax_f32(xd.data_ptr(), 368, out.data_ptr(), 1000, 365, stream()))
    src = bf(torch.randn(2, 50, 256, generator=g))
    idx = torch.randint(0, 50, (2, 7), generator=g, dtype=torch.int32)
    sd_, idd = to_dev(src), to_dev(idx)
    got = torch.empty(2, 7, 256, dtype=torch.bfloat16, device=DEV)
    check(lib.fx_gather_rows_bf16(sd_.data_ptr(), 256, 50, idd.data_ptr(), 7, got.data_ptr(), 256, 2, 256, stream()))
    rows = to_dev(torch.tensor([3, 49], dtype=torch.int32))
    rowv = to_dev(bf(torch.arange(256).float()))
    check(lib.fx_fill_rows_bf16(sd_.data_ptr(), 256, 50, rows.data_ptr(), 2, rowv.data_ptr(), 2, 256, stream()))
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), x[:, :365].max(-1).values)
    assert torch.equal(got.cpu(), src.gather(1, idx.long()[..., None].expand(-1, -1, 256)))
    filled = sd_.cpu()
    assert torch.equal(filled[:, 3], rowv.cpu().expand(2, -1)) and torch.equal(filled[:, 49], rowv.cpu().expand(2, -1))
    assert torch.equal(filled[:, 4], src[:, 4])


def test_decoder_small_heads(lib):
    g = torch.Generator().manual_seed(6)
    rows = 601
    ref = torch.rand(rows, 4, generator=g)
    ref[0] = torch.tensor([0.0, 1.0, 1e-7, 0.5])  # inverse_sigmoid clamps (functional.py:4-6)
    W0, b0 = torch.randn(512, 4, generator=g), torch.randn(512, generator=g)
    out = torch.empty(rows, 512, dtype=torch.bfloat16, device=DEV)
    rd, wd, bd = to_dev(ref), to_dev(W0), to_dev(b0)
    check(lib.fx_linear_k4_relu(rd.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr(), 512, rows, 512, stream()))
    torch.cuda.synchronize()
    r0 = F.relu(F.linear(ref, W0, b0))
    assert (out.float().cpu() - r0).abs().max() / r0.abs().max() < 6e-3
    h = bf(torch.randn(rows, 256, generator=g))
    W2, b2 = torch.randn(4, 256, generator=g) / 16, torch.randn(4, generator=g) * 0.1
    hd, w2d, b2d = to_dev(h), to_dev(W2), to_dev(b2)
    new = torch.empty(rows, 4, dtype=torch.float32, device=DEV)
    check(lib.fx_bbox_head(hd.data_ptr(), 256, w2d.data_ptr(), b2d.data_ptr(), rd.data_ptr(), None, None, 300, 0, new.data_ptr(), None, rows, 256, stream()))
    torch.cuda.synchronize()
    r1 = torch.sigmoid(F.linear(h.float(), W2, b2) + O.inverse_sigmoid(ref))
    assert (new.cpu() - r1).abs().max() < 2e-5
    anchors = torch.randn(50, 4, generator=g)
    idx = torch.randint(0, 50, (rows,), generator=g, dtype=torch.int32)
    ad, idd = to_dev(anchors), to_dev(idx)
    un = torch.empty(rows, 4, dtype=torch.float32, device=DEV)
    check(lib.fx_bbox_head(hd.data_ptr(), 256, w2d.data_ptr(), b2d.data_ptr(), None, ad.data_ptr(), idd.data_ptr(), 300, 1, new.data_ptr(), un.data_ptr(), rows, 256,
                           stream()))
    torch.cuda.synchronize()
    u = F.linear(h.float(), W2, b2) + anchors[idx.long()]
    assert (un.cpu() - u).abs().max() < 2e-5 and (new.cpu() - torch.sigmoid(u)).abs().max() < 2e-5


def test_head_out_and_postprocess_vs_oracle(lib):
    g = torch.Generator().manual_seed(12)
    B, Q, K = 3, 300, 365
    logits = torch.randn(B * Q, 368, generator=g) * 2 - 4
    refb = torch.rand(B * Q, 4, generator=g) * torch.tensor([1.0, 1.0, 0.5, 0.5])
    ld, rd = to_dev(logits), to_dev(refb)
    probs = torch.empty(B, Q, K, dtype=torch.float32, device=DEV)
    boxes = torch.empty(B, Q, 4, dtype=torch.float32, device=DEV)
    check(lib.fx_detr_head_out(ld.data_ptr(), 368, rd.data_ptr(), probs.data_ptr(), boxes.data_ptr(), B * Q, K, stream()))
    torch.cuda.synchronize()
    p_ref = torch.sigmoid(logits[:, :K]).view(B, Q, K)
    b_ref = O.box_cxcywh_to_xyxy(refb).view(B, Q, 4)
    assert (probs.cpu() - p_ref).abs().max() < 1e-6 and (boxes.cpu() - b_ref).abs().max() < 1e-6
    # device post-process on the *oracle's* probabilities/boxes -> indices and integer boxes must be bit-exact
    sizes = [(480, 640), (640, 640), (333, 517)]
    thr = 0.35
    pd_, bd_ = to_dev(p_ref), to_dev(b_ref)
    val = torch.empty(B, 300, dtype=torch.float32, device=DEV)
    idx = torch.empty(B, 300, dtype=torch.int32, device=DEV)
    lab, qq = torch.empty_like(idx), torch.empty_like(idx)
    ob = torch.empty(B, 300, 4, dtype=torch.int32, device=DEV)
    cnt = torch.empty(B, dtype=torch.int32, device=DEV)
    sz = to_dev(torch.tensor(sizes, dtype=torch.int32))
    check(lib.fx_topk_rows_f32(pd_.data_ptr(), Q * K, B, Q * K, 300, val.data_ptr(), idx.data_ptr(), stream()))
    check(lib.fx_detr_postprocess(val.data_ptr(), idx.data_ptr(), bd_.data_ptr(), sz.data_ptr(), B, Q, K, 300, thr, lab.data_ptr(), qq.data_ptr(), ob.data_ptr(),
                                  cnt.data_ptr(), stream()))
    torch.cuda.synchronize()
    ref = O.postprocess(p_ref, b_ref, sizes, 300, thr)
    for i, (s, l, q, bp) in enumerate(ref):
        n = int(cnt[i])
        assert n == len(s) and n > 0
        assert torch.equal(val[i, :n].cpu(), s)
        assert torch.equal(lab[i, :n].cpu().long(), l) and torch.equal(qq[i, :n].cpu().long(), q)
        assert torch.equal(ob[i, :n].cpu(), bp)


def test_library_exports_and_device(lib):
    cu, arch = C.c_int(0), C.create_string_buffer(64)
    check(lib.fx_device_info(0, C.byref(cu), arch, 64))
    assert arch.value.decode().startswith("gfx950") and cu.value >= 200


# ---------------------------------------------------------------------------------------------------------------------
# fx_pw_chain_bf16: branch2c (+ shortcut conv as second K segment) + residual + ReLU -> next branch2a + ReLU in one launch
def frag_pack(W2):
    """[N,K] -> MFMA fragment order [N/32][K/16][64][8] (include/focoos_amd.h, fx_pw_chain_desc)."""
    N, K = W2.shape
    w = W2.float().reshape(N // 32, 32, K // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous()
    return to_dev(w.reshape(N // 32, K // 16, 64, 8), torch.bfloat16)


PW_CHAIN_CASES = [
    # M, K1a, K1b, N1, N2, residual, act1, pad (extra row stride)
    (64 * 7, 64, 0, 256, 64, True, "relu", 0),       # res2 mid block
    (64 * 5 + 37, 64, 0, 256, 64, True, "relu", 8),  # M tail, strided rows
    (333, 64, 64, 256, 64, False, "relu", 0),        # res2 block 0: shortcut conv as second source
    (200, 64, 0, 256, 128, True, "relu", 0),         # res2 -> res3 seam
    (130, 64, 0, 256, 0, True, None, 0),             # first GEMM only, no activation
    (256, 128, 0, 512, 128, True, "relu", 0),        # res3 mid (two 256-channel groups)
    (190, 128, 256, 512, 128, False, "relu", 16),    # res3 block 0 (pooled shortcut source)
    (129, 128, 0, 512, 256, True, "relu", 0),        # res3 -> res4 seam
    (100, 256, 0, 1024, 256, True, "relu", 0),       # res4 mid (four groups)
    (70, 256, 512, 1024, 256, False, "relu", 0),     # res4 block 0
    (64, 256, 0, 1024, 0, True, "relu", 0),
]


@pytest.mark.parametrize("case", PW_CHAIN_CASES)
def test_pw_chain_matches_two_convs(lib, case):
    from focoos_amd._lib import FxPwChainDesc

    M, K1a, K1b, N1, N2, has_res, act1, pad = case
    assert lib.fx_pw_chain_supported(K1a, K1b, N1, N2) == 1
    g = torch.Generator().manual_seed(M * 7 + K1a + N2)
    K1 = K1a + K1b
    # asymmetric, non-separable data: a transposed / row-swapped write shows up as an O(1) error
    x1 = torch.randn(M, K1a, generator=g) + torch.linspace(-1, 1, M)[:, None]
    x2 = torch.randn(M, K1b, generator=g) * 0.5 if K1b else None
    res = torch.randn(M, N1, generator=g) if has_res else None
    W1 = torch.randn(N1, K1, generator=g) / math.sqrt(K1) + torch.linspace(-0.05, 0.05, N1)[:, None]
    b1 = torch.randn(N1, generator=g) * 0.3
    W2 = torch.randn(max(N2, 1), N1, generator=g) / math.sqrt(N1)
    b2 = torch.randn(max(N2, 1), generator=g) * 0.3

    def strided(t, ld):
        buf = torch.full((t.shape[0], ld), 7.0, dtype=torch.bfloat16)
        buf[:, : t.shape[1]] = bf(t)
        return to_dev(buf)

    x1d = strided(x1, K1a + pad)
    x2d = strided(x2, K1b + pad) if K1b else None
    rd = strided(res, N1 + pad) if has_res else None
    y1d = torch.full((M + 3, N1 + pad), float("nan"), dtype=torch.bfloat16, device=DEV)   # +3 guard rows: nothing may be written past M
    y2d = torch.full((M + 3, N2 + pad), float("nan"), dtype=torch.bfloat16, device=DEV) if N2 else None
    w1d, b1d = frag_pack(W1), to_dev(b1)
    w2d, b2d = (frag_pack(W2), to_dev(b2)) if N2 else (None, None)
    d = FxPwChainDesc()
    d.x1, d.ldx1, d.K1a = x1d.data_ptr(), K1a + pad, K1a
    if K1b:
        d.x2, d.ldx2, d.K1b = x2d.data_ptr(), K1b + pad, K1b
    if has_res:
        d.residual, d.ldr = rd.data_ptr(), N1 + pad
    d.w1, d.bias1, d.y1, d.ldy1, d.N1, d.M = w1d.data_ptr(), b1d.data_ptr(), y1d.data_ptr(), N1 + pad, N1, M
    d.act1, d.act2 = FX_ACT[act1], FX_ACT["relu"]
    if N2:
        d.w2, d.bias2, d.y2, d.ldy2, d.N2 = w2d.data_ptr(), b2d.data_ptr(), y2d.data_ptr(), N2 + pad, N2
    check(lib.fx_pw_chain_bf16(C.byref(d), stream()), "pw_chain")
    torch.cuda.synchronize()
    # reference: bf16 operands, fp32 accumulation, y1 rounded to bf16 before it feeds the second GEMM
    X = torch.cat([bf(x1).float()] + ([bf(x2).float()] if K1b else []), 1)
    r1 = X @ bf(W1).float().T + b1
    if has_res:
        r1 = r1 + bf(res).float()
    if act1 == "relu":
        r1 = r1.relu()
    y1 = y1d.float().cpu()
    assert torch.isnan(y1[M:]).all() and (pad == 0 or torch.isnan(y1[:, N1:]).all()), "wrote outside [M, N1]"
    got1 = y1[:M, :N1]
    assert not torch.isnan(got1).any()
    tol1 = 1e-2 * r1.abs().max().item()
    assert (got1 - r1).abs().max().item() <= tol1, ((got1 - r1).abs().max().item(), tol1)
    if N2:
        r2 = (got1 @ bf(W2).float().T + b2).relu()   # from the kernel's own (bf16) y1: isolates the second GEMM
        y2 = y2d.float().cpu()
        assert torch.isnan(y2[M:]).all() and (pad == 0 or torch.isnan(y2[:, N2:]).all())
        got2 = y2[:M, :N2]
        assert not torch.isnan(got2).any()
        tol2 = 1e-2 * r2.abs().max().item()
        assert (got2 - r2).abs().max().item() <= tol2, ((got2 - r2).abs().max().item(), tol2)


@pytest.mark.parametrize("case", [(2, 12, 20, 64, 256, 128), (1, 8, 8, 128, 512, 256), (3, 6, 10, 64, 512, 128)])
def test_pw_chain_quad_tiles_with_pooled_output(lib, case):
    """The QUAD form of fx_pw_chain_bf16 (tiles of 16 2x2 pixel quads, AvgPool2d(2,2) of y1 as a third output - the variant-d shortcut input of
    the next stage, resnet.py:46,95): y1 and y2 are bit-identical to the flat-tile launch (same arithmetic per pixel, another pixel -> tile map),
    and the pooled tensor is bit-identical to fx_avgpool2x2_nhwc_bf16 of the stored y1 (same summation order)."""
    from focoos_amd._lib import FxPwChainDesc

    B, H, W, K1a, N1, N2 = case
    assert lib.fx_pw_chain_pool_supported(K1a, 0, N1, N2) == 1 and lib.fx_pw_chain_pool_supported(K1a, 64, N1, N2) == 0
    M = B * H * W
    g = torch.Generator().manual_seed(M + N1)
    x1 = to_dev(bf(torch.randn(M, K1a, generator=g) + torch.linspace(-1, 1, M)[:, None]))
    res = to_dev(bf(torch.randn(M, N1, generator=g)))
    W1 = torch.randn(N1, K1a, generator=g) / math.sqrt(K1a) + torch.linspace(-0.05, 0.05, N1)[:, None]
    W2 = torch.randn(N2, N1, generator=g) / math.sqrt(N1)
    w1d, b1d, w2d, b2d = frag_pack(W1), to_dev(torch.randn(N1, generator=g) * 0.3), frag_pack(W2), to_dev(torch.randn(N2, generator=g) * 0.3)
    outs = []
    for quad in (False, True):
        y1 = torch.full((M + 3, N1), float("nan"), dtype=torch.bfloat16, device=DEV)
        y2 = torch.full((M + 3, N2), float("nan"), dtype=torch.bfloat16, device=DEV)
        pool = torch.full((M // 4 + 2, N1), float("nan"), dtype=torch.bfloat16, device=DEV)
        d = FxPwChainDesc()
        d.x1, d.ldx1, d.K1a, d.residual, d.ldr = x1.data_ptr(), K1a, K1a, res.data_ptr(), N1
        d.w1, d.bias1, d.y1, d.ldy1, d.N1, d.M = w1d.data_ptr(), b1d.data_ptr(), y1.data_ptr(), N1, N1, M
        d.w2, d.bias2, d.y2, d.ldy2, d.N2 = w2d.data_ptr(), b2d.data_ptr(), y2.data_ptr(), N2, N2
        d.act1 = d.act2 = FX_ACT["relu"]
        if quad:
            d.pool, d.ldp, d.img_h, d.img_w = pool.data_ptr(), N1, H, W
        check(lib.fx_pw_chain_bf16(C.byref(d), stream()), "pw_chain")
        torch.cuda.synchronize()
        outs.append((y1, y2, pool))
    (y1a, y2a, _), (y1b, y2b, pool) = outs
    assert not torch.isnan(y1a[:M].float()).any() and torch.isnan(y1b[M:].float()).all() and torch.isnan(y2b[M:].float()).all()
    assert torch.equal(y1a[:M], y1b[:M]) and torch.equal(y2a[:M], y2b[:M])
    ref = torch.full((B, H // 2, W // 2, N1), float("nan"), dtype=torch.bfloat16, device=DEV)
    check(lib.fx_avgpool2x2_nhwc_bf16(y1b.data_ptr(), N1, ref.data_ptr(), N1, B, H, W, N1, stream()), "avgpool")
    torch.cuda.synchronize()
    assert torch.isnan(pool[M // 4:].float()).all()
    assert torch.equal(pool[: M // 4], ref.reshape(M // 4, N1))
    # and against torch on the stored bf16 y1
    want = F.avg_pool2d(y1b[:M].float().reshape(B, H, W, N1).permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1).reshape(M // 4, N1)
    assert (pool[: M // 4].float() - want).abs().max().item() <= 8e-3 * want.abs().max().item()
    # round 5: y1 = NULL - the block output is not stored (RT-DETR never reads res2 itself); y2 and the pooled tensor are unchanged
    y2c = torch.full_like(y2b, float("nan"))
    poolc = torch.full_like(pool, float("nan"))
    d.y1, d.ldy1, d.y2, d.pool = None, 0, y2c.data_ptr(), poolc.data_ptr()
    check(lib.fx_pw_chain_bf16(C.byref(d), stream()), "pw_chain without y1")
    torch.cuda.synchronize()
    assert torch.equal(y2c[:M], y2b[:M]) and torch.isnan(y2c[M:].float()).all()
    assert torch.equal(poolc[: M // 4], pool[: M // 4]) and torch.isnan(poolc[M // 4:].float()).all()
    d.pool = None            # ... but only where y2 and the pooled tensor carry it on
    assert lib.fx_pw_chain_bf16(C.byref(d), stream()) == -1
    d.pool, d.y1, d.ldy1 = poolc.data_ptr(), y1b.data_ptr(), N1
    d.img_h = H + 1   # odd height: refused
    assert lib.fx_pw_chain_bf16(C.byref(d), stream()) == -1


def test_pw_chain_rejects_bad_arguments(lib):
    from focoos_amd._lib import FxPwChainDesc

    assert lib.fx_pw_chain_supported(64, 0, 256, 64) == 1 and lib.fx_pw_chain_supported(64, 0, 200, 64) == 0
    assert lib.fx_pw_chain_supported(96, 0, 256, 64) == 0 and lib.fx_pw_chain_supported(64, 0, 256, 512) == 0
    d = FxPwChainDesc()
    assert lib.fx_pw_chain_bf16(C.byref(d), stream()) == -1   # FX_ERR_INVALID_ARGUMENT, nothing launched


@pytest.mark.parametrize("M,S,K", [(64 * 3, 64, 365), (8400 + 500, 8400, 365), (333, 111, 80), (200, 200, 150)])
def test_enc_score_head_fused(lib, M, S, K):
    """fx_enc_score_head_bf16 = valid_mask*memory -> Linear -> LayerNorm -> class Linear -> max (modelling.py:1202-1214) in one launch."""
    g = torch.Generator().manual_seed(M + K)
    mem = torch.randn(M, 256, generator=g) * 1.5 + torch.linspace(-0.5, 0.5, 256)[None]
    valid = (torch.rand(S, generator=g) > 0.1).to(torch.uint8)
    W1 = torch.randn(256, 256, generator=g) / 16 + torch.linspace(-0.02, 0.02, 256)[:, None]
    b1 = torch.randn(256, generator=g) * 0.1
    gam, bet = torch.rand(256, generator=g) * 0.4 + 0.8, torch.randn(256, generator=g) * 0.05
    W2 = torch.randn(K, 256, generator=g) * 2 / 16
    b2 = -6.0 + 0.5 * torch.randn(K, generator=g)
    ncp = (K + 127) // 128 * 128
    W2p = torch.zeros(ncp, 256)
    W2p[:K] = W2
    b2p = torch.full((ncp,), -3e38)
    b2p[:K] = b2
    memd = to_dev(bf(mem))
    om = torch.full((M + 2, 256), float("nan"), dtype=torch.bfloat16, device=DEV)
    sc = torch.full((M + 2,), float("nan"), dtype=torch.float32, device=DEV)
    w1d, w2d = frag_pack(W1), frag_pack(W2p)
    args = [to_dev(t) for t in (b1, gam, bet, b2p)]
    vd = to_dev(valid)
    check(lib.fx_enc_score_head_bf16(memd.data_ptr(), 256, vd.data_ptr(), S, w1d.data_ptr(), args[0].data_ptr(), args[1].data_ptr(), args[2].data_ptr(),
                                     C.c_float(1e-5), w2d.data_ptr(), args[3].data_ptr(), ncp, om.data_ptr(), 256, sc.data_ptr(), M, stream()), "score head")
    torch.cuda.synchronize()
    assert torch.isnan(om[M:].float()).all() and torch.isnan(sc[M:]).all()
    x = bf(mem).float() * valid[torch.arange(M) % S].float()[:, None]
    y = F.layer_norm(x @ bf(W1).float().T + b1, (256,), gam, bet, 1e-5)
    got_om = om[:M].float().cpu()
    assert (got_om - y).abs().max().item() <= 2e-2 * y.abs().max().item()          # one bf16 rounding of an O(4) value
    # the class GEMM + max on the kernel's own bf16 output_memory: fp32 accumulation only differs in summation order
    ref_sc = (got_om @ bf(W2).float().T + b2).max(-1).values
    got_sc = sc[:M].cpu()
    assert (got_sc - ref_sc).abs().max().item() <= 2e-4 * max(1.0, ref_sc.abs().max().item()), (got_sc - ref_sc).abs().max().item()
    # and against the all-fp32 pipeline (what the reference computes): only output_memory's bf16 rounding in between
    full = (y @ bf(W2).float().T + b2).max(-1).values
    assert (got_sc - full).abs().max().item() <= 3e-2


# 3x3 / stride 1 layers on the halo kernel (conv3x3_flat.hip): B,H,W,C,N,act,residual-after-act,ldx pad
FLAT_CASES = [
    (2, 24, 20, 64, 64, "relu", False, 0),      # res2 branch2b class: one channel chunk, M = 960 (partial last tile)
    (1, 17, 13, 64, 64, "relu", False, 8),      # odd sizes, M = 221 < one tile, strided input rows
    (3, 16, 16, 128, 128, "relu", False, 0),    # two chunks (double-buffered halo), 256 x 128 tile
    (2, 9, 33, 128, 64, None, False, 0),        # N = 64 with two chunks, no activation
    (1, 12, 80, 256, 128, "silu", False, 0),    # W = 80 (the 80x80 level), four chunks
    (2, 20, 20, 256, 256, "silu", True, 0),     # RepVGG / CSP: SiLU then + residual, 128 x 256 tile, image boundary inside a tile
    (1, 40, 40, 256, 256, "silu", False, 0),
    (5, 6, 7, 64, 256, "relu", False, 0),       # several tiny images per tile: every tap crosses image borders
    (2, 20, 20, 512, 512, "relu", False, 0),    # res5 branch2b: eight channel chunks, two 256-channel output tiles per pixel tile
    (1, 5, 200, 256, 256, "silu", True, 0),     # W = 200 (MaskFormer FPN): the 576-row plane instantiation, residual epilogue through LDS
    (1, 7, 160, 64, 64, "relu", False, 0),      # W = 160 (res2): 512-pixel tile, one chunk
    (1, 9, 100, 128, 128, None, False, 16),     # W = 100 (MaskFormer res3), strided input rows, no activation
    (2, 20, 24, 32, 32, "relu", False, 0),      # 32 input channels (conv3x3_c32.hip): conv1_2 class, several images per 512-pixel tile
    (1, 5, 320, 32, 64, "relu", False, 0),      # conv1_3 at the benchmark's width, M = 1600 (3 full tiles + a tail)
    (1, 3, 400, 32, 64, None, False, 32),       # MaskFormer's 400-wide stem: the 1344-row plane, strided input rows, no activation
    # round 5: 64 -> 64 layers on the LDS-resident-filter kernel (conv3x3_c64.hip): 16 x 32 tiles - partial bands / strips, several images
    (2, 33, 47, 64, 64, "relu", False, 8),
    (1, 40, 200, 64, 64, None, False, 0),       # MaskFormer res2 width, no activation (the input-gradient form of the training graph)
    (3, 16, 32, 64, 64, "relu", False, 0),      # exactly one tile per image
    # round 6: M >= 40 000 without a residual = the loader-less multi-chunk form (two workgroups per CU, FX_C3K_DUO256_MIN_M): four channel
    # chunks through ONE halo buffer, partial last tile (M = 40 960 + 37 rows of a ninth image would not divide - 8 x 64 x 81 = 41 472 does not either)
    (8, 64, 81, 256, 256, "silu", False, 0),
    (13, 40, 80, 256, 512, "relu", False, 8),   # N = 512 (two n-tiles per pixel tile), strided input rows, M = 41 600
    (7, 80, 80, 256, 256, "silu", True, 0),     # the CSP tail at large M: SiLU then + residual (output tile through LDS; the two-workgroup form under FX_C3K_DUO256_RES=1)
]


@pytest.mark.parametrize("case", FLAT_CASES)
def test_conv3x3_flat_halo_kernel(lib, case, flat_small_shapes):
    B, H, W, Cc, N, act, res_after, pad = case
    assert lib.fx_conv3x3_flat_supported(Cc, N, W) == 1
    g = torch.Generator().manual_seed(300 + FLAT_CASES.index(case))
    # position-dependent input: a wrong tap offset / border mask is an O(1) error, not noise
    x = torch.randn(B, H, W, Cc, generator=g) + torch.linspace(-1, 1, W)[None, None, :, None] + torch.linspace(-0.5, 0.5, H)[None, :, None, None]
    W4 = torch.randn(N, Cc, 3, 3, generator=g) / math.sqrt(Cc * 9) + torch.linspace(-0.03, 0.03, 9).view(1, 1, 3, 3)
    bias = torch.randn(N, generator=g) * 0.5
    res = torch.randn(B, H, W, N, generator=g) if res_after else None
    ref = ref_conv(x, W4, bias, 1, act, res, res_after, False)
    got = run_conv(lib, x, W4, bias, 1, act, res, res_after, ldx=Cc + pad, frag=True)[..., :N]
    assert not torch.isnan(got).any()
    err = (got - ref).abs().max() / ref.abs().max()
    assert err < 1.2e-2, f"rel err {err}"
    # same layer through the implicit-GEMM kernel: the two agree to bf16 rounding of the output
    igemm = run_conv(lib, x, W4, bias, 1, act, res, res_after, ldx=Cc + pad, frag=False)[..., :N]
    assert (got - igemm).abs().max() / ref.abs().max() < 1.2e-2


# 3x3 / STRIDE 2 layers on the parity-plane k-plane kernel (conv3x3s2_kplane.hip): B,H,W,C,N,act,ldx pad
S2_CASES = [
    (2, 16, 16, 128, 128, "relu", 0),      # res3 branch2b class: 256 x 128 tile, two channel chunks x 4 planes, M = 128 < one tile
    (1, 48, 40, 128, 128, "relu", 8),      # several tiles (M = 480), strided input rows
    (2, 80, 80, 256, 256, "relu", 0),      # res4 block 0 at the benchmark's width (Wo = 40): 128 x 256 tiles, image boundary inside a tile
    (3, 40, 40, 512, 512, "relu", 0),      # res5 block 0 (Wo = 20): the 64-pixel small-M tile, two 256-channel output tiles, 32 chunks
    (5, 6, 10, 64, 256, None, 0),          # several tiny images per tile: top / left border masks everywhere, no activation, one chunk per plane
    (1, 12, 200, 128, 128, "silu", 16),    # Wo = 100 (MaskFormer res3 at 800 x 800): the widest plane instance
    (1, 34, 26, 256, 256, "relu", 0),      # Wo = 13: odd output width, M tail
]


@pytest.mark.parametrize("case", S2_CASES)
def test_conv3x3_stride2_kplane_kernel(lib, case, flat_small_shapes):
    """Stride-2 3x3 layers as a stride-1 correlation over the four parity planes of the input vs fp32 torch and vs the implicit-GEMM
    kernel; position-dependent inputs so that a wrong plane / tap offset / border mask is an O(1) error."""
    B, H, W, Cc, N, act, pad = case
    g = torch.Generator().manual_seed(700 + S2_CASES.index(case))
    x = torch.randn(B, H, W, Cc, generator=g) + torch.linspace(-1, 1, W)[None, None, :, None] + torch.linspace(-0.5, 0.5, H)[None, :, None, None]
    W4 = torch.randn(N, Cc, 3, 3, generator=g) / math.sqrt(Cc * 9) + torch.linspace(-0.03, 0.03, 9).view(1, 1, 3, 3)
    bias = torch.randn(N, generator=g) * 0.5
    ref = ref_conv(x, W4, bias, 2, act, None, False, False)
    d = FxConvDesc()
    got = run_conv(lib, x, W4, bias, 2, act, None, False, ldx=Cc + pad, frag=True)[..., :N]
    assert got.shape == ref.shape and not torch.isnan(got).any()
    err = (got - ref).abs().max() / ref.abs().max()
    assert err < 1.2e-2, f"rel err {err}"
    igemm = run_conv(lib, x, W4, bias, 2, act, None, False, ldx=Cc + pad, frag=False)[..., :N]
    assert (got - igemm).abs().max() / ref.abs().max() < 1.2e-2


def test_conv3x3_stride2_routing(lib, flat_small_shapes):
    """The library's own routing label: stride-2 3x3 layers with a fragment-ordered weight copy and even input sizes run on the
    parity-plane kernel; odd input sizes (no parity-plane view) and layers without the copy stay on the implicit-GEMM tiles."""
    def label(H, W, Cc, N, frag):
        d = FxConvDesc()
        buf = torch.zeros(16, dtype=torch.bfloat16, device=DEV)
        d.x = d.w = d.y = buf.data_ptr()
        d.B, d.H, d.W, d.C, d.ldx = 2, H, W, Cc, Cc
        d.Ho, d.Wo, d.N, d.ldy = (H + 1) // 2, (W + 1) // 2, N, N
        d.KH, d.KW, d.stride, d.pad, d.act = 3, 3, 2, 1, FX_ACT["relu"]
        if frag:
            d.w_frag = buf.data_ptr()
        out = C.create_string_buffer(64)
        check(lib.fx_conv2d_variant(C.byref(d), out, 64), "variant")
        return out.value.decode()

    assert label(80, 80, 256, 256, True) == "conv3x3s2_kplane<256>"
    assert label(160, 160, 128, 128, True) == "conv3x3s2_kplane<128>"
    assert label(80, 80, 256, 256, False).startswith("conv_igemm")
    assert label(81, 80, 256, 256, True).startswith("conv_igemm")


# 1x1 layers with C, N multiples of 256 on the pointwise variant of the same kernel: B,H,W,C,N,act,residual(before act)
PW_FLAT_CASES = [
    (1, 1, 300, 256, 256, None, False),
    (2, 20, 20, 256, 1024, "relu", True),      # res4 branch2c: ReLU(conv + residual), four 256-channel output tiles
    (1, 1, 1000, 512, 512, "silu", False),     # CSP conv1|conv2: two K chunks (double-buffered), M tail
    (1, 37, 11, 1024, 256, "relu", False),     # res4 branch2a: four K chunks
    (1, 1, 777, 256, 1536, None, False),       # the six value_proj's as one GEMM
    (2, 13, 9, 512, 2048, "relu", True),       # res5 branch2c: K = 512 resident + residual (direct 16-byte residual loads), M tail
    (1, 1, 130, 256, 512, "relu", True),       # one full tile + a 2-row tail: residual rows past M are clamped, not read out of bounds
]


@pytest.mark.parametrize("case", PW_FLAT_CASES)
def test_pointwise_flat_kernel(lib, case, flat_small_shapes):
    B, H, W, Cc, N, act, has_res = case
    g = torch.Generator().manual_seed(400 + PW_FLAT_CASES.index(case))
    x = torch.randn(B, H, W, Cc, generator=g) + torch.linspace(-1, 1, Cc)[None, None, None, :]
    W4 = torch.randn(N, Cc, 1, 1, generator=g) / math.sqrt(Cc) + torch.linspace(-0.02, 0.02, N).view(N, 1, 1, 1)
    bias = torch.randn(N, generator=g) * 0.5
    res = torch.randn(B, H, W, N, generator=g) if has_res else None
    ref = ref_conv(x, W4, bias, 1, act, res, False, False)
    got = run_conv(lib, x, W4, bias, 1, act, res, False, frag=True)[..., :N]
    assert not torch.isnan(got).any()
    err = (got - ref).abs().max() / ref.abs().max()
    assert err < 1.2e-2, f"rel err {err}"
    igemm = run_conv(lib, x, W4, bias, 1, act, res, False, frag=False)[..., :N]
    assert (got - igemm).abs().max() / ref.abs().max() < 1.2e-2


def test_pointwise_flat_batch_stride(lib, flat_small_shapes):
    """y_batch_stride (a level writing its rows of the [B, sum(HW), C] decoder memory) through the pointwise kernel."""
    B, H, W, Cc, N, S = 3, 5, 8, 256, 256, 100
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, H, W, Cc, generator=g)
    W4 = torch.randn(N, Cc, 1, 1, generator=g) / 16
    bias = torch.randn(N, generator=g)
    buf = torch.full((B, S, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    run_conv(lib, x, W4, bias, ybs=S * N, y_buf=buf, y_off=20 * N, frag=True)
    got = buf.float().cpu()
    ref = ref_conv(x, W4, bias).reshape(B, H * W, N)
    assert torch.isnan(got[:, :20]).all() and torch.isnan(got[:, 20 + H * W:]).all()
    assert (got[:, 20:20 + H * W] - ref).abs().max() / ref.abs().max() < 1.2e-2


def test_row_chain_interpreter(lib):
    """fx_row_chain: every stage type of the decoder's row chains against torch (bf16 operands, fp32 accumulation, bf16 hand-over
    between stages exactly where the kernel stores bf16)."""
    from focoos_amd._lib import FxRcStage

    M = 32 * 5 + 7
    g = torch.Generator().manual_seed(11)
    x = torch.randn(M, 256, generator=g)
    ref = torch.rand(M, 4, generator=g) * 0.8 + 0.1
    W1, b1 = torch.randn(1024, 256, generator=g) / 16, torch.randn(1024, generator=g) * 0.1
    W2, b2 = torch.randn(256, 1024, generator=g) / 32, torch.randn(256, generator=g) * 0.1
    gam, bet = torch.rand(256, generator=g) * 0.4 + 0.8, torch.randn(256, generator=g) * 0.05
    Wk, bk = torch.randn(512, 4, generator=g), torch.randn(512, generator=g) * 0.1
    W3, b3 = torch.randn(288, 512, generator=g) / 22, torch.randn(288, generator=g) * 0.1
    Wb, bb = torch.randn(4, 256, generator=g) / 16, torch.randn(4, generator=g) * 0.1
    S0, S1, S2, BIG, REF, RED, LDS = 0, 16384, 32768, 65536, 131072, 131584, 132608
    xd, refd = to_dev(bf(x)), to_dev(ref)
    y1 = torch.full((M + 1, 1024), float("nan"), dtype=torch.bfloat16, device=DEV)
    y2 = torch.full((M + 1, 256), float("nan"), dtype=torch.bfloat16, device=DEV)
    y3 = torch.full((M + 1, 288), float("nan"), dtype=torch.float32, device=DEV)
    nref = torch.full((M + 1, 4), float("nan"), dtype=torch.float32, device=DEV)
    keep = [frag_pack(W1), to_dev(b1), frag_pack(W2), to_dev(b2), to_dev(gam), to_dev(bet), to_dev(Wk), to_dev(bk), frag_pack(W3), to_dev(b3), to_dev(Wb), to_dev(bb)]
    w1d, b1d, w2d, b2d, gd, bd, wkd, bkd, w3d, b3d, wbd, bbd = keep

    def st(**kw):
        s = FxRcStage()
        s.src, s.dst, s.aux = -1, -1, -1
        for k, v in kw.items():
            setattr(s, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
        return s

    prog = [
        st(type=0, K=256, dst=S0, g0=xd, ld=256),                                                          # x -> S0
        st(type=1, K=256, N=1024, act=1, src=S0, dst=BIG, w=w1d, bias=b1d, g0=y1, ld=1024),                  # relu(x W1^T + b1) -> BIG, y1
        st(type=2, K=1024, N=256, src=BIG, dst=S1, aux=S0, w=w2d, bias=b2d, gamma=gd, beta=bd, g0=y2, ld=256, ld2=RED),   # LN(h W2^T + b2 + x) -> S1, y2
        st(type=5, K=256, src=S1, aux=REF, w=wbd, bias=bbd, g0=refd, g1=nref),                               # refined boxes -> nref, REF
        st(type=4, N=512, dst=BIG, aux=REF, w=wkd, bias=bkd),                                                # relu(ref' Wk^T + bk) -> BIG
        st(type=3, K=256, src=S0, aux=S1, dst=S2),                                                           # x + y2 -> S2 (not stored; feeds nothing here)
        st(type=1, K=512, N=288, src=BIG, w=w3d, bias=b3d, g0=y3, ld=288, flags=1),                          # fp32 output, N = 9 blocks of 32
    ]
    arr = (FxRcStage * len(prog))(*prog)
    pd = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(DEV)
    check(lib.fx_row_chain(pd.data_ptr(), len(prog), M, LDS, stream()), "row_chain")
    torch.cuda.synchronize()
    xb = bf(x).float()
    h = (xb @ bf(W1).float().T + b1).relu()
    got1 = y1[:M].float().cpu()
    assert torch.isnan(y1[M:].float()).all() and (got1 - h).abs().max() <= 1e-2 * h.abs().max()
    r2 = F.layer_norm(got1 @ bf(W2).float().T + b2 + xb, (256,), gam, bet, 1e-5)      # from the kernel's own bf16 h
    got2 = y2[:M].float().cpu()
    assert (got2 - r2).abs().max() <= 2e-2 * r2.abs().max(), (got2 - r2).abs().max()
    u = got2 @ Wb.T + bb + torch.log(ref.clamp(1e-5) / (1 - ref).clamp(1e-5))
    rn = torch.sigmoid(u)
    gotn = nref[:M].cpu()
    assert torch.isnan(nref[M:]).all() and (gotn - rn).abs().max() <= 2e-5
    k4 = bf((gotn @ Wk.T + bk).relu()).float()
    r3 = k4 @ bf(W3).float().T + b3
    got3 = y3[:M].cpu()
    assert (got3 - r3).abs().max() <= 1e-2 * r3.abs().max(), (got3 - r3).abs().max()


@pytest.mark.parametrize("hidden", [1024, 2048, 256])
def test_row_chain_lean_stages(lib, hidden):
    """Round-5 stage forms of fx_row_chain that keep a decoder chain under 80 KiB of LDS (two workgroups per CU): RC_FFN_LN (type 7: the whole
    feed-forward block, hidden layer in two ping-pong [32][256] slots, second GEMM accumulated in registers across the chunks), RC_K4 writing its
    512-wide result into two slots and RC_GEMM reading a K = 512 operand from two slots (flags bit 2).  Against torch, and against the SAME
    arithmetic done by the round-3 stages with the wide slot (bit-identical: same fragment order, same fp32 accumulation order per output)."""
    from focoos_amd._lib import FxRcStage

    M = 32 * 7 + 5
    g = torch.Generator().manual_seed(17 + hidden)
    x = torch.randn(M, 256, generator=g)
    ref = torch.rand(M, 4, generator=g) * 0.8 + 0.1
    W1, b1 = torch.randn(hidden, 256, generator=g) / 16, torch.randn(hidden, generator=g) * 0.1
    W2, b2 = torch.randn(256, hidden, generator=g) / (hidden ** 0.5), torch.randn(256, generator=g) * 0.1
    gam, bet = torch.rand(256, generator=g) * 0.4 + 0.8, torch.randn(256, generator=g) * 0.05
    Wk, bk = torch.randn(512, 4, generator=g), torch.randn(512, generator=g) * 0.1
    W3, b3 = torch.randn(256, 512, generator=g) / 22, torch.randn(256, generator=g) * 0.1
    S0, S1, S2, S3, REF, RED, LDS = 0, 16384, 32768, 49152, 65536, 66048, 67072
    xd, refd = to_dev(bf(x)), to_dev(ref)
    keep = [frag_pack(W1), frag_pack(W2), to_dev(torch.cat([b1, b2])), to_dev(gam), to_dev(bet), to_dev(Wk), to_dev(bk), frag_pack(W3), to_dev(b3), to_dev(b1), to_dev(b2)]
    w1d, w2d, b12d, gd, bd, wkd, bkd, w3d, b3d, b1d, b2d = keep

    def st(**kw):
        s = FxRcStage()
        s.src, s.dst, s.aux = -1, -1, -1
        for k, v in kw.items():
            setattr(s, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
        return s

    def run(prog, lds):
        arr = (FxRcStage * len(prog))(*prog)
        pd = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(DEV)
        check(lib.fx_row_chain(pd.data_ptr(), len(prog), M, lds, stream()), "row_chain")
        torch.cuda.synchronize()

    y_lean = torch.full((M + 1, 256), float("nan"), dtype=torch.bfloat16, device=DEV)
    q_lean = torch.full((M + 1, 256), float("nan"), dtype=torch.float32, device=DEV)
    lean = [
        st(type=0, K=256, dst=S2, g0=xd, ld=256),
        st(type=7, K=256, N=hidden, act=S0, flags=S1, src=S2, aux=S2, dst=S3, w=w1d, g1=w2d, bias=b12d, gamma=gd, beta=bd, g0=y_lean, ld=256, ld2=RED),
        st(type=4, N=512, dst=S0, ld2=S1, flags=4, aux=-1, w=wkd, bias=bkd, g0=refd),                       # relu(ref Wk^T + bk) -> (S0 | S1)
        st(type=1, K=512, N=256, src=S0, aux=S1, flags=4 | 1, w=w3d, bias=b3d, g0=q_lean, ld=256),          # fp32 out
    ]
    run(lean, LDS)
    got = y_lean[:M].float().cpu()
    assert torch.isnan(y_lean[M:].float()).all() and torch.isnan(q_lean[M:]).all()
    xb = bf(x).float()
    h = bf((xb @ bf(W1).float().T + b1).relu()).float()                  # the kernel hands the hidden layer over in bf16
    r = F.layer_norm(h @ bf(W2).float().T + b2 + xb, (256,), gam, bet, 1e-5)
    assert (got - r).abs().max() <= 2e-2 * r.abs().max(), (got - r).abs().max()
    k4 = bf((ref @ Wk.T + bk).relu()).float()
    rq = k4 @ bf(W3).float().T + b3
    gq = q_lean[:M].cpu()
    assert (gq - rq).abs().max() <= 1e-2 * rq.abs().max(), (gq - rq).abs().max()
    if hidden == 1024:
        # the round-3 program of the same arithmetic (wide slot): bit-identical outputs
        BIG, RED3, LDS3 = 65536, 131584, 132608
        y_old = torch.full((M + 1, 256), float("nan"), dtype=torch.bfloat16, device=DEV)
        q_old = torch.full((M + 1, 256), float("nan"), dtype=torch.float32, device=DEV)
        old = [
            st(type=0, K=256, dst=S2, g0=xd, ld=256),
            st(type=1, K=256, N=1024, act=1, src=S2, dst=BIG, w=w1d, bias=b1d),
            st(type=2, K=1024, N=256, src=BIG, dst=S3, aux=S2, w=w2d, bias=b2d, gamma=gd, beta=bd, g0=y_old, ld=256, ld2=RED3),
            st(type=4, N=512, dst=BIG, aux=-1, w=wkd, bias=bkd, g0=refd),
            st(type=1, K=512, N=256, src=BIG, flags=1, w=w3d, bias=b3d, g0=q_old, ld=256),
        ]
        run(old, LDS3)
        assert torch.equal(y_old[:M], y_lean[:M]) and torch.equal(q_old[:M], q_lean[:M])


@pytest.mark.parametrize("case", [(2, 64, 96, 0), (1, 320, 320, 0), (3, 33, 47, 8), (1, 400, 400, 0), (2, 17, 30, 0), (1, 2, 2, 0)])
def test_stem_conv_relu_maxpool_fused(lib, case, flat_small_shapes):
    """fx_stem_conv3x3_relu_maxpool_bf16 (csrc/stem_pool.hip, round 5): conv1_3 + ReLU + max_pool2d(3, 2, 1) in one launch - BIT-identical to the
    two launches it replaces (fx_conv2d_nhwc_bf16 on the c32 kernel, then fx_maxpool3x3s2_nhwc_bf16) and equal to fp32 torch within one bf16
    rounding; odd sizes (partial bands / strips, odd H and W), the benchmark's 320 x 320 and MaskFormer's 400 x 400, strided input rows.
    Position-dependent inputs: a wrong tap offset / tile coordinate is an O(1) error."""
    B, H, W, pad = case
    Cc, N = 32, 64
    assert lib.fx_stem_conv_pool_supported(Cc, N, H, W) == 1
    g = torch.Generator().manual_seed(900 + H + W)
    x = torch.randn(B, H, W, Cc, generator=g) + torch.linspace(-1, 1, W)[None, None, :, None] + torch.linspace(-0.5, 0.5, H)[None, :, None, None]
    W4 = torch.randn(N, Cc, 3, 3, generator=g) / math.sqrt(Cc * 9) + torch.linspace(-0.03, 0.03, 9).view(1, 1, 3, 3)
    bias = torch.randn(N, generator=g) * 0.5 - 0.3     # a good share of negative pre-activations: the ReLU / zero-padding rule matters
    ldx = Cc + pad
    xb = torch.zeros(B, H, W, ldx, dtype=torch.bfloat16)
    xb[..., :Cc] = bf(x)
    xd = to_dev(xb)
    wf = frag_pack(W4.permute(0, 2, 3, 1).reshape(N, 9 * Cc))
    bd = to_dev(bias)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.full((B, Ho, Wo, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    check(lib.fx_stem_conv3x3_relu_maxpool_bf16(xd.data_ptr(), ldx, wf.data_ptr(), bd.data_ptr(), y.data_ptr(), N, B, H, W, stream()), "stem_conv_pool")
    torch.cuda.synchronize()
    got = y.float().cpu()
    assert not torch.isnan(got).any()
    # (1) fp32 torch on the bf16-rounded operands
    ref = F.max_pool2d(F.relu(F.conv2d(bf(x).float().permute(0, 3, 1, 2), bf(W4).float(), bias, padding=1)), 3, 2, 1).permute(0, 2, 3, 1)
    assert tuple(ref.shape) == (B, Ho, Wo, N)
    assert (got - ref).abs().max() <= 1.0e-2 * ref.abs().max(), (got - ref).abs().max()
    # (2) the two launches it replaces: identical bits
    conv = run_conv(lib, x, W4, bias, 1, "relu", ldx=ldx, frag=True)     # [B,H,W,64] through fx_conv2d_nhwc_bf16 (c32 kernel where W allows)
    cd = to_dev(bf(conv))
    y2 = torch.full((B, Ho, Wo, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    check(lib.fx_maxpool3x3s2_nhwc_bf16(cd.data_ptr(), N, y2.data_ptr(), N, B, H, W, N, stream()), "maxpool")
    torch.cuda.synchronize()
    assert torch.equal(y2, y), (y2.float() - y.float()).abs().max()


def test_stem_conv_relu_maxpool_fused_four_wave_form():
    """The 4-wave form of the fused stem kernel (both channel blocks per wave, software-pipelined K loop; FX_STEM_POOL_8WAVE=0 - the routing
    knob is read once per process, hence the subprocess): the same bit-exact parity cases as the default 8-wave form."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_kernels.py"), "-q", "-x", "-k", "test_stem_conv_relu_maxpool_fused and not four_wave"],
                       cwd=root, env=dict(os.environ, FX_STEM_POOL_8WAVE="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "6 passed" in r.stdout, r.stdout[-1500:] + r.stderr[-500:]


@pytest.mark.parametrize("hw", [(640, 640), (128, 160), (75, 94), (33, 130), (16, 16)])
def test_stem_conv1_1_conv1_2_fused(lib, hw, flat_small_shapes):
    """fx_stem_conv12_u8_bf16 (csrc/stem12.hip, round 5): normalise + conv1_1 + conv1_2 from the uint8 image in one launch - BIT-identical to
    fx_stem_conv3x3s2 followed by fx_conv2d_nhwc_bf16 on the c32 kernel (same operand arithmetic, same K-slot assignment and accumulation order in
    both layers) and equal to fp32 torch within the bf16 roundings; odd sizes (partial bands / strips, bottom / right zero padding of both layers)."""
    H, W = hw
    B = 2
    g = torch.Generator().manual_seed(40 + H + W)
    img = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
    img[0, :, :, :] = (img[0].float() * torch.linspace(0.2, 1.0, W)[None, :, None]).to(torch.uint8)      # position-dependent content
    W1 = torch.randn(32, 3, 3, 3, generator=g) * 0.2
    b1 = torch.randn(32, generator=g) * 0.1
    W2 = torch.randn(32, 32, 3, 3, generator=g) / math.sqrt(288) + torch.linspace(-0.03, 0.03, 9).view(1, 1, 3, 3)
    b2 = torch.randn(32, generator=g) * 0.3 - 0.1
    mean, std = torch.tensor([123.675, 116.28, 103.53]), torch.tensor([58.395, 57.12, 57.375])
    xin = to_dev(img)
    H1, W1_ = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    w1d, b1d, md, sd_ = to_dev(W1.permute(2, 3, 1, 0).contiguous()), to_dev(b1), to_dev(mean), to_dev(1.0 / std)
    wf2 = frag_pack(W2.permute(0, 2, 3, 1).reshape(32, 288))
    b2d = to_dev(torch.cat([b2, torch.zeros(96)]))
    y = torch.full((B, H1, W1_, 32), float("nan"), dtype=torch.bfloat16, device=DEV)
    check(lib.fx_stem_conv12_u8_bf16(xin.data_ptr(), w1d.data_ptr(), b1d.data_ptr(), md.data_ptr(), sd_.data_ptr(), wf2.data_ptr(), b2d.data_ptr(),
                                     y.data_ptr(), 32, B, H, W, stream()), "stem12")
    torch.cuda.synchronize()
    got = y.float().cpu()
    assert not torch.isnan(got).any()
    # (1) the two launches it replaces: identical bits
    c1 = torch.empty(B, H1, W1_, 32, dtype=torch.bfloat16, device=DEV)
    check(lib.fx_stem_conv3x3s2(xin.data_ptr(), 0, w1d.data_ptr(), b1d.data_ptr(), md.data_ptr(), sd_.data_ptr(), c1.data_ptr(), B, H, W, 32, stream()))
    torch.cuda.synchronize()
    two = run_conv(lib, c1.float().cpu(), W2, b2, 1, "relu", frag=True)
    assert torch.equal(bf(two), bf(got)), (two - got).abs().max()
    # (2) fp32 torch with the engine's roundings (normalised input, both weights and the conv1_1 activation in bf16)
    xn = bf((img.float() - mean) / std).float().permute(0, 3, 1, 2)
    a1 = bf(F.relu(F.conv2d(xn, bf(W1).float(), b1, stride=2, padding=1))).float()
    ref = F.relu(F.conv2d(a1, bf(W2).float(), b2, padding=1)).permute(0, 2, 3, 1)
    assert (got - ref).abs().max() <= 1.2e-2 * ref.abs().max(), (got - ref).abs().max()
