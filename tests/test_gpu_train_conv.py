"""Training-path convolution kernels on a real MI355X vs torch CPU fp32 autograd of the same op
(F.conv2d on bf16-rounded operands): weight gradient (transpose-read MFMA kernel), input gradient, activation / pooling
backward.  Tolerance: products of bf16 operands accumulated in fp32 -> relative error <= 2e-3 of the gradient's max."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from focoos_amd import _lib  # noqa: E402
from focoos_amd._lib import check  # noqa: E402

DEV = "cuda:0"


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available()
    return _lib.load()


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


WGRAD_CASES = [
    # B, H, W, C, N, k, stride
    (2, 20, 24, 64, 64, 3, 1),      # C = 64: a 128-column tile spans two filter taps
    (2, 17, 19, 128, 256, 1, 1),    # odd sizes: pixel tail
    (1, 32, 32, 32, 64, 3, 1),      # C = 32 (stem conv1_3): four taps per tile, Ktot = 288 (tile tail)
    (3, 16, 16, 128, 128, 3, 2),    # stride 2
    (2, 40, 40, 256, 64, 1, 1),     # N = 64: half-empty n tile
    (1, 28, 28, 512, 2048, 1, 1),   # many tiles
    (4, 64, 64, 64, 256, 1, 1),     # M = 16384: several pixel chunks accumulate atomically
    (2, 12, 12, 3 * 8, 40, 3, 1),   # C = 24, N = 40: both below a tile, C not a power of two
    # wide layers (N % 256 == 0, C % 256 == 0): the partial-slab call runs conv_wgrad_dma_kernel (256 x 256 tiles, LDS-DMA ring)
    (2, 40, 40, 256, 256, 3, 1),    # res4-level 3x3: nine taps = nine K tiles, image borders inside a pixel range
    (3, 17, 23, 256, 256, 3, 1),    # ragged: rows wrap inside a 32-pixel step, W < 32, pixel tail (M = 1173)
    (1, 20, 20, 512, 512, 3, 1),    # res5 3x3: two K tiles per tap, 2 x 18 tiles
    (2, 25, 31, 256, 512, 1, 1),    # pointwise, M = 1550 (tail), N = 512
    (1, 40, 40, 1024, 256, 1, 1),   # pointwise, K = 1024
    (2, 20, 20, 256, 256, 3, 2),    # same shape class at stride 2: the 128 x 128 kernel with the wide class's split plan
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_conv_wgrad(lib, case):
    B, H, W, Cc, N, k, stride = case
    pad = (k - 1) // 2
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, H, W, Cc, generator=g).bfloat16()
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    dz = torch.randn(B, Ho, Wo, N, generator=g).bfloat16()
    w = torch.zeros(N, Cc, k, k, requires_grad=True)
    y = F.conv2d(x.float().permute(0, 3, 1, 2), w, None, stride=stride, padding=pad)
    y.backward(dz.float().permute(0, 3, 1, 2))
    ref = w.grad.permute(0, 2, 3, 1).contiguous()  # [N][KH][KW][C]
    xd, dzd = x.to(DEV), dz.to(DEV)
    dw = torch.zeros(N, k, k, Cc, dtype=torch.float32, device=DEV)
    check(lib.fx_conv2d_wgrad_nhwc_bf16(xd.data_ptr(), Cc, dzd.data_ptr(), N, dw.data_ptr(), B, H, W, Cc, Ho, Wo, N, k, k, stride, pad, stream()))
    torch.cuda.synchronize()
    err = (dw.cpu() - ref).abs().max() / ref.abs().max()
    assert err < 2e-3, f"rel err {err}"
    # accumulation semantics: a second call doubles the result
    check(lib.fx_conv2d_wgrad_nhwc_bf16(xd.data_ptr(), Cc, dzd.data_ptr(), N, dw.data_ptr(), B, H, W, Cc, Ho, Wo, N, k, k, stride, pad, stream()))
    torch.cuda.synchronize()
    assert (dw.cpu() - 2 * ref).abs().max() / ref.abs().max() < 4e-3
    # partial-slab path (plain stores per pixel range, no zeroing needed) + the summing / scaling / re-layout pass
    S = lib.fx_conv2d_wgrad_splits(B, Ho, Wo, Cc, N, k, k)
    assert S >= 1
    slab = N * k * k * Cc + 64
    ws = torch.full((S * slab,), float("nan"), dtype=torch.float32, device=DEV)
    check(lib.fx_conv2d_wgrad_partial_nhwc_bf16(xd.data_ptr(), Cc, dzd.data_ptr(), N, ws.data_ptr(), slab, S, B, H, W, Cc, Ho, Wo, N, k, k, stride, pad,
                                                stream()))
    scale = (torch.rand(N, generator=g) + 0.5).to(DEV)
    out = torch.ones(N, Cc, k, k, dtype=torch.float32, device=DEV)
    check(lib.fx_unpack_conv_wgrad_sum_f32(ws.data_ptr(), slab, S, scale.data_ptr(), out.data_ptr(), N, Cc, k, k, Cc, 1, stream()))
    torch.cuda.synchronize()
    want = 1 + w.grad * scale.cpu().view(-1, 1, 1, 1)
    assert (out.cpu() - want).abs().max() / ref.abs().max() < 3e-3
    with pytest.raises(_lib.FocoosAmdError):   # a split count other than the planned one is rejected
        check(lib.fx_conv2d_wgrad_partial_nhwc_bf16(xd.data_ptr(), Cc, dzd.data_ptr(), N, ws.data_ptr(), slab, S + 1, B, H, W, Cc, Ho, Wo, N, k, k, stride,
                                                    pad, stream()))


def test_conv_wgrad_asymmetric_and_strided_views(lib):
    """Transpose-detecting check (one-hot pixels / channels) and channel-slice views (ldx > C, lddz > N)."""
    B, H, W, Cc, N = 1, 4, 8, 16, 24
    x = torch.zeros(B, H, W, 32)
    dz = torch.zeros(B, H, W, 40)
    x[0, 1, 2, 5] = 3.0       # pixel (1,2), input channel 5
    dz[0, 1, 2, 7] = 2.0      # same pixel, output channel 7
    dz[0, 3, 7, 11] = 1.0     # another pixel with no activation
    xd, dzd = x.bfloat16().to(DEV), dz.bfloat16().to(DEV)
    dw = torch.zeros(N, 1, 1, Cc, dtype=torch.float32, device=DEV)
    check(lib.fx_conv2d_wgrad_nhwc_bf16(xd.data_ptr(), 32, dzd.data_ptr(), 40, dw.data_ptr(), B, H, W, Cc, H, W, N, 1, 1, 1, 0, stream()))
    torch.cuda.synchronize()
    want = torch.zeros(N, 1, 1, Cc)
    want[7, 0, 0, 5] = 6.0
    assert torch.equal(dw.cpu(), want)


def test_pack_unpack_and_elementwise_backward(lib):
    g = torch.Generator().manual_seed(0)
    N, Cc, k = 40, 24, 3
    w = torch.randn(N, Cc, k, k, generator=g)
    s = torch.rand(N, generator=g) + 0.5
    wf = torch.zeros(128, k, k, Cc, dtype=torch.bfloat16, device=DEV)
    wd = torch.zeros(128, k, k, N, dtype=torch.bfloat16, device=DEV)
    wdv, sd = w.to(DEV), s.to(DEV)
    check(lib.fx_pack_conv_weights_f32(wdv.data_ptr(), sd.data_ptr(), wf.data_ptr(), wd.data_ptr(), None, None, N, Cc, k, k, stream()))
    torch.cuda.synchronize()
    ws = (w * s.view(-1, 1, 1, 1)).bfloat16()
    assert torch.equal(wf[:N].cpu(), ws.permute(0, 2, 3, 1)) and float(wf[N:].abs().max()) == 0
    assert torch.equal(wd[:Cc].cpu(), ws.flip(2, 3).permute(1, 2, 3, 0))
    dwe = torch.randn(N, k, k, 32, generator=g)
    out = torch.ones(N, Cc, k, k, device=DEV)
    dd = dwe.to(DEV)
    check(lib.fx_unpack_conv_wgrad_f32(dd.data_ptr(), sd.data_ptr(), out.data_ptr(), N, Cc, k, k, 32, 1, stream()))
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), (1 + dwe[..., :Cc].permute(0, 3, 1, 2) * s.view(-1, 1, 1, 1)).numpy(), rtol=1e-6)
    # relu backward with a second upstream gradient
    y = torch.randn(5, 7, 9, 64, generator=g).clamp_min(0).bfloat16()
    dy, dy2 = torch.randn(5, 7, 9, 64, generator=g).bfloat16(), torch.randn(5, 7, 9, 64, generator=g).bfloat16()
    dz = torch.empty_like(y, device=DEV)
    a, b, c = dy.to(DEV), dy2.to(DEV), y.to(DEV)
    check(lib.fx_relu_bwd_bf16(a.data_ptr(), 64, b.data_ptr(), 64, c.data_ptr(), 64, dz.data_ptr(), 64, 5 * 7 * 9, 64, 1, stream()))
    torch.cuda.synchronize()
    ref = ((dy.float() + dy2.float()) * (y.float() > 0)).bfloat16()
    assert torch.equal(dz.cpu(), ref)


@pytest.mark.parametrize("hw", [(12, 16), (13, 15)])
def test_pool_backward(lib, hw):
    H, W = hw
    g = torch.Generator().manual_seed(H)
    # max pool: ReLU-like input with many exact ties (zeros) -> the first-max rule matters
    x = torch.randn(2, H, W, 16, generator=g).clamp_min(0).bfloat16()
    xt = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    y = F.max_pool2d(xt, 3, 2, 1)
    dy = torch.randn(y.shape, generator=g).bfloat16()
    y.backward(dy.float())
    ref = xt.grad.permute(0, 2, 3, 1)
    xd, dyd = x.to(DEV), dy.permute(0, 2, 3, 1).contiguous().to(DEV)
    dx = torch.empty_like(xd)
    arg = torch.empty(dyd.numel(), dtype=torch.uint8, device=DEV)   # workspace: arg-max tap per output element
    check(lib.fx_maxpool3x3s2_bwd_nhwc_bf16(xd.data_ptr(), 16, dyd.data_ptr(), 16, dx.data_ptr(), 16, 2, H, W, 16, arg.data_ptr(), stream()))
    torch.cuda.synchronize()
    assert (dx.float().cpu() - ref).abs().max() <= 2e-2 * ref.abs().max()  # sums of up to 4 bf16 gradients, rounded once
    assert ((dx.float().cpu() != 0) == (ref != 0)).all()               # identical routing (arg-max choice incl. ties)
    # average pool (ceil mode)
    xt = torch.zeros(2, 16, H, W, requires_grad=True)
    y = F.avg_pool2d(xt, 2, 2, 0, ceil_mode=True)
    dy = torch.randn(y.shape, generator=g).bfloat16()
    y.backward(dy.float())
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(DEV)
    dx = torch.empty(2, H, W, 16, dtype=torch.bfloat16, device=DEV)
    check(lib.fx_avgpool2x2_bwd_nhwc_bf16(dyd.data_ptr(), 16, dx.data_ptr(), 16, 2, H, W, 16, stream()))
    torch.cuda.synchronize()
    assert torch.equal(dx.cpu(), xt.grad.permute(0, 2, 3, 1).bfloat16())


def test_resnet_vd_backward_vs_torch_autograd():
    """ResNet50-vd (frozen BN) forward + backward through the HIP autograd nodes vs torch CPU fp32 autograd of the oracle's
    restatement of the reference backbone, same seeded weights / images / loss.  bf16 activations and gradients through
    ~50 layers: per-parameter relative L2 error of the weight gradients <= 6e-2 (typically 1-2e-2), features <= 2e-2."""
    from focoos_amd.registry import ModelRegistry
    from focoos_amd.synth import synth_image_structured, synth_state_dict
    from focoos_amd.train_nn import ResNetVd
    from oracle import detr_oracle as O
    from tests.helpers import rel_l2

    cfg = ModelRegistry.get_model_info("fai-detr-l-coco")["config"]
    sd = synth_state_dict(cfg, 11)
    pre = "pixel_decoder.backbone."
    bsd = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
    net = ResNetVd(50).to(DEV)
    missing = net.load_state_dict(bsd, strict=True)
    assert list(net.state_dict().keys()) == list(bsd.keys())
    imgs = [synth_image_structured(40 + i, 128, 160) for i in range(2)]
    x_u8 = torch.from_numpy(np.stack(imgs)).to(DEV)
    g = torch.Generator().manual_seed(3)
    proj = {k: torch.randn(c, generator=g) for k, c in (("res2", 256), ("res3", 512), ("res4", 1024), ("res5", 2048))}
    outs = net(x_u8)
    loss = sum((outs[k].float() * proj[k].to(DEV)).sum() for k in proj) * 1e-2
    loss.backward()
    torch.cuda.synchronize()
    # CPU fp32 reference
    ref_sd = {k: (v.clone().requires_grad_(True) if k.endswith("conv.weight") else v) for k, v in sd.items() if k.startswith(pre)}
    mean = torch.tensor(cfg["pixel_mean"]).view(-1, 1, 1)
    std = torch.tensor(cfg["pixel_std"]).view(-1, 1, 1)
    xi = (O.get_torch_batch(imgs, None) - mean) / std
    feats = O.resnet_vd(ref_sd, pre[:-1], xi, O.RESNET_BLOCKS[50])
    ref_loss = sum((feats[k] * proj[k].view(1, -1, 1, 1)).sum() for k in proj) * 1e-2
    ref_loss.backward()
    for k in proj:
        assert rel_l2(outs[k].detach().float().cpu().permute(0, 3, 1, 2), feats[k].detach()) <= 2e-2, k
    assert abs(float(loss) - float(ref_loss)) <= 2e-2 * abs(float(ref_loss))
    worst = 0.0
    for name, p in net.named_parameters():
        if not p.requires_grad:
            continue
        assert p.grad is not None, name
        e = rel_l2(p.grad.cpu(), ref_sd[pre + name].grad)
        worst = max(worst, e)
        assert e <= 6e-2, (name, e)
    print("worst weight-gradient rel-L2:", worst)


def test_token_layer_backward_kernels(lib):
    """LayerNorm / activation / attention / resize / linear backward vs torch fp32 autograd on bf16-rounded inputs."""
    from focoos_amd.train_nn import LayerNorm, Linear, MultiheadAttention, _ResizeFn

    g = torch.Generator().manual_seed(5)
    # LayerNorm
    x = (torch.randn(3, 50, 256, generator=g) * 2 + 0.3).bfloat16()
    ln = LayerNorm(lib).to(DEV)
    with torch.no_grad():
        ln.weight.copy_(torch.rand(256, generator=g) + 0.5)
        ln.bias.copy_(torch.randn(256, generator=g) * 0.1)
    xd = x.to(DEV).requires_grad_(True)
    dy = torch.randn(3, 50, 256, generator=g).bfloat16()
    ln(xd).backward(dy.to(DEV))
    xr = x.float().requires_grad_(True)
    wr, br = ln.weight.detach().cpu().requires_grad_(True), ln.bias.detach().cpu().requires_grad_(True)
    F.layer_norm(xr, (256,), wr, br, 1e-5).backward(dy.float())
    assert (xd.grad.float().cpu() - xr.grad).abs().max() <= 2e-2 * xr.grad.abs().max()
    assert (ln.weight.grad.cpu() - wr.grad).abs().max() <= 1e-2 * wr.grad.abs().max()
    assert (ln.bias.grad.cpu() - br.grad).abs().max() <= 1e-2 * br.grad.abs().max()
    # Linear with GELU / odd shapes (N = 365 -> padded to 368, K = 4 -> padded to 32)
    for cin, cout, act in ((256, 1024, "gelu"), (256, 365, None), (4, 512, "relu"), (256, 4, None), (256, 256, "silu")):
        lin = Linear(lib, cin, cout, act=act).to(DEV)
        with torch.no_grad():
            lin.weight.copy_(torch.randn(cout, cin, generator=g) / math.sqrt(cin))
            lin.bias.copy_(torch.randn(cout, generator=g) * 0.1)
        x = torch.randn(2, 77, cin, generator=g).bfloat16()
        dy = torch.randn(2, 77, cout, generator=g).bfloat16()
        xd = x.to(DEV).requires_grad_(True)
        y = lin(xd)
        y.backward(dy.to(DEV))
        xr = x.float().requires_grad_(True)
        wr = lin.weight.detach().cpu().bfloat16().float().requires_grad_(True)
        br = lin.bias.detach().cpu().requires_grad_(True)
        z = F.linear(xr, wr, br)
        yr = {"gelu": F.gelu, "relu": F.relu, "silu": F.silu, None: lambda t: t}[act](z)
        yr.backward(dy.float())
        assert (y.float().cpu() - yr).abs().max() <= 2e-2 * yr.abs().max(), (cin, cout, act)
        for got, ref, nm in ((xd.grad.float().cpu(), xr.grad, "dx"), (lin.weight.grad.cpu(), wr.grad, "dw"), (lin.bias.grad.cpu(), br.grad, "db")):
            assert (got - ref).abs().max() <= 2.5e-2 * ref.abs().max(), (cin, cout, act, nm)
    # attention (q = k path and separate q / k path)
    mha = MultiheadAttention(lib).to(DEV)
    with torch.no_grad():
        mha.in_proj_weight.copy_(torch.randn(768, 256, generator=g) / 16)
        mha.in_proj_bias.copy_(torch.randn(768, generator=g) * 0.1)
        mha.out_proj.weight.copy_(torch.randn(256, 256, generator=g) / 16)
        mha.out_proj.bias.copy_(torch.randn(256, generator=g) * 0.1)
    ref = torch.nn.MultiheadAttention(256, 8, batch_first=True)
    with torch.no_grad():
        ref.in_proj_weight.copy_(mha.in_proj_weight.cpu().bfloat16().float()); ref.in_proj_bias.copy_(mha.in_proj_bias.cpu())
        ref.out_proj.weight.copy_(mha.out_proj.weight.cpu().bfloat16().float()); ref.out_proj.bias.copy_(mha.out_proj.bias.cpu())
    qk = torch.randn(2, 130, 256, generator=g).bfloat16()
    vv = torch.randn(2, 130, 256, generator=g).bfloat16()
    dy = torch.randn(2, 130, 256, generator=g).bfloat16()
    qd, vd = qk.to(DEV).requires_grad_(True), vv.to(DEV).requires_grad_(True)
    out = mha(qd, qd, vd)
    out.backward(dy.to(DEV))
    qr, vr = qk.float().requires_grad_(True), vv.float().requires_grad_(True)
    outr = ref(qr, qr, vr, need_weights=False)[0]
    outr.backward(dy.float())
    assert (out.float().cpu() - outr).abs().max() <= 3e-2 * outr.abs().max()
    assert (qd.grad.float().cpu() - qr.grad).abs().max() <= 4e-2 * qr.grad.abs().max()
    assert (vd.grad.float().cpu() - vr.grad).abs().max() <= 4e-2 * vr.grad.abs().max()
    assert (mha.in_proj_weight.grad.cpu() - ref.in_proj_weight.grad).abs().max() <= 4e-2 * ref.in_proj_weight.grad.abs().max()
    assert (mha.in_proj_bias.grad.cpu() - ref.in_proj_bias.grad).abs().max() <= 4e-2 * ref.in_proj_bias.grad.abs().max()
    # bilinear resize backward (x2 up and x0.5 down)
    for (H, W, Ho, Wo) in ((10, 12, 20, 24), (20, 24, 10, 12), (7, 9, 13, 17)):
        x = torch.randn(2, H, W, 64, generator=g).bfloat16()
        dy = torch.randn(2, Ho, Wo, 64, generator=g).bfloat16()
        xd = x.to(DEV).requires_grad_(True)
        _ResizeFn.apply(xd, Ho, Wo, lib).backward(dy.to(DEV))
        xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
        F.interpolate(xr, size=(Ho, Wo), mode="bilinear", align_corners=False).backward(dy.float().permute(0, 3, 1, 2))
        assert (xd.grad.float().cpu() - xr.grad.permute(0, 2, 3, 1)).abs().max() <= 1e-2 * xr.grad.abs().max()


def test_backbone_plus_hybrid_encoder_backward_vs_torch_autograd():
    """ResNet50-vd + the RT-DETR hybrid encoder (AIFI + CSP-Rep FPN/PAN, frozen BN) forward + backward on the HIP autograd
    nodes vs torch CPU fp32 autograd of the oracle's restatement (88 % of the model's FLOPs)."""
    from focoos_amd.registry import ModelRegistry
    from focoos_amd.synth import synth_image_structured, synth_state_dict
    from focoos_amd.train_nn import HybridEncoder, ResNetVd
    from focoos_amd import _lib as L
    from oracle import detr_oracle as O
    from tests.helpers import rel_l2

    cfg = ModelRegistry.get_model_info("fai-detr-l-coco")["config"]
    sd = synth_state_dict(cfg, 12)
    # The seeded weights feed the AIFI layer tokens of std ~12, i.e. attention logits in the hundreds: its softmax is one-hot
    # and d(logits) is ill-conditioned under ANY rounding of q / k (measured: 24 % gradient error from bf16 q, k alone, 0.5 %
    # everywhere else).  Trained models keep logits O(1-10); scale the q / k projections to that regime for this check.
    k_qk = "pixel_decoder.encoder.0.layers.0.self_attn.in_proj_weight"
    sd[k_qk] = sd[k_qk].clone()
    sd[k_qk][:512] *= 0.05
    pre = "pixel_decoder.backbone."
    net = ResNetVd(50).to(DEV)
    net.load_state_dict({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}, strict=True)
    enc = HybridEncoder(L.load()).to(DEV)
    esd = {k[len("pixel_decoder."):]: v for k, v in sd.items() if k.startswith("pixel_decoder.") and not k.startswith(pre)}
    enc.load_state_dict(esd, strict=True)
    assert list(enc.state_dict().keys()) == list(esd.keys())
    imgs = [synth_image_structured(60 + i, 128, 160) for i in range(2)]
    x_u8 = torch.from_numpy(np.stack(imgs)).to(DEV)
    g = torch.Generator().manual_seed(4)
    proj = [torch.randn(256, generator=g) for _ in range(3)]
    f = net(x_u8)
    outs = enc([f["res3"], f["res4"], f["res5"]])
    loss = sum((o.float() * p.to(DEV)).sum() for o, p in zip(outs, proj)) * 1e-2
    loss.backward()
    torch.cuda.synchronize()
    trainable = lambda k: k.endswith("conv.weight") or ".0.weight" in k or k.endswith(("in_proj_weight", "in_proj_bias", "out_proj.weight", "out_proj.bias")) \
        or ".linear" in k or (".norm1." in k or ".norm2." in k) and "encoder.0" in k
    ref_sd = {k: (v.clone().requires_grad_(True) if (v.dtype == torch.float32 and trainable(k) and "mask_features" not in k) else v) for k, v in sd.items()}
    mean = torch.tensor(cfg["pixel_mean"]).view(-1, 1, 1)
    std = torch.tensor(cfg["pixel_std"]).view(-1, 1, 1)
    xi = (O.get_torch_batch(imgs, None) - mean) / std
    feats = O.resnet_vd(ref_sd, pre[:-1], xi, O.RESNET_BLOCKS[50])
    ref_outs = O.hybrid_encoder(ref_sd, [feats["res3"], feats["res4"], feats["res5"]], cfg)
    ref_loss = sum((o * p.view(1, -1, 1, 1)).sum() for o, p in zip(ref_outs, proj)) * 1e-2
    ref_loss.backward()
    for o, r in zip(outs, ref_outs):
        assert rel_l2(o.detach().float().cpu().permute(0, 3, 1, 2), r.detach()) <= 2.5e-2
    worst, n, errs = 0.0, 0, []
    for prefix, mod in (("pixel_decoder.backbone.", net), ("pixel_decoder.", enc)):
        for name, p in mod.named_parameters():
            if not p.requires_grad:
                continue
            r = ref_sd[prefix + name]
            assert p.grad is not None and r.grad is not None, name
            e = rel_l2(p.grad.cpu(), r.grad)
            worst, n = max(worst, e), n + 1
            errs.append((e, prefix + name))
    bad = [(round(e, 4), nm) for e, nm in errs if e > 8e-2]
    print(f"{n} parameter tensors, worst gradient rel-L2 {worst:.4f}")
    assert not bad, bad


@pytest.mark.parametrize("act,with_res,C,rows_shape", [("relu", True, 64, (3, 9, 11)), ("silu", True, 256, (2, 5, 7)), (None, False, 24, (4, 33, 17)),
                                                        ("gelu", False, 520, (1, 6, 5))])
@pytest.mark.parametrize("z_f32", [0, 1])
def test_batchnorm_train_kernels(lib, act, with_res, C, rows_shape, z_f32):
    """fx_bn_stats / finalize / apply and fx_bn_bwd_stats / apply vs torch fp32 autograd of
    act(F.batch_norm(z, training=True) [+ residual]) on the same bf16-rounded inputs; z read as bf16 or as fp32 (the form the
    trainable graphs use: the conv epilogue writes the pre-normalisation tensor in fp32)."""
    from focoos_amd._lib import FX_ACT

    g = torch.Generator().manual_seed(C)
    Bn, H, W_ = rows_shape
    rows = Bn * H * W_
    z = (torch.randn(Bn, H, W_, C, generator=g) * (torch.rand(C, generator=g) * 3 + 0.2) + torch.randn(C, generator=g)).bfloat16()
    res = torch.randn(Bn, H, W_, C, generator=g).bfloat16() if with_res else None
    dy = torch.randn(Bn, H, W_, C, generator=g).bfloat16()
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    # reference
    zt = z.float().permute(0, 3, 1, 2).requires_grad_(True)
    gt, bt = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    a = F.batch_norm(zt, rm_ref, rv_ref, gt, bt, training=True, momentum=0.1, eps=1e-5)
    rt = None
    if with_res:
        rt = res.float().permute(0, 3, 1, 2).requires_grad_(True)
        a = a + rt
    yt = {"relu": F.relu, "silu": F.silu, "gelu": F.gelu, None: lambda t: t}[act](a)
    yt.backward(dy.float().permute(0, 3, 1, 2))
    # kernels
    st = stream()
    zd, dyd = z.to(DEV), dy.to(DEV)
    if z_f32:
        zd = zd.float()
    rd = res.to(DEV) if with_res else None
    sums = torch.zeros(2, C, device=DEV)
    check(lib.fx_bn_stats_bf16(zd.data_ptr(), C, z_f32, sums.data_ptr(), rows, C, st))
    stats = torch.empty(4, C, device=DEV)
    gd, bd, rmd, rvd = gamma.to(DEV), beta.to(DEV), rm.to(DEV), rv.to(DEV)
    nbt = torch.zeros((), dtype=torch.long, device=DEV)
    check(lib.fx_bn_finalize_f32(sums.data_ptr(), float(rows), gd.data_ptr(), bd.data_ptr(), 1e-5, 0.1, rmd.data_ptr(), rvd.data_ptr(), nbt.data_ptr(),
                                 stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr(), stats[3].data_ptr(), C, st))
    y = torch.empty_like(dyd)
    rp = rd.data_ptr() if with_res else None
    check(lib.fx_bn_apply_bf16(zd.data_ptr(), C, z_f32, stats[2].data_ptr(), stats[3].data_ptr(), rp, C, FX_ACT[act], y.data_ptr(), C, rows, C, st))
    bs = torch.zeros(2, C, device=DEV)
    check(lib.fx_bn_bwd_stats_bf16(dyd.data_ptr(), C, zd.data_ptr(), C, z_f32, rp, C, stats[2].data_ptr(), stats[3].data_ptr(), stats[0].data_ptr(),
                                   stats[1].data_ptr(), FX_ACT[act], bs.data_ptr(), rows, C, st))
    dz = torch.empty_like(dyd)
    da = torch.empty_like(dyd) if with_res else None
    gacc, bacc = torch.full((C,), 0.5, device=DEV), torch.full((C,), -0.25, device=DEV)   # accumulated INTO: dgamma / dbeta land on top
    check(lib.fx_bn_bwd_apply_bf16(dyd.data_ptr(), C, zd.data_ptr(), C, z_f32, rp, C, stats[2].data_ptr(), stats[3].data_ptr(), stats[0].data_ptr(),
                                   stats[1].data_ptr(), FX_ACT[act], bs.data_ptr(), 1.0 / rows, da.data_ptr() if with_res else None, C, dz.data_ptr(), C,
                                   rows, C, gacc.data_ptr(), bacc.data_ptr(), st))
    torch.cuda.synchronize()
    assert int(nbt) == 1
    np.testing.assert_allclose(rmd.cpu().numpy(), rm_ref.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(rvd.cpu().numpy(), rv_ref.numpy(), rtol=1e-4, atol=1e-5)
    nhwc = lambda t: t.detach().permute(0, 2, 3, 1)
    assert (y.float().cpu() - nhwc(yt)).abs().max() <= 2e-2 * nhwc(yt).abs().max()      # one bf16 rounding of the output
    np.testing.assert_allclose(bs[0].cpu().numpy(), bt.grad.numpy(), rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(bs[1].cpu().numpy(), gt.grad.numpy(), rtol=2e-3, atol=2e-3 * float(gt.grad.abs().max()))
    assert torch.equal(gacc, bs[1] + 0.5) and torch.equal(bacc, bs[0] - 0.25)          # the launch's own accumulation into the parameter gradients
    assert (dz.float().cpu() - nhwc(zt.grad)).abs().max() <= 1e-2 * nhwc(zt.grad).abs().max() + 1e-6
    if with_res:
        assert (da.float().cpu() - nhwc(rt.grad)).abs().max() <= 1e-2 * nhwc(rt.grad).abs().max()


def _bn_layer_case(lib, cin, cout, k, stride, act, with_res, B=4, H=20, W=24, seed=0):
    from focoos_amd.train_nn import ConvNormLayer, set_norm_mode
    from tests.helpers import rel_l2

    g = torch.Generator().manual_seed(seed)
    layer = ConvNormLayer(lib, cin, cout, k, stride, act)
    set_norm_mode(layer, "BN")
    with torch.no_grad():
        layer._conv_h.weight.copy_((torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).bfloat16().float())
        layer._norm_h.weight.copy_(torch.rand(cout, generator=g) + 0.5)
        layer._norm_h.bias.copy_(torch.randn(cout, generator=g) * 0.3)
    layer = layer.to(DEV)
    x = torch.randn(B, H, W, cin, generator=g).bfloat16()
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    res = torch.randn(B, Ho, Wo, cout, generator=g).bfloat16() if with_res else None
    cot = torch.randn(B, Ho, Wo, cout, generator=g).bfloat16()
    xd = x.to(DEV).requires_grad_(True)
    rd = res.to(DEV).requires_grad_(True) if with_res else None
    y = layer(xd, residual=rd)
    y.backward(cot.to(DEV))
    torch.cuda.synchronize()
    xt = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    w = layer._conv_h.weight.detach().cpu().clone().requires_grad_(True)
    ga = layer._norm_h.weight.detach().cpu().clone().requires_grad_(True)
    be = layer._norm_h.bias.detach().cpu().clone().requires_grad_(True)
    a = F.batch_norm(F.conv2d(xt, w, None, stride, k // 2), torch.zeros(cout), torch.ones(cout), ga, be, True, 0.1, 1e-5)
    rt = None
    if with_res:
        rt = res.float().permute(0, 3, 1, 2).requires_grad_(True)
        a = a + rt
    yt = {"relu": F.relu, "silu": F.silu, None: lambda t: t}[act](a)
    yt.backward(cot.float().permute(0, 3, 1, 2))
    nchw = lambda t: t.detach().float().cpu().permute(0, 3, 1, 2)
    tol = 3.5e-2 if act == "relu" else 8e-3   # ReLU: sign flips of near-zero inputs from the bf16 storage of the conv output
    assert rel_l2(nchw(y), yt.detach()) <= 5e-3
    assert rel_l2(nchw(xd.grad), xt.grad) <= tol
    assert rel_l2(layer._conv_h.weight.grad.cpu(), w.grad) <= tol
    assert rel_l2(layer._norm_h.weight.grad.cpu(), ga.grad) <= tol
    assert rel_l2(layer._norm_h.bias.grad.cpu(), be.grad) <= tol
    if with_res:
        assert rel_l2(nchw(rd.grad), rt.grad) <= tol


@pytest.mark.parametrize("case", [(64, 64, 3, 1, "relu", False), (64, 128, 3, 2, "relu", False), (64, 256, 1, 1, "relu", True),
                                  (256, 64, 1, 1, None, False), (256, 256, 1, 1, "silu", True), (32, 32, 3, 1, "silu", False)])
def test_conv_norm_layer_batch_stat(lib, case):
    """One ConvNormLayer with live BatchNorm (conv -> batch statistics -> affine [+ residual] -> act) forward + backward
    (input, weight, gamma, beta, residual gradients) vs torch fp32 autograd on identical bf16-representable inputs."""
    _bn_layer_case(lib, *case)


def test_bottleneck_pair_batch_stat(lib):
    """Two ResNet-vd bottlenecks (projection shortcut, then identity shortcut; 7 live BatchNorm layers) vs torch autograd of
    the same composition, twice: with straight-through bf16 rounding where the engine stores tensors (tight: proves the
    residual / two-consumer gradient wiring and the BN backward in composition) and in pure fp32 (loose: documents how far
    bf16 storage of the layer outputs moves gradients through ReLU sign flips and the BN backward's cancellations)."""
    from focoos_amd.train_nn import BottleNeck, _Blocks, set_norm_mode
    from tests.helpers import rel_l2

    g = torch.Generator().manual_seed(5)
    net = _Blocks([BottleNeck(lib, 64, 32, 1, False, True), BottleNeck(lib, 128, 32, 1, True, True)])
    set_norm_mode(net, "BN")
    with torch.no_grad():
        for n, p in net.named_parameters():
            if n.endswith("conv.weight"):
                p.copy_((torch.randn(p.shape, generator=g) / (p.shape[1] * p.shape[2] * p.shape[3]) ** 0.5).bfloat16().float())
            elif n.endswith("norm.weight"):
                p.copy_(torch.rand(p.shape, generator=g) + 0.5)
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.3)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.to(DEV)
    x = torch.randn(4, 24, 28, 64, generator=g).clamp_min(0).bfloat16()
    cot = torch.randn(4, 24, 28, 128, generator=g).bfloat16()
    xd = x.to(DEV).requires_grad_(True)
    y = net(xd)
    y.backward(cot.to(DEV))
    torch.cuda.synchronize()
    nchw = lambda t: t.detach().float().cpu().permute(0, 3, 1, 2)

    def reference(rb):
        """The same composition in torch fp32; ``rb`` = identity (the reference's arithmetic) or straight-through bf16 rounding
        at the points where the engine STORES a tensor in bf16 (weight image, layer output; the conv output in front of a live
        BatchNorm stays fp32)."""
        ref = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and "running" not in k else v.clone()) for k, v in sd.items()}
        xt = x.float().permute(0, 3, 1, 2).requires_grad_(True)

        def cbn(p, h, act, res=None):
            z = F.conv2d(h, rb(ref[f"{p}.conv.weight"]), None, 1, (ref[f"{p}.conv.weight"].shape[-1] - 1) // 2)   # kept in fp32 by the engine
            a = F.batch_norm(z, ref[f"{p}.norm.running_mean"], ref[f"{p}.norm.running_var"], ref[f"{p}.norm.weight"], ref[f"{p}.norm.bias"], True, 0.1, 1e-5)
            if res is not None:
                a = a + res
            return rb(F.relu(a) if act else a)

        h = xt
        for bi in range(2):
            p = f"blocks.{bi}"
            short = cbn(f"{p}.short", h, False) if bi == 0 else h
            h = cbn(f"{p}.branch2c", cbn(f"{p}.branch2b", cbn(f"{p}.branch2a", h, True), True), True, short)
        h.backward(cot.float().permute(0, 3, 1, 2))
        return ref, xt, h

    def compare(ref, xt, h):
        e_y, e_x = rel_l2(nchw(y), h.detach()), rel_l2(nchw(xd.grad), xt.grad)
        errs = sorted(((rel_l2(p.grad.cpu(), ref[n].grad), n) for n, p in net.named_parameters()), reverse=True)
        assert len(errs) == 21
        return e_y, e_x, errs

    # (1) against the engine's own storage precision: the ReLU sign pattern is reproduced, what is left is bf16 gradient storage
    e_y, e_x, errs = compare(*reference(lambda t: t + (t.bfloat16().float() - t).detach()))
    print(f"bottleneck pair vs bf16-storage emulation: y {e_y:.4f} dx {e_x:.4f} worst {errs[:2]}")
    assert e_y <= 4e-3 and e_x <= 2.5e-2 and errs[0][0] <= 2.5e-2, (e_y, e_x, errs[:4])
    # (2) against pure fp32 (the reference's arithmetic): six ReLU layers of sign flips near zero, ~2-4% each (see _bn_layer_case)
    e_y, e_x, errs = compare(*reference(lambda t: t))
    print(f"bottleneck pair vs fp32: y {e_y:.4f} dx {e_x:.4f} worst {errs[:2]}")
    assert e_y <= 8e-3 and e_x <= 0.25 and errs[0][0] <= 0.25, (e_y, e_x, errs[:4])


def test_resnet_vd_batch_stat_batchnorm_backward_vs_torch_autograd():
    """ResNet50-vd with LIVE BatchNorm (batch statistics, trainable affine, running-statistics update) through the HIP
    autograd nodes vs torch CPU fp32 autograd of the oracle in BN-training mode (itself pinned to the reference in .train()).
    Tolerances are wide by construction: the batch-statistics backward subtracts from each gradient its per-channel mean and
    its component along xhat, so what is compared is a small remainder of the cotangent, and bf16 storage of the conv
    output (before the normalisation) flips ReLU signs near zero.  Measured: 0.2-5% at the last block growing to ~20% worst /
    9% median over 50 layers for this loss (0.5 * sum feat^2).  The tight checks of the same nodes are
    test_batchnorm_train_kernels, test_conv_norm_layer_batch_stat and test_bottleneck_pair_batch_stat above; this test
    guards the composite wiring (every parameter gets a gradient of the right scale; running statistics; .eval() folding)."""
    from focoos_amd.registry import ModelRegistry
    from focoos_amd.synth import synth_image_structured, synth_state_dict
    from focoos_amd.train_nn import ResNetVd, set_norm_mode
    from oracle import detr_oracle as O
    from tests.helpers import rel_l2

    cfg = ModelRegistry.get_model_info("fai-detr-l-coco")["config"]
    sd = synth_state_dict(cfg, 12)
    pre = "pixel_decoder.backbone."
    bsd = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
    net = set_norm_mode(ResNetVd(50), "BN").to(DEV)
    net.load_state_dict(bsd, strict=True)
    imgs = [synth_image_structured(50 + i, 160, 192) for i in range(4)]
    x_u8 = torch.from_numpy(np.stack(imgs)).to(DEV)
    g = torch.Generator().manual_seed(4)
    outs = net(x_u8)
    proj = ("res2", "res3", "res4", "res5")
    loss = sum((outs[k].float() ** 2).sum() for k in proj) * 0.5e-3
    loss.backward()
    torch.cuda.synchronize()
    ref_sd = {k: (v.clone().requires_grad_(True) if k.endswith(("conv.weight", "norm.weight", "norm.bias")) else v.clone())
              for k, v in sd.items() if k.startswith(pre)}
    mean = torch.tensor(cfg["pixel_mean"]).view(-1, 1, 1)
    std = torch.tensor(cfg["pixel_std"]).view(-1, 1, 1)
    xi = (O.get_torch_batch(imgs, None) - mean) / std
    O.BN_TRAINING[0] = True
    try:
        feats = O.resnet_vd(ref_sd, pre[:-1], xi, O.RESNET_BLOCKS[50])
    finally:
        O.BN_TRAINING[0] = False
    ref_loss = sum((feats[k] ** 2).sum() for k in proj) * 0.5e-3
    ref_loss.backward()
    for k in proj:
        assert rel_l2(outs[k].detach().float().cpu().permute(0, 3, 1, 2), feats[k].detach()) <= 3e-2, k
    errs = []
    for name, p in net.named_parameters():
        assert p.requires_grad and p.grad is not None, name
        errs.append((rel_l2(p.grad.cpu(), ref_sd[pre + name].grad), name))
    errs.sort(reverse=True)
    print("worst gradient rel-L2:", errs[:4], "median", errs[len(errs) // 2][0])
    assert len(errs) == 3 * 55 and errs[0][0] <= 0.35 and errs[len(errs) // 2][0] <= 0.15, errs[:6]
    last = {n: e for e, n in errs if n.startswith("res_layers.3.blocks.2.branch2c")}
    assert max(last.values()) <= 0.08, last
    msd = net.state_dict()
    for k in ("conv1.conv1_1.norm.running_mean", "res_layers.0.blocks.0.short.norm.running_var", "res_layers.3.blocks.2.branch2c.norm.running_mean"):
        assert rel_l2(msd[k].cpu(), ref_sd[pre + k]) <= 1e-2, k
    # .eval() folds the UPDATED running statistics into the conv images
    net.eval()
    with torch.no_grad():
        ev = net(x_u8)
    feats_e = O.resnet_vd({k: v.detach() for k, v in ref_sd.items()}, pre[:-1], xi, O.RESNET_BLOCKS[50])
    assert rel_l2(ev["res5"].float().cpu().permute(0, 3, 1, 2), feats_e["res5"]) <= 3e-2


def test_fused_bottleneck_node_matches_per_layer_nodes():
    """ResNet50-vd backward with every bottleneck as ONE autograd node (ReLU backward and the shortcut-gradient add fused into the
    dgrad convolutions' epilogues) vs the per-layer nodes: same gradients up to the bf16 rounding that the fusion removes."""
    from focoos_amd import train_nn
    from focoos_amd.registry import ModelRegistry
    from focoos_amd.synth import synth_image_structured, synth_state_dict
    from focoos_amd.train_nn import ResNetVd
    from tests.helpers import rel_l2

    cfg = ModelRegistry.get_model_info("fai-detr-l-coco")["config"]
    sd = synth_state_dict(cfg, 11)
    pre = "pixel_decoder.backbone."
    bsd = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
    x_u8 = torch.from_numpy(np.stack([synth_image_structured(40 + i, 128, 160) for i in range(2)])).to(DEV)
    g = torch.Generator().manual_seed(3)
    proj = {k: torch.randn(c, generator=g).to(DEV) for k, c in (("res2", 256), ("res3", 512), ("res4", 1024), ("res5", 2048))}
    grads = {}
    hits0 = train_nn.PREMASK_HITS[0]
    for fused in (True, False):
        train_nn.FUSED_BLOCKS[0] = fused
        try:
            net = ResNetVd(50).to(DEV)
            net.load_state_dict(bsd, strict=True)
            outs = net(x_u8)
            (sum((outs[k].float() * proj[k]).sum() for k in proj) * 1e-2).backward()
            torch.cuda.synchronize()
            grads[fused] = {n: p.grad.clone() for n, p in net.named_parameters() if p.requires_grad}
            feats = {k: v.detach().clone() for k, v in outs.items()}
        finally:
            train_nn.FUSED_BLOCKS[0] = True
        if fused:
            feats_fused = feats
    assert all(torch.equal(feats[k], feats_fused[k]) for k in feats)          # identical forward launches
    # res2 (3 blocks) and res3 (4 blocks): the input gradient of blocks 1.. arrives at block i-1 with relu'(output) already applied
    # (fx_conv_desc.mask) - the hand-over by tensor identity must really have happened: 2 + 3 backward passes without their ReLU pass
    assert train_nn.PREMASK_HITS[0] - hits0 == (5 if train_nn.PREMASK[0] else 0)
    worst = max(rel_l2(grads[True][n], grads[False][n]) for n in grads[True])
    print("fused vs per-layer weight gradients, worst rel-L2:", worst)
    assert worst <= 2e-2


def test_training_convs_on_the_flat_kernels_match_the_implicit_gemm_routing(monkeypatch):
    """Training forward / input-gradient convolutions given the fragment-order weight copies (fx_pack_conv_weights_f32's w_*_frag outputs)
    run on the halo 3x3 / flat pointwise kernels from 20000 output pixels; here the thresholds are lowered so that a small ResNet50-vd
    pass goes through them (incl. the training-only epilogues: ReLU mask of a saved activation, shortcut-gradient add), and the result is
    compared with the same pass routed to the implicit-GEMM kernels: same products, different summation order."""
    from focoos_amd.registry import ModelRegistry
    from focoos_amd.synth import synth_image_structured, synth_state_dict
    from focoos_amd.train_nn import ResNetVd
    from tests.helpers import rel_l2

    cfg = ModelRegistry.get_model_info("fai-detr-l-coco")["config"]
    sd = synth_state_dict(cfg, 12)
    pre = "pixel_decoder.backbone."
    bsd = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
    x_u8 = torch.from_numpy(np.stack([synth_image_structured(50 + i, 128, 160) for i in range(2)])).to(DEV)
    g = torch.Generator().manual_seed(4)
    proj = {k: torch.randn(c, generator=g).to(DEV) for k, c in (("res2", 256), ("res3", 512), ("res4", 1024), ("res5", 2048))}
    res = {}
    for flat in (True, False):
        monkeypatch.setenv("FX_CONV3_MIN_M", "0" if flat else "2000000000")
        monkeypatch.setenv("FX_PW_MIN_M", "0" if flat else "2000000000")
        net = ResNetVd(50).to(DEV)
        net.load_state_dict(bsd, strict=True)
        outs = net(x_u8)
        (sum((outs[k].float() * proj[k]).sum() for k in proj) * 1e-2).backward()
        torch.cuda.synchronize()
        res[flat] = ({k: v.detach().float().clone() for k, v in outs.items()}, {n: p.grad.clone() for n, p in net.named_parameters() if p.requires_grad})
        frags = [m for m in net.modules() if getattr(m, "w_fwd_frag", None) is not None or getattr(m, "w_dgrad_frag", None) is not None]
        assert len(frags) >= 20, len(frags)   # 3x3 of res2-res4 and the 256-multiple pointwise layers of res4 / res5 carry fragment copies
    for k in proj:
        assert rel_l2(res[True][0][k], res[False][0][k]) <= 1e-2, k
    worst = max((rel_l2(res[True][1][n], res[False][1][n]), n) for n in res[True][1])
    print("flat vs implicit-GEMM routing, worst weight-gradient rel-L2:", worst)
    assert worst[0] <= 3e-2, worst
    assert any(not torch.equal(res[True][1][n], res[False][1][n]) for n in res[True][1])   # the routing did change


def test_pack_all_matches_the_per_layer_packing():
    """train_nn.WeightPacker (fx_pack_weights_many_f32: every stale weight image of the registered layers in one launch, fragment-order copies
    written straight from the masters) vs the lazy per-layer path (fx_pack_conv_weights_f32 / fx_pack_linear_weights_f32 +
    fx_pack_frag_bf16): bit-identical images, folded-BatchNorm scale included; versions stamped so that the lazy path has nothing to do."""
    from focoos_amd import train_nn
    from focoos_amd.train_nn import BottleNeck, Linear

    lib = _lib.load()
    g = torch.Generator().manual_seed(9)
    blocks = [BottleNeck(lib, 256, 64, 1, True, False), BottleNeck(lib, 1024, 256, 1, True, False), BottleNeck(lib, 256, 128, 2, False, False)]
    lins = [Linear(lib, 256, 256), Linear(lib, 256, 365), Linear(lib, 4, 512, act="relu"), Linear(lib, 256, 1024, bias=False)]
    mods = torch.nn.ModuleList(blocks + lins).to(DEV)
    with torch.no_grad():
        for p in mods.parameters():
            p.copy_(torch.randn(p.shape, generator=g).to(DEV) * 0.1)
        for m in mods.modules():
            if hasattr(m, "running_var"):
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g).to(DEV) + 0.5)
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g).to(DEV) * 0.1)
    x = {256: torch.randn(2, 8, 8, 256, generator=g).bfloat16().to(DEV), 1024: torch.randn(2, 8, 8, 1024, generator=g).bfloat16().to(DEV),
         4: torch.randn(2, 8, 8, 4, generator=g).bfloat16().to(DEV)}
    for b in blocks:
        b(x[b.branch2a.cin])                       # lazy packing registers the layers
    for l in lins:
        l(x[l.weight.shape[1]])
    convs = [m for m in mods.modules() if isinstance(m, train_nn.ConvNormLayer)]
    packs = [l._pack for l in lins]
    # three equally shaped Linears sharing one image (the decoder's value projections): a group object on one of the modules
    glins = [Linear(lib, 256, 256).to(DEV) for _ in range(3)]
    with torch.no_grad():
        for l in glins:
            l.weight.copy_(torch.randn(256, 256, generator=g).to(DEV) * 0.1)
            l.bias.copy_(torch.randn(256, generator=g).to(DEV) * 0.1)
    mods.extend(glins)
    group = train_nn._PackedLinearGroup()
    glins[0]._group = group
    train_nn._LinearGroupFn.apply(x[256], group, lib, *[l.weight for l in glins], *[l.bias for l in glins])

    def images():
        out = []
        for c in convs:
            out += [c.w_fwd, c.w_dgrad, c.w_fwd_frag, c.w_dgrad_frag, c.shift]
        for pk in packs + [group]:
            out += [pk.w_fwd, pk.w_t, pk.w_fwd_frag, pk.w_t_frag, pk.bias]
        return [None if t is None else t.clone() for t in out]

    with torch.no_grad():
        for p in mods.parameters():
            p.mul_(1.25).add_(0.01)                 # new weights (in place: versions move)
    packer = train_nn.WeightPacker(mods)
    n = packer.pack(DEV)
    assert n == len(convs) + len(packs) + 3, (n, len(convs), len(packs))   # the group contributes one table entry per master
    multi = images()
    assert all(c.pack_fields(torch.device(DEV)) is None for c in convs) and all(pk.pack_fields(torch.device(DEV)) is None for pk in packs + [group])
    assert packer.pack(DEV) == 0                   # nothing stale
    for c in convs:
        c._packed_version = None
        c.sync_packed()
    for l, pk in zip(lins, packs):
        pk.ver = None
        pk.sync(lib, l.weight, l.bias, 0, l.weight.shape[0])
    group.ver = None
    group.sync(lib, [l.weight for l in glins], [l.bias for l in glins])
    torch.cuda.synchronize()
    lazy = images()
    assert sum(t is not None for t in multi) >= 40
    for a, b in zip(multi, lazy):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(a, b)
    assert any(c.w_fwd_frag is not None for c in convs) and any(c.w_dgrad_frag is not None for c in convs) and packs[0].w_fwd_frag is not None


def test_syncbn_sibling_nodes_equal_per_layer_nodes(lib, monkeypatch):
    """SyncBN sibling fusion (train_nn._SiblingConvBnFn): a bottleneck with a pooled projection shortcut (variant d: the
    shortcut conv reads AvgPool(x), branch2a reads x), one with a plain projection shortcut and a CSP layer (conv1 | conv2 + three RepVGG
    blocks) under norm = "SyncBN" with a data-parallel group of TWO ranks faked on one GPU (the collective is a counting stand-in that
    doubles the buffer - what two identical ranks would sum to), once through the sibling nodes and once through the per-layer nodes:
    outputs, input gradient, every parameter gradient and the running statistics are bit-identical, and the collectives drop from two per
    BatchNorm layer (16 + 16) to 10 forward (two shortcut pairs, the CSP pair, three RepVGG pairs) + 13 backward (the RepVGG pairs stay apart)."""
    import torch.distributed as dist

    from focoos_amd import train_nn
    from focoos_amd.train_nn import BottleNeck, CSPRepLayer, _Blocks, set_norm_mode

    calls = []

    def fake_all_reduce(buf, *a, **k):
        calls.append(buf.numel())
        buf.mul_(2.0)          # two ranks holding the same batch

    monkeypatch.setattr(train_nn, "_bn_sync_group", lambda layer: 2 if layer.norm_mode == "SyncBN" else 1)
    monkeypatch.setattr(dist, "all_reduce", fake_all_reduce)

    def build():
        g = torch.Generator().manual_seed(11)
        net = torch.nn.Sequential(_Blocks([BottleNeck(lib, 64, 32, 1, False, True), BottleNeck(lib, 128, 64, 2, False, False)]), CSPRepLayer(lib, 256, 256, 3))
        set_norm_mode(net, "SyncBN")
        with torch.no_grad():
            for n, p in net.named_parameters():
                if n.endswith("conv.weight"):
                    p.copy_((torch.randn(p.shape, generator=g) / (p.shape[1] * p.shape[2] * p.shape[3]) ** 0.5).bfloat16().float())
                elif n.endswith("norm.weight"):
                    p.copy_(torch.rand(p.shape, generator=g) + 0.5)
                else:
                    p.copy_(torch.randn(p.shape, generator=g) * 0.3)
        return net.to(DEV).train()

    g = torch.Generator().manual_seed(12)
    x = torch.randn(2, 24, 28, 64, generator=g).clamp_min(0).bfloat16()
    cot = torch.randn(2, 12, 14, 256, generator=g).bfloat16()
    res = {}
    for on in (True, False):
        monkeypatch.setattr(train_nn, "BN_SIBLINGS", [on])
        del calls[:]
        net = build()
        n_bn = sum(1 for m in net.modules() if isinstance(m, train_nn.ConvNormLayer))
        xd = x.to(DEV).requires_grad_(True)
        y = net(xd)
        fwd_calls = len(calls)
        y.backward(cot.to(DEV))
        torch.cuda.synchronize()
        res[on] = {"y": y.detach().clone(), "dx": xd.grad.clone(), "grads": {n: p.grad.clone() for n, p in net.named_parameters()},
                   "state": {k: v.clone() for k, v in net.state_dict().items()}, "fwd": fwd_calls, "bwd": len(calls) - fwd_calls, "n_bn": n_bn}
    a, b = res[True], res[False]
    assert b["n_bn"] == 16 and b["fwd"] == 16 and b["bwd"] == 16                       # per layer: one collective each way
    # pairs: 2 bottleneck shortcuts + CSP conv1|conv2 (both directions) + 3 RepVGG blocks (forward only)
    assert a["fwd"] == 16 - 6 and a["bwd"] == 16 - 3, (a["fwd"], a["bwd"])
    assert torch.equal(a["y"], b["y"]) and torch.equal(a["dx"], b["dx"])
    for n in b["grads"]:
        assert torch.equal(a["grads"][n], b["grads"][n]), n
    for k in b["state"]:
        assert torch.equal(a["state"][k], b["state"][k]), k
    assert float(b["y"].float().abs().max()) > 0 and float(b["dx"].float().abs().max()) > 0
