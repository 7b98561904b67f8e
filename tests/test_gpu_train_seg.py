"""Training-direction kernels of the mask families (BASELINE config 5) on a real MI355X, each against
torch CPU fp32 autograd of the same op on bf16-rounded operands:
  * attention backward on the matrix cores, unmasked and with the boolean attention mask (incl. fully masked rows and partial tiles);
  * depthwise 3x3 stride-2 conv / AvgPool2d(3,2,1) backward, gate / mean / broadcast reductions, plane -> row transposition;
  * gradients of the point-sampled mask criterion (fx_mask_set_loss_bwd_f32) against autograd through the oracle's restatement.
Tolerances: products of bf16 operands accumulated in fp32, outputs rounded to bf16 -> <= 2-3 % of the gradient's max (attention:
P and dS enter the second MFMA as bf16, like the forward kernel); fp32 criterion gradients 2e-4."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from focoos_amd import _lib  # noqa: E402
from focoos_amd._lib import check  # noqa: E402
from focoos_amd.engine_maskdec import pack_mask_bits  # noqa: E402

DEV = "cuda:0"


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available()
    return _lib.load()


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dev(t, dtype=None):
    return t.to(device=DEV, dtype=dtype or t.dtype).contiguous()


MHA_CASES = [
    # B, Lq, Lk, masked
    (2, 130, 130, False),     # RT-DETR-like self-attention, partial tiles on both sides
    (1, 300, 300, False),
    (2, 400, 400, False),     # AIFI
    (2, 100, 64, True),       # BiSeNetFormer level 0 at 256x256
    (2, 100, 1024, True),     # level 0 at 1024x1024 (several LDS chunks of keys)
    (1, 100, 4100, True),     # many chunks, partial last tile
    (3, 37, 449, True),
    (1, 300, 77, False),      # more queries than keys (several LDS chunks of queries in the dK/dV kernel)
]


@pytest.mark.parametrize("case", MHA_CASES)
def test_mha_backward_mfma(lib, case):
    B, Lq, Lk, masked = case
    g = torch.Generator().manual_seed(Lq * 7 + Lk)
    q = (torch.randn(B, Lq, 256, generator=g) * 1.5).bfloat16()
    k = (torch.randn(B, Lk, 256, generator=g) * 1.5).bfloat16()
    v = torch.randn(B, Lk, 256, generator=g).bfloat16()
    do = torch.randn(B, Lq, 256, generator=g).bfloat16()
    mask, bits, words = None, None, (Lk + 31) // 32
    if masked:
        mask = torch.rand(B, Lq, Lk, generator=g) < 0.7
        mask[:, 3] = True           # fully masked query: attends everywhere
        mask[:, 5] = False
        mask[:, 7, : Lk - 1] = True  # a single allowed key, in the last (partial) tile
        mask[:, 7, Lk - 1] = False
        bits = dev(pack_mask_bits(mask.reshape(B * Lq, Lk), words))
    qd, kd, vd, dod = dev(q), dev(k), dev(v), dev(do)
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    nb = lib.fx_mha_bwd_workspace_bytes(B, Lq, Lk, 8)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    check(lib.fx_mha_masked_bwd_bf16(qd.data_ptr(), 256, kd.data_ptr(), 256, vd.data_ptr(), 256, dod.data_ptr(), 256, dq.data_ptr(), 256, dk.data_ptr(), 256,
                                     dv.data_ptr(), 256, B, Lq, Lk, 8, bits.data_ptr() if masked else None, words, ws.data_ptr(), nb, stream()))
    torch.cuda.synchronize()
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
    qh, kh, vh = (t.view(B, -1, 8, 32).transpose(1, 2) for t in (qr, kr, vr))
    s = qh @ kh.transpose(-1, -2) / math.sqrt(32)
    if masked:
        eff = mask & (mask.sum(-1, keepdim=True) != Lk)
        s = s.masked_fill(eff[:, None], float("-inf"))
    out = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, Lq, 256)
    out.backward(do.float())
    for got, ref, nm in ((dq, qr.grad, "dq"), (dk, kr.grad, "dk"), (dv, vr.grad, "dv")):
        got = got.float().cpu()
        assert not torch.isnan(got).any(), nm
        err = (got - ref).abs().max() / ref.abs().max()
        assert err <= 3e-2, (nm, float(err))
        rel = (got - ref).norm() / ref.norm()
        assert rel <= 1.5e-2, (nm, float(rel))


@pytest.mark.parametrize("shape", [(2, 17, 20, 64), (1, 32, 32, 128), (3, 9, 7, 256), (2, 64, 64, 32)])
def test_dwconv3x3s2_backward(lib, shape):
    B, H, W, Cc = shape
    g = torch.Generator().manual_seed(H * W)
    x = torch.randn(B, H, W, Cc, generator=g).bfloat16()
    w = torch.randn(Cc, 1, 3, 3, generator=g) * 0.3
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    dy = torch.randn(B, Ho, Wo, Cc, generator=g).bfloat16()
    w9 = w[:, 0].permute(1, 2, 0).reshape(9, Cc).contiguous()
    xd, dyd, wd = dev(x), dev(dy), dev(w9)
    dx = torch.empty_like(xd)
    dw = torch.zeros(9, Cc, dtype=torch.float32, device=DEV)
    check(lib.fx_dwconv3x3s2_bwd_nhwc_bf16(dyd.data_ptr(), Cc, xd.data_ptr(), Cc, wd.data_ptr(), dx.data_ptr(), Cc, dw.data_ptr(), B, H, W, Cc, stream()))
    torch.cuda.synchronize()
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    F.conv2d(xr, wr, None, stride=2, padding=1, groups=Cc).backward(dy.float().permute(0, 3, 1, 2))
    assert (dx.float().cpu() - xr.grad.permute(0, 2, 3, 1)).abs().max() <= 1e-2 * xr.grad.abs().max()
    ref_dw = wr.grad[:, 0].permute(1, 2, 0).reshape(9, Cc)
    assert (dw.cpu() - ref_dw).abs().max() <= 2e-3 * ref_dw.abs().max()
    # AvgPool2d(3, 2, 1) backward = the same dgrad with w = 1/9
    pw = torch.full((9, Cc), 1.0 / 9.0, dtype=torch.float32, device=DEV)
    check(lib.fx_dwconv3x3s2_bwd_nhwc_bf16(dyd.data_ptr(), Cc, None, 0, pw.data_ptr(), dx.data_ptr(), Cc, None, B, H, W, Cc, stream()))
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    F.avg_pool2d(xr, 3, 2, 1).backward(dy.float().permute(0, 3, 1, 2))
    assert (dx.float().cpu() - xr.grad.permute(0, 2, 3, 1)).abs().max() <= 1e-2 * xr.grad.abs().max()


def test_rowdot_bcast_planes_to_rows(lib):
    g = torch.Generator().manual_seed(3)
    B, P, Cc = 3, 1000, 128
    a = torch.randn(B, P, Cc, generator=g).bfloat16()
    b = torch.randn(B, P, Cc, generator=g).bfloat16()
    ad, bd = dev(a), dev(b)
    for splits in (1, 5):
        for second in (True, False):
            out = torch.zeros(B, Cc, dtype=torch.float32, device=DEV)
            check(lib.fx_rowdot_nhwc_bf16(ad.data_ptr(), Cc, bd.data_ptr() if second else None, Cc, 0.5, out.data_ptr(), Cc, B, P, Cc, splits, stream()))
            ref = 0.5 * ((a.float() * b.float()) if second else a.float()).sum(1)
            assert (out.cpu() - ref).abs().max() <= 1e-4 * ref.abs().max() + 1e-4
    vec = torch.randn(B, Cc, generator=g)
    y = torch.empty(B, P, Cc, dtype=torch.bfloat16, device=DEV)
    check(lib.fx_bcast_vec_nhwc_bf16(dev(vec).data_ptr(), Cc, 0.25, y.data_ptr(), Cc, B, P, Cc, stream()))
    assert torch.equal(y.cpu(), (vec * 0.25).bfloat16()[:, None, :].expand(B, P, Cc))
    Q, Pp, Qp = 100, 777, 128
    planes = torch.randn(B, Q, Pp, generator=g)
    rows = torch.full((B, Pp, Qp), float("nan"), dtype=torch.bfloat16, device=DEV)
    check(lib.fx_planes_to_rows_bf16(dev(planes).data_ptr(), Q, Pp, rows.data_ptr(), Qp, Qp, B, stream()))
    want = torch.zeros(B, Pp, Qp)
    want[:, :, :Q] = planes.transpose(1, 2)
    assert torch.equal(rows.cpu(), want.bfloat16())


def test_mask_set_loss_gradients_vs_oracle_autograd():
    """fx_mask_set_loss_bwd_f32 through the host mirror's autograd node vs torch autograd through oracle/mask_criterion_oracle.py on
    the same random draws and the same matches (the oracle is bit-pinned to the reference on the reference's recorded torch.rand stream,
    tests/test_mask_criterion_oracle.py)."""
    from focoos_amd.mask_criterion import MaskHungarianMatcher, SetCriterion
    from oracle import mask_criterion_oracle as O

    K, Q, P = 80, 100, 1024
    outputs, tgt_labels, tgt_masks = O.synth_mask_predictions_and_targets(seed=5, B=2, Q=Q, K=K, hw=(40, 48), scale=4, counts=(6, 3), n_aux=2)
    draws = []

    def rec_rand(*shape, device=None):
        t = torch.rand(*shape, generator=rec_rand.g)
        draws.append(t)
        return t.to(device) if device is not None else t

    rec_rand.g = torch.Generator().manual_seed(11)

    class T:
        def __init__(self, l, m):
            self.labels, self.masks = l, m

    targets = [T(l, m) for l, m in zip(tgt_labels, tgt_masks)]
    wd = {"loss_ce": 2.0, "loss_mask": 5.0, "loss_dice": 5.0}
    crit = SetCriterion(K, MaskHungarianMatcher(2.0, 5.0, 5.0, P, rand=rec_rand), wd, eos_coef=0.1, num_points=P, oversample_ratio=3.0,
                        importance_sample_ratio=0.75, rand=rec_rand)

    def leafs(o):
        return o["pred_logits"].clone().to(DEV).requires_grad_(True), o["pred_masks"].clone().to(DEV).requires_grad_(True)

    sets = [outputs] + list(outputs["aux_outputs"])
    dl = [leafs(o) for o in sets]
    out_d = {"pred_logits": dl[0][0], "pred_masks": dl[0][1], "aux_outputs": [{"pred_logits": a, "pred_masks": b} for a, b in dl[1:]]}
    losses = crit(out_d, targets)
    coef = {k: 0.5 + 0.1 * i for i, k in enumerate(sorted(losses))}   # distinct upstream gradients per loss
    sum(coef[k] * v for k, v in losses.items()).backward()
    torch.cuda.synchronize()
    # oracle on the recorded draws
    rl = [(o["pred_logits"].clone().requires_grad_(True), o["pred_masks"].clone().requires_grad_(True)) for o in sets]
    out_r = {"pred_logits": rl[0][0], "pred_masks": rl[0][1], "aux_outputs": [{"pred_logits": a, "pred_masks": b} for a, b in rl[1:]]}
    ref = O.criterion(out_r, tgt_labels, tgt_masks, O.RandStream(draws), K, num_points=P, weights=(wd["loss_ce"], wd["loss_mask"], wd["loss_dice"]))
    ref_losses = ref[0] if isinstance(ref, tuple) else ref
    sum(coef[k] * v for k, v in ref_losses.items()).backward()
    for k in ref_losses:
        assert abs(float(losses[k]) - float(ref_losses[k])) <= 3e-5 * max(1.0, abs(float(ref_losses[k]))), k
    for (gl, gm), (rl_, rm) in zip(dl, rl):
        for got, want, nm in ((gl.grad.cpu(), rl_.grad, "dlogits"), (gm.grad.cpu(), rm.grad, "dmasks")):
            assert (got - want).abs().max() <= 2e-4 * want.abs().max() + 1e-9, (nm, float((got - want).abs().max()), float(want.abs().max()))
