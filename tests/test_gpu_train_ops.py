"""GPU parity of the training-side kernels: ms_deform_attn_core fp32 forward + backward (vs autograd through the oracle's
restatement of the reference function) and the fused flat AdamW + grad-norm clip (vs torch.optim.AdamW + clip_grad_norm_)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from focoos_amd.train import FlatAdamW, ms_deform_attn_core  # noqa: E402
from oracle import detr_oracle as O  # noqa: E402
from tests._cases import MSDA_SHAPES, msda_case_inputs  # noqa: E402
from tests.helpers import load_golden  # noqa: E402

DEV = "cuda:0"


def test_msda_forward_backward_vs_reference_function():
    value, loc, w = (torch.from_numpy(a) for a in msda_case_inputs())
    gold = torch.from_numpy(load_golden("msda_core.npz")["out"])
    g = torch.Generator().manual_seed(3)
    go = torch.randn(gold.shape, generator=g)
    # CPU: autograd through the restated reference function
    vc, lc, wc = value.clone().requires_grad_(), loc.clone().requires_grad_(), w.clone().requires_grad_()
    out_c = O.ms_deform_attn_core(vc, MSDA_SHAPES, lc, wc)
    out_c.backward(go)
    # GPU kernel pair behind the same signature
    vg, lg, wg = (t.to(DEV).requires_grad_() for t in (value, loc, w))
    out_g = ms_deform_attn_core(vg, MSDA_SHAPES, lg, wg)
    out_g.backward(go.to(DEV))
    torch.cuda.synchronize()
    assert (out_g.cpu() - gold).abs().max() < 2e-5          # forward vs the reference's golden output (fp32)
    for name, a, b in (("value", vg.grad.cpu(), vc.grad), ("loc", lg.grad.cpu(), lc.grad), ("attn", wg.grad.cpu(), wc.grad)):
        err = (a - b).abs().max() / b.abs().max()
        assert err < 2e-5, (name, float(err))


def test_flat_adamw_matches_torch_adamw_with_clipping():
    g = torch.Generator().manual_seed(0)
    shapes = [("backbone.w", (64, 32, 3, 3), 1e-5, 1e-4), ("head.w", (365, 256), 1e-4, 1e-4), ("norm.g", (256,), 1e-4, 0.0), ("big", (300, 700), 1e-4, 5e-2)]
    opt = FlatAdamW(shapes, DEV, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=0.1)
    ref_params = []
    for name, shape, lr, wd in shapes:
        p = torch.randn(*shape, generator=g)
        opt.params[name].copy_(p)
        ref_params.append(torch.nn.Parameter(p.clone()))
    ref = torch.optim.AdamW([{"params": [p], "lr": s[2], "weight_decay": s[3]} for p, s in zip(ref_params, shapes)], betas=(0.9, 0.999), eps=1e-8)
    for step in range(4):
        scale = [3.0, 0.02, 1.0, 0.3][step]  # steps with and without active clipping
        for (name, shape, _, _), p in zip(shapes, ref_params):
            gr = torch.randn(*shape, generator=g) * scale * 1e-2
            p.grad = gr.clone()
            opt.grads[name].copy_(gr)
        total = torch.nn.utils.clip_grad_norm_(ref_params, 0.1)
        ref.step()
        opt.step()
        torch.cuda.synchronize()
        assert abs(float(opt.total_norm) - float(total)) <= 1e-5 * float(total)
        for (name, _, _, _), p in zip(shapes, ref_params):
            assert (opt.params[name].cpu() - p.detach()).abs().max() < 2e-6, (step, name)


def test_flat_adamw_skips_the_update_on_non_finite_gradients():
    """An inf / NaN anywhere in the flat gradient: parameters and both moments stay bit-identical, the reported norm is inf, and the next
    finite step updates as usual (what GradScaler does for the reference's default amp training, trainer.py:645,735-773; without a loss
    scale there is nothing to back off, the step is only dropped)."""
    g = torch.Generator().manual_seed(1)
    shapes = [("a.w", (64, 32, 3, 3), 1e-4, 1e-4), ("b.w", (300, 700), 1e-4, 5e-2)]
    opt = FlatAdamW(shapes, DEV, max_grad_norm=0.1)
    for name, shape, _, _ in shapes:
        opt.params[name].copy_(torch.randn(*shape, generator=g))
        opt.grads[name].copy_(torch.randn(*shape, generator=g) * 1e-2)
    opt.step()
    torch.cuda.synchronize()
    p0, m0, v0 = opt.flat_p.clone(), opt.flat_m.clone(), opt.flat_v.clone()
    for bad in (float("nan"), float("inf"), -float("inf")):
        opt.grads["b.w"][17, 3] = bad
        opt.step()
        torch.cuda.synchronize()
        assert torch.equal(opt.flat_p, p0) and torch.equal(opt.flat_m, m0) and torch.equal(opt.flat_v, v0)
        assert float(opt.total_norm) == float("inf")
    opt.grads["b.w"][17, 3] = 0.5
    opt.step()
    torch.cuda.synchronize()
    assert torch.isfinite(opt.flat_p).all() and not torch.equal(opt.flat_p, p0) and np.isfinite(float(opt.total_norm))


def test_box_refine_forward_backward_vs_torch_autograd():
    """sigmoid(delta + inverse_sigmoid(ref)) (fai_detr/modelling.py:1003, 1010; functional.py:4-6) as fx_box_refine_f32 / _bwd vs torch
    autograd of the reference expression on the same bf16 deltas: fp32 both sides -> 1e-6 absolute forward, 1e-5 relative backward.
    ref includes the clamp edges (0, 1, below eps, outside [0, 1])."""
    from focoos_amd.train_detr import _BoxRefineFn

    g = torch.Generator().manual_seed(5)
    delta = (torch.randn(3, 50, 4, generator=g) * 2).bfloat16()
    ref = torch.rand(3, 50, 4, generator=g)
    ref[0, 0] = torch.tensor([0.0, 1.0, 1e-7, 0.5])
    ref[0, 1] = torch.tensor([-0.1, 1.2, 1.0 - 1e-7, 1e-5])
    go = torch.randn(3, 50, 4, generator=g)
    dc, rc = delta.float().requires_grad_(), ref.clone().requires_grad_()
    out_c = torch.sigmoid(dc + O.inverse_sigmoid(rc))
    out_c.backward(go)
    dg, rg = delta.to(DEV).requires_grad_(), ref.to(DEV).requires_grad_()
    out_g = _BoxRefineFn.apply(dg, rg)
    out_g.backward(go.to(DEV))
    torch.cuda.synchronize()
    assert (out_g.detach().cpu() - out_c.detach()).abs().max() <= 1e-6
    assert (dg.grad.float().cpu() - dc.grad).abs().max() <= 8e-3 * dc.grad.abs().max()          # bf16 gradient
    assert (rg.grad.cpu() - rc.grad).abs().max() <= 1e-5 * rc.grad.abs().max(), (rg.grad.cpu() - rc.grad).abs().max()
    # detached reference points: no gradient tensor for them
    dg2 = delta.to(DEV).requires_grad_()
    _BoxRefineFn.apply(dg2, ref.to(DEV)).backward(go.to(DEV))
    assert torch.equal(dg2.grad, dg.grad)


@pytest.mark.parametrize("slab", ["1", "0"])
@pytest.mark.parametrize("shapes,Q,crowd", [(MSDA_SHAPES, 40, False), ([[80, 80], [40, 40], [20, 20]], 300, False), ([[3, 400], [7, 9]], 161, False),
                                            ([[80, 80], [40, 40], [20, 20]], 300, True)])
def test_grouped_msda_matches_the_per_layer_function(shapes, Q, crowd, slab, monkeypatch):
    _grouped_msda_case(shapes, Q, crowd, slab, monkeypatch)


@pytest.mark.parametrize("Q", [300, 75])
def test_grouped_msda_fp32_grad_out(Q):
    """The fp32-grad_out instantiation of the binning backward (msda_bwd_value_kernel<float>: 16-byte LDS stores of the staged gradient
    rows - ADVICE r3: their base was only 8-byte aligned for Q*P = 1200 and no test passed fp32 gradients).  Autograd always hands the
    node a bf16 grad_out (its output is bf16), so the backward core is called directly: the same gradient values as fp32 and as bf16 must
    give identical point gradients and the same value gradient (up to the summation order)."""
    from focoos_amd.train import ValueGradSink, _msda_group_backward, _shape_host, _shape_tensors

    shapes = [[80, 80], [40, 40], [20, 20]]
    B, M, D, L, P = 2, 8, 32, 3, 4
    S = sum(h * w for h, w in shapes)
    g = torch.Generator().manual_seed(80 + Q)
    value = torch.randn(B, S, 256, generator=g).bfloat16().to(DEV)
    lc = (torch.rand(B, Q, M, L, P, 2, generator=g) * 1.2 - 0.1).to(DEV)
    aw = torch.softmax(torch.randn(B, Q, M, L * P, generator=g), -1).view(B, Q, M, L, P).to(DEV)
    go = torch.randn(B, Q, 256, generator=g).bfloat16().to(DEV)
    st, ss = _shape_tensors(shapes, value.device)
    res = []
    for grad in (go, go.float()):
        sink = ValueGradSink(1)
        gv, gl, ga = _msda_group_backward(value, st, ss, lc, aw, grad, sink, 0, (B, S, Q, M, D, L, P, 256), _shape_host(shapes))
        torch.cuda.synchronize()
        assert sink.slab is True and gv.dtype == torch.bfloat16 and torch.isfinite(gv.float()).all()
        res.append((gv, gl, ga))
    (gv_a, gl_a, ga_a), (gv_b, gl_b, ga_b) = res
    assert torch.equal(gl_a, gl_b) and torch.equal(ga_a, ga_b)     # point gradients: same arithmetic on the same values
    # value gradient: a pixel's taps are filed in the order the LDS atomics of a launch happen to retire, so the fp32 summation order (and
    # the bf16 rounding of the result) may differ between ANY two launches - same tolerance as the grouped test above
    a, b = gv_a.float(), gv_b.float()
    assert ((a - b).abs() <= 8e-3 * b.abs() + 2e-5 * b.abs().max()).all(), float((a - b).abs().max())


def _grouped_msda_case(shapes, Q, crowd, slab, monkeypatch):
    """ms_deform_attn_grouped (G layers reading column slices of one bf16 value tensor, value gradients delivered together by the last
    layer to run) vs ms_deform_attn_core on the same bf16-rounded values, layer by layer: identical forward (same kernel code, typed
    loads), identical location / weight gradients, value gradient equal up to the order of the fp32 additions and the bf16 rounding of
    the delivered tensor.  Both forms of the value gradient: LDS slabs (fx_msda_train_bwd_slab: several slabs per level, more queries
    than one staging pass, sampling points outside the map) and fp32 L2 atomics (fx_msda_train_bwd)."""
    from focoos_amd.train import ValueGradSink, ms_deform_attn_grouped

    monkeypatch.setenv("FX_MSDA_BWD_SLAB", slab)
    G, B = 3, 2
    S = sum(h * w for h, w in shapes)
    g = torch.Generator().manual_seed(8)
    value_all = torch.randn(B, S, G * 256, generator=g).bfloat16().to(DEV).requires_grad_()
    if crowd:   # every query samples around one spot: thousands of taps on a handful of pixels (the set-aside path of the binned kernel)
        locs = [(0.37 + 0.004 * torch.randn(B, Q, 8, len(shapes), 4, 2, generator=g)).to(DEV).requires_grad_() for _ in range(G)]
    else:
        locs = [(torch.rand(B, Q, 8, len(shapes), 4, 2, generator=g) * 1.2 - 0.1).to(DEV).requires_grad_() for _ in range(G)]
    aws = [torch.softmax(torch.randn(B, Q, 8, len(shapes) * 4, generator=g), -1).view(B, Q, 8, len(shapes), 4).to(DEV).requires_grad_() for _ in range(G)]
    gos = [torch.randn(B, Q, 256, generator=g).bfloat16().to(DEV) for _ in range(G)]
    sink = ValueGradSink(G)
    outs = [ms_deform_attn_grouped(value_all, sink, i, shapes, locs[i], aws[i]) for i in range(G)]
    torch.autograd.backward(outs, gos)
    torch.cuda.synchronize()
    assert sink.count == 0 and sink.buf is None            # delivered and reset
    gv_all = value_all.grad.float()
    assert torch.isfinite(gv_all).all()
    for i in range(G):
        v = value_all.detach()[:, :, i * 256:(i + 1) * 256].float().reshape(B, S, 8, 32).requires_grad_()
        lc, aw = locs[i].detach().clone().requires_grad_(), aws[i].detach().clone().requires_grad_()
        ref = ms_deform_attn_core(v, shapes, lc, aw)
        ref.backward(gos[i].float())
        torch.cuda.synchronize()
        assert torch.equal(outs[i].detach(), ref.detach().to(torch.bfloat16))
        if slab == "0":   # same kernel code for the point gradients
            assert torch.equal(locs[i].grad, lc.grad) and torch.equal(aws[i].grad, aw.grad)
        else:             # the slab form reduces over (8 lanes x 4 channels) instead of 32 lanes: fp32 summation order
            assert (locs[i].grad - lc.grad).abs().max() <= 2e-5 * lc.grad.abs().max()
            assert (aws[i].grad - aw.grad).abs().max() <= 2e-5 * aw.grad.abs().max()
        gref = v.grad.reshape(B, S, 256)
        # bf16 rounding of the delivered gradient (2^-9 relative) + fp32 summation order
        assert ((gv_all[:, :, i * 256:(i + 1) * 256] - gref).abs() <= 4e-3 * gref.abs() + 2e-5 * gref.abs().max()).all()


def test_raw_projection_msda_node_matches_the_torch_op_prelude():
    """ms_deform_attn_grouped_raw (softmax + location arithmetic of MSDeformableAttention.forward inside the node: fx_msda_prep_bf16 /
    fx_msda_prep_bwd_bf16) vs the same arithmetic as torch ops in front of ms_deform_attn_grouped: equal forward, gradients of the raw
    projections equal to bf16 rounding."""
    from focoos_amd.train import ValueGradSink, ms_deform_attn_grouped, ms_deform_attn_grouped_raw

    shapes = [[80, 80], [40, 40], [20, 20]]
    B, Q, M, L, P = 2, 77, 8, 3, 4
    S = sum(h * w for h, w in shapes)
    g = torch.Generator().manual_seed(31)
    value_all = torch.randn(B, S, 256, generator=g).bfloat16().to(DEV)
    off0 = (torch.randn(B, Q, M * L * P * 2, generator=g) * 3).bfloat16().to(DEV)
    lg0 = (torch.randn(B, Q, M * L * P, generator=g) * 2).bfloat16().to(DEV)
    ref = torch.cat([torch.rand(B, Q, 2, generator=g), torch.rand(B, Q, 2, generator=g) * 0.4 + 0.02], -1).to(DEV)
    go = torch.randn(B, Q, 256, generator=g).bfloat16().to(DEV)
    outs, grads = [], []
    for raw in (False, True):
        v = value_all.clone().requires_grad_()
        off, lg = off0.clone().requires_grad_(), lg0.clone().requires_grad_()
        sink = ValueGradSink(1)
        if raw:
            out = ms_deform_attn_grouped_raw(v, sink, 0, shapes, off, lg, ref, M, L, P)
        else:
            o = off.float().view(B, Q, M, L, P, 2)
            aw = torch.softmax(lg.float().view(B, Q, M, L * P), -1).view(B, Q, M, L, P)
            r4 = ref.unsqueeze(2)
            loc = r4[:, :, None, :, None, :2] + o / P * r4[:, :, None, :, None, 2:] * 0.5
            out = ms_deform_attn_grouped(v, sink, 0, shapes, loc, aw)
        out.backward(go)
        torch.cuda.synchronize()
        outs.append(out.detach().float())
        grads.append((v.grad.float(), off.grad.float(), lg.grad.float()))
    assert (outs[0] - outs[1]).abs().max() <= 2e-2 * outs[0].abs().max()
    for a, b in zip(grads[0], grads[1]):
        assert (a - b).abs().max() <= 1.2e-2 * a.abs().max(), (a - b).abs().max() / a.abs().max()
