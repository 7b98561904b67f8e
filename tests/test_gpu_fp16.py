"""The fp16 element type (libfocoos_amd_fp16.so: the same kernels compiled with -DFX_FP16=1) and the loss-scaled training step - BASELINE
configs[4] names fp16; the reference trains under torch.autocast(float16) + GradScaler(init_scale=2**10) (trainer/trainer.py:645,735-773).

What is checked on a real MI355X:
* the library identifies itself (fx_build_flags bit 1) and the loader refuses a library of the other element type;
* weight-gradient / forward / input-gradient convolution kernels on fp16 operands vs torch fp32 on the SAME fp16-rounded operands (a kernel that
  still decoded its 16-bit words as bfloat16 would be off by orders of magnitude, not by a rounding);
* ResNet-50-vd forward + backward through the HIP autograd nodes vs fp32 torch autograd of the oracle: with 11 significand bits the gates are
  HALF of the bf16 test's (features 1e-2, weight gradients 3e-2);
* the whole BiSeNetFormer training step (frozen BatchNorm, full STDC-2, teacher-forced attention masks / matches / draws) against the fp32
  training oracle: 21 losses within 1 % (the real reference's own fp16-autocast losses deviate 0.66 % from fp32:
  tests/test_oracle_vs_reference.py::test_reference_fp16_amp_losses_vs_fp32_and_bf16_autocast), gradient rel-L2 median <= 3 %;
  and the RT-DETR training step (deformable attention, box criterion) with the bf16 test's gates;
* the dynamic loss scale: gradients carry the scale, the fused AdamW launch unscales (update identical to an unscaled bf16-free reference
  computed in torch), an injected inf skips the step on the device (parameters, moments and Adam's step count untouched, scale halved),
  `growth_interval` good steps double it.
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from focoos_amd import _lib  # noqa: E402
from focoos_amd._lib import check  # noqa: E402
from tests.helpers import rel_l2  # noqa: E402

DEV = "cuda:0"


@pytest.fixture(autouse=True)
def fp16_element_type():
    prev = _lib.set_compute_dtype("fp16")
    yield
    _lib.set_compute_dtype(prev)


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def test_fp16_library_identifies_itself():
    lib = _lib.load()
    assert lib.fx_build_flags() & 2 and _lib.act_dtype() == torch.float16
    _lib.set_compute_dtype("bf16")
    lib_bf = _lib.load()
    assert not (lib_bf.fx_build_flags() & 2) and lib_bf is not lib and _lib.act_dtype() == torch.bfloat16
    with pytest.raises(_lib.FocoosAmdError):
        _lib.set_compute_dtype("fp8")
    _lib.set_compute_dtype("fp16")
    from focoos_amd.engine import DetrEngine
    from focoos_amd.registry import ModelRegistry

    with pytest.raises(_lib.FocoosAmdError):   # inference engines are bf16: refused loudly under the fp16 element type
        DetrEngine(ModelRegistry.get_model_info("fai-detr-l-coco")["config"], {}, DEV)


@pytest.mark.parametrize("case", [(2, 20, 24, 64, 64, 3, 1), (2, 17, 19, 128, 256, 1, 1), (1, 32, 32, 32, 64, 3, 1), (3, 16, 16, 128, 128, 3, 2),
                                  (2, 40, 40, 256, 256, 3, 1), (2, 25, 31, 256, 512, 1, 1)])
def test_fp16_conv_wgrad(case):
    """tests/test_gpu_train_conv.py::test_conv_wgrad on fp16 operands (incl. the wide-layer conv_wgrad_dma_kernel cases)."""
    lib = _lib.load()
    B, H, W, Cc, N, k, stride = case
    pad = (k - 1) // 2
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, H, W, Cc, generator=g).half()
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    dz = torch.randn(B, Ho, Wo, N, generator=g).half()
    w = torch.zeros(N, Cc, k, k, requires_grad=True)
    F.conv2d(x.float().permute(0, 3, 1, 2), w, None, stride=stride, padding=pad).backward(dz.float().permute(0, 3, 1, 2))
    ref = w.grad.permute(0, 2, 3, 1).contiguous()
    xd, dzd = x.to(DEV), dz.to(DEV)
    S = lib.fx_conv2d_wgrad_splits(B, Ho, Wo, Cc, N, k, k)
    slab = N * k * k * Cc
    ws = torch.full((S * slab,), float("nan"), dtype=torch.float32, device=DEV)
    check(lib.fx_conv2d_wgrad_partial_nhwc_bf16(xd.data_ptr(), Cc, dzd.data_ptr(), N, ws.data_ptr(), slab, S, B, H, W, Cc, Ho, Wo, N, k, k, stride, pad,
                                                stream()))
    out = torch.zeros(N, Cc, k, k, dtype=torch.float32, device=DEV)
    check(lib.fx_unpack_conv_wgrad_sum_f32(ws.data_ptr(), slab, S, None, out.data_ptr(), N, Cc, k, k, Cc, 0, stream()))
    torch.cuda.synchronize()
    err = (out.cpu().permute(0, 2, 3, 1) - ref).abs().max() / ref.abs().max()
    assert err < 1e-3, f"rel err {err}"     # fp16 operands are exact inputs: only the fp32 accumulation order differs


@pytest.mark.parametrize("case", [(2, 40, 48, 256, 256, 3, 1, "relu"), (2, 40, 48, 64, 64, 3, 1, "relu"), (2, 40, 48, 256, 1024, 1, 1, None),
                                  (2, 40, 48, 512, 256, 1, 1, "silu"), (2, 20, 24, 128, 128, 3, 2, "relu"), (2, 32, 32, 32, 64, 3, 1, "relu"),
                                  (4, 12, 12, 96, 192, 1, 1, None)])
def test_fp16_conv_layer_forward_and_backward(case, flat_small_shapes):
    """One ConvNormLayer (frozen BatchNorm folded) on every kernel route of the training graph - k-plane 3x3, stem c32, pointwise k-plane / flat,
    stride-2 k-plane, implicit GEMM - forward, input gradient and weight gradient vs torch fp32 autograd on the fp16-rounded operands."""
    from focoos_amd import train_nn
    from focoos_amd.train_nn import ConvNormLayer, set_norm_mode

    B, H, W, Cc, N, k, stride, act = case
    g = torch.Generator().manual_seed(sum(c for c in case if isinstance(c, int)))
    layer = ConvNormLayer(_lib.load(), Cc, N, k, stride, act).to(DEV)
    set_norm_mode(layer, "FrozenBN")
    with torch.no_grad():
        layer._conv_h.weight.copy_(torch.randn(N, Cc, k, k, generator=g) / (Cc * k * k) ** 0.5)
        layer._norm_h.weight.copy_(torch.rand(N, generator=g) + 0.5)
        layer._norm_h.bias.copy_(torch.randn(N, generator=g) * 0.1)
        layer._norm_h.running_mean.copy_(torch.randn(N, generator=g) * 0.1)
        layer._norm_h.running_var.copy_(torch.rand(N, generator=g) + 0.5)
    layer.train()
    x = (torch.randn(B, H, W, Cc, generator=g)).half()
    xd = x.to(DEV).requires_grad_(True)
    train_nn.WEIGHTS_EPOCH[0] += 1
    y = layer(xd)
    assert y.dtype == torch.float16
    dy = torch.randn(y.shape, generator=g).half()
    y.backward(dy.to(DEV))
    torch.cuda.synchronize()
    # reference: fp32 autograd with the folded weights rounded to fp16 (what the packed image holds), fp16 input, fp16 output gradient
    wt = layer._conv_h.weight.detach().cpu().clone().requires_grad_(True)
    nm = layer._norm_h
    s = (nm.weight / torch.sqrt(nm.running_var + 1e-5)).detach().cpu()
    shift = (nm.bias - nm.running_mean * nm.weight / torch.sqrt(nm.running_var + 1e-5)).detach().cpu()
    xr = x.float().permute(0, 3, 1, 2).clone().requires_grad_(True)
    w_eff = wt * s.view(-1, 1, 1, 1)
    w16 = w_eff + (w_eff.half().float() - w_eff).detach()           # straight-through rounding
    z = F.conv2d(xr, w16, shift, stride=stride, padding=(k - 1) // 2)
    yr = {"relu": F.relu, "silu": F.silu, None: lambda t: t}[act](z)
    yr.backward(dy.float().permute(0, 3, 1, 2))
    got = y.detach().float().cpu().permute(0, 3, 1, 2)
    assert (got - yr.detach()).abs().max() <= 2e-3 * yr.detach().abs().max() + 1e-3, (got - yr.detach()).abs().max()
    gx = xd.grad.float().cpu().permute(0, 3, 1, 2)
    assert rel_l2(gx, xr.grad) <= 3e-3, rel_l2(gx, xr.grad)
    assert rel_l2(layer._conv_h.weight.grad.cpu(), wt.grad) <= 3e-3, rel_l2(layer._conv_h.weight.grad.cpu(), wt.grad)


def test_fp16_resnet_vd_backward_vs_torch_autograd():
    """tests/test_gpu_train_conv.py::test_resnet_vd_backward_vs_torch_autograd under fp16: gates at half the bf16 ones."""
    from focoos_amd.registry import ModelRegistry
    from focoos_amd.synth import synth_image_structured, synth_state_dict
    from focoos_amd.train_nn import ResNetVd
    from oracle import detr_oracle as O

    cfg = ModelRegistry.get_model_info("fai-detr-l-coco")["config"]
    sd = synth_state_dict(cfg, 11)
    pre = "pixel_decoder.backbone."
    bsd = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
    net = ResNetVd(50).to(DEV)
    net.load_state_dict(bsd, strict=True)
    imgs = [synth_image_structured(40 + i, 128, 160) for i in range(2)]
    x_u8 = torch.from_numpy(np.stack(imgs)).to(DEV)
    g = torch.Generator().manual_seed(3)
    proj = {k: torch.randn(c, generator=g) for k, c in (("res2", 256), ("res3", 512), ("res4", 1024), ("res5", 2048))}
    outs = net(x_u8)
    assert all(outs[k].dtype == torch.float16 for k in proj)
    loss = sum((outs[k].float() * proj[k].to(DEV)).sum() for k in proj) * 1e-2
    loss.backward()
    torch.cuda.synchronize()
    ref_sd = {k: (v.clone().requires_grad_(True) if k.endswith("conv.weight") else v) for k, v in sd.items() if k.startswith(pre)}
    mean = torch.tensor(cfg["pixel_mean"]).view(-1, 1, 1)
    std = torch.tensor(cfg["pixel_std"]).view(-1, 1, 1)
    feats = O.resnet_vd(ref_sd, pre[:-1], (O.get_torch_batch(imgs, None) - mean) / std, O.RESNET_BLOCKS[50])
    ref_loss = sum((feats[k] * proj[k].view(1, -1, 1, 1)).sum() for k in proj) * 1e-2
    ref_loss.backward()
    fe = {k: rel_l2(outs[k].detach().float().cpu().permute(0, 3, 1, 2), feats[k].detach()) for k in proj}
    worst = max(rel_l2(p.grad.cpu(), ref_sd[pre + n].grad) for n, p in net.named_parameters() if p.requires_grad)
    print("fp16 ResNet-50-vd: feature rel-L2", {k: round(v, 5) for k, v in fe.items()}, "worst weight-gradient rel-L2", round(worst, 5))
    assert max(fe.values()) <= 1e-2 and worst <= 3e-2


class _DrawAndRecord:
    def __init__(self, seed):
        self.g = torch.Generator().manual_seed(seed)
        self.rec = []

    def take(self, *shape):
        t = torch.rand(*shape, generator=self.g)
        self.rec.append(t)
        return t


class _Replay:
    def __init__(self, tensors):
        self.t, self.i = list(tensors), 0

    def __call__(self, *shape, device):
        t = self.t[self.i % len(self.t)]
        assert tuple(t.shape) == tuple(shape), (self.i, tuple(t.shape), shape)
        self.i += 1
        return t.to(device)


def _fixed(matches):
    return [(torch.tensor(np.concatenate([np.asarray(i) for i, _ in m]), dtype=torch.int32, device=DEV),
             torch.tensor(np.concatenate([np.asarray(j) for _, j in m]), dtype=torch.int32, device=DEV)) for m in matches]


def _grad_errs(named_grads, sdg):
    errs = []
    for name, gr in named_grads:
        r = sdg[name]
        if isinstance(r, torch.Tensor) and r.requires_grad and r.grad is not None and gr is not None:
            errs.append((rel_l2(gr.float().cpu(), r.grad), name, float(r.grad.norm())))
    floor = 1e-3 * sorted(n for _, _, n in errs)[len(errs) // 2]
    return sorted(((e, n) for e, n, g in errs if g >= floor), reverse=True)


def test_bf_train_step_fp16_losses_gradients_and_loss_scale():
    """BASELINE configs[4]'s model through the fp16 step (frozen BatchNorm for the comparison, as in tests/test_gpu_train_bf.py): losses within
    1 % of the fp32 oracle, gradients (read from the flat buffer, divided by the loss scale) tighter than the bf16 step's; then three
    optimizer steps with the dynamic scale: finite, parameters move, no step skipped."""
    from focoos_amd.ports import MaskFormerTargets
    from focoos_amd.registry import ModelRegistry
    from focoos_amd.synth import synth_image_structured, synth_state_dict
    from focoos_amd.train_bf import BisenetFormerTrainable
    from focoos_amd.train_detr import TrainStep
    from oracle import detr_oracle as O
    from oracle import train_oracle as T

    cfg = dict(ModelRegistry.get_model_info("bisenetformer-l-ade")["config"], criterion_num_points=2048)
    sd = synth_state_dict(cfg, 31, family="bisenetformer")
    nimg, (ih, iw) = 2, (192, 256)
    imgs = [synth_image_structured(60 + i, ih, iw) for i in range(nimg)]
    labels, masks = T.synth_mask_targets(5, nimg, int(cfg["num_classes"]), (ih, iw), counts=(3, 5, 2, 4))

    def trainable(k, v):
        if not (v.dtype == torch.float32 and v.dim() > 0) or any(t in k for t in ("running_", "empty_weight")):
            return False
        return not k.endswith((".bn.weight", ".bn.bias", ".bn_atten.weight", ".bn_atten.bias", ".avd_layer.1.weight", ".avd_layer.1.bias"))

    sdg = {k: (v.clone().requires_grad_(True) if trainable(k, v) else v.clone()) for k, v in sd.items()}
    col = {}
    outs = T.bf_train_outputs(sdg, cfg, O.get_torch_batch(imgs, None), collect=col)
    rs = _DrawAndRecord(77)
    losses_o, matches = T.bf_criterion(outs, labels, masks, rs, cfg)
    sum(losses_o.values()).backward()

    model = BisenetFormerTrainable(cfg, norm="FrozenBN", rand=_Replay(rs.rec)).to(DEV)
    model.load_state_dict(sd, strict=True)
    model.train()
    stepper = TrainStep(model)
    assert stepper.dtype_name == "fp16" and stepper.opt.scaler is not None
    st0 = stepper.opt.scaler_state()
    assert st0 == {"scale": 1024.0, "growth_tracker": 0, "good_steps": 0, "skipped_steps": 0}
    targets = [MaskFormerTargets(labels=l.to(DEV), masks=m.to(DEV)) for l, m in zip(labels, masks)]
    x = torch.from_numpy(np.stack(imgs)).to(DEV)
    losses = stepper.forward_backward(x, targets, forced_attn=col["attn_masks"], fixed_matches=_fixed(matches))
    torch.cuda.synchronize()
    assert model.last_outputs["pred_masks"].dtype in (torch.float16, torch.float32)
    worst = max(abs(float(losses[k]) - float(losses_o[k])) / abs(float(losses_o[k])) for k in losses_o)
    errs = _grad_errs([(n, stepper.opt.grads[n] / 1024.0) for n, _ in stepper.named], sdg)
    print(f"fp16 BiSeNetFormer step: worst loss deviation {worst:.4f}; {len(errs)} gradient tensors, worst 4 {[(round(e, 4), n) for e, n in errs[:4]]}, "
          f"quartiles {[round(errs[len(errs) * q // 4][0], 4) for q in (1, 2, 3)]}")
    assert worst <= 1e-2
    assert len(errs) > 180 and errs[0][0] <= 0.15 and errs[len(errs) // 2][0] <= 3e-2
    # ---- optimizer steps under the dynamic scale
    before = stepper.opt.flat_p.clone()
    for _ in range(3):
        out = stepper.step(x, targets)
    torch.cuda.synchronize()
    st = stepper.opt.scaler_state()
    assert st["good_steps"] == 3 and st["skipped_steps"] == 0 and st["scale"] == 1024.0 and st["growth_tracker"] == 3, st
    assert all(np.isfinite(float(v)) for v in out.values())
    assert torch.isfinite(stepper.opt.flat_p).all() and not torch.equal(before, stepper.opt.flat_p)


def test_detr_train_step_fp16_losses_and_gradients():
    """RT-DETR training step under fp16 (deformable attention forward / binning backward, box criterion, encoder heads on selected rows) vs the
    fp32 oracle, discrete choices teacher-forced: the bf16 test's gates (tests/test_gpu_train_detr.py) must hold with room to spare."""
    from focoos_amd.ports import DETRTargets
    from focoos_amd.registry import ModelRegistry
    from focoos_amd.synth import synth_image_structured, synth_state_dict
    from focoos_amd.train_detr import FAIDetrTrainable, TrainStep
    from oracle import detr_oracle as O
    from oracle import train_oracle as T

    cfg = ModelRegistry.get_model_info("fai-detr-l-coco")["config"]
    sd = synth_state_dict(cfg, 21)
    k_qk = "pixel_decoder.encoder.0.layers.0.self_attn.in_proj_weight"
    sd[k_qk] = sd[k_qk].clone()
    sd[k_qk][:512] *= 0.05
    nimg, (ih, iw) = 2, (128, 160)
    imgs = [synth_image_structured(80 + i, ih, iw) for i in range(nimg)]
    labels, boxes = T.synth_targets(2, nimg, 80, counts=(4, 6, 2, 5))

    def trainable(k, v):
        if not (v.dtype == torch.float32 and v.dim() > 0) or any(t in k for t in ("running_", "empty_weight", "mask_features")):
            return False
        return not (k.endswith((".norm.weight", ".norm.bias")) or (".input_proj." in k and k.split(".")[-2] == "1"))

    sdg = {k: (v.clone().requires_grad_(True) if trainable(k, v) else v.clone()) for k, v in sd.items()}
    outs = T.detr_train_outputs(sdg, cfg, O.get_torch_batch(imgs, None))
    losses_o, matches = T.criterion(outs, labels, boxes)
    sum(losses_o.values()).backward()
    model = FAIDetrTrainable(cfg, norm="FrozenBN").to(DEV)
    model.load_state_dict(sd, strict=True)
    model.train()
    stepper = TrainStep(model)
    targets = [DETRTargets(labels=l.to(DEV), boxes=b.to(DEV)) for l, b in zip(labels, boxes)]
    x = torch.from_numpy(np.stack(imgs)).to(DEV)
    losses = stepper.forward_backward(x, targets, forced_topk=outs["topk_ind"].to(DEV), fixed_matches=_fixed(matches))
    torch.cuda.synchronize()
    worst = max(abs(float(losses[k]) - float(losses_o[k])) / (abs(float(losses_o[k])) + 1e-3) for k in losses_o)
    errs = _grad_errs([(n, stepper.opt.grads[n] / 1024.0) for n, _ in stepper.named], sdg)
    print(f"fp16 RT-DETR step: worst loss deviation {worst:.4f}; {len(errs)} gradient tensors, worst 4 {[(round(e, 4), n) for e, n in errs[:4]]}, "
          f"quartiles {[round(errs[len(errs) * q // 4][0], 4) for q in (1, 2, 3)]}")
    assert worst <= 3e-2
    assert len(errs) > 250 and errs[0][0] <= 0.25 and errs[len(errs) // 2][0] <= 0.08


def test_loss_scale_unscale_skip_and_growth():
    """fx_adamw_step_scaled_f32 against torch: (1) gradients carrying the scale give the update torch.optim.AdamW + clip_grad_norm_ give on the
    unscaled gradients; (2) an inf anywhere skips the step on the device - parameters, moments, Adam's step count untouched - and halves the
    scale; (3) the step after a skip uses bias-correction step 2, not 3; (4) growth_interval good steps double the scale."""
    from focoos_amd.train import FlatAdamW

    g = torch.Generator().manual_seed(0)
    shapes = [("a", (300, 7), 1e-3, 1e-2), ("b", (70000,), 5e-4, 0.0)]
    opt = FlatAdamW(shapes, DEV, max_grad_norm=0.5, loss_scale=256.0, growth_interval=2)
    ref_p = [torch.randn(*s, generator=g).requires_grad_(True) for _, s, _, _ in shapes]
    ref = torch.optim.AdamW([{"params": [p], "lr": lr, "weight_decay": wd} for p, (_, _, lr, wd) in zip(ref_p, shapes)], betas=(0.9, 0.999), eps=1e-8)
    for p, (n, _, _, _) in zip(ref_p, shapes):
        opt.params[n].copy_(p.detach().to(DEV))

    def grads():
        return [torch.randn(*s, generator=g) for _, s, _, _ in shapes]

    def ref_step(gs):
        for p, gr in zip(ref_p, gs):
            p.grad = gr.clone()
        torch.nn.utils.clip_grad_norm_(ref_p, 0.5)
        ref.step()

    def dev_step(gs, scale, poison=False):
        for (n, _, _, _), gr in zip(shapes, gs):
            opt.grads[n].copy_((gr * scale).to(DEV))
        if poison:
            opt.grads["b"][12345] = float("inf")
        opt.step()
        torch.cuda.synchronize()

    gs = grads()
    ref_step(gs)
    dev_step(gs, 256.0)
    for p, (n, _, _, _) in zip(ref_p, shapes):
        assert (opt.params[n].cpu() - p.detach()).abs().max() <= 2e-6, n
    st = opt.scaler_state()
    assert st == {"scale": 256.0, "growth_tracker": 1, "good_steps": 1, "skipped_steps": 0}
    snap = (opt.flat_p.clone(), opt.flat_m.clone(), opt.flat_v.clone())
    dev_step(grads(), 256.0, poison=True)                       # skipped: nothing moves, the scale backs off
    assert torch.equal(snap[0], opt.flat_p) and torch.equal(snap[1], opt.flat_m) and torch.equal(snap[2], opt.flat_v)
    st = opt.scaler_state()
    assert st == {"scale": 128.0, "growth_tracker": 0, "good_steps": 1, "skipped_steps": 1}
    assert not np.isfinite(float(opt.total_norm))
    gs = grads()
    ref_step(gs)                                               # torch's second step: bias correction with t = 2
    dev_step(gs, 128.0)
    for p, (n, _, _, _) in zip(ref_p, shapes):
        assert (opt.params[n].cpu() - p.detach()).abs().max() <= 4e-6, n
    gs = grads()
    ref_step(gs)
    dev_step(gs, 128.0)                                        # second consecutive good step: growth_interval = 2 -> the scale doubles
    st = opt.scaler_state()
    assert st == {"scale": 256.0, "growth_tracker": 0, "good_steps": 3, "skipped_steps": 1}
    for p, (n, _, _, _) in zip(ref_p, shapes):
        assert (opt.params[n].cpu() - p.detach()).abs().max() <= 6e-6, n
