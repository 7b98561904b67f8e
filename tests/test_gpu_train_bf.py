"""The whole BiSeNetFormer training step (BASELINE config 5: forward in training mode, 7 supervised prediction
heads, point-sampled Hungarian criterion, backward, optimizer) on the HIP autograd graph vs the CPU fp32 training oracle
(oracle/train_oracle.bf_train_outputs / bf_criterion, pinned against the real reference fully in .train() by
tests/test_oracle_vs_reference.py::test_bf_train_oracle_matches_reference_losses_and_gradients).
Discrete choices are teacher-forced to the oracle's: the boolean attention masks (their `< 0` test flips under any rounding change), the
Hungarian matches (the GPU matcher is checked bit-exactly in tests/test_gpu_mask_criterion.py) and the torch.rand draws of the point
sampling (inputs of the C ABI).  Tolerances (bf16 activations / gradients, fp32 losses and weight gradients): each of the 21 losses within
3 % (+1e-3) with frozen BatchNorm, 6 % with batch statistics; per-parameter gradient relative L2: see the asserts (measured values in
DESIGN.md §2)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from focoos_amd.ports import MaskFormerTargets  # noqa: E402
from focoos_amd.registry import ModelRegistry  # noqa: E402
from focoos_amd.synth import synth_image_structured, synth_state_dict  # noqa: E402
from oracle import detr_oracle as O  # noqa: E402
from oracle import train_oracle as T  # noqa: E402
from tests.helpers import rel_l2  # noqa: E402

DEV = "cuda:0"


class _DrawAndRecord:
    """mask_criterion_oracle.RandStream interface that draws from a seeded generator and records, for the engine's replay."""

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
        t = self.t[self.i]
        assert tuple(t.shape) == tuple(shape), (self.i, tuple(t.shape), shape)
        self.i += 1
        return t.to(device)


def _cfg(num_points=2048, variant="bisenetformer-l-ade"):
    cfg = ModelRegistry.get_model_info(variant)["config"]
    return dict(cfg, criterion_num_points=num_points)   # 12544 in the registry: minutes on the CPU oracle, same code path


@pytest.mark.parametrize("norm,size,variant", [("FrozenBN", (192, 256), "bisenetformer-l-ade"), ("BN", (192, 256), "bisenetformer-l-ade"),
                                               ("FrozenBN", (150, 200), "bisenetformer-l-ade"), ("FrozenBN", (192, 256), "bisenetformer-m-ade")])
def test_bf_train_step_losses_and_gradients(norm, size, variant):
    """size (150, 200): not a multiple of 32 - ceil(H/2) at every stride-2 layer of the forward AND of its adjoints (tests/test_gpu_odd_sizes.py
    covers the inference engine).  bisenetformer-m-ade: 96-channel pixel decoder / mask dimension, four decoder layers."""
    from focoos_amd.train_bf import BisenetFormerTrainable

    cfg = _cfg(variant=variant)
    n_losses = 3 * (int(cfg["transformer_predictor_dec_layers"]) + 1)
    if norm == "BN":
        # Batch statistics on a random-weight STDC (no residual connections) amplify ANY perturbation ~1.5x per CatBottleneck: through the
        # 12 blocks of STDC-2 the 0.4 % bf16 rounding of the first activations grows to 60 % at res5 (measured; every block reproduces the
        # oracle to 0.6 % when fed the oracle's input, and the frozen-statistics run of the same 12 blocks stays at 0.9 %).  The live-BN
        # plumbing of the whole model - conv / depthwise / pooled-vector BatchNorms, their gradients and running statistics - is
        # therefore compared on the 3-block STDC (layers 1-1-1: the same modules, a composition shallow enough to be well conditioned).
        cfg = dict(cfg, backbone_config=dict(cfg["backbone_config"], layers=[1, 1, 1]))
    sd = synth_state_dict(cfg, 31, family="bisenetformer")
    nimg, (ih, iw) = (4 if norm == "BN" else 2), size
    imgs = [synth_image_structured(60 + i, ih, iw) for i in range(nimg)]
    labels, masks = T.synth_mask_targets(5, nimg, int(cfg["num_classes"]), (ih, iw), counts=(3, 5, 2, 4))

    def trainable(k, v):
        if not (v.dtype == torch.float32 and v.dim() > 0) or any(t in k for t in ("running_", "empty_weight")):
            return False
        is_bn = k.endswith((".bn.weight", ".bn.bias", ".bn_atten.weight", ".bn_atten.bias", ".avd_layer.1.weight", ".avd_layer.1.bias"))
        return norm != "FrozenBN" or not is_bn

    sdg = {k: (v.clone().requires_grad_(True) if trainable(k, v) else v.clone()) for k, v in sd.items()}
    x = O.get_torch_batch(imgs, None)
    col = {}
    O.BN_TRAINING[0] = norm != "FrozenBN"
    try:
        outs = T.bf_train_outputs(sdg, cfg, x, collect=col)
    finally:
        O.BN_TRAINING[0] = False
    rs = _DrawAndRecord(77)
    losses_o, matches = T.bf_criterion(outs, labels, masks, rs, cfg)
    sum(losses_o.values()).backward()
    sens = None
    if variant != "bisenetformer-l-ade" or size != (192, 256):   # (the odd-size case measured 0.248 against the absolute 0.25: gate it per tensor too)
        # this configuration's own conditioning: the fp32 oracle again with nothing but the weights rounded to bf16 (same attention masks,
        # matches and draws); each tensor is then gated against ITS sensitivity as well as the absolute gate (tests/test_gpu_detr_variants.py)
        from oracle.mask_criterion_oracle import RandStream

        sdb = {k: ((v.detach().bfloat16().float() if v.dim() >= 2 else v.detach().clone()).requires_grad_(v.requires_grad)) if v.dtype == torch.float32
               else v.clone() for k, v in sdg.items()}
        outs_w = T.bf_train_outputs(sdb, cfg, x, forced_attn=col["attn_masks"])
        lw, _ = T.bf_criterion(outs_w, labels, masks, RandStream(rs.rec), cfg, fixed_matches=matches)
        sum(lw.values()).backward()
        sens = {k: rel_l2(sdb[k].grad, sdg[k].grad) for k in sdg if isinstance(sdg[k], torch.Tensor) and sdg[k].requires_grad and sdb[k].grad is not None}
    # ---- HIP autograd graph
    model = BisenetFormerTrainable(cfg, norm=norm, rand=_Replay(rs.rec)).to(DEV)
    model.load_state_dict(sd, strict=True)
    assert sorted(model.state_dict().keys()) == sorted(sd.keys())
    model.train()
    targets = [MaskFormerTargets(labels=l.to(DEV), masks=m.to(DEV)) for l, m in zip(labels, masks)]
    fixed = []
    for m in matches:
        pi = torch.tensor(np.concatenate([np.asarray(i) for i, _ in m]), dtype=torch.int32, device=DEV)
        ti = torch.tensor(np.concatenate([np.asarray(j) for _, j in m]), dtype=torch.int32, device=DEV)
        fixed.append((pi, ti))
    x_u8 = torch.from_numpy(np.stack(imgs)).to(DEV)
    losses = model(x_u8, targets, forced_attn=col["attn_masks"], fixed_matches=fixed)
    sum(losses.values()).backward()
    torch.cuda.synchronize()
    assert sorted(losses) == sorted(losses_o) and len(losses) == n_losses
    for k in losses_o:
        a, b = float(losses[k]), float(losses_o[k])
        assert abs(a - b) <= (6e-2 if norm == "BN" else 3e-2) * abs(b) + 1e-3, (k, a, b)
    # the supervised mask logits of the last head against the oracle's (teacher-forced attention): a direct forward check
    pm = model.last_outputs["pred_masks"].detach().float().cpu()
    pm_err = rel_l2(pm, outs["pred_masks"].detach())
    print(f"{norm}: last-head mask logits rel-L2 {pm_err:.4f}")
    errs = []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        r = sdg[name]
        if not (isinstance(r, torch.Tensor) and r.requires_grad):
            continue
        assert p.grad is not None, name
        assert r.grad is not None, name
        errs.append((rel_l2(p.grad.cpu(), r.grad), name, float(r.grad.norm())))
    floor = 1e-3 * sorted(n for _, _, n in errs)[len(errs) // 2]
    print("zero-gradient tensors skipped:", [n for _, n, g in errs if g < floor])
    errs = [(e, n) for e, n, g in errs if g >= floor]
    errs.sort(reverse=True)
    print(f"{norm}: {len(errs)} parameter tensors; worst 8: {[(round(e, 4), n) for e, n in errs[:8]]}; median {errs[len(errs) // 2][0]:.4f}")
    print("quartiles:", [round(errs[len(errs) * q // 4][0], 4) for q in (1, 2, 3)])
    assert len(errs) > (180 if sens is None else 140)
    if norm == "FrozenBN":
        assert pm_err <= 4e-2
    if norm == "BN":
        dec = sorted(e for e, n in errs if n.startswith("head.predictor."))
        assert dec[len(dec) // 2] <= 0.12, dec[len(dec) // 2]
        assert errs[len(errs) // 2][0] <= 0.40 and errs[len(errs) // 10][0] <= 0.60, errs[:8]
        msd = model.state_dict()
        for k in ("pixel_decoder.backbone.features.0.bn.running_mean", "pixel_decoder.backbone.features.3.avd_layer.1.running_var",
                  "pixel_decoder.cp.arm16.bn_atten.running_mean", "pixel_decoder.conv_out.bn.running_var"):
            assert rel_l2(msd[k].cpu(), sdg[k]) <= 2e-2, k
            assert not torch.equal(sdg[k], sd[k])
    elif sens is None:
        assert errs[0][0] <= 0.25, errs[:8]
        assert errs[len(errs) // 2][0] <= 0.08
    else:
        sw = sorted(sens.values(), reverse=True)
        print(f"bf16-weights-only oracle: worst {sw[0]:.4f}, median {sw[len(sw) // 2]:.4f}")
        assert errs[len(errs) // 2][0] <= 0.08
        bad = [(round(e, 4), round(sens.get(n, 0.0), 4), n) for e, n in errs if e > max(0.25, 3.0 * sens.get(n, 0.0))]
        assert not bad, bad


def test_bf_train_step_free_running_and_optimizer():
    """TrainStep on the BiSeNetFormer graph, nothing teacher-forced (GPU Hungarian matcher, torch.rand on the device, attention masks
    from the engine's own mask logits): losses finite and close to the teacher-forced values' scale, every trainable parameter receives
    a gradient and moves, a second step runs on the updated weights, running statistics move under norm="BN"."""
    from focoos_amd.train_bf import BisenetFormerTrainable
    from focoos_amd.train_detr import TrainStep

    cfg = _cfg(1024)
    sd = synth_state_dict(cfg, 32, family="bisenetformer")
    model = BisenetFormerTrainable(cfg, norm="BN").to(DEV)
    model.load_state_dict(sd, strict=True)
    model.train()
    ts = TrainStep(model, lr=1e-4, max_grad_norm=0.1)
    imgs = [synth_image_structured(90 + i, 128, 160) for i in range(4)]
    labels, masks = T.synth_mask_targets(6, 4, int(cfg["num_classes"]), (128, 160), counts=(2, 0, 4, 1))   # one image without targets
    targets = [MaskFormerTargets(labels=l.to(DEV), masks=m.to(DEV)) for l, m in zip(labels, masks)]
    x_u8 = torch.from_numpy(np.stack(imgs)).to(DEV)
    p0 = ts.opt.flat_p.clone()
    l1 = {k: float(v) for k, v in ts.step(x_u8, targets).items()}
    l2 = {k: float(v) for k, v in ts.step(x_u8, targets).items()}
    torch.cuda.synchronize()
    assert len(l1) == 21 and all(np.isfinite(v) for v in l1.values()) and all(np.isfinite(v) for v in l2.values())
    assert not torch.equal(ts.opt.flat_p, p0)
    dead = [n for n, _ in ts.named if float(ts.opt.grads[n].abs().max()) == 0.0]   # the flat gradient views still hold step 2's gradients
    assert not dead, dead[:10]
    assert int(model.state_dict()["pixel_decoder.backbone.features.0.bn.num_batches_tracked"]) == 2


@pytest.mark.parametrize("counts", [(0, 0), (0, 4)])
def test_mask_train_step_with_empty_targets(counts):
    """Images without a single ground-truth mask - both images, or one of two (mask SetCriterion: num_masks clamped to 1, empty matches, the
    mask / dice losses sum over nothing; bisenetformer/loss.py = fai_mf/loss.py:345-607, 661-723): losses of the BiSeNetFormer training
    graph against the fp32 training oracle (attention masks and point draws teacher-forced), finite gradients everywhere."""
    from focoos_amd.train_bf import BisenetFormerTrainable

    cfg = _cfg(variant="bisenetformer-m-ade")
    sd = synth_state_dict(cfg, 35, family="bisenetformer")
    ih, iw = 128, 160
    imgs = [synth_image_structured(70 + i, ih, iw) for i in range(2)]
    labels, masks = T.synth_mask_targets(9, 2, int(cfg["num_classes"]), (ih, iw), counts=counts)
    x = O.get_torch_batch(imgs, None)
    col = {}
    with torch.no_grad():
        outs = T.bf_train_outputs(sd, cfg, x, collect=col)
        rs = _DrawAndRecord(79)
        losses_o, matches = T.bf_criterion(outs, labels, masks, rs, cfg)
    # (the reference's torch.rand of a (0, n, 2) tensor for a set without masks draws nothing from the generator; the engine skips the call)
    model = BisenetFormerTrainable(cfg, norm="FrozenBN", rand=_Replay([t for t in rs.rec if t.numel() > 0])).to(DEV)
    model.load_state_dict(sd, strict=True)
    model.train()
    targets = [MaskFormerTargets(labels=l.to(DEV), masks=m.to(DEV)) for l, m in zip(labels, masks)]
    fixed = None
    if sum(counts) > 0:
        fixed = []
        for m in matches:
            pi = torch.tensor(np.concatenate([np.asarray(i, dtype=np.int64) for i, _ in m]), dtype=torch.int32, device=DEV)
            ti = torch.tensor(np.concatenate([np.asarray(j, dtype=np.int64) for _, j in m]), dtype=torch.int32, device=DEV)
            fixed.append((pi, ti))
    losses = model(torch.from_numpy(np.stack(imgs)).to(DEV), targets, forced_attn=col["attn_masks"], fixed_matches=fixed)
    sum(losses.values()).backward()
    torch.cuda.synchronize()
    assert sorted(losses) == sorted(losses_o)
    for k in losses_o:
        a, b = float(losses[k]), float(losses_o[k])
        assert np.isfinite(a) and abs(a - b) <= 3e-2 * abs(b) + 1e-3, (k, a, b)
        if sum(counts) == 0 and ("mask" in k or "dice" in k):
            assert a == 0.0, (k, a)
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.requires_grad and p.grad is not None)
