"""The MaskFormer (fai-mf-*) training step on the HIP autograd graph vs the CPU fp32 training oracle
(oracle/train_oracle.mf_train_outputs / bf_criterion, pinned against the real reference in .train() by
tests/test_oracle_vs_reference.py::test_mf_train_oracle_matches_reference_losses_and_gradients).  Attention masks, Hungarian matches and
the point-sampling draws are teacher-forced as in tests/test_gpu_train_bf.py.  Tolerances: 30 losses within 3 % (+1e-3) with frozen
BatchNorm; per-parameter gradient relative L2 as asserted below (measured values in DESIGN.md §2)."""
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
from tests.test_gpu_train_bf import _DrawAndRecord, _Replay  # noqa: E402

DEV = "cuda:0"


def _cfg(depth, num_points=2048):
    cfg = dict(ModelRegistry.get_model_info("fai-mf-l-coco-ins")["config"], criterion_num_points=num_points)
    cfg["backbone_config"] = dict(cfg["backbone_config"], depth=depth)
    return cfg


@pytest.mark.parametrize("norm,variant,size", [("FrozenBN", "r50", (192, 256)), ("BN", "r50", (192, 256)), ("FrozenBN", "fai-mf-m-ade", (192, 256)),
                                               ("FrozenBN", "r50", (150, 200)), ("FrozenBN", "fai-mf-s-coco-ins", (192, 256))])
def test_mf_train_step_losses_and_gradients(norm, variant, size):
    """size (150, 200): not a multiple of 32 (ceil-size forward and adjoints: partial AvgPool2d(2,2,ceil_mode) windows, nearest up-sampling with
    non-integer ratios, 5 x 7 encoder tokens).  fai-mf-s-coco-ins: the 128-channel pixel decoder WITH its three-layer transformer encoder
    (8 heads of 16 channels zero-padded onto the head-dim-32 attention kernels, LayerNorm over 128 columns) and six decoder layers."""
    from focoos_amd.train_mf import FAIMaskFormerTrainable

    if variant == "r50":
        cfg = _cfg(50)   # R50: the R101 of fai-mf-l is the same blocks, 17 more of them in res4
    else:                # STDC-2 backbone under the 128-channel FPN, three decoder layers (focoos/model_registry/fai-mf-m-ade.json)
        cfg = dict(ModelRegistry.get_model_info(variant)["config"], criterion_num_points=2048)
    n_losses = 3 * (int(cfg["transformer_predictor_dec_layers"]) + 1)
    sd = synth_state_dict(cfg, 41, family="fai_mf")
    for k in sd:     # keep the six pre-norm encoder layers' attention logits O(1) (see tests/test_gpu_train_conv.py on AIFI)
        if ".transformer.encoder.layers." in k and k.endswith("self_attn.in_proj_weight"):
            sd[k] = sd[k].clone()
            sd[k][: 2 * sd[k].shape[1]] *= 0.05      # the q and k rows (512 at 256 channels, 256 at the 128 channels of fai-mf-{m,s}-coco-ins)
    if variant != "r50":   # no pixel-decoder encoder (and no LayerNorm) in front of the decoder: keep ITS attention logits O(1) as well
        for k in sd:
            if k.startswith("head.predictor.") and k.endswith("in_proj_weight"):
                sd[k] = sd[k].clone()
                sd[k][:512] *= 0.25
    nimg, (ih, iw) = (4 if norm == "BN" else 2), size
    imgs = [synth_image_structured(160 + i, ih, iw) for i in range(nimg)]
    labels, masks = T.synth_mask_targets(7, nimg, int(cfg["num_classes"]), (ih, iw), counts=(3, 5, 2, 4))

    def trainable(k, v):
        if not (v.dtype == torch.float32 and v.dim() > 0) or any(t in k for t in ("running_", "empty_weight")):
            return False
        is_bn = (k.endswith((".norm.weight", ".norm.bias")) and ".transformer." not in k) or k.endswith(
            (".bn.weight", ".bn.bias", ".avd_layer.1.weight", ".avd_layer.1.bias"))       # the latter: STDC's BatchNorms
        return norm != "FrozenBN" or not is_bn

    sdg = {k: (v.clone().requires_grad_(True) if trainable(k, v) else v.clone()) for k, v in sd.items()}
    x = O.get_torch_batch(imgs, None)
    col = {}
    O.BN_TRAINING[0] = norm != "FrozenBN"
    try:
        outs = T.mf_train_outputs(sdg, cfg, x, collect=col)
    finally:
        O.BN_TRAINING[0] = False
    rs = _DrawAndRecord(78)
    losses_o, matches = T.bf_criterion(outs, labels, masks, rs, cfg)
    sum(losses_o.values()).backward()
    sens = None
    if variant != "r50":
        # Conditioning of THIS configuration at random-init weights (scripts/dev/mf_variant_grad_sensitivity.py): the fp32 oracle again with
        # nothing but the weights rounded to bf16 - same forced attention masks, matches and draws.  fai-mf-l-coco-ins moves by 0.8 % in
        # the mask logits and 4-6 % in the gradients (quartiles), fai-mf-m-ade by 1.8 % and 10-20 % (worst tensor 0.5: the first STDC
        # stage).  The engine additionally stores activations in bf16; its gates below are relative to this measured floor.
        from oracle.mask_criterion_oracle import RandStream

        sdb = {k: ((v.detach().bfloat16().float() if v.dim() >= 2 else v.detach().clone()).requires_grad_(v.requires_grad)) if v.dtype == torch.float32
               else v.clone() for k, v in sdg.items()}
        outs_w = T.mf_train_outputs(sdb, cfg, x, forced_attn=col["attn_masks"])
        lw, _ = T.bf_criterion(outs_w, labels, masks, RandStream(rs.rec), cfg, fixed_matches=matches)
        sum(lw.values()).backward()
        ew = sorted((rel_l2(sdb[k].grad, sdg[k].grad) for k in sdg if isinstance(sdg[k], torch.Tensor) and sdg[k].requires_grad), reverse=True)
        sens = {"pm": rel_l2(outs_w["pred_masks"].detach(), outs["pred_masks"].detach()), "worst": ew[0], "median": ew[len(ew) // 2],
                "q1": ew[len(ew) // 4]}
        print("bf16-weights-only oracle vs fp32 oracle:", {k: round(v, 4) for k, v in sens.items()})
    model = FAIMaskFormerTrainable(cfg, norm=norm, rand=_Replay(rs.rec)).to(DEV)
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
    pm_err = rel_l2(model.last_outputs["pred_masks"].detach().float().cpu(), outs["pred_masks"].detach())
    print(f"{norm}: last-head mask logits rel-L2 {pm_err:.4f}")
    worst_loss = max(abs(float(losses[k]) - float(losses_o[k])) / (abs(float(losses_o[k])) + 1e-3) for k in losses_o)
    print(f"{norm}: worst relative loss deviation {worst_loss:.4f}")
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
    assert len(errs) > (250 if variant == "r50" else 100)
    for k in losses_o:
        a, b = float(losses[k]), float(losses_o[k])
        assert abs(a - b) <= (6e-2 if norm == "BN" else 3e-2) * abs(b) + 1e-3, (k, a, b)
    if norm == "BN":
        # live BatchNorm through ~60 normalised layers: same regime as the RT-DETR BN-mode test (tests/test_gpu_train_detr.py)
        dec = sorted(e for e, n in errs if n.startswith("head.predictor."))
        assert dec[len(dec) // 2] <= 0.15, dec[len(dec) // 2]
        assert errs[len(errs) // 2][0] <= 0.40 and errs[len(errs) // 10][0] <= 0.60, errs[:8]
    elif sens is None:
        assert pm_err <= 4e-2
        assert errs[0][0] <= 0.25, errs[:8]
        assert errs[len(errs) // 2][0] <= 0.08
    else:
        assert pm_err <= max(4e-2, 2.5 * sens["pm"]), (pm_err, sens)
        assert errs[len(errs) // 2][0] <= max(0.08, 2.0 * sens["median"]), (errs[len(errs) // 2], sens)
        assert errs[len(errs) // 4][0] <= max(0.15, 2.0 * sens["q1"]), (errs[len(errs) // 4], sens)
        dec = sorted(e for e, n in errs if n.startswith("head.predictor."))
        assert dec[len(dec) // 2] <= 0.10, dec[len(dec) // 2]


def test_mf_train_step_free_running_r101():
    """TrainStep on the full fai-mf-l graph (R101), nothing teacher-forced: finite losses, every trainable tensor receives a gradient and moves."""
    from focoos_amd.train_detr import TrainStep
    from focoos_amd.train_mf import FAIMaskFormerTrainable

    cfg = _cfg(101, 1024)
    model = FAIMaskFormerTrainable(cfg, norm="FrozenBN").to(DEV)
    model.load_state_dict(synth_state_dict(cfg, 42, family="fai_mf"), strict=True)
    model.train()
    ts = TrainStep(model, lr=1e-4, max_grad_norm=0.1)
    imgs = [synth_image_structured(190 + i, 128, 160) for i in range(2)]
    labels, masks = T.synth_mask_targets(8, 2, int(cfg["num_classes"]), (128, 160), counts=(3, 0))
    targets = [MaskFormerTargets(labels=l.to(DEV), masks=m.to(DEV)) for l, m in zip(labels, masks)]
    x_u8 = torch.from_numpy(np.stack(imgs)).to(DEV)
    p0 = ts.opt.flat_p.clone()
    l1 = {k: float(v) for k, v in ts.step(x_u8, targets).items()}
    l2 = {k: float(v) for k, v in ts.step(x_u8, targets).items()}
    torch.cuda.synchronize()
    assert len(l1) == 30 and all(np.isfinite(v) for v in l1.values()) and all(np.isfinite(v) for v in l2.values())
    assert not torch.equal(ts.opt.flat_p, p0)
    dead = [n for n, _ in ts.named if float(ts.opt.grads[n].abs().max()) == 0.0]
    assert not dead, dead[:10]
