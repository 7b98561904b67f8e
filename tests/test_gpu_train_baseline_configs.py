"""Oracle parity of the TRAINING step at the BASELINE.json shapes themselves, run exactly as `bench.py --train` runs them (VERDICT r4
"weak #1" / "next #1"): BASELINE configs[3] = fai-detr-l-obj365, 16 images of 640x640 per GPU, FrozenBN, bench.py's own weights (seed 0),
images (`synth_image(i)`) and targets (`bench.synth_train_targets`, step 0 of rank 0); configs[4]'s model and shape = bisenetformer-l-ade,
8 images of 1024x1024, the registry's 12544 sample points.  The step goes through `TrainStep.forward_backward` = the production
`TrainStep.step` minus all-reduce / optimizer: flat gradient views (DIRECT_GRAD), one-launch weight re-packing, production kernel routing
(a layer's kernel is chosen by M = B*H*W, so the small-batch tests never exercise `conv_wgrad_dma`, the k-plane input-gradient kernels or
the wide pointwise forms), weight gradients on the side stream, `FX_ENC_SELECT_ROWS=1`.

What is compared (as in tests/test_gpu_train_detr.py / test_gpu_train_bf.py, same gates): the CPU fp32 training oracle
(oracle/train_oracle.py, pinned live against the real reference in train mode) free-running; its discrete choices - encoder top-k /
attention bitmaps, Hungarian matches, torch.rand draws - teacher-forced on the engine; every loss within 3 %; gradient relative L2 of
every trainable tensor (worst <= 0.25, median <= 0.08).  Each test prints the census of convolution kernels the step routed to
(train_nn.VARIANT_CENSUS: forward / input-gradient launches by `fx_conv2d_variant`, weight gradients by `fx_conv2d_wgrad_variant`).

Reference: FAIDetr.forward (train) fai_detr/modelling.py:1344-1358, SetCriterion.forward :553-612; BisenetFormer.forward
bisenetformer/modelling.py:594-609; TrainerLoop.run_step trainer/trainer.py:723-773.
"""
import os
import sys
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from focoos_amd.registry import ModelRegistry  # noqa: E402
from focoos_amd.synth import synth_image, synth_state_dict  # noqa: E402
from oracle import detr_oracle as O  # noqa: E402
from oracle import train_oracle as T  # noqa: E402
from tests.helpers import rel_l2  # noqa: E402

DEV = "cuda:0"


def _grad_table(stepper, sdg, tag):
    errs = []
    for name, _ in stepper.named:
        r = sdg[name]
        if not (isinstance(r, torch.Tensor) and r.requires_grad and r.grad is not None):
            continue
        errs.append((rel_l2(stepper.opt.grads[name].float().cpu(), r.grad), name, float(r.grad.norm())))
    floor = 1e-3 * sorted(n for _, _, n in errs)[len(errs) // 2]
    skipped = [n for _, n, g in errs if g < floor]
    errs = sorted(((e, n) for e, n, g in errs if g >= floor), reverse=True)
    print(f"{tag}: {len(errs)} parameter tensors ({len(skipped)} with a mathematically zero gradient skipped); worst 6: "
          f"{[(round(e, 4), n) for e, n in errs[:6]]}; quartiles {[round(errs[len(errs) * q // 4][0], 4) for q in (1, 2, 3)]}")
    return errs


def _fixed(matches):
    out = []
    for m in matches:
        pi = torch.tensor(np.concatenate([np.asarray(i) for i, _ in m]), dtype=torch.int32, device=DEV)
        ti = torch.tensor(np.concatenate([np.asarray(j) for _, j in m]), dtype=torch.int32, device=DEV)
        out.append((pi, ti))
    return out


def test_config3_detr_train_step_bs16_640_production_path_vs_oracle():
    """BASELINE configs[3] per GPU: fai-detr-l-obj365, bs=16, 640x640, norm=FrozenBN - the step `bench.py --train` times."""
    import bench
    from focoos_amd import train_nn
    from focoos_amd.train_detr import FAIDetrTrainable, TrainStep

    assert os.environ.get("FX_ENC_SELECT_ROWS", "1") == "1" and os.environ.get("FX_WGRAD_STREAM", "1") == "1"
    cfg = ModelRegistry.get_model_info("fai-detr-l-obj365")["config"]
    K, B, S = int(cfg["num_classes"]), 16, 640
    sd = synth_state_dict(cfg, 0)
    imgs = [synth_image(i, S, S) for i in range(B)]
    targets = bench.synth_train_targets("fai_detr", 0, 0, B, S, K, DEV)
    labels, boxes = [t.labels.cpu() for t in targets], [t.boxes.cpu() for t in targets]

    # ---- oracle: fp32, free-running (its encoder top-k and Hungarian matches are then forced on the engine)
    def trainable(k, v):
        if not (v.dtype == torch.float32 and v.dim() > 0) or any(t in k for t in ("running_", "empty_weight", "mask_features")):
            return False
        return not (k.endswith((".norm.weight", ".norm.bias")) or (".input_proj." in k and k.split(".")[-2] == "1"))   # FrozenBN affine

    sdg = {k: (v.clone().requires_grad_(True) if trainable(k, v) else v.clone()) for k, v in sd.items()}
    t0 = time.time()
    outs = T.detr_train_outputs(sdg, cfg, O.get_torch_batch(imgs, None))
    losses_o, matches = T.criterion(outs, labels, boxes)
    sum(losses_o.values()).backward()
    print(f"oracle forward + criterion + backward: {time.time() - t0:.1f} s on {torch.get_num_threads()} threads")
    # this configuration's own conditioning (as in tests/test_gpu_detr_variants.py): the fp32 oracle again with NOTHING but the weights rounded
    # to bf16 (same top-k, same matches).  bench.py's seed-0 weights put the box heads in an ill-conditioned spot - their gradients come from
    # the ~170 matched rows per prediction set only and move by 14-41 % under weight rounding alone (median over all tensors: 4 %) - so each
    # tensor is gated against 3 x ITS sensitivity as well as the absolute 0.25
    sdb = {k: ((v.detach().bfloat16().float() if v.dim() >= 2 else v.detach().clone()).requires_grad_(v.requires_grad)) if v.dtype == torch.float32
           else v.clone() for k, v in sdg.items()}
    outs_w = T.detr_train_outputs(sdb, cfg, O.get_torch_batch(imgs, None), forced_topk=outs["topk_ind"])
    lw, _ = T.criterion(outs_w, labels, boxes, fixed_matches=matches)
    sum(lw.values()).backward()
    sens = {k: rel_l2(sdb[k].grad, sdg[k].grad) for k in sdg if isinstance(sdg[k], torch.Tensor) and sdg[k].requires_grad and sdb[k].grad is not None}
    del sdb, outs_w, lw

    # ---- the production step
    model = FAIDetrTrainable(cfg, norm="FrozenBN").to(DEV)
    model.load_state_dict(sd, strict=True)
    model.train()
    stepper = TrainStep(model)
    assert stepper.wgrad_stream is not None, "weight gradients must run on the side stream as in the timed step"
    x = torch.from_numpy(np.stack(imgs)).to(DEV)
    forced = dict(forced_topk=outs["topk_ind"].to(DEV), fixed_matches=_fixed(matches))
    stepper.forward_backward(x, targets, **forced)         # first pass: lazy packing, first-use kernel attributes
    train_nn.VARIANT_CENSUS[0] = {}
    try:
        losses = stepper.forward_backward(x, targets, **forced)
    finally:
        census, train_nn.VARIANT_CENSUS[0] = train_nn.VARIANT_CENSUS[0], None
    torch.cuda.synchronize()
    print("conv kernels of the step (launches):", dict(sorted(census.items(), key=lambda kv: -kv[1])))
    assert any(k.startswith("wgrad:conv_wgrad_dma") for k in census), census
    assert any("conv3x3_kplane<256>" in k for k in census) and any("pw_kplane" in k for k in census), census

    assert sorted(losses) == sorted(losses_o) and len(losses) == 21
    worst = 0.0
    for k in losses_o:
        a, b = float(losses[k]), float(losses_o[k])
        worst = max(worst, abs(a - b) / max(abs(b), 1e-6))
        assert abs(a - b) <= 3e-2 * abs(b) + 1e-3, (k, a, b)
    print(f"21 losses: worst relative deviation {worst:.4f}; total {float(sum(losses.values())):.4f} vs oracle {float(sum(losses_o.values())):.4f}")
    errs = _grad_table(stepper, sdg, "configs[3] fai-detr-l-obj365 bs=16 640^2 FrozenBN")
    assert len(errs) > 250
    sw = sorted(sens.values(), reverse=True)
    print(f"bf16-weights-only sensitivity of the fp32 oracle's gradients: worst {sw[0]:.4f}, median {sw[len(sw) // 2]:.4f}; of the engine's worst 6 tensors: "
          f"{[round(sens.get(n, 0.0), 4) for _, n in errs[:6]]}")
    bad = [(round(e, 4), round(sens.get(n, 0.0), 4), n) for e, n in errs if e > max(0.25, 3.0 * sens.get(n, 0.0))]
    assert not bad, bad[:8]
    assert errs[len(errs) // 2][0] <= 0.08
    # outside the box heads (whose conditioning the sensitivity pass measures) the absolute gate of the small-batch test holds as it stands
    rest = [(e, n) for e, n in errs if "dec_bbox_classifier" not in n and "enc_bbox" not in n]
    assert rest[0][0] <= 0.25, rest[:8]


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


def test_config4_bisenetformer_train_step_bs8_1024_production_path_vs_oracle():
    """BASELINE configs[4]'s model and shape per GPU: bisenetformer-l-ade, bs=8, 1024x1024, 12544 sample points, bench.py's weights / images /
    targets.  BatchNorm frozen: under batch statistics a random-init STDC-2 (12 blocks, no residual connections) amplifies ANY 0.4 % rounding to
    61 % at res5, so a live-BN comparison against fp32 at this depth measures the network, not the kernels; the live-BN plumbing is
    compared on the 3-block STDC in tests/test_gpu_train_bf.py."""
    import bench
    from focoos_amd import train_nn
    from focoos_amd.train_bf import BisenetFormerTrainable
    from focoos_amd.train_detr import TrainStep

    cfg = ModelRegistry.get_model_info("bisenetformer-l-ade")["config"]
    K, B, S = int(cfg["num_classes"]), 8, 1024
    assert int(cfg.get("criterion_num_points", 12544)) == 12544
    sd = synth_state_dict(cfg, 0, family="bisenetformer")
    imgs = [synth_image(i, S, S) for i in range(B)]
    targets = bench.synth_train_targets("bisenetformer", 0, 0, B, S, K, DEV)
    labels, masks = [t.labels.cpu() for t in targets], [t.masks.cpu() for t in targets]

    def trainable(k, v):
        if not (v.dtype == torch.float32 and v.dim() > 0) or any(t in k for t in ("running_", "empty_weight")):
            return False
        return not k.endswith((".bn.weight", ".bn.bias", ".bn_atten.weight", ".bn_atten.bias", ".avd_layer.1.weight", ".avd_layer.1.bias"))

    sdg = {k: (v.clone().requires_grad_(True) if trainable(k, v) else v.clone()) for k, v in sd.items()}
    col = {}
    t0 = time.time()
    outs = T.bf_train_outputs(sdg, cfg, O.get_torch_batch(imgs, None), collect=col)
    rs = _DrawAndRecord(77)
    losses_o, matches = T.bf_criterion(outs, labels, masks, rs, cfg)
    sum(losses_o.values()).backward()
    print(f"oracle forward + criterion + backward: {time.time() - t0:.1f} s on {torch.get_num_threads()} threads")

    model = BisenetFormerTrainable(cfg, norm="FrozenBN", rand=_Replay(rs.rec)).to(DEV)
    model.load_state_dict(sd, strict=True)
    model.train()
    stepper = TrainStep(model)
    x = torch.from_numpy(np.stack(imgs)).to(DEV)
    forced = dict(forced_attn=col["attn_masks"], fixed_matches=_fixed(matches))
    stepper.forward_backward(x, targets, **forced)
    train_nn.VARIANT_CENSUS[0] = {}
    try:
        losses = stepper.forward_backward(x, targets, **forced)
    finally:
        census, train_nn.VARIANT_CENSUS[0] = train_nn.VARIANT_CENSUS[0], None
    torch.cuda.synchronize()
    print("conv kernels of the step (launches):", dict(sorted(census.items(), key=lambda kv: -kv[1])))
    assert sorted(losses) == sorted(losses_o) and len(losses) == 21
    worst = 0.0
    for k in losses_o:
        a, b = float(losses[k]), float(losses_o[k])
        worst = max(worst, abs(a - b) / max(abs(b), 1e-6))
        assert abs(a - b) <= 3e-2 * abs(b) + 1e-3, (k, a, b)
    pm_err = rel_l2(model.last_outputs["pred_masks"].detach().float().cpu(), outs["pred_masks"].detach())
    print(f"21 losses: worst relative deviation {worst:.4f}; last-head mask logits rel-L2 {pm_err:.4f}")
    assert pm_err <= 4e-2
    errs = _grad_table(stepper, sdg, "configs[4] bisenetformer-l-ade bs=8 1024^2 FrozenBN")
    assert len(errs) > 180
    assert errs[0][0] <= 0.25, errs[:8]
    assert errs[len(errs) // 2][0] <= 0.08
