"""`.train()` behind the public surface (focoos_model.py:221-274) and the trainer-side `eval_postprocess` (fai_detr/processor.py:121-151)
on a real MI355X."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _entries(n, size, nc, seed=0):
    from focoos_amd.ports import Boxes, DatasetEntry, Instances
    from focoos_amd.synth import synth_image_structured

    rs = np.random.RandomState(seed)
    out = []
    for i in range(n):
        t = rs.randint(1, 6)
        cx, cy = rs.uniform(0.25, 0.75, t) * size, rs.uniform(0.25, 0.75, t) * size
        w, h = rs.uniform(0.08, 0.3, t) * size, rs.uniform(0.08, 0.3, t) * size
        xyxy = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], -1).astype(np.float32)
        img = torch.from_numpy(synth_image_structured(i, size, size)).permute(2, 0, 1).contiguous()   # CHW uint8 like the reference's mappers
        out.append(DatasetEntry(image=img, height=size, width=size,
                                instances=Instances((size, size), boxes=Boxes(torch.from_numpy(xyxy)), classes=torch.from_numpy(rs.randint(0, nc, t)))))
    return out


def test_focoos_model_train_runs_steps_and_reloads_weights(tmp_path):
    from focoos_amd.model import ModelManager
    from focoos_amd.ports import TrainerArgs

    fm = ModelManager.get("fai-detr-l-coco", seed=1)
    nc = fm.model.num_classes
    data = _entries(8, 256, nc)
    before = {k: v.clone() for k, v in fm.model.state_dict().items()}
    args = TrainerArgs(run_name="t_run", output_dir=str(tmp_path), num_gpus=1, max_iters=3, batch_size=4, learning_rate=1e-4, freeze_bn=True,
                       scheduler="FIXED", log_period=1, seed=3)
    out = fm.train(args, data, data)
    assert out is fm and not fm.model.training
    folder = os.path.join(str(tmp_path), "t_run")
    assert os.path.exists(os.path.join(folder, "model_final.pth")) and os.path.exists(os.path.join(folder, "model_info.json"))
    ck = torch.load(os.path.join(folder, "model_final.pth"), map_location="cpu", weights_only=True)
    assert list(ck["model"]) == list(before)                       # the reference's state-dict keys, in order
    after = fm.model.state_dict()
    changed = sum(not torch.equal(before[k], after[k]) for k in before if before[k].is_floating_point() and "norm" not in k)
    assert changed > 100                                           # three AdamW steps moved the trainable weights, and they were reloaded
    k = "head.predictor.dec_score_classifier.5.weight"
    assert torch.allclose(ck["model"][k], after[k])
    # the engine now infers with the trained weights
    dets = fm.infer_batch([np.asarray(e.image.permute(1, 2, 0)) for e in data[:2]], threshold=0.01)
    assert len(dets) == 2


def test_training_preprocess_targets_are_normalised_cxcywh():
    from focoos_amd.processor import DETRProcessor

    p = DETRProcessor({"top_k": 300, "threshold": 0.5}, image_size=128).train(True)
    ents = _entries(3, 128, 80, seed=5)
    images, targets = p.preprocess(ents, device=torch.device(DEV))
    assert images.shape == (3, 128, 128, 3) and images.dtype == torch.uint8 and len(targets) == 3
    for e, t in zip(ents, targets):
        b = e.instances.boxes.tensor / 128.0
        ref = torch.stack([(b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2, b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], -1)
        assert torch.allclose(t.boxes.cpu(), ref, atol=1e-6) and torch.equal(t.labels.cpu(), e.instances.classes)


def test_eval_postprocess_matches_reference_arithmetic():
    """fai_detr/processor.py:121-151 restated with torch ops (topk over Q*K, label = idx % K, query = idx // K, scale, clip, nonempty)."""
    from focoos_amd.ports import DETRModelOutput
    from focoos_amd.processor import DETRProcessor

    g = torch.Generator().manual_seed(0)
    B, Q, K = 3, 300, 80
    probs = torch.rand(B, Q, K, generator=g)
    cxcywh = torch.rand(B, Q, 4, generator=g) * torch.tensor([1.0, 1.0, 0.4, 0.4])
    boxes = torch.cat([cxcywh[..., :2] - cxcywh[..., 2:] / 2, cxcywh[..., :2] + cxcywh[..., 2:] / 2], -1)
    boxes[0, :5, 2] = boxes[0, :5, 0]          # some empty boxes
    probs[0, :5] += 1.0                        # ... that would otherwise be selected
    sizes = [(480, 640), (333, 500), (1, 1)]
    ents = [{"height": h, "width": w} for h, w in sizes]
    res = DETRProcessor({"top_k": 100, "threshold": 0.5}, 640).eval_postprocess(DETRModelOutput(logits=probs.to(DEV), boxes=boxes.to(DEV), loss=None), ents)
    for i, (h, w) in enumerate(sizes):
        sc, idx = torch.topk(probs[i].flatten(), 100)
        lab, q = idx % K, idx // K
        bx = boxes[i][q] * torch.tensor([w, h, w, h], dtype=torch.float32)
        bx = torch.stack([bx[:, 0].clamp(0, w), bx[:, 1].clamp(0, h), bx[:, 2].clamp(0, w), bx[:, 3].clamp(0, h)], -1)
        keep = ((bx[:, 2] - bx[:, 0]) > 0) & ((bx[:, 3] - bx[:, 1]) > 0)
        inst = res[i]["instances"]
        assert inst.image_size == (h, w) and len(inst) == int(keep.sum())
        assert torch.equal(inst.classes.cpu(), lab[keep]) and torch.equal(inst.scores.cpu(), sc[keep])
        assert torch.allclose(inst.boxes.tensor.cpu(), bx[keep], atol=1e-4)


def _mask_entries(n, size, nc, seed=0):
    """DatasetEntry list for the mask families: CHW uint8 image + Instances(classes, masks = [T, H, W] bool rectangles)."""
    from focoos_amd.ports import DatasetEntry, Instances
    from focoos_amd.synth import synth_image_structured

    rs = np.random.RandomState(seed)
    out = []
    for i in range(n):
        t = rs.randint(1, 5)
        m = np.zeros((t, size, size), bool)
        for j in range(t):
            y0, x0 = rs.randint(0, size // 2), rs.randint(0, size // 2)
            m[j, y0:y0 + rs.randint(16, size // 2), x0:x0 + rs.randint(16, size // 2)] = True
        img = torch.from_numpy(synth_image_structured(50 + i, size, size)).permute(2, 0, 1).contiguous()
        out.append(DatasetEntry(image=img, height=size, width=size,
                                instances=Instances((size, size), classes=torch.from_numpy(rs.randint(0, nc, t)), masks=torch.from_numpy(m))))
    return out


def test_bisenetformer_model_train_runs_steps_and_reloads_weights(tmp_path):
    """BASELINE config 5 through the public surface: ModelManager.get("bisenetformer-l-ade").train(args, data): the mask-family
    branch of the processor's training preprocess, train_bf.BisenetFormerTrainable under TrainStep (live BatchNorm: freeze_bn=False),
    artifacts with the reference's key order, weights reloaded into the inference engine."""
    import json

    from focoos_amd.model import ModelManager
    from focoos_amd.ports import TrainerArgs

    fm = ModelManager.get("bisenetformer-l-ade", seed=4, criterion_num_points=1024)
    nc = fm.model.num_classes
    data = _mask_entries(8, 256, nc)
    before = {k: v.clone() for k, v in fm.model.state_dict().items()}
    args = TrainerArgs(run_name="bf_run", output_dir=str(tmp_path), num_gpus=1, max_iters=3, batch_size=4, learning_rate=1e-4, freeze_bn=False,
                       scheduler="FIXED", log_period=1, seed=3)
    out = fm.train(args, data, data)
    assert out is fm and not fm.model.training
    folder = os.path.join(str(tmp_path), "bf_run")
    ck = torch.load(os.path.join(folder, "model_final.pth"), map_location="cpu", weights_only=True)
    assert list(ck["model"]) == list(before)
    after = fm.model.state_dict()
    changed = sum(not torch.equal(before[k], after[k]) for k in before if before[k].is_floating_point())
    assert changed > 300                                           # weights, BatchNorm affine and running statistics moved
    assert int(after["pixel_decoder.backbone.features.0.bn.num_batches_tracked"]) == int(before["pixel_decoder.backbone.features.0.bn.num_batches_tracked"]) + 3
    info = json.load(open(os.path.join(folder, "model_info.json")))
    assert len(info["final_losses"]) == 21 and all(np.isfinite(v) for v in info["final_losses"].values())
    dets = fm.infer_batch([np.asarray(e.image.permute(1, 2, 0)) for e in data[:2]], threshold=0.01)
    assert len(dets) == 2


class _StubReferenceModule:
    """What integration.share_parameters needs from the reference module - named_parameters() / named_buffers() with the reference's
    names - holding its OWN tensors, like the real FAIDetr / BisenetFormer nn.Module that an external trainer (TrainerLoop, DDP, EMA)
    owns.  The reference package does not exist on the GPU box; the name-for-name match against the REAL modules is asserted on CPU
    (tests/test_integration_reference.py)."""

    def __init__(self, cfg, family, seed, frozen_bn):
        from focoos_amd.state_spec import state_spec
        from focoos_amd.synth import synth_state_dict

        sd = synth_state_dict(cfg, seed, family=family)
        self.params, self.buffers = {}, {}
        for k, (_, kind) in state_spec(cfg, family).items():
            if kind in ("bn_mean", "bn_var", "bn_nbt", "buf"):
                self.buffers[k] = sd[k].to(DEV)
            else:
                self.params[k] = torch.nn.Parameter(sd[k].to(DEV), requires_grad=not (frozen_bn and kind in ("bn_w", "bn_b")) and not (family == "fai_detr" and "mask_features" in k))

    def named_parameters(self):
        return self.params.items()

    def named_buffers(self):
        return self.buffers.items()


@pytest.mark.parametrize("family", ["fai_detr", "bisenetformer", "fai_mf"])
def test_adapter_training_mechanics_with_external_optimizer(family):
    """Seam B2 in training mode on the GPU: the HIP autograd graph runs over tensors OWNED by another module (share_parameters), gradients
    land in that module's ``.grad`` fields, and an external torch optimizer - the reference's TrainerLoop keeps its own - steps them;
    the engine notices the change (WEIGHTS_EPOCH) and the next forward uses the new weights."""
    from focoos_amd import train_nn
    from focoos_amd.integration import share_parameters
    from focoos_amd.ports import DETRTargets, MaskFormerTargets
    from focoos_amd.registry import ModelRegistry
    from focoos_amd.synth import synth_image_structured

    if family == "fai_detr":
        from focoos_amd.train_detr import FAIDetrTrainable as Net

        cfg, norm = ModelRegistry.get_model_info("fai-detr-l-coco")["config"], "FrozenBN"
    elif family == "bisenetformer":
        from focoos_amd.train_bf import BisenetFormerTrainable as Net

        cfg, norm = dict(ModelRegistry.get_model_info("bisenetformer-l-ade")["config"], criterion_num_points=1024), "BN"
    else:
        from focoos_amd.train_mf import FAIMaskFormerTrainable as Net

        cfg, norm = dict(ModelRegistry.get_model_info("fai-mf-l-coco-ins")["config"], criterion_num_points=1024), "FrozenBN"
    ref = _StubReferenceModule(cfg, family, 8, norm == "FrozenBN")
    net = Net(cfg, norm=norm).to(DEV)
    n = share_parameters(net, ref)
    assert n == len(ref.params) + len(ref.buffers)
    assert all(p is ref.params[k] for k, p in net.named_parameters())
    net.train()
    opt = torch.optim.AdamW([p for p in ref.params.values() if p.requires_grad], lr=1e-4)
    rs = np.random.RandomState(0)
    x = torch.from_numpy(np.stack([synth_image_structured(70 + i, 128, 160) for i in range(4)])).to(DEV)
    if family == "fai_detr":
        tg = [DETRTargets(labels=torch.from_numpy(rs.randint(0, 80, (3,))).to(DEV),
                          boxes=torch.from_numpy(np.concatenate([rs.uniform(0.3, 0.7, (3, 2)), rs.uniform(0.1, 0.3, (3, 2))], -1).astype(np.float32)).to(DEV)) for _ in range(4)]
    else:
        tg = []
        for _ in range(4):
            m = np.zeros((2, 128, 160), bool)
            m[0, 10:70, 20:90] = True
            m[1, 60:120, 80:150] = True
            tg.append(MaskFormerTargets(labels=torch.from_numpy(rs.randint(0, int(cfg["num_classes"]), (2,))).to(DEV), masks=torch.from_numpy(m).to(DEV)))
    before = {k: p.detach().clone() for k, p in ref.params.items() if p.requires_grad}
    totals = []
    for _ in range(3):
        train_nn.WEIGHTS_EPOCH[0] += 1   # what the adapters do before every training forward
        losses = net(x, tg)
        total = sum(losses.values())
        opt.zero_grad(set_to_none=True)
        total.backward()
        missing = [k for k, p in ref.params.items() if p.requires_grad and p.grad is None]
        assert not missing, missing[:5]
        opt.step()
        totals.append(float(total))
    torch.cuda.synchronize()
    assert all(np.isfinite(t) for t in totals)
    moved = sum(not torch.equal(before[k], ref.params[k].detach()) for k in before)
    assert moved == len(before)
    assert totals[-1] != totals[0]     # the forward sees the stepped weights (packed bf16 images rebuilt)


class _TreeModule(torch.nn.Module):
    """A plain, picklable nn.Module tree carrying a state dict under the reference's dotted key names (stand-in for the reference's FAIDetr /
    FAIMaskFormer / BisenetFormer module on a box without the reference)."""

    @staticmethod
    def build(cls, sd, spec):
        root = cls()
        for k, v in sd.items():
            mod, parts = root, k.split(".")
            for p in parts[:-1]:
                if p not in mod._modules:
                    mod.add_module(p, torch.nn.Module())
                mod = mod._modules[p]
            if spec[k][1] in ("bn_mean", "bn_var", "bn_nbt", "buf"):
                mod.register_buffer(parts[-1], v.clone())
            else:
                mod.register_parameter(parts[-1], torch.nn.Parameter(v.clone(), requires_grad=v.is_floating_point()))
        return root


from focoos_amd.integration import _FxAdapterState  # noqa: E402


class _AdapterStandIn(_FxAdapterState, _TreeModule):
    """The adapters' own state handling (integration._FxAdapterState) on a module that is NOT the reference's."""


@pytest.mark.parametrize("family, model_name", [("fai_detr", "fai-detr-l-coco"), ("fai_mf", "fai-mf-l-coco-ins"), ("bisenetformer", "bisenetformer-l-ade")])
def test_adapter_state_survives_deepcopy_and_pickle_after_real_forwards(family, model_name):
    """VERDICT r4 weak #3 on the GPU with the REAL engine objects: after an inference forward (engine with a loaded CDLL, captured hipGraphs)
    and a training forward (HIP autograd graph sharing the module's parameters) the module deep-copies and pickles; the copy carries the same
    weights, no engine, and its lazily re-built engine reproduces the original's outputs bit for bit."""
    import copy
    import pickle

    from focoos_amd.integration import share_parameters
    from focoos_amd.registry import ModelRegistry
    from focoos_amd.state_spec import state_spec
    from focoos_amd.synth import synth_image_structured, synth_state_dict

    cfg = ModelRegistry.get_model_info(model_name)["config"]
    sd = {k: v.to(DEV) for k, v in synth_state_dict(cfg, 5, family=family).items()}
    mod = _TreeModule.build(_AdapterStandIn, sd, state_spec(cfg, family)).to(DEV)
    if family == "fai_detr":
        from focoos_amd.engine import DetrEngine as Eng
        from focoos_amd.train_detr import FAIDetrTrainable as Net
        size, kw = (640, 640), {}
    elif family == "fai_mf":
        from focoos_amd.engine_mf import MfEngine as Eng
        from focoos_amd.train_mf import FAIMaskFormerTrainable as Net
        size, kw = (128, 160), {"full_masks": True}
    else:
        from focoos_amd.engine_bf import BfEngine as Eng
        from focoos_amd.train_bf import BisenetFormerTrainable as Net
        size, kw = (128, 160), {"full_masks": True}
    x = torch.from_numpy(np.stack([synth_image_structured(30 + i, *size) for i in range(2)])).to(DEV)

    def run(m):
        if m._fx_engine is None:   # what _fx_sync does
            m._fx_engine = Eng(cfg, m.state_dict(), DEV, **kw)
        pl = m._fx_engine.forward(x, **kw)
        return pl.probs.clone(), (pl.boxes if family == "fai_detr" else pl.masks).clone()

    mod._fx_engine, mod._fx_version = None, None
    p0, o0 = run(mod)
    mod._fx_version = ("cuda:0", 1)
    net = Net(cfg, norm="FrozenBN").to(DEV)        # what _fx_train_graph does
    share_parameters(net, mod)
    mod.__dict__["_fx_train"] = (net, DEV)
    with torch.no_grad():
        net.train()
        net.forward_outputs(x)
    torch.cuda.synchronize()
    with pytest.raises(Exception):
        copy.deepcopy(mod._fx_engine)             # the engine itself cannot travel (ctypes handles) - which is why the adapter drops it

    for clone in (copy.deepcopy(mod), pickle.loads(pickle.dumps(mod))):
        assert clone._fx_engine is None and clone._fx_version is None and "_fx_train" not in clone.__dict__
        a, b = mod.state_dict(), clone.state_dict()
        assert list(a) == list(b) and all(torch.equal(a[k], b[k]) and a[k].data_ptr() != b[k].data_ptr() for k in a)
        p1, o1 = run(clone)
        assert torch.equal(p0, p1) and torch.equal(o0, o1)
    assert mod._fx_engine is not None and "_fx_train" in mod.__dict__
