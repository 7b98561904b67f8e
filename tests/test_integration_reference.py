"""Drop-in registration against the real reference (build container only)."""
import pytest

from oracle import ref_import

pytestmark = pytest.mark.skipif(not ref_import.reference_available(), reason="/root/reference not present")


def test_register_replaces_detr_family_and_keeps_state_dict():
    ref_import.install()
    import torch
    from focoos.model_manager import ConfigManager, ModelManager
    from focoos.ports import ModelFamily

    import focoos_amd.integration as fx
    from focoos_amd.registry import ModelRegistry
    from focoos_amd.state_spec import detr_state_spec

    fx.register()
    cls = ModelManager._models_family_map[ModelFamily.DETR.value]()
    assert cls.__name__ == "EngineFAIDetr"
    cfgd = ModelRegistry.get_model_info("fai-detr-l-coco")["config"]
    model = cls(ConfigManager.from_dict(ModelFamily.DETR, dict(cfgd))).eval()
    assert list(model.state_dict()) == list(detr_state_spec(cfgd))   # checkpoint keys unchanged
    n = fx.bind_msda_core(model)
    assert n == 6
    if not torch.cuda.is_available():
        from focoos_amd._lib import FocoosAmdError

        with pytest.raises(FocoosAmdError):  # loud, never a silent CPU fallback
            with torch.no_grad():
                model(torch.zeros(1, 3, 64, 64))


def test_register_replaces_maskformer_family_and_keeps_state_dict():
    ref_import.install()
    from focoos.model_manager import ConfigManager, ModelManager
    from focoos.ports import ModelFamily

    import focoos_amd.integration as fx
    from focoos_amd.registry import ModelRegistry
    from focoos_amd.state_spec import mf_state_spec

    fx.register()
    cls = ModelManager._models_family_map[ModelFamily.MASKFORMER.value]()
    assert cls.__name__ == "EngineFAIMaskFormer"
    cfgd = ModelRegistry.get_model_info("fai-mf-l-coco-ins")["config"]
    model = cls(ConfigManager.from_dict(ModelFamily.MASKFORMER, dict(cfgd))).eval()
    assert list(model.state_dict()) == list(mf_state_spec(cfgd))   # checkpoint keys unchanged


def test_register_replaces_bisenetformer_family_and_keeps_state_dict():
    ref_import.install()
    from focoos.model_manager import ConfigManager, ModelManager
    from focoos.ports import ModelFamily

    import focoos_amd.integration as fx
    from focoos_amd.registry import ModelRegistry
    from focoos_amd.state_spec import bf_state_spec

    fx.register()
    cls = ModelManager._models_family_map[ModelFamily.BISENETFORMER.value]()
    assert cls.__name__ == "EngineBisenetFormer"
    cfgd = ModelRegistry.get_model_info("bisenetformer-l-ade")["config"]
    cfg_ref = {k: v for k, v in cfgd.items() if k != "resolution"}
    model = cls(ConfigManager.from_dict(ModelFamily.BISENETFORMER, cfg_ref)).eval()
    assert list(model.state_dict()) == list(bf_state_spec(cfgd))   # checkpoint keys unchanged


def test_register_installs_engine_processors_and_training_forward_contract():
    """register() also replaces the three processors in the reference's ProcessorManager (device post-process behind model.infer()),
    and the adapter's training forward is wired to the HIP training graph over the module's OWN parameters (share_parameters)."""
    ref_import.install()
    import torch
    from focoos.model_manager import ConfigManager, ModelManager
    from focoos.ports import ModelFamily
    from focoos.processor.processor_manager import ProcessorManager

    import focoos_amd.integration as fx
    from focoos_amd.registry import ModelRegistry

    fx.register()
    cfgd = ModelRegistry.get_model_info("fai-detr-l-coco")["config"]
    cfg = ConfigManager.from_dict(ModelFamily.DETR, dict(cfgd))
    proc = ProcessorManager.get_processor(ModelFamily.DETR, cfg, 640)
    assert type(proc).__name__ == "EngineDETRProcessor" and hasattr(proc, "eval_postprocess")   # the reference's own methods are inherited
    assert type(ProcessorManager.get_processor(ModelFamily.MASKFORMER, ConfigManager.from_dict(
        ModelFamily.MASKFORMER, dict(ModelRegistry.get_model_info("fai-mf-l-coco-ins")["config"])), 1024)).__name__ == "EngineMaskFormerProcessor"
    # CPU tensors go through the reference's post-process (same class hierarchy), identical to the stock processor
    from focoos.models.fai_detr.ports import DETRModelOutput

    g = torch.Generator().manual_seed(0)
    out = DETRModelOutput(logits=torch.rand(1, 300, 80, generator=g), boxes=torch.rand(1, 300, 4, generator=g), loss=None)
    import numpy as np

    dets = proc.postprocess(out, [np.zeros((480, 640, 3), np.uint8)], threshold=0.9)
    assert len(dets) == 1 and all(d.conf > 0.9 for d in dets[0].detections)
    # share_parameters: engine graph parameters become the reference module's Parameter objects (CPU stand-in for the HIP graph)
    cls = ModelManager._models_family_map[ModelFamily.DETR.value]()
    model = cls(cfg)
    import copy

    standin = copy.deepcopy(model.head.predictor.dec_score_classifier)     # any sub-tree with the same names works for the mechanism
    n = fx.share_parameters(standin, model.head.predictor.dec_score_classifier)
    assert n == len(list(standin.parameters())) and n > 0
    for (k, p), (k2, q) in zip(standin.named_parameters(), model.head.predictor.dec_score_classifier.named_parameters()):
        assert k == k2 and p is q
    loss = sum((p.float() ** 2).sum() for p in standin.parameters())
    loss.backward()
    assert all(q.grad is not None for q in model.head.predictor.dec_score_classifier.parameters())   # gradients land in the reference module
    # training forward on CPU is loud (no GPU here): the adapter never silently trains on a CPU path
    if not torch.cuda.is_available():
        from focoos_amd._lib import FocoosAmdError

        model.train()
        with pytest.raises(FocoosAmdError):
            model(torch.zeros(1, 3, 64, 64), [])


def test_bisenetformer_adapter_shares_every_parameter_with_the_training_graph(monkeypatch):
    """The BiSeNetFormer adapter's training forward runs train_bf.BisenetFormerTrainable over the reference module's OWN tensors: every
    parameter and buffer of the HIP graph is matched by name and shape in the REAL reference module (share_parameters raises otherwise).
    Structure only - built here without a GPU by stubbing the library handle; the numerics are tests/test_gpu_train_bf.py."""
    ref_import.install()
    import torch
    from focoos.model_manager import ConfigManager, ModelManager
    from focoos.ports import ModelFamily

    import focoos_amd.integration as fx
    from focoos_amd import _lib
    from focoos_amd.registry import ModelRegistry

    fx.register()
    cfgd = ModelRegistry.get_model_info("bisenetformer-l-ade")["config"]
    ref = ModelManager._models_family_map[ModelFamily.BISENETFORMER.value]()(ConfigManager.from_dict(ModelFamily.BISENETFORMER, {k: v for k, v in cfgd.items() if k != "resolution"}))
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(_lib, "load", lambda: None)
    from focoos_amd.train_bf import BisenetFormerTrainable

    net = BisenetFormerTrainable(cfgd, norm="BN")
    n = fx.share_parameters(net, ref)
    assert n == len(list(ref.state_dict())) == len(list(net.state_dict()))
    refp = dict(ref.named_parameters())
    assert all(p is refp[k] for k, p in net.named_parameters())
    assert net.pixel_decoder.backbone.features[2].avd_layer._norm_h.running_mean is dict(ref.named_buffers())["pixel_decoder.backbone.features.2.avd_layer.1.running_mean"]


def test_maskformer_adapter_shares_every_parameter_with_the_training_graph(monkeypatch):
    """Same structural check for the MaskFormer family: train_mf.FAIMaskFormerTrainable's parameter / buffer tree against the REAL FAIMaskFormer."""
    ref_import.install()
    import torch
    from focoos.model_manager import ConfigManager, ModelManager
    from focoos.ports import ModelFamily

    import focoos_amd.integration as fx
    from focoos_amd import _lib
    from focoos_amd.registry import ModelRegistry

    fx.register()
    cfgd = ModelRegistry.get_model_info("fai-mf-l-coco-ins")["config"]
    ref = ModelManager._models_family_map[ModelFamily.MASKFORMER.value]()(ConfigManager.from_dict(ModelFamily.MASKFORMER, {k: v for k, v in cfgd.items() if k != "resolution"}))
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(_lib, "load", lambda: None)
    from focoos_amd.train_mf import FAIMaskFormerTrainable

    net = FAIMaskFormerTrainable(cfgd, norm="BN")
    n = fx.share_parameters(net, ref)
    assert n == len(list(ref.state_dict())) == len(list(net.state_dict()))
    refp = dict(ref.named_parameters())
    assert all(p is refp[k] for k, p in net.named_parameters())


@pytest.mark.parametrize("family,ptype", [("fai_mf", "instance"), ("bisenetformer", "semantic"), ("bisenetformer", "instance")])
def test_mask_eval_postprocess_equals_reference(family, ptype):
    """Trainer-side eval_postprocess of the mask families (fai_mf/processor.py:99-166, bisenetformer/processor.py:95-157) on the same random
    model output: our processors against the REAL reference processors - instance fields (scores, classes, boxes, masks) and the semantic
    score maps, incl. the crop to the un-padded extent and the resize to the original (height, width)."""
    ref_import.install()
    import torch
    from focoos.model_manager import ConfigManager
    from focoos.ports import DatasetEntry as RefEntry
    from focoos.ports import ModelFamily

    from focoos_amd.ports import BisenetFormerOutput, DatasetEntry, MaskFormerModelOutput
    from focoos_amd.processor import BisenetFormerProcessor, MaskFormerProcessor
    from focoos_amd.registry import ModelRegistry

    name = "fai-mf-l-coco-ins" if family == "fai_mf" else "bisenetformer-l-ade"
    cfgd = dict(ModelRegistry.get_model_info(name)["config"])
    cfgd.update(postprocessing_type=ptype, top_k=20)
    fam = ModelFamily.MASKFORMER if family == "fai_mf" else ModelFamily.BISENETFORMER
    rcfg = ConfigManager.from_dict(fam, {k: v for k, v in cfgd.items() if k != "resolution"})
    if family == "fai_mf":
        from focoos.models.fai_mf.ports import MaskFormerModelOutput as RefOut
        from focoos.models.fai_mf.processor import MaskFormerProcessor as RefProc
        mine, mk = MaskFormerProcessor(cfgd), MaskFormerModelOutput
    else:
        from focoos.models.bisenetformer.ports import BisenetFormerOutput as RefOut
        from focoos.models.bisenetformer.processor import BisenetFormerProcessor as RefProc
        mine, mk = BisenetFormerProcessor(cfgd), BisenetFormerOutput
    ref = RefProc(rcfg)
    g = torch.Generator().manual_seed(5)
    B, Q, K = 2, 12, int(cfgd["num_classes"])
    logits = torch.softmax(torch.randn(B, Q, K + 1, generator=g) * 2, -1)[..., :-1]
    masks = torch.sigmoid(torch.randn(B, Q, 24, 32, generator=g) * 3)          # stride 4 of a padded 96 x 128 batch
    sizes = [((90, 120), (181, 240)), ((96, 128), (96, 128))]                   # (augmented image size, original size)
    ents_r = [RefEntry(image=torch.zeros(3, *a), height=o[0], width=o[1]) for a, o in sizes]
    ents_m = [DatasetEntry(image=torch.zeros(3, *a), height=o[0], width=o[1]) for a, o in sizes]
    want = ref.eval_postprocess(RefOut(masks=masks, logits=logits, loss=None), ents_r)
    got = mine.eval_postprocess(mk(masks=masks, logits=logits, loss=None), ents_m)
    assert len(want) == len(got) == 2
    for w, m_ in zip(want, got):
        assert set(w) == set(m_)
        if ptype == "semantic":
            assert torch.allclose(w["sem_seg"], m_["sem_seg"], atol=1e-6) and w["sem_seg"].shape[0] == K
        else:
            wi, mi = w["instances"], m_["instances"]
            assert wi.image_size == mi.image_size and len(wi) == len(mi) == 20
            assert torch.equal(wi.classes, mi.classes) and torch.allclose(wi.scores, mi.scores, atol=1e-6)
            assert torch.equal(wi.masks.tensor, mi.masks.tensor) and torch.equal(wi.boxes.tensor, mi.boxes.tensor)


@pytest.mark.parametrize("family", ["fai_detr", "fai_mf", "bisenetformer"])
@pytest.mark.parametrize("sizes", [[(64, 96), (64, 96)], [(64, 96), (80, 72), (33, 120)]])
def test_entry_batches_are_padded_like_imagelist(family, sizes):
    """A list of DatasetEntry through `preprocess` in training mode, against the REAL processors on the same entries: images of different
    sizes zero-padded (top-left) to the batch's largest height / width - ImageList.from_tensors, focoos/structures.py:730-803 -, RT-DETR's
    boxes normalised by the PADDED size (fai_detr/processor.py:91-96), the mask families' ground-truth masks padded to it
    (fai_mf/processor.py:76-88), an entry without ground truth included."""
    ref_import.install()
    import numpy as np
    import torch
    from focoos.ports import DatasetEntry as RefEntry
    from focoos.structures import BitMasks as RefBitMasks
    from focoos.structures import Boxes as RefBoxes
    from focoos.structures import Instances as RefInstances

    from focoos_amd.ports import BitMasks, Boxes, DatasetEntry, Instances
    from focoos_amd.processor import BisenetFormerProcessor, DETRProcessor, MaskFormerProcessor
    from focoos_amd.registry import ModelRegistry

    g = torch.Generator().manual_seed(len(sizes))
    ents_r, ents_m = [], []
    for i, (h, w) in enumerate(sizes):
        img = torch.randint(0, 256, (3, h, w), generator=g, dtype=torch.uint8)
        t = 0 if i == 1 else 3
        cls = torch.randint(0, 80, (t,), generator=g)
        xy = torch.rand(t, 2, generator=g) * torch.tensor([w * 0.5, h * 0.5])
        bx = torch.cat([xy, xy + torch.rand(t, 2, generator=g) * torch.tensor([w * 0.5, h * 0.5]) + 1.0], 1)
        ms = torch.rand(t, h, w, generator=g) > 0.6
        if family == "fai_detr":
            ents_r.append(RefEntry(image=img, height=h, width=w, instances=RefInstances((h, w), boxes=RefBoxes(bx.clone()), classes=cls.clone())))
            ents_m.append(DatasetEntry(image=img.clone(), height=h, width=w, instances=Instances((h, w), boxes=Boxes(bx.clone()), classes=cls.clone())))
        else:
            ents_r.append(RefEntry(image=img, height=h, width=w, instances=RefInstances((h, w), masks=RefBitMasks(ms.clone()), classes=cls.clone())))
            ents_m.append(DatasetEntry(image=img.clone(), height=h, width=w, instances=Instances((h, w), masks=BitMasks(ms.clone()), classes=cls.clone())))
    if family == "fai_detr":
        import focoos.models.fai_detr.processor as rp

        _, ref, _ = ref_import.build_reference_detr(dict(ModelRegistry.get_model_info("fai-detr-l-coco")["config"]))
        mine = DETRProcessor(ModelRegistry.get_model_info("fai-detr-l-coco")["config"], 640)
    elif family == "fai_mf":
        cfg = dict(ModelRegistry.get_model_info("fai-mf-l-coco-ins")["config"])
        cfg["backbone_config"] = dict(cfg["backbone_config"], depth=50)
        _, ref, _ = ref_import.build_reference_mf(cfg)
        mine = MaskFormerProcessor(cfg)
    else:
        import json
        import os

        name = "bisenetformer-s-ade"
        _, ref, _ = ref_import.build_reference_bf(json.load(open(os.path.join(ref_import.REFERENCE_ROOT, f"focoos/model_registry/{name}.json")))["config"])
        mine = BisenetFormerProcessor(ModelRegistry.get_model_info(name)["config"])
    cpu = torch.device("cpu")
    xr, tr = ref.train().preprocess(ents_r, device=cpu)
    xm, tm = mine.train().preprocess(ents_m, device=cpu)
    H, W = max(h for h, _ in sizes), max(w for _, w in sizes)
    assert tuple(xr.shape) == (len(sizes), 3, H, W) and tuple(xm.shape) == (len(sizes), H, W, 3)
    assert torch.equal(xm.permute(0, 3, 1, 2).to(xr.dtype), xr)
    assert len(tr) == len(tm) == len(sizes)
    for a, b in zip(tm, tr):
        assert torch.equal(torch.as_tensor(a.labels), b.labels)
        if family == "fai_detr":
            assert a.boxes.shape == b.boxes.shape
            np.testing.assert_allclose(a.boxes.numpy(), b.boxes.numpy(), rtol=0, atol=1e-6)
        else:
            assert tuple(a.masks.shape) == tuple(b.masks.shape) and torch.equal(a.masks.bool(), b.masks.bool())
            assert a.masks.shape[0] == 0 or tuple(a.masks.shape[1:]) == (H, W)
    # evaluation mode: the same padded batch, no targets
    xe, te = mine.eval().preprocess(ents_m, device=cpu)
    assert torch.equal(xe, xm) and te == []


def test_get_image_sizes_equals_reference_for_every_input_kind():
    """Processor.get_image_sizes (base_processor.py:176-221) - the original sizes every post-process scales its boxes / masks to - ours against
    the REAL one for each accepted input kind: ndarray HWC and BHWC (the reference reports ONE size for a 4-D array), tensor CHW and BCHW,
    a PIL image, lists of each and a mixed list; an unsupported type raises ValueError in both."""
    ref_import.install()
    import numpy as np
    import torch
    from PIL import Image

    from focoos_amd.processor import DETRProcessor
    from focoos_amd.registry import ModelRegistry

    cfg = ModelRegistry.get_model_info("fai-detr-l-coco")["config"]
    _, ref, _ = ref_import.build_reference_detr(dict(cfg))
    mine = DETRProcessor(cfg, 640)
    pil = Image.fromarray(np.zeros((30, 50, 3), np.uint8))
    cases = [np.zeros((48, 60, 3), np.uint8), np.zeros((2, 64, 32, 3), np.uint8), torch.zeros(3, 100, 50), torch.zeros(2, 3, 40, 70), pil,
             [np.zeros((48, 60, 3), np.uint8), np.zeros((10, 20, 3), np.uint8)], [torch.zeros(3, 100, 50), torch.zeros(3, 7, 9)], [pil, pil],
             [pil, np.zeros((48, 60, 3), np.uint8), torch.zeros(3, 11, 13)]]
    for c in cases:
        want = [tuple(int(v) for v in s) for s in ref.get_image_sizes(c)]
        got = [tuple(int(v) for v in s) for s in mine.get_image_sizes(c)]
        assert got == want, (type(c), got, want)
    for bad in ("x", [1, 2]):
        with pytest.raises(ValueError):
            ref.get_image_sizes(bad)
        with pytest.raises(ValueError):
            mine.get_image_sizes(bad)


@pytest.mark.parametrize("family, model_name", [("DETR", "fai-detr-l-coco"), ("MASKFORMER", "fai-mf-l-coco-ins"), ("BISENETFORMER", "bisenetformer-l-ade")])
def test_adapter_survives_deepcopy_and_pickle_after_a_forward(family, model_name):
    """SURVEY §8(b) B2's survival contract: the reference deep-copies the model (FocoosModel.export models/focoos_model.py:465,
    EMAState trainer/solver/ema.py:49) and pickles it into spawned ranks (utils/distributed/dist.py:78-91).  After a forward the adapter holds
    an engine (ctypes.CDLL, function pointers, graph handles) and a training graph sharing its parameters: neither may travel - a stand-in
    engine holding a REAL ctypes.CDLL and a function pointer is planted exactly where ``_fx_sync`` / ``_fx_train_graph`` put theirs."""
    ref_import.install()
    import copy
    import ctypes
    import pickle

    import torch
    from focoos.model_manager import ConfigManager, ModelManager
    from focoos.ports import ModelFamily

    import focoos_amd.integration as fx
    from focoos_amd.registry import ModelRegistry

    fx.register()
    fam = getattr(ModelFamily, family)
    cls = ModelManager._models_family_map[fam.value]()
    cfgd = {k: v for k, v in ModelRegistry.get_model_info(model_name)["config"].items() if k != "resolution" or family == "DETR"}
    model = cls(ConfigManager.from_dict(fam, dict(cfgd))).eval()

    class StandInEngine:
        def __init__(self):
            self.lib = ctypes.CDLL(None)
            self.fn = self.lib.strlen
            self.dev = "cpu"

    with pytest.raises(ValueError):
        copy.deepcopy(StandInEngine())           # the stand-in reproduces the defect: a CDLL cannot be deep-copied ...
    with pytest.raises(Exception):
        pickle.dumps(StandInEngine())            # ... or pickled
    model._fx_engine = StandInEngine()
    model._fx_version = ("cpu", 1, 2, 3)
    model.__dict__["_fx_train"] = (StandInEngine(), "cpu")

    for clone in (copy.deepcopy(model), pickle.loads(pickle.dumps(model))):
        assert type(clone) is cls and clone._fx_engine is None and clone._fx_version is None and "_fx_train" not in clone.__dict__
        sd, sc = model.state_dict(), clone.state_dict()
        assert list(sd) == list(sc) and all(torch.equal(sd[k], sc[k]) for k in sd)
        assert all(a.data_ptr() != b.data_ptr() for a, b in zip(model.parameters(), clone.parameters()))   # a copy, not a view
    assert model._fx_engine is not None and "_fx_train" in model.__dict__      # the original keeps its engine
    # the class pickles by reference through the module-level __getattr__ (what a spawned rank resolves)
    assert getattr(fx, cls.__name__) is cls and pickle.loads(pickle.dumps(cls)) is cls


def test_adapters_have_no_stock_graph_fallback():
    """no branch of an adapter's forward may call the reference's own PyTorch graph - with ONE exception (VERDICT r5
    missing #1): the export branch.  ``FocoosModel.export`` traces a deep copy of the model (models/focoos_model.py:464-468) and a tracer
    cannot see through the ctypes engine, so a copy marked by ``switch_to_export`` (or a forward under a running tracer) delegates to the
    stock module graph.  Exactly one ``super().forward`` per family, each behind ``_fx_exporting()``."""
    import inspect
    import re

    import focoos_amd.integration as fx

    src = inspect.getsource(fx)
    code = "\n".join(line.split("#")[0] for line in src.splitlines() if not line.strip().startswith(("#", '"', "``", "(")))
    calls = [m.start() for m in re.finditer(r"super\(\)\.forward", code)]
    assert len(calls) == 3            # EngineFAIDetr.forward + the two mask-family adapters handing theirs to _mask_family_forward
    # the DETR one sits directly under the export test; the mask families' bound methods are only ever called by the export branch
    assert re.search(r"if self\._fx_exporting\(\):\s+return super\(\)\.forward\(images, targets\)", code)
    assert re.search(r"if self\._fx_exporting\(\):\s+return stock_forward\(images, targets\)", code)
    assert code.count("stock_forward(") == 1 and code.count("stock_forward") == 2      # the parameter + its single guarded call


@pytest.mark.parametrize("family, model_name, size", [("DETR", "fai-detr-l-coco", 192), ("MASKFORMER", "fai-mf-l-coco-ins", 64),
                                                      ("BISENETFORMER", "bisenetformer-l-ade", 64)])
def test_export_traces_the_stock_graph_through_the_adapter(family, model_name, size):
    """VERDICT r5 missing #1 / SURVEY 8(b) B2: ``.export()`` through a registered adapter.  The reference's own sequence
    (models/focoos_model.py:40-85,464-468,487): ``ExportableModel(copy.deepcopy(model), device, input_size)`` -> one warm-up call ->
    ``torch.jit.trace``.  The traced module of the ADAPTER must equal the un-adapted reference module of the same weights, on the traced
    input and on a second one (a tracer that had recorded the engine's outputs as constants would fail the second)."""
    ref_import.install()
    import copy
    import importlib

    import torch
    from focoos.model_manager import ConfigManager, ModelManager
    from focoos.models.focoos_model import ExportableModel
    from focoos.ports import ModelFamily

    import focoos_amd.integration as fx
    from focoos_amd.registry import ModelRegistry

    fx.register()
    fam = getattr(ModelFamily, family)
    cls = ModelManager._models_family_map[fam.value]()
    stock = {"DETR": ("focoos.models.fai_detr.modelling", "FAIDetr"), "MASKFORMER": ("focoos.models.fai_mf.modelling", "FAIMaskFormer"),
             "BISENETFORMER": ("focoos.models.bisenetformer.modelling", "BisenetFormer")}[family]
    RefCls = getattr(importlib.import_module(stock[0]), stock[1])
    assert issubclass(cls, RefCls) and cls is not RefCls
    cfgd = {k: v for k, v in ModelRegistry.get_model_info(model_name)["config"].items() if k != "resolution" or family == "DETR"}
    torch.manual_seed(0)
    adapter = cls(ConfigManager.from_dict(fam, dict(cfgd))).eval()
    ref = RefCls(ConfigManager.from_dict(fam, dict(cfgd))).eval()
    ref.load_state_dict(adapter.state_dict())

    exportable = ExportableModel(copy.deepcopy(adapter), device="cpu", input_size=size)
    assert exportable.model._fx_exporting() and not adapter._fx_exporting()      # the COPY is the export model; the live one keeps the engine
    assert exportable.model._fx_engine is None
    torch.manual_seed(1)
    data = 128 * torch.randn(1, 3, size, size)
    data2 = 255 * torch.rand(1, 3, size, size)
    with torch.no_grad():
        warm = exportable(data)                        # focoos_model.py:487 ("hack to warm up the model")
        traced = torch.jit.trace(exportable, data, check_trace=False)
        for x in (data, data2):
            want = ref(x).to_tuple()
            got = traced(x)
            assert len(got) == len(want)
            for g, w in zip(got, want):
                if w is None:
                    continue
                torch.testing.assert_close(g, w, rtol=1e-4, atol=1e-4)
        for g, w in zip(warm, ref(data).to_tuple()):
            if w is not None:
                torch.testing.assert_close(g, w, rtol=1e-4, atol=1e-4)
    # a tracer running over the LIVE adapter (no switch_to_export) takes the same branch - never the engine with constant outputs
    class Tuple(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, x):
            return self.m(x).to_tuple()

    with torch.no_grad():
        traced_live = torch.jit.trace(Tuple(adapter), data, check_trace=False)
        for g, w in zip(traced_live(data2), ref(data2).to_tuple()):
            if w is not None:
                torch.testing.assert_close(g, w, rtol=1e-4, atol=1e-4)
    assert not adapter._fx_exporting() and adapter._fx_engine is None     # tracing left no mark on the live model
    if not torch.cuda.is_available():
        from focoos_amd._lib import FocoosAmdError

        with pytest.raises(FocoosAmdError), torch.no_grad():      # outside export the live adapter still refuses a CPU forward loudly
            adapter(data)


def test_registry_serves_the_reference_label_names_when_focoos_is_installed():
    """`classes` of a registry entry = the reference registry file's list (read from the installed package's own JSON, located without
    importing it), so FocoosDet.label and a training run's model_info.json carry 'person' / 'wall' ... - placeholders only where no focoos exists."""
    import json
    import os

    ref_import.install()
    from focoos_amd.registry import ModelRegistry

    for name in ModelRegistry.list_models():
        ref = json.load(open(os.path.join(ref_import.REFERENCE_ROOT, "focoos", "model_registry", f"{name}.json")))
        got = ModelRegistry.get_model_info(name)
        assert got["classes"] == ref["classes"] and len(got["classes"]) == int(got["config"]["num_classes"]), name
