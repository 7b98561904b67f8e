"""TEST INFRASTRUCTURE — shim importer for the *real* reference (FocoosAI/focoos).

Only usable where ``/root/reference`` exists (the build container); the GPU box
does not have it, so nothing that runs under ``-m gpu``, ``smoke()`` or
``bench.py`` may call into this module.  It is used to
  * validate ``oracle/detr_oracle.py`` (the CPU restatement) against the
    reference's own PyTorch-CPU implementation, and
  * generate the golden fixtures under ``tests/golden/`` (``scripts/make_golden.py``).

The reference cannot be imported as-is in this image: ``focoos/__init__.py``
eagerly imports hub/infer/vision modules that need packages which are not
installed (pydantic_settings, torchvision, cv2, supervision, pycocotools,
fvcore, orjson, colorama, termcolor, tensorboard ...).  We therefore pre-seed
``sys.modules["focoos"]`` with an empty namespace package whose ``__path__``
points at the reference tree (so ``focoos/__init__.py`` is skipped) and stub the
missing third-party modules.  The reference tree itself is never modified
(``sys.dont_write_bytecode`` keeps ``__pycache__`` out of it).
"""
from __future__ import annotations

import importlib
import importlib.machinery
import importlib.metadata
import os
import sys
import types
from unittest import mock

REFERENCE_ROOT = os.environ.get("FOCOOS_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "focoos", "models", "fai_detr"))


_installed = False


def _stub_module(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _MockModule(types.ModuleType):
    """A module whose every attribute is a MagicMock, but which still carries a real
    ``__spec__`` (torch._dynamo's trace rules call importlib.util.find_spec on e.g. "onnx")."""

    def __init__(self, name):
        super().__init__(name)
        self.__spec__ = importlib.machinery.ModuleSpec(name, None)
        self.__path__ = []

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        v = mock.MagicMock(name=f"{self.__name__}.{item}")
        setattr(self, item, v)
        return v


def install() -> None:
    """Make ``import focoos.models.fai_detr...`` work against /root/reference."""
    global _installed
    if _installed:
        return
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True
    import pydantic
    import torch

    # namespace stub: skips focoos/__init__.py (which pulls hub/infer/cv2/...)
    pkg = types.ModuleType("focoos")
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, "focoos")]
    sys.modules["focoos"] = pkg

    # importlib.metadata.version("focoos") -> the surveyed version
    _orig_version = importlib.metadata.version

    def _version(name):
        if name == "focoos":
            return "0.25.0"
        return _orig_version(name)

    importlib.metadata.version = _version

    if "pydantic_settings" not in sys.modules:
        try:
            importlib.import_module("pydantic_settings")
        except ImportError:
            _stub_module("pydantic_settings", BaseSettings=pydantic.BaseModel)

    for name in [
        "pycocotools", "pycocotools.mask", "pycocotools.coco", "pycocotools.cocoeval",
        "cv2", "supervision", "orjson", "colorama", "IPython", "IPython.display", "faster_coco_eval",
    ]:
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                sys.modules[name] = _MockModule(name)

    if "termcolor" not in sys.modules:
        try:
            importlib.import_module("termcolor")
        except ImportError:
            _stub_module("termcolor", colored=lambda s, *a, **k: s)

    # fvcore.transforms.transform: real (empty) classes — the reference subclasses them
    try:
        importlib.import_module("fvcore.transforms.transform")
    except ImportError:
        class Transform:  # noqa: D401 - stub
            @classmethod
            def register_type(cls, *a, **k):
                return None

        names = ["CropTransform", "HFlipTransform", "NoOpTransform", "PadTransform",
                 "TransformList", "VFlipTransform", "BlendTransform"]
        tmod = _stub_module("fvcore.transforms.transform", Transform=Transform,
                            **{n: type(n, (Transform,), {}) for n in names})
        fv = _stub_module("fvcore")
        fvt = _stub_module("fvcore.transforms", transform=tmod)
        fv.transforms = fvt

    # torchvision: only box_area is on the hot path (focoos/utils/box.py:4)
    try:
        importlib.import_module("torchvision")
    except ImportError:
        def box_area(boxes):
            return (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])

        boxes_mod = _stub_module("torchvision.ops.boxes", box_area=box_area)
        ops_mod = _stub_module("torchvision.ops", boxes=boxes_mod, box_area=box_area,
                               nms=mock.MagicMock(), sigmoid_focal_loss=mock.MagicMock())
        io_image = _stub_module("torchvision.io.image", read_image=mock.MagicMock())
        io_mod = _stub_module("torchvision.io", ImageReadMode=mock.MagicMock(), image=io_image,
                              read_image=mock.MagicMock())
        tv = _stub_module("torchvision", ops=ops_mod, io=io_mod, _is_tracing=lambda: False)
        tv.__version__ = "0.0-stub"
        _stub_module("torchvision.transforms", functional=mock.MagicMock())
        _stub_module("torchvision.transforms.functional")

    _ = torch  # keep the import (first import pages the image in)
    _installed = True


def build_reference_detr(config_dict: dict, name: str = "fai-detr-l-obj365"):
    """Build the reference's FAIDetr + DETRProcessor from a registry-style config dict.

    Follows ModelManager._from_model_info (focoos/model_manager.py:129-155) but
    without ``FocoosModel`` (whose weight loading would try the network).
    Returns ``(model, processor, config)``; the model is in eval mode on CPU.
    """
    install()
    from focoos.model_manager import ConfigManager
    from focoos.models.fai_detr.modelling import FAIDetr
    from focoos.models.fai_detr.processor import DETRProcessor
    from focoos.ports import ModelFamily

    cfg = ConfigManager.from_dict(ModelFamily.DETR, dict(config_dict))
    model = FAIDetr(cfg)
    model.eval()
    im = config_dict.get("resolution") or 640
    proc = DETRProcessor(cfg, image_size=im).eval()
    return model, proc, cfg


def build_reference_mf(config_dict: dict):
    """Build the reference's FAIMaskFormer + MaskFormerProcessor from a registry-style config dict
    (same recipe as ``build_reference_detr``).  Returns ``(model, processor, config)``; eval mode, CPU."""
    install()
    import focoos.models.fai_mf as fam
    from focoos.model_manager import ConfigManager
    from focoos.models.fai_mf.modelling import FAIMaskFormer
    from focoos.models.fai_mf.processor import MaskFormerProcessor
    from focoos.ports import ModelFamily

    for attr in dir(fam):  # family registration hooks (model_manager.py:108-126)
        if attr.startswith("_register"):
            getattr(fam, attr)()
    cfg = ConfigManager.from_dict(ModelFamily.MASKFORMER, dict(config_dict))
    model = FAIMaskFormer(cfg)
    model.eval()
    proc = MaskFormerProcessor(cfg).eval()
    return model, proc, cfg


def build_reference_bf(config_dict: dict):
    """Build the reference's BisenetFormer + BisenetFormerProcessor from a registry-style config dict
    (same recipe as ``build_reference_detr``).  Returns ``(model, processor, config)``; eval mode, CPU."""
    install()
    import focoos.models.bisenetformer as fam
    from focoos.model_manager import ConfigManager
    from focoos.models.bisenetformer.modelling import BisenetFormer
    from focoos.models.bisenetformer.processor import BisenetFormerProcessor
    from focoos.ports import ModelFamily

    for attr in dir(fam):  # family registration hooks (model_manager.py:108-126)
        if attr.startswith("_register"):
            getattr(fam, attr)()
    cfg = ConfigManager.from_dict(ModelFamily.BISENETFORMER, dict(config_dict))
    model = BisenetFormer(cfg)
    model.eval()
    proc = BisenetFormerProcessor(cfg).eval()
    return model, proc, cfg
