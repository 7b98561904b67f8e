"""Host-side mirror of the reference's public surface for the RT-DETR and MaskFormer paths:

  ModelManager.get()        focoos/model_manager.py:42-155
  FocoosModel.__call__/infer focoos/models/focoos_model.py:370-416,575-621
  FAIDetr (BaseModelNN)     focoos/models/fai_detr/modelling.py:1273-1358, focoos/models/base_model.py:17-143
  FAIMaskFormer             focoos/models/fai_mf/modelling.py:633-725

Same names, argument meaning and error behaviour; the compute is the HIP engine (engine.py).
``FocoosModel.train`` drives the HIP training step (trainer.py; RT-DETR and BiSeNetFormer families); ``.export()`` is out of scope and raises
NotImplementedError (loudly — never a silent fallback)."""
from __future__ import annotations

import os
import warnings
from collections import OrderedDict
from time import perf_counter
from typing import Callable, Dict, List, Optional, Tuple, Type, Union

import numpy as np
import torch

from .engine import DetrEngine
from .engine_bf import BfEngine
from .engine_mf import MfEngine
from .ports import MODELS_DIR, BisenetFormerOutput, DETRModelOutput, FocoosDetections, InferLatency, MaskFormerModelOutput, ModelInfo
from .processor import BisenetFormerProcessor, DETRProcessor, MaskFormerProcessor
from .registry import ModelRegistry
from .state_spec import state_spec
from .synth import synth_state_dict


class IncompatibleKeys:
    def __init__(self, missing_keys, unexpected_keys, incorrect_shapes):
        self.missing_keys, self.unexpected_keys, self.incorrect_shapes = missing_keys, unexpected_keys, incorrect_shapes

    def __repr__(self):
        return f"IncompatibleKeys(missing={self.missing_keys}, unexpected={self.unexpected_keys}, incorrect_shapes={self.incorrect_shapes})"


class _EngineModel:
    """BaseModelNN surface shared by the engine-backed model classes (focoos/models/base_model.py:17-143).
    ``state_dict()`` keeps the reference's key names and fp32 values, so checkpoints round-trip; the packed bf16 copies the
    kernels read are rebuilt on ``load_state_dict``."""

    family = ""

    def _make_engine(self):  # pragma: no cover - overridden
        raise NotImplementedError

    def __init__(self, config: dict, device: Union[str, torch.device] = "cuda:0", seed: int = 0, **engine_kwargs):
        self.config = dict(config)
        self.training = False
        self._spec = state_spec(self.config, self.family)
        self._state = synth_state_dict(self.config, seed, family=self.family)  # "random init" (no network -> no pretrained weights)
        self._device = torch.device(device)
        self._engine_kwargs = engine_kwargs
        self.engine = self._make_engine()
        self.num_classes = int(self.config["num_classes"])

    # ---- BaseModelNN surface
    @property
    def device(self) -> torch.device:
        return self._device

    @property
    def dtype(self) -> torch.dtype:
        return torch.float32  # pixel_mean.dtype in the reference (modelling.py:1340-1342)

    def __call__(self, *args, **kwargs):
        """nn.Module semantics of the reference's BaseModelNN: ``model(images, targets)`` is ``model.forward(images, targets)``."""
        return self.forward(*args, **kwargs)

    def eval(self):
        self.training = False
        return self

    def train(self, mode: bool = True):
        """nn.Module.train: the engine-backed module itself is the inference graph; the trainable graph (train_detr.FAIDetrTrainable)
        is built from its state_dict by FocoosModel.train / trainer.run_train.  Calling forward in training mode raises."""
        self.training = bool(mode)
        return self

    def cuda(self, device=None):
        return self

    def to(self, *a, **k):
        return self

    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        return OrderedDict((k, v.clone()) for k, v in self._state.items())

    def load_state_dict(self, checkpoint_state_dict: Dict[str, torch.Tensor], strict: bool = True) -> IncompatibleKeys:
        """base_model.py:98-143: strips a DDP ``module.`` prefix, skips shape-mismatched tensors (reported),
        reports missing / unexpected keys; raises only when ``strict`` and something is off."""
        sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in checkpoint_state_dict.items()}
        missing, unexpected, bad = [], [], []
        new = OrderedDict(self._state)
        for k, (shape, _) in self._spec.items():
            if k not in sd:
                missing.append(k)
                continue
            v = sd[k].detach().cpu()
            if tuple(v.shape) != tuple(shape):
                bad.append((k, tuple(v.shape), tuple(shape)))
                continue
            new[k] = v.to(self._state[k].dtype).clone()
        unexpected = [k for k in sd if k not in self._spec]
        if strict and (missing or unexpected or bad):
            raise RuntimeError(f"Error(s) in loading state_dict: missing={missing[:5]} unexpected={unexpected[:5]} incorrect_shapes={bad[:5]}")
        self._state = new
        self.engine.load_state_dict(new)
        return IncompatibleKeys(missing, unexpected, bad)

    # ---- forward
    def _to_nhwc(self, images: torch.Tensor) -> torch.Tensor:
        if images.dim() != 4:
            raise ValueError("images must be [B,3,H,W] float (reference contract) or [B,H,W,3] uint8/float32")
        if images.shape[1] == 3 and images.shape[-1] != 3:  # reference contract: NCHW, 0..255, un-normalised
            images = images.permute(0, 2, 3, 1)
        if images.dtype != torch.uint8:
            images = images.to(torch.float32)
        return images.to(self._device).contiguous()



class FAIDetr(_EngineModel):
    """Engine-backed RT-DETR (focoos/models/fai_detr/modelling.py:1273-1358)."""

    family = "fai_detr"

    def _make_engine(self):
        return DetrEngine(self.config, self._state, str(self._device))

    def forward(self, images: torch.Tensor, targets: list = [], forced_topk: Optional[torch.Tensor] = None,
                use_graph: bool = True) -> DETRModelOutput:
        if self.training or (targets is not None and len(targets) > 0):
            raise NotImplementedError("the engine-backed module is the inference graph; the training forward (losses) runs in "
                                      "train_detr.FAIDetrTrainable, reached through FocoosModel.train / trainer.run_train")
        pl = self.engine.forward(self._to_nhwc(images), forced_topk=forced_topk, use_graph=use_graph)
        self.last_plan = pl
        return DETRModelOutput(logits=pl.probs.clone(), boxes=pl.boxes.clone(), loss=None)

    __call__ = forward

    def detect(self, images: torch.Tensor, sizes: Optional[torch.Tensor] = None, threshold: Optional[float] = None):
        """Fused forward + device post-process.  Returns the plan; ``det_scores/det_labels/det_boxes/det_count``
        hold the packed results."""
        pl = self.engine.forward(self._to_nhwc(images), sizes=sizes, threshold=threshold)
        self.last_plan = pl
        return pl


class FAIMaskFormer(_EngineModel):
    """Engine-backed MaskFormer (focoos/models/fai_mf/modelling.py:633-725).  ``forward`` returns the reference's
    ``MaskFormerModelOutput`` (full-resolution fp32 ``masks``, a separate plan that adds the upsample kernel);
    ``detect`` is the fused forward + device post-process that never materialises the [B,Q,H,W] tensor."""

    family = "fai_mf"

    def _make_engine(self):
        return MfEngine(self.config, self._state, str(self._device), **self._engine_kwargs)

    def forward(self, images: torch.Tensor, targets: list = [], forced_attn=None, use_graph: bool = True) -> MaskFormerModelOutput:
        if self.training or (targets is not None and len(targets) > 0):
            raise NotImplementedError("the engine-backed module is the inference graph; the training forward (losses) runs in "
                                      "train_detr.FAIDetrTrainable, reached through FocoosModel.train / trainer.run_train")
        pl = self.engine.forward(self._to_nhwc(images), forced_attn=forced_attn, use_graph=use_graph, full_masks=True)
        self.last_plan = pl
        return MaskFormerModelOutput(masks=pl.masks.clone(), logits=pl.probs.clone(), loss=None)

    __call__ = forward

    def detect(self, images: torch.Tensor, sizes: Optional[torch.Tensor] = None, threshold: Optional[float] = None):
        pl = self.engine.forward(self._to_nhwc(images), threshold=threshold, full_masks=False)
        self.last_plan = pl
        return pl


class BisenetFormer(FAIMaskFormer):
    """Engine-backed BiSeNetFormer (focoos/models/bisenetformer/modelling.py:523-609): same contract as FAIMaskFormer
    (``forward`` -> BisenetFormerOutput with the full-resolution fp32 ``masks``; ``detect`` -> fused forward + device post-process,
    whose plan also holds ``winner`` [B,H,W] uint8, the per-pixel query index of the predict_all_pixels branch)."""

    family = "bisenetformer"

    def _make_engine(self):
        return BfEngine(self.config, self._state, str(self._device), **self._engine_kwargs)

    def forward(self, images: torch.Tensor, targets: list = [], forced_attn=None, use_graph: bool = True) -> BisenetFormerOutput:
        out = super().forward(images, targets, forced_attn, use_graph)
        return BisenetFormerOutput(masks=out.masks, logits=out.logits, loss=None)

    __call__ = forward


class ProcessorManager:
    """focoos/processor/processor_manager.py:8-46 — family -> processor class registry (seam B1) with lazy loaders."""

    _PROCESSOR_MAPPING: Dict[str, Callable[[], Type]] = {
        "fai_detr": lambda: DETRProcessor, "fai_mf": lambda: MaskFormerProcessor, "bisenetformer": lambda: BisenetFormerProcessor}

    @classmethod
    def register_processor(cls, model_family, processor_loader: Callable[[], Type]):
        cls._PROCESSOR_MAPPING[getattr(model_family, "value", model_family)] = processor_loader

    @classmethod
    def get_processor(cls, model_family, model_config: dict, image_size=None):
        fam = getattr(model_family, "value", model_family)
        if fam not in cls._PROCESSOR_MAPPING:
            raise ValueError(f"Processor for {model_family} not supported")
        return cls._PROCESSOR_MAPPING[fam]()(config=model_config, image_size=image_size)


class FocoosModel:
    """focoos/models/focoos_model.py:88-147 — model + processor + model_info."""

    def __init__(self, model: _EngineModel, model_info: ModelInfo):
        self.model = model
        self.model_info = model_info
        # the mask families' processors ignore image_size (fai_mf/processor.py:96: no resize)
        self.processor = ProcessorManager.get_processor(model.family, model_info.config, image_size=model_info.im_size).eval()
        self.model.eval()

    @property
    def device(self):
        return self.model.device

    def infer_batch(self, inputs: list, threshold: Optional[float] = None) -> List[FocoosDetections]:
        """Batched inference: list of images in, list of FocoosDetections out (per-image result == the
        reference at B=1; the reference's own __call__ drops all but image 0 — focoos_model.py:615-621)."""
        t0 = perf_counter()
        images, _ = self.processor.preprocess(inputs, device=self.model.device, dtype=self.model.dtype)
        t1 = perf_counter()
        sizes = torch.tensor(self.processor.get_image_sizes(inputs), dtype=torch.int32)
        thr = threshold or self.processor.threshold
        pl = self.model.detect(images, sizes=sizes, threshold=thr)
        torch.cuda.current_stream(self.model.device).synchronize()
        t2 = perf_counter()
        if self.model.family in ("fai_mf", "bisenetformer"):
            out = self.processor.pack_detections(pl, self.model_info.classes)
        else:
            out = self.processor.pack_detections(pl.det_scores, pl.det_labels, pl.det_boxes, pl.det_count, self.model_info.classes)
        t3 = perf_counter()
        for o in out:
            o.latency = InferLatency(preprocess=round(t1 - t0, 3), inference=round(t2 - t1, 3), postprocess=round(t3 - t2, 3))
        return out

    def infer_stream(self, batches, threshold: Optional[float] = None, depth: Optional[int] = None):
        """Throughput inference (round 6): iterate over ``batches`` (each a list of images of ONE common shape after preprocessing and one
        batch size) and yield each batch's ``List[FocoosDetections]`` in order, with up to ``depth`` batches in flight on the engine's
        pipeline (engine.pipeline(): batch i+1's backbone runs beside batch i's decoder tail; default depth 3).  Same results as
        ``infer_batch`` per batch, bit for bit - the lanes share nothing but the read-only weights; what changes is that the GPU is not
        drained between batches (RT-DETR-L bs = 32: +8 % images/s, MaskFormer-L +15 %, BiSeNetFormer-L +36 %)."""
        from collections import deque

        thr = threshold or self.processor.threshold
        mask_family = self.model.family in ("fai_mf", "bisenetformer")
        pipe, pending = None, deque()

        def finish():
            ticket = pending.popleft()
            pl = pipe.wait(ticket)
            if mask_family:
                return self.processor.pack_detections(pl, self.model_info.classes)
            return self.processor.pack_detections(pl.det_scores, pl.det_labels, pl.det_boxes, pl.det_count, self.model_info.classes)

        for inputs in batches:
            images, _ = self.processor.preprocess(inputs, device=self.model.device, dtype=self.model.dtype)
            x = self.model._to_nhwc(images)
            sizes = torch.tensor(self.processor.get_image_sizes(inputs), dtype=torch.int32, device=x.device)
            B, H, W, _ = x.shape
            if pipe is None or (pipe.B, pipe.H, pipe.W) != (B, H, W):
                while pending:
                    yield finish()
                eng = self.model.engine
                pipe = (eng.pipeline(B, H, W, depth, x.dtype == torch.float32, 1, False) if mask_family
                        else eng.pipeline(B, H, W, depth, x.dtype == torch.float32))
            if len(pending) >= pipe.depth:      # the lane about to be reused still holds an unread result
                yield finish()
            pending.append(pipe.submit(x, sizes, thr))
        while pending:
            yield finish()

    def __call__(self, inputs, **kwargs) -> FocoosDetections:
        lst = inputs if isinstance(inputs, list) else [inputs]
        return self.infer_batch(lst, threshold=kwargs.get("threshold"))[0]

    def infer(self, image, threshold: Optional[float] = None, annotate: bool = False) -> FocoosDetections:
        if isinstance(image, (str, os.PathLike)):
            from PIL import Image

            image = np.array(Image.open(image).convert("RGB"))
        if annotate:
            raise NotImplementedError("annotation/drawing is outside the hot path")
        return self(image, threshold=threshold)

    def train(self, args, data_train, data_val=None, hub=None):
        """focoos_model.py:221-274: fine-tune on ``data_train`` (indexable, entries = ports.DatasetEntry) with ``args`` (ports.TrainerArgs):
        one process per GPU, TrainStep on every rank, artifacts in ``args.output_dir/args.run_name``, weights reloaded into the engine."""
        from .trainer import train as _train

        return _train(self, args, data_train, data_val, hub)

    def export(self, *a, **k):
        raise NotImplementedError("FocoosModel.export: the ONNX/TensorRT export path is out of scope (BASELINE north_star)")


class ModelManager:
    """focoos/model_manager.py:17-155 — lazy family registry + ``get``."""

    _MODEL_MAPPING: Dict[str, Callable[[], Type]] = {"fai_detr": lambda: FAIDetr, "fai_mf": lambda: FAIMaskFormer,
                                                              "bisenetformer": lambda: BisenetFormer}

    @classmethod
    def register_model(cls, model_family: str, model_loader: Callable[[], Type]):
        cls._MODEL_MAPPING[getattr(model_family, "value", model_family)] = model_loader

    @classmethod
    def _from_local_dir(cls, name: str) -> ModelInfo:
        """model_manager.py:158-188: a run directory (as given, or under the models directory) holding ``model_info.json``; a bare
        ``model_final.pth`` in it is resolved against the directory."""
        run_dir = name
        if not os.path.exists(run_dir):
            run_dir = os.path.join(MODELS_DIR, name)
            if not os.path.exists(run_dir):
                raise ValueError(f"Run {run_dir} not exists.")
        info_path = os.path.join(run_dir, "model_info.json")
        if not os.path.exists(info_path):
            raise ValueError(f"Model info not found in {run_dir}")
        info = ModelInfo.from_json(info_path)
        if info.weights_uri == "model_final.pth":
            info.weights_uri = os.path.join(run_dir, info.weights_uri)
        return info

    @classmethod
    def get(cls, name: str, model_info: Optional[ModelInfo] = None, config: Optional[dict] = None, device: str = "cuda:0", seed: int = 0,
            **kwargs) -> FocoosModel:
        if model_info is None:
            # model_manager.py:74-91: hub reference / registry name / local run directory, in this order
            name = os.fspath(name)
            if name.startswith("hub://"):
                raise NotImplementedError("hub:// references need the Focoos Hub client (network; outside this engine): download the run and pass its folder")
            if ModelRegistry.exists(name):
                model_info = ModelInfo.from_json(ModelRegistry.get_model_info(name))
            else:
                model_info = cls._from_local_dir(name)
        fam = getattr(model_info.model_family, "value", model_info.model_family)
        if fam not in cls._MODEL_MAPPING:
            raise ValueError(f"Model {fam} not supported")
        cfg = dict(model_info.config)
        if config:
            cfg.update(config)
        cfg.update(kwargs)
        model_info.config = cfg
        nn = cls._MODEL_MAPPING[fam]()(cfg, device=device, seed=seed)
        fm = FocoosModel(nn, model_info)
        if model_info.weights_uri:
            if not os.path.exists(model_info.weights_uri):
                raise FileNotFoundError(f"Weights file not found: {model_info.weights_uri}")
            state = torch.load(model_info.weights_uri, map_location="cpu", weights_only=True)
            nn.load_state_dict(state.get("model", state) if isinstance(state, dict) else state, strict=False)
        else:
            warnings.warn(f"⚠️ Model {model_info.name} has no pretrained weights (offline): seeded synthetic weights (seed={seed})")
        return fm
