"""Boundary types of the hot path — field-for-field mirrors of the reference's dataclasses
(focoos/ports.py:303-333 FocoosDet, :360-373 InferLatency, :373-… FocoosDetections;
focoos/models/fai_detr/ports.py:9-19 DETRModelOutput / DETRTargets).  When the reference package is
installed next to this one, ``focoos_amd.integration`` uses the reference's own classes instead."""
from __future__ import annotations

import json
import os
from collections import OrderedDict
from dataclasses import dataclass, field, fields
from typing import Any, List, Optional, Tuple, Union

import torch


# focoos/ports.py:22-24: where runs live when a model is named by its run folder (ModelManager._from_local_dir)
MODELS_DIR = os.path.join(os.path.expanduser("~"), "FocoosAI", "models")


class DictClass(OrderedDict):
    """The reference's container base (focoos/ports.py:875-922): a dataclass that is ALSO an ordered mapping of its fields - what
    ``BaseModelNN.forward`` returns (``ModelOutput``) and what the data pipeline hands over (``DatasetEntry``).  Same observable
    behaviour: ``obj["field"]`` and ``obj.field`` are one value, an integer (or slice) index addresses ``to_tuple()`` - the fields that
    are not None, in declaration order (what the export path traces) -, attribute assignment of a non-None value and item assignment
    keep both views in step, and the object pickles by its fields.  (One addition: ``__setstate__`` restores the mapping view too; the
    reference's unpickled objects come back with attributes only.)"""

    def __post_init__(self):
        names = [f.name for f in fields(self)]
        if not names:
            raise ValueError(f"{type(self).__name__} has no fields.")
        for n in names:
            OrderedDict.__setitem__(self, n, getattr(self, n))

    def to_tuple(self) -> tuple:
        return tuple(v for v in OrderedDict.values(self) if v is not None)

    def __getitem__(self, k):
        if isinstance(k, str):
            return OrderedDict.__getitem__(self, k)
        return self.to_tuple()[k]

    def __setattr__(self, name, value):
        if value is not None and OrderedDict.__contains__(self, name):
            OrderedDict.__setitem__(self, name, value)
        object.__setattr__(self, name, value)

    def __setitem__(self, key, value):
        OrderedDict.__setitem__(self, key, value)
        object.__setattr__(self, key, value)

    def __reduce__(self):
        return (type(self).__new__, (type(self),), {f.name: getattr(self, f.name) for f in fields(self)})

    def __setstate__(self, state):
        for k, v in state.items():
            object.__setattr__(self, k, v)
            OrderedDict.__setitem__(self, k, v)


@dataclass
class ModelOutput(DictClass):
    """focoos/ports.py:930-934 - model output base container."""
    loss: Optional[dict]


@dataclass
class FocoosDet:
    bbox: Optional[List[int]] = None
    conf: Optional[float] = None
    cls_id: Optional[int] = None
    label: Optional[str] = None
    mask: Optional[str] = None
    keypoints: Optional[List[Tuple[int, int, float]]] = None


@dataclass
class InferLatency:
    imload: Optional[float] = None
    preprocess: Optional[float] = None
    inference: Optional[float] = None
    postprocess: Optional[float] = None
    annotate: Optional[float] = None


@dataclass
class FocoosDetections:
    detections: List[FocoosDet] = field(default_factory=list)
    latency: Optional[InferLatency] = None
    image: Optional[Any] = None

    def __len__(self):
        return len(self.detections)


@dataclass
class DETRModelOutput(ModelOutput):
    """focoos/models/fai_detr/ports.py:9-13."""
    boxes: torch.Tensor   # [N, num_queries, 4] XYXY normalised to [0,1]
    logits: torch.Tensor  # [N, num_queries, num_classes] (probabilities, like the reference)
    loss: Optional[dict]   # (declared in ModelOutput: stays the FIRST field, like in the reference - construct by keyword)


@dataclass
class MaskFormerModelOutput(ModelOutput):
    """focoos/models/fai_mf/ports.py:9-13."""
    masks: torch.Tensor   # [N, num_queries, H, W] mask probabilities (sigmoid, bilinearly upsampled to the input size)
    logits: torch.Tensor  # [N, num_queries, num_classes] class probabilities (softmax, no-object column dropped)
    loss: Optional[dict]   # (declared in ModelOutput: stays the FIRST field, like in the reference - construct by keyword)


@dataclass
class BisenetFormerOutput(ModelOutput):
    """focoos/models/bisenetformer/ports.py (same fields as MaskFormerModelOutput)."""
    masks: torch.Tensor   # [N, num_queries, H, W] mask probabilities (sigmoid at 1/8 resolution, bilinearly upsampled to the input size)
    logits: torch.Tensor  # [N, num_queries, num_classes] class probabilities (softmax, no-object column dropped)
    loss: Optional[dict]   # (declared in ModelOutput: stays the FIRST field, like in the reference - construct by keyword)


@dataclass
class MaskFormerTargets:
    """focoos/models/fai_mf/ports.py:16-19 (BisenetFormerTargets has the same fields)."""
    labels: torch.Tensor  # [T] class ids
    masks: torch.Tensor   # [T, H, W] binary masks (bool / uint8 / float)


BisenetFormerTargets = MaskFormerTargets


@dataclass
class DynamicAxes:
    """focoos/ports.py:1357-1363 - dynamic axes for model export."""
    input_names: List[str]
    output_names: List[str]
    dynamic_axes: dict


@dataclass
class DETRTargets:
    labels: torch.Tensor
    boxes: torch.Tensor


@dataclass
class ModelInfo:
    name: str
    model_family: str
    classes: List[str]
    im_size: Union[int, Tuple[int, int]]
    task: str
    config: dict
    weights_uri: Optional[str] = None
    description: Optional[str] = None

    @classmethod
    def from_json(cls, data) -> "ModelInfo":
        """focoos/ports.py:32-38 (PydanticBase.from_json): a path to a ``model_info.json`` - what the reference's trainer and this one write
        next to ``model_final.pth`` - or the parsed dict.  The fields the engine does not use (ref, status, train_args, metrics, latency ...)
        are accepted and dropped; enum-valued fields arrive as their string values."""
        if isinstance(data, (str, os.PathLike)):
            with open(data, encoding="utf-8") as f:
                data = json.load(f)
        missing = [k for k in ("name", "model_family", "classes", "im_size", "task", "config") if k not in data]
        if missing:
            raise ValueError(f"model info without the required field(s) {missing}")
        return cls(**{f.name: data[f.name] for f in fields(cls) if f.name in data})


# ---- evaluation-side structures (focoos/structures.py: Boxes :18-170, Instances :430-560), the subset eval_postprocess needs
class Boxes:
    """[N,4] xyxy boxes with the in-place scale / clip / nonempty helpers detector_postprocess uses (structures.py:57-100,156-161)."""

    def __init__(self, tensor):
        import torch

        if not isinstance(tensor, torch.Tensor):
            tensor = torch.as_tensor(tensor, dtype=torch.float32)
        if tensor.numel() == 0:
            tensor = tensor.reshape((-1, 4)).to(dtype=torch.float32)
        assert tensor.dim() == 2 and tensor.size(-1) == 4, tensor.size()
        self.tensor = tensor

    def __len__(self):
        return self.tensor.shape[0]

    def scale(self, scale_x: float, scale_y: float) -> None:
        self.tensor[:, 0::2] *= scale_x
        self.tensor[:, 1::2] *= scale_y

    def clip(self, box_size) -> None:
        import torch

        assert torch.isfinite(self.tensor).all(), "Box tensor contains infinite or NaN!"
        h, w = box_size
        self.tensor = torch.stack((self.tensor[:, 0].clamp(min=0, max=w), self.tensor[:, 1].clamp(min=0, max=h),
                                   self.tensor[:, 2].clamp(min=0, max=w), self.tensor[:, 3].clamp(min=0, max=h)), dim=-1)

    def nonempty(self, threshold: float = 0.0):
        b = self.tensor
        return ((b[:, 2] - b[:, 0]) > threshold) & ((b[:, 3] - b[:, 1]) > threshold)

    def __getitem__(self, item) -> "Boxes":
        if isinstance(item, int):
            return Boxes(self.tensor[item].view(1, -1))
        return Boxes(self.tensor[item])


class BitMasks:
    """[N,H,W] boolean masks of one image (structures.py:292-420, the subset the mask families' eval_postprocess returns): indexing,
    length and the tight boxes of ``get_bounding_boxes`` (:384-398: [x0, y0, x1 + 1, y1 + 1]; an empty mask gives zeros)."""

    def __init__(self, tensor):
        import torch

        tensor = torch.as_tensor(tensor)
        assert tensor.dim() == 3, tensor.size()
        self.tensor = tensor.to(torch.bool)
        self.image_size = tuple(tensor.shape[1:])

    def __len__(self):
        return self.tensor.shape[0]

    def __getitem__(self, item):
        import torch

        if isinstance(item, int):
            return BitMasks(self.tensor[item].unsqueeze(0))
        m = self.tensor[item]
        assert m.dim() == 3, f"indexing with {item} gave shape {tuple(m.shape)}"
        return BitMasks(m)

    def nonempty(self):
        return self.tensor.flatten(1).any(dim=1)

    def get_bounding_boxes(self) -> "Boxes":
        import torch

        n, h, w = self.tensor.shape
        boxes = torch.zeros(n, 4, dtype=torch.float32)
        if n:
            xs, ys = self.tensor.any(dim=1), self.tensor.any(dim=2)            # [N,W], [N,H]
            ok = (xs.any(dim=1) & ys.any(dim=1)).cpu()
            ar_w, ar_h = torch.arange(w, device=xs.device), torch.arange(h, device=ys.device)
            x0 = torch.where(xs, ar_w, w).amin(dim=1)
            x1 = torch.where(xs, ar_w, -1).amax(dim=1) + 1
            y0 = torch.where(ys, ar_h, h).amin(dim=1)
            y1 = torch.where(ys, ar_h, -1).amax(dim=1) + 1
            b = torch.stack([x0, y0, x1, y1], dim=1).to(torch.float32).cpu()
            boxes[ok] = b[ok]
        return Boxes(boxes)


class Instances:
    """Per-image container of equally long fields (structures.py Instances): ``Instances(image_size, boxes=..., scores=..., classes=...)``."""

    def __init__(self, image_size, **fields):
        self.image_size = tuple(image_size)
        self._fields = {}
        for k, v in fields.items():
            self.set(k, v)

    def set(self, name, value):
        if self._fields:
            assert len(value) == len(self), f"field {name}: length {len(value)} != {len(self)}"
        self._fields[name] = value

    def __getattr__(self, name):
        f = self.__dict__.get("_fields", {})
        if name in f:
            return f[name]
        raise AttributeError(name)

    def has(self, name):
        return name in self._fields

    def get_fields(self):
        return self._fields

    def __len__(self):
        for v in self._fields.values():
            return len(v)
        return 0

    def __getitem__(self, item) -> "Instances":
        return Instances(self.image_size, **{k: v[item] for k, v in self._fields.items()})


# ---- training-side ports (focoos/ports.py: DatasetEntry :938-944, TrainerArgs :970-1065)
@dataclass
class DatasetEntry(DictClass):
    """One training / evaluation sample: ``image`` CHW uint8 (or float) tensor, ``instances`` with ``boxes`` (Boxes, absolute xyxy) and
    ``classes`` (int64 tensor), ``height`` / ``width`` of the original image."""
    image: Optional[object] = None
    height: Optional[int] = None
    width: Optional[int] = None
    instances: Optional[Instances] = None
    file_name: Optional[str] = None
    image_id: Optional[int] = None


@dataclass
class TrainerArgs:
    """focoos/ports.py:970-1065 - same fields and defaults (hub syncing / visualisation fields are accepted and ignored by the engine's
    trainer).  ``batch_size`` is the TOTAL batch over all GPUs (data/loaders.py:61-65)."""
    run_name: str
    output_dir: str = "./focoos_amd_runs"
    ckpt_dir: Optional[str] = None
    init_checkpoint: Optional[str] = None
    resume: bool = False
    num_gpus: int = 1
    device: str = "cuda"
    workers: int = 4
    amp_enabled: bool = True
    ddp_broadcast_buffers: bool = False
    ddp_find_unused: bool = True
    checkpointer_period: int = 1000
    checkpointer_max_to_keep: int = 1
    eval_period: int = 200
    log_period: int = 20
    samples: int = 9
    seed: int = 42
    early_stop: bool = True
    patience: int = 10
    ema_enabled: bool = False
    ema_decay: float = 0.999
    ema_warmup: int = 2000
    learning_rate: float = 5e-4
    weight_decay: float = 0.02
    max_iters: int = 3000
    batch_size: int = 16
    scheduler: str = "MULTISTEP"
    scheduler_extra: Optional[dict] = None
    optimizer: str = "ADAMW"
    optimizer_extra: Optional[dict] = None
    weight_decay_norm: float = 0.0
    weight_decay_embed: float = 0.0
    backbone_multiplier: float = 0.1
    decoder_multiplier: float = 1.0
    head_multiplier: float = 1.0
    freeze_bn: bool = False
    clip_gradients: float = 0.1
    size_divisibility: int = 0
    gather_metric_period: int = 1
    zero_grad_before_forward: bool = False
    sync_to_hub: bool = False
