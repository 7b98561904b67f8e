"""Boundary types of the hot path — field-for-field mirrors of the reference's dataclasses
(focoos/ports.py:303-333 FocoosDet, :360-373 InferLatency, :373-… FocoosDetections;
focoos/models/fai_detr/ports.py:9-19 DETRModelOutput / DETRTargets).  When the reference package is
installed next to this one, ``focoos_amd.integration`` uses the reference's own classes instead."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, List, Optional, Tuple, Union

import torch


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
class DETRModelOutput:
    boxes: torch.Tensor   # [N, num_queries, 4] XYXY normalised to [0,1]
    logits: torch.Tensor  # [N, num_queries, num_classes] (probabilities, like the reference)
    loss: Optional[dict] = None


@dataclass
class MaskFormerModelOutput:
    """focoos/models/fai_mf/ports.py:9-13."""
    masks: torch.Tensor   # [N, num_queries, H, W] mask probabilities (sigmoid, bilinearly upsampled to the input size)
    logits: torch.Tensor  # [N, num_queries, num_classes] class probabilities (softmax, no-object column dropped)
    loss: Optional[dict] = None


@dataclass
class BisenetFormerOutput:
    """focoos/models/bisenetformer/ports.py (same fields as MaskFormerModelOutput)."""
    masks: torch.Tensor   # [N, num_queries, H, W] mask probabilities (sigmoid at 1/8 resolution, bilinearly upsampled to the input size)
    logits: torch.Tensor  # [N, num_queries, num_classes] class probabilities (softmax, no-object column dropped)
    loss: Optional[dict] = None


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
