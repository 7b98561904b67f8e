"""Seeded synthetic weights and inputs (no network -> no pretrained checkpoints).

Every tensor is drawn from ``numpy.random.RandomState`` seeded by (seed, crc32(name)),
so the same ``state_dict`` is reproduced bit-for-bit on any host (the build
container that generates the golden fixtures and the GPU box that checks them)
independently of tensor order.  Distributions are chosen so that the random
network is numerically well conditioned end to end (activations stay O(1)-O(10),
BatchNorm statistics are non-trivial so BN folding is really exercised, the
deformable-attention offsets are data dependent, and class scores straddle the
0.5 post-processing threshold).
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict

import numpy as np
import torch

from .state_spec import state_spec


def _rs(seed: int, name: str) -> np.random.RandomState:
    return np.random.RandomState((seed * 1000003 + zlib.crc32(name.encode())) % (2**32))


def synth_state_dict(config: Dict, seed: int = 0, family: str = "fai_detr") -> "OrderedDict[str, torch.Tensor]":
    spec = state_spec(config, family)
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, (shape, kind) in spec.items():
        rs = _rs(seed, name)
        if kind == "conv_w":
            fan_in = shape[1] * shape[2] * shape[3]
            a = rs.standard_normal(shape).astype(np.float32) * np.float32(np.sqrt(2.0 / fan_in))
        elif kind == "bn_w":
            if ".branch2c." in name:  # damp the residual branch so 16 blocks do not blow up
                a = rs.uniform(0.15, 0.35, shape).astype(np.float32)
            elif ".adapter_" in name:  # MaskFormer FPN laterals: bring the O(10) backbone features back to O(1)
                a = rs.uniform(0.03, 0.07, shape).astype(np.float32)
            elif ".cp." in name or ".ffm." in name:  # BiSeNet context path / fusion: keep the decoder memory O(1)
                a = rs.uniform(0.15, 0.35, shape).astype(np.float32)
            else:
                a = rs.uniform(0.6, 1.4, shape).astype(np.float32)
        elif kind == "bn_b":
            a = (rs.standard_normal(shape) * 0.1).astype(np.float32)
        elif kind == "bn_mean":
            a = (rs.standard_normal(shape) * 0.1).astype(np.float32)
        elif kind == "bn_var":
            a = rs.uniform(0.5, 1.5, shape).astype(np.float32)
        elif kind == "bn_nbt":
            a = np.zeros(shape, dtype=np.int64)
        elif kind == "lin_w":
            fan_in = shape[1]
            gain = 1.0
            if "score_classifier" in name:
                gain = 2.0
            elif name.endswith("forward_prediction_heads.classifier.weight"):
                gain = 3.0  # softmax over K+1 classes: make some queries confident enough to pass the 0.5 threshold
            elif "bbox_classifier" in name and name.endswith("layers.2.weight"):
                gain = 0.5
            elif "query_pos_head.layers.0" in name:
                gain = 2.0
            a = rs.standard_normal(shape).astype(np.float32) * np.float32(gain / np.sqrt(fan_in))
        elif kind == "lin_b":
            if "score_classifier" in name:
                a = (-6.0 + 0.5 * rs.standard_normal(shape)).astype(np.float32)
            elif "sampling_offsets" in name:
                a = (2.0 * rs.standard_normal(shape)).astype(np.float32)
            else:
                a = (0.02 * rs.standard_normal(shape)).astype(np.float32)
        elif kind == "ln_w":
            a = rs.uniform(0.8, 1.2, shape).astype(np.float32)
        elif kind == "ln_b":
            a = (0.05 * rs.standard_normal(shape)).astype(np.float32)
        elif kind == "emb":
            a = rs.standard_normal(shape).astype(np.float32)
        elif kind == "buf":
            a = np.ones(shape, dtype=np.float32)
            a[-1] = 0.1
        else:  # pragma: no cover
            raise ValueError(kind)
        out[name] = torch.from_numpy(np.ascontiguousarray(a)).reshape(shape)
    return out


def synth_image(index: int, height: int = 640, width: int = 640) -> np.ndarray:
    """Image ``index`` of the synthetic COCO-shaped stream: HWC uint8."""
    return np.random.RandomState(index).randint(0, 256, (height, width, 3)).astype(np.uint8)


def synth_image_structured(index: int, height: int = 640, width: int = 640) -> np.ndarray:
    """Smoother synthetic image (random rectangles + gradient) — exercises the
    detector on content with spatial structure rather than white noise."""
    rs = np.random.RandomState(10_000 + index)
    yy, xx = np.mgrid[0:height, 0:width]
    img = np.stack([(xx * 255 // max(width - 1, 1)), (yy * 255 // max(height - 1, 1)),
                    ((xx + yy) * 255 // max(height + width - 2, 1))], -1).astype(np.int32)
    for _ in range(12):
        x0, y0 = rs.randint(0, width - 8), rs.randint(0, height - 8)
        w, h = rs.randint(8, max(9, width // 2)), rs.randint(8, max(9, height // 2))
        col = rs.randint(0, 256, 3)
        img[y0:y0 + h, x0:x0 + w] = col
    img = img + rs.randint(-12, 13, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)
