"""DETRProcessor for the engine — same interface and semantics as
focoos/models/fai_detr/processor.py:60-217 (+ Processor.get_torch_batch / get_image_sizes,
focoos/processor/base_processor.py:176-296), with the arithmetic on the GPU:

* ``preprocess`` keeps images as HWC uint8 in HBM when no resize is needed (the stem kernel fuses the
  float conversion and (x-mean)/std), and resizes with ``fx_resize_bilinear_u8`` otherwise;
* ``postprocess`` reads the packed device results of ``fx_topk_rows_f32`` + ``fx_detr_postprocess``
  (one D2H copy per batch) and only builds the Python ``FocoosDet`` objects on the host.

Reference quirks kept on purpose: ``threshold or self.threshold`` (processor.py:171: 0.0 means
"default"), and a batched 4-D tensor input yields ONE image size (base_processor.py:196-202), so
batched inference must pass a list of images.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _lib
from ._lib import check
from .ports import DETRModelOutput, FocoosDet, FocoosDetections

try:  # PIL is optional
    from PIL import Image
except Exception:  # pragma: no cover
    Image = None

ImageInput = Union[torch.Tensor, np.ndarray, "Image.Image", list]


class DETRProcessor:
    def __init__(self, config: dict, image_size: Optional[Union[int, Tuple[int, int]]] = None):
        self.config = config
        self.image_size = image_size
        self.top_k = int(config.get("top_k", 300))
        self.threshold = float(config.get("threshold", 0.5))
        self.training = False

    def eval(self):
        self.training = False
        return self

    def train(self, training: bool = True):
        self.training = training
        return self

    # ---- base_processor.py:176-221
    def get_image_sizes(self, inputs: ImageInput) -> List[Tuple[int, int]]:
        def one(img):
            if isinstance(img, torch.Tensor):
                return tuple(int(v) for v in img.shape[-2:])
            if isinstance(img, np.ndarray):
                return tuple(int(v) for v in (img.shape[-3:-1] if img.ndim > 3 else img.shape[:2]))
            if Image is not None and isinstance(img, Image.Image):
                w, h = img.size
                return (h, w)
            raise ValueError(f"Unsupported input type: {type(img)}")

        if isinstance(inputs, list):
            return [one(i) for i in inputs]
        return [one(inputs)]

    def _target_size(self) -> Optional[Tuple[int, int]]:
        if self.image_size is None:
            return None
        return (self.image_size, self.image_size) if isinstance(self.image_size, int) else tuple(self.image_size)

    # ---- fai_detr/processor.py:66-119 (inference branch) + base_processor.py:223-296
    def preprocess(self, inputs: ImageInput, device: torch.device, dtype: torch.dtype = torch.float32):
        """Returns (images, targets).  ``images`` is NHWC on ``device``: uint8 [B,H,W,3] when every input
        already has the target size (fused fast path), float32 [B,H,W,3] (0..255 scale, bilinearly resized)
        otherwise.  The engine's ``FAIDetr.forward`` accepts both, as well as the reference's NCHW float."""
        if self.training:
            raise ValueError("During training, inputs should be a list of DetectionDatasetDict")  # training path: later round
        lst = inputs if isinstance(inputs, list) else [inputs]
        arrs = []
        for inp in lst:
            if Image is not None and isinstance(inp, Image.Image):
                inp = np.array(inp)
            if isinstance(inp, np.ndarray):
                inp = torch.from_numpy(np.ascontiguousarray(inp))
            if inp.dim() == 4:
                if inp.shape[0] != 1:
                    raise ValueError("pass a list of images for batched inference (reference H6)")
                inp = inp[0]
            if inp.shape[0] == 3 and inp.shape[-1] != 3:  # CHW -> HWC
                inp = inp.permute(1, 2, 0)
            arrs.append(inp.contiguous())
        tgt = self._target_size() or tuple(arrs[0].shape[:2])
        all_u8_same = all(a.dtype == torch.uint8 and tuple(a.shape[:2]) == tgt for a in arrs)
        if all_u8_same:
            batch = torch.stack(arrs, 0).to(device, non_blocking=True)
            return batch, []
        lib = _lib.load()
        out = torch.empty(len(arrs), tgt[0], tgt[1], 3, dtype=torch.float32, device=device)
        stream = torch.cuda.current_stream(device).cuda_stream
        for i, a in enumerate(arrs):
            if a.dtype != torch.uint8:
                raise ValueError("engine preprocess resizes uint8 images; float inputs must already have the target size")
            d = a.to(device, non_blocking=True)
            check(lib.fx_resize_bilinear_u8(d.data_ptr(), a.shape[0], a.shape[1], out[i].data_ptr(), tgt[0], tgt[1], stream), "fx_resize_bilinear_u8")
            d.record_stream(torch.cuda.current_stream(device))
        return out, []

    # ---- fai_detr/processor.py:153-217
    def postprocess(self, output: DETRModelOutput, inputs: ImageInput, class_names: Sequence[str] = (), top_k: Optional[int] = None,
                    threshold: Optional[float] = None) -> List[FocoosDetections]:
        top_k = top_k or self.top_k
        threshold = threshold or self.threshold
        image_sizes = self.get_image_sizes(inputs)
        B, Q, K = output.logits.shape
        assert len(image_sizes) == B, f"Expected image sizes {len(image_sizes)} to match batch size {B}"
        dev = output.logits.device
        lib = _lib.load()
        probs = output.logits.contiguous()
        boxes = output.boxes.contiguous()
        tk = min(top_k, Q * K)
        val = torch.empty(B, tk, dtype=torch.float32, device=dev)
        idx = torch.empty(B, tk, dtype=torch.int32, device=dev)
        labels = torch.empty(B, tk, dtype=torch.int32, device=dev)
        queries = torch.empty(B, tk, dtype=torch.int32, device=dev)
        obox = torch.empty(B, tk, 4, dtype=torch.int32, device=dev)
        count = torch.empty(B, dtype=torch.int32, device=dev)
        sizes = torch.tensor(image_sizes, dtype=torch.int32).to(dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        check(lib.fx_topk_rows_f32(probs.data_ptr(), Q * K, B, Q * K, tk, val.data_ptr(), idx.data_ptr(), stream), "fx_topk_rows_f32")
        check(lib.fx_detr_postprocess(val.data_ptr(), idx.data_ptr(), boxes.data_ptr(), sizes.data_ptr(), B, Q, K, tk, float(threshold),
                                      labels.data_ptr(), queries.data_ptr(), obox.data_ptr(), count.data_ptr(), stream), "fx_detr_postprocess")
        return self.pack_detections(val, labels, obox, count, class_names)

    @staticmethod
    def pack_detections(scores, labels, boxes, count, class_names: Sequence[str] = ()) -> List[FocoosDetections]:
        """One D2H copy of the packed device results, then Python object creation (processor.py:199-217)."""
        n = count.cpu().tolist()
        s, l, b = scores.cpu().tolist(), labels.cpu().tolist(), boxes.cpu().tolist()
        res = []
        for i, ni in enumerate(n):
            res.append(FocoosDetections(detections=[
                FocoosDet(bbox=b[i][j], conf=s[i][j], cls_id=l[i][j], label=class_names[l[i][j]] if class_names else None)
                for j in range(ni)]))
        return res
