"""DETRProcessor / MaskFormerProcessor for the engine — same interface and semantics as
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
import torch.nn.functional as F

from . import _lib
from ._lib import check
from .ports import DatasetEntry, DETRModelOutput, DynamicAxes, FocoosDet, FocoosDetections, MaskFormerModelOutput  # noqa: F401

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
        lst = inputs if isinstance(inputs, list) else [inputs]
        if len(lst) > 0 and isinstance(lst[0], DatasetEntry):
            return self._preprocess_entries(lst, device)
        if self.training:
            raise ValueError("During training, inputs should be a list of DetectionDatasetDict")
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
        if self._target_size() is None and any(tuple(a.shape[:2]) != tuple(arrs[0].shape[:2]) for a in arrs):
            raise ValueError("images of different sizes need an image_size to resize to (the reference fails in torch.stack here)")
        tgt = self._target_size() or tuple(arrs[0].shape[:2])
        all_u8_same = all(a.dtype == torch.uint8 and tuple(a.shape[:2]) == tgt for a in arrs)
        if all_u8_same:
            batch = torch.stack(arrs, 0).to(device, non_blocking=True)
            return batch, []
        if all(a.dtype != torch.uint8 and tuple(a.shape[:2]) == tgt for a in arrs):   # float images already at the target size
            return torch.stack([a.to(torch.float32) for a in arrs], 0).to(device, non_blocking=True), []
        lib = _lib.load("bf16")   # inference surface: the bf16 product library whatever a training step pinned
        out = torch.empty(len(arrs), tgt[0], tgt[1], 3, dtype=torch.float32, device=device)
        stream = torch.cuda.current_stream(device).cuda_stream
        for i, a in enumerate(arrs):
            if a.dtype != torch.uint8:
                raise ValueError("engine preprocess resizes uint8 images; float inputs must already have the target size")
            d = a.to(device, non_blocking=True)
            check(lib.fx_resize_bilinear_u8(d.data_ptr(), a.shape[0], a.shape[1], out[i].data_ptr(), tgt[0], tgt[1], stream), "fx_resize_bilinear_u8")
            d.record_stream(torch.cuda.current_stream(device))
        return out, []

    def _preprocess_entries(self, entries, device: torch.device):
        """fai_detr/processor.py:81-101: a list of DatasetEntry -> (uint8 NHWC batch, [DETRTargets]): images of different sizes zero-padded to the
        batch's largest height / width like ImageList.from_tensors (_stack_entry_images), ground-truth boxes absolute xyxy -> cxcywh normalised
        by the PADDED size (:91-96), classes as they are.  Targets only in training mode."""
        from .ports import DETRTargets

        batch = _stack_entry_images(entries).to(device, non_blocking=True)
        if batch.dtype != torch.uint8:
            batch = batch.to(torch.float32)
        targets = []
        if self.training:
            h, w = batch.shape[1:3]
            scale = torch.tensor([w, h, w, h], dtype=torch.float32, device=device)
            for e in entries:
                inst = e.instances
                assert inst is not None and inst.has("boxes") and inst.has("classes"), "boxes and classes are required for training"
                bx = inst.boxes.tensor.to(device, torch.float32) / scale
                cxcywh = torch.stack([(bx[:, 0] + bx[:, 2]) / 2, (bx[:, 1] + bx[:, 3]) / 2, bx[:, 2] - bx[:, 0], bx[:, 3] - bx[:, 1]], -1)
                targets.append(DETRTargets(labels=inst.classes.to(device), boxes=cxcywh))
        return batch, targets

    # ---- fai_detr/processor.py:153-217
    def postprocess(self, output: DETRModelOutput, inputs: ImageInput, class_names: Sequence[str] = (), top_k: Optional[int] = None,
                    threshold: Optional[float] = None) -> List[FocoosDetections]:
        top_k = top_k or self.top_k
        threshold = threshold or self.threshold
        image_sizes = self.get_image_sizes(inputs)
        B, Q, K = output.logits.shape
        assert len(image_sizes) == B, f"Expected image sizes {len(image_sizes)} to match batch size {B}"
        dev = output.logits.device
        lib = _lib.load("bf16")   # inference surface: the bf16 product library whatever a training step pinned
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
        ws = torch.empty(max(int(lib.fx_topk_rows_workspace_bytes(B, Q * K, tk)), 8), dtype=torch.uint8, device=probs.device)
        check(lib.fx_topk_rows_ws_f32(probs.data_ptr(), Q * K, B, Q * K, tk, val.data_ptr(), idx.data_ptr(), ws.data_ptr(), ws.numel(), stream), "fx_topk_rows_ws_f32")
        check(lib.fx_detr_postprocess(val.data_ptr(), idx.data_ptr(), boxes.data_ptr(), sizes.data_ptr(), B, Q, K, tk, float(threshold),
                                      labels.data_ptr(), queries.data_ptr(), obox.data_ptr(), count.data_ptr(), stream), "fx_detr_postprocess")
        return self.pack_detections(val, labels, obox, count, class_names)

    # ---- fai_detr/processor.py:19-57,121-151 (trainer-side evaluation; seam B3)
    def eval_postprocess(self, output: DETRModelOutput, batched_inputs: Sequence, top_k: Optional[int] = None):
        """Per image: top-k over the flattened [Q*K] scores (fx_topk_rows_f32, exact torch.topk order), label = idx % K, box of
        query idx // K, then detector_postprocess: scale the normalised boxes to the entry's (height, width), clip, drop empty
        boxes.  ``batched_inputs[i]`` has ``.height`` / ``.width`` (DatasetEntry) or is a dict with those keys.
        Returns ``[{"instances": Instances(image_size, boxes=Boxes, scores, classes)}, ...]``."""
        from .ports import Boxes, Instances

        top_k = top_k or self.top_k
        probs, boxes = output.logits.contiguous().float(), output.boxes.contiguous().float()
        B, Q, K = probs.shape
        assert len(batched_inputs) == B, (len(batched_inputs), B)
        dev = probs.device
        if dev.type != "cuda":
            raise _lib.FocoosAmdError("eval_postprocess runs the top-k on the GPU (fx_topk_rows_f32); no CPU fallback exists")
        lib = _lib.load("bf16")   # inference surface: the bf16 product library whatever a training step pinned
        tk = min(top_k, Q * K)
        val = torch.empty(B, tk, dtype=torch.float32, device=dev)
        idx = torch.empty(B, tk, dtype=torch.int32, device=dev)
        ws = torch.empty(max(int(lib.fx_topk_rows_workspace_bytes(B, Q * K, tk)), 8), dtype=torch.uint8, device=dev)
        check(lib.fx_topk_rows_ws_f32(probs.data_ptr(), Q * K, B, Q * K, tk, val.data_ptr(), idx.data_ptr(), ws.data_ptr(), ws.numel(),
                                      torch.cuda.current_stream(dev).cuda_stream), "fx_topk_rows_ws_f32")
        idx = idx.long()
        labels, queries = idx % K, idx // K
        results = []
        for i in range(B):
            ent = batched_inputs[i]
            h = (ent.get("height") if isinstance(ent, dict) else getattr(ent, "height", None)) or 1
            w = (ent.get("width") if isinstance(ent, dict) else getattr(ent, "width", None)) or 1
            bx = Boxes(boxes[i][queries[i]].clone())
            bx.scale(w / 1.0, h / 1.0)            # detector_postprocess: scale = output size / results.image_size, image_size = (1, 1)
            bx.clip((h, w))
            inst = Instances((h, w), boxes=bx, scores=val[i], classes=labels[i])
            results.append({"instances": inst[bx.nonempty()]})
        return results

    def _device(self) -> torch.device:
        if not torch.cuda.is_available():
            raise _lib.FocoosAmdError("focoos_amd needs a ROCm GPU (gfx950); no CPU fallback exists")
        return torch.device("cuda", torch.cuda.current_device())

    def export_postprocess(self, output, inputs: ImageInput, class_names: Sequence[str] = (), top_k: Optional[int] = None,
                           threshold: float = 0.5) -> List[FocoosDetections]:
        """fai_detr/processor.py:219-240: raw runtime outputs ``[boxes, logits]`` (tensors or ndarrays, any device) -> detections."""
        boxes, logits = output[0], output[1]
        if isinstance(boxes, np.ndarray):
            boxes = torch.from_numpy(boxes)
        if isinstance(logits, np.ndarray):
            logits = torch.from_numpy(logits)
        dev = self._device()
        model_output = DETRModelOutput(boxes=boxes.to(dev, torch.float32), logits=logits.to(dev, torch.float32), loss=None)
        return self.postprocess(model_output, inputs, class_names, 300 if top_k is None else top_k, threshold)

    def get_dynamic_axes(self) -> DynamicAxes:
        """fai_detr/processor.py:242-251."""
        return DynamicAxes(input_names=["images"], output_names=["boxes", "logits"],
                           dynamic_axes={"images": {0: "batch", 2: "height", 3: "width"}, "boxes": {0: "batch"}, "logits": {0: "batch"}})

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


# ------------------------------------------------------------------------------------------------ MaskFormer
def _stack_entry_images(entries) -> torch.Tensor:
    """The images of a DatasetEntry batch as ONE channels-last tensor [B, Hmax, Wmax, 3]: every image in the top-left corner of a zero
    canvas of the batch's largest height and width - ImageList.from_tensors with its defaults (focoos/structures.py:730-803: pad_value 0,
    no size divisibility), which is what both families' processors call for entry lists (fai_detr/processor.py:82-86,
    fai_mf/processor.py:66-70).  The pad is applied to the raw 0..255 pixels (normalisation happens inside the model), so a zero is a
    black pixel here as there.  Accepts CHW / HWC tensors and HWC ndarrays; equal sizes stack without a copy of the canvas."""
    imgs = []
    for e in entries:
        im = e.image
        if isinstance(im, np.ndarray):
            im = torch.from_numpy(np.ascontiguousarray(im))
        if im.dim() == 3 and im.shape[0] == 3 and im.shape[-1] != 3:
            im = im.permute(1, 2, 0)
        imgs.append(im.contiguous())
    if any(i.dtype != imgs[0].dtype or i.shape[-1] != imgs[0].shape[-1] for i in imgs):
        raise ValueError("the images of a batch must share dtype and channel count")
    if all(tuple(i.shape) == tuple(imgs[0].shape) for i in imgs):
        return torch.stack(imgs, 0)
    H, W = max(i.shape[0] for i in imgs), max(i.shape[1] for i in imgs)
    batch = imgs[0].new_zeros((len(imgs), H, W, imgs[0].shape[-1]))
    for k, im in enumerate(imgs):
        batch[k, : im.shape[0], : im.shape[1]] = im
    return batch


def trim_mask(mask: np.ndarray, bbox) -> np.ndarray:
    """focoos/utils/vision.py:264-267 (note: the inclusive box is used as an exclusive slice end, as in the reference)."""
    x1, y1, x2, y2 = map(int, bbox)
    y2, x2 = min(y2, mask.shape[0]), min(x2, mask.shape[1])
    return mask[y1:y2, x1:x2]


def binary_mask_to_base64(binary_mask: np.ndarray) -> str:
    """focoos/utils/vision.py:270-293: 8-bit grayscale PNG (0 / 255) of the mask, base64-encoded.

    The reference calls ``cv2.imencode(".png", mask)``; cv2 is not a dependency of this package, so the PNG is written here with the encoder
    settings OpenCV's PngEncoder uses when no parameter is given (modules/imgcodecs/src/grfmt_png.cpp: filter SUB on every row, zlib level
    Z_BEST_SPEED, strategy Z_RLE, no ancillary chunks) and libpng's framing (IDAT chunks of 8 192 bytes, the zlib header's window size
    reduced to the smallest that covers the image - libpng's optimize_cmf).  GUARANTEE: the string decodes to exactly the reference's image
    (tests/test_host_cpu.py decodes it with PIL).  Byte-equality with cv2's output is what these settings aim at but it cannot be pinned
    here (no cv2 in this image, and the reference's own test builds its expectation with cv2 at run time, tests/utils/conftest.py:18-26): it
    additionally depends on the zlib build inside the cv2 wheel.  Where the reference is installed (integration mode) its own function is used."""
    import base64
    import struct
    import zlib

    img = (np.asarray(binary_mask) * 255).astype(np.uint8)
    if img.ndim != 2:
        raise ValueError("binary_mask must be 2-D")
    h, w = img.shape
    # filter type 1 (SUB, bytes per pixel = 1): byte - left neighbour (mod 256), the first byte of a row as it is
    sub = img.copy()
    if w > 1:
        sub[:, 1:] = img[:, 1:] - img[:, :-1]
    raw = np.concatenate([np.ones((h, 1), dtype=np.uint8), sub], axis=1).tobytes()
    co = zlib.compressobj(1, zlib.DEFLATED, 15, 8, zlib.Z_RLE)
    z = bytearray(co.compress(raw) + co.flush())
    # libpng's optimize_cmf: claim the smallest window >= the uncompressed size in the zlib header (the deflate stream itself is unchanged)
    if len(z) >= 2 and (z[0] & 0x0F) == 8 and len(raw) <= 16384:
        cinfo, half = z[0] >> 4, 1 << ((z[0] >> 4) + 7)
        while len(raw) <= half and cinfo > 0:
            cinfo -= 1
            half >>= 1
        z[0] = (cinfo << 4) | 8
        z[1] &= 0xE0
        z[1] += 0x1F - ((z[0] << 8) + z[1]) % 0x1F
    z = bytes(z)

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    idat = b"".join(chunk(b"IDAT", z[i:i + 8192]) for i in range(0, max(len(z), 1), 8192))
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) + idat + chunk(b"IEND", b"")
    return base64.b64encode(png).decode("utf-8")


class MaskFormerProcessor(DETRProcessor):
    """Mirror of focoos/models/fai_mf/processor.py:34-306 (threshold branch and the predict_all_pixels branch).  ``preprocess`` does not resize
    (processor.py:96: "we are not using image_size input"); ``postprocess`` runs the reference's mask thresholding,
    empty-mask filter, mask score and score filter on the GPU (``fx_mf_postprocess``) and only the PNG/base64 packaging on
    the host.  The reference's gather-based filtering only works for batch 1 (index tensors are [1, n]); here every image
    of the batch gets the batch-1 behaviour."""

    _postprocessing_types = ("instance", "semantic")   # fai-mf-l-coco-ins / fai-mf-l-ade (predict_all_pixels): one post-process, two branches
    _export_output_names = ("masks", "logits")

    def __init__(self, config: dict, image_size=None):
        super().__init__(config, None)
        if config.get("postprocessing_type", "instance") not in self._postprocessing_types:
            raise NotImplementedError(f"engine {type(self).__name__} covers postprocessing_type in {self._postprocessing_types}")
        self.num_classes = int(config["num_classes"])
        self.mask_threshold = float(config.get("mask_threshold", 0.5))
        self.top_k = int(config.get("top_k", 100))
        self.threshold = float(config.get("threshold", 0.5))
        self.use_mask_score = bool(config.get("use_mask_score", False))
        self.predict_all_pixels = bool(config.get("predict_all_pixels", False))

    # ---- trainer-side evaluation (fai_mf/processor.py:99-166 == bisenetformer/processor.py:95-157): plain tensor arithmetic on the model
    # output, on whatever device it lives - the reference's own code path here is PyTorch too
    def semantic_inference(self, mask_cls: torch.Tensor, mask_pred: torch.Tensor) -> torch.Tensor:
        """:99-105 - per-class score maps [K,H,W] = sum_q class probability x mask probability."""
        return torch.einsum("qc,qhw->chw", mask_cls, mask_pred)

    def instance_inference(self, mask_cls: torch.Tensor, mask_pred: torch.Tensor):
        """:107-144 - the top_k (query, class) pairs of the [Q,K] scores; score = class score x mean mask probability inside the
        thresholded mask (with the reference's 1e-3 scaling of the binary mask and its + 1e-6); tight boxes of the masks."""
        from .ports import BitMasks, Instances

        image_size = tuple(mask_pred.shape[-2:])
        nq = mask_pred.shape[0]
        labels = torch.arange(self.num_classes, device=mask_cls.device).unsqueeze(0).repeat(nq, 1).flatten(0, 1)
        scores_per_image, topk_indices = mask_cls.flatten(0, 1).topk(self.top_k, sorted=False)
        labels_per_image = labels[topk_indices]
        mask_pred = mask_pred[topk_indices // self.num_classes]
        bin_masks = (mask_pred > self.mask_threshold) * 1e-3
        mask_scores = (bin_masks.flatten(1) * mask_pred.flatten(1)).sum(1) / (bin_masks.flatten(1).sum(1) + 1e-6)
        masks = BitMasks(bin_masks.float())
        return Instances(image_size, boxes=masks.get_bounding_boxes(), masks=masks, scores=scores_per_image * mask_scores, classes=labels_per_image)

    def eval_postprocess(self, output, batched_inputs: Sequence, top_k: Optional[int] = None):
        """:146-166 (ABC: base_processor.py:161): per image, crop the predicted masks to the un-padded extent, resize them bilinearly to
        the original (height, width) and run the configured inference ("instance" -> {"instances": Instances}, "semantic" -> {"sem_seg":
        [K,H,W]}).  ``output``: MaskFormerModelOutput / BisenetFormerOutput (``logits`` [B,Q,K], ``masks`` [B,Q,h,w])."""
        semantic = self.config.get("postprocessing_type", "instance") == "semantic"
        fn = self.semantic_inference if semantic else self.instance_inference
        results = []
        for i, entry in enumerate(batched_inputs):
            size = tuple(entry.image.shape[-2:])
            m = output.masks[i]
            stride = size[1] // m.shape[2]
            m = m[:, : 1 + size[0] // stride, : 1 + size[1] // stride]
            m = F.interpolate(m.unsqueeze(0), size=(int(entry.height), int(entry.width)), mode="bilinear", align_corners=False)[0]
            results.append({("sem_seg" if semantic else "instances"): fn(output.logits[i], m)})
        return results

    def preprocess(self, inputs: ImageInput, device: torch.device, dtype: torch.dtype = torch.float32):
        """fai_mf/processor.py:60-97 (inference branch): images are batched at their own size; mixed sizes cannot be stacked
        (base_processor.py:294 torch.stack) and are rejected here as well."""
        lst = inputs if isinstance(inputs, list) else [inputs]
        if len(lst) > 0 and isinstance(lst[0], DatasetEntry):
            return self._preprocess_mask_entries(lst, device)
        sizes = set(self.get_image_sizes(lst))
        if len(sizes) > 1:
            raise ValueError(f"MaskFormerProcessor does not resize: all images of a batch must share one size, got {sorted(sizes)}")
        return super().preprocess(inputs, device, dtype)

    def _preprocess_mask_entries(self, entries, device: torch.device):
        """fai_mf/processor.py:60-90 == bisenetformer/processor.py:60-86: a list of DatasetEntry -> (uint8 NHWC batch, [MaskFormerTargets]):
        per image the classes and the ground-truth masks (``instances.masks``: a [T,h,w] tensor or an object with ``.tensor``) padded to
        the batch's (h, w) - images of different sizes are zero-padded to the largest height / width like ImageList.from_tensors
        (_stack_entry_images).  Targets only in training mode."""
        from .ports import MaskFormerTargets

        batch = _stack_entry_images(entries).to(device, non_blocking=True)
        if batch.dtype != torch.uint8:
            batch = batch.to(torch.float32)
        targets = []
        if self.training:
            h, w = batch.shape[1:3]
            for e in entries:
                inst = e.instances
                assert inst is not None and inst.has("masks"), "masks are required for training"
                assert inst.has("classes"), "classes are required for training"
                gm = inst.masks.tensor if hasattr(inst.masks, "tensor") else inst.masks
                gm = torch.as_tensor(gm).to(device)
                if len(gm) > 0 and tuple(gm.shape[1:]) != (h, w):
                    pad = torch.zeros(gm.shape[0], h, w, dtype=gm.dtype, device=device)
                    pad[:, : gm.shape[1], : gm.shape[2]] = gm
                    gm = pad
                targets.append(MaskFormerTargets(labels=torch.as_tensor(inst.classes).to(device), masks=gm))
        return batch, targets

    def postprocess(self, output, inputs: ImageInput, class_names: Sequence[str] = (), top_k: Optional[int] = None,
                    threshold: Optional[float] = None, use_mask_score: Optional[bool] = None,
                    predict_all_pixels: Optional[bool] = None) -> List[FocoosDetections]:
        threshold = threshold or self.threshold
        use_mask_score = use_mask_score or self.use_mask_score
        predict_all_pixels = predict_all_pixels or self.predict_all_pixels
        image_sizes = self.get_image_sizes(inputs)
        masks = output.masks.contiguous()
        probs = output.logits.contiguous()
        B, Q, H, W = masks.shape
        assert len(image_sizes) == B, f"Expected image sizes {len(image_sizes)} to match batch size {B}"
        if any(tuple(sz) != (H, W) for sz in image_sizes):
            raise NotImplementedError("mask resize to a different original size is not on the engine path (the processor never resizes)")
        dev = masks.device
        lib = _lib.load("bf16")   # inference surface: the bf16 product library whatever a training step pinned
        score, label = probs.max(-1)  # processor.py:212
        score, label = score.contiguous(), label.to(torch.int32).contiguous()
        res = _MfDeviceResults(B, Q, H, W, dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        if predict_all_pixels:  # every pixel goes to the query maximising score x probability (processor.py:215-229)
            nb = lib.fx_seg_postprocess_workspace_bytes(B, Q, H, W, H, W)
            ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
            check(lib.fx_seg_postprocess(masks.data_ptr(), H, W, H, W, score.data_ptr(), label.data_ptr(), B, Q, float(threshold),
                                         int(bool(use_mask_score)), ws.data_ptr(), ws.numel(), res.det_count.data_ptr(), res.det_query.data_ptr(),
                                         res.det_scores.data_ptr(), res.det_labels.data_ptr(), res.det_boxes.data_ptr(), res.det_area.data_ptr(),
                                         res.mask_words.data_ptr(), None, stream), "fx_seg_postprocess")
            return self.pack_detections(res, class_names)
        nb = lib.fx_mf_postprocess_workspace_bytes(B, Q, H)
        ws = torch.empty(max(nb, 8), dtype=torch.uint8, device=dev)
        # full-resolution probabilities: the kernel's bilinear tap degenerates to the identity (scale 1)
        check(lib.fx_mf_postprocess(masks.data_ptr(), H, W, H, W, score.data_ptr(), label.data_ptr(), B, Q, float(self.mask_threshold),
                                    float(threshold), int(bool(use_mask_score)), ws.data_ptr(), ws.numel(), res.det_count.data_ptr(),
                                    res.det_query.data_ptr(), res.det_scores.data_ptr(), res.det_labels.data_ptr(), res.det_boxes.data_ptr(),
                                    res.det_area.data_ptr(), res.mask_words.data_ptr(), stream), "fx_mf_postprocess")
        return self.pack_detections(res, class_names)

    def export_postprocess(self, output, inputs: ImageInput, class_names: Sequence[str] = (), threshold: Optional[float] = None,
                           **kwargs) -> List[FocoosDetections]:
        """fai_mf/processor.py:308-336 (== bisenetformer/processor.py:312-339): raw runtime outputs ``[masks, logits]`` -> detections."""
        masks, logits = output[0], output[1]
        if isinstance(logits, np.ndarray):
            logits = torch.from_numpy(logits)
        if isinstance(masks, np.ndarray):
            masks = torch.from_numpy(masks)
        dev = self._device()
        model_output = MaskFormerModelOutput(logits=logits.to(dev, torch.float32), masks=masks.to(dev, torch.float32), loss=None)
        return self.postprocess(model_output, inputs, class_names, threshold=threshold, **kwargs)

    def get_dynamic_axes(self) -> DynamicAxes:
        """fai_mf/processor.py:338-345; the BiSeNetFormer processor lists its outputs as ["logits", "masks"] (bisenetformer/processor.py:302-310)."""
        return DynamicAxes(input_names=["images"], output_names=list(self._export_output_names), dynamic_axes={"images": {0: "batch", 2: "height", 3: "width"}})

    @staticmethod
    def unpack_masks(words: torch.Tensor, H: int, W: int) -> np.ndarray:
        """int32 [n, H, ceil(W/32)] bit-packed -> bool [n, H, W]."""
        w = words.cpu().numpy().view(np.uint8)
        return np.unpackbits(w, axis=-1, bitorder="little").reshape(words.shape[0], H, -1)[..., :W].astype(bool)

    def pack_detections(self, res, class_names: Sequence[str] = (), encode_masks: bool = True) -> List[FocoosDetections]:
        """D2H of the packed device results (counts, scores, labels, boxes, bit-packed masks of the kept detections only),
        then the host tail of processor.py:270-303: trim to the box, PNG + base64."""
        n = res.det_count.cpu().tolist()
        s, l, b = res.det_scores.cpu().tolist(), res.det_labels.cpu().tolist(), res.det_boxes.cpu().tolist()
        H, W = res.mask_words.shape[2], int(getattr(res, "mask_width", res.mask_words.shape[3] * 32))
        out = []
        for i, ni in enumerate(n):
            if ni == 0:
                out.append(FocoosDetections(detections=[]))
                continue
            masks = self.unpack_masks(res.mask_words[i, :ni], H, W)
            out.append(FocoosDetections(detections=[
                FocoosDet(bbox=b[i][j], conf=s[i][j], cls_id=l[i][j], label=class_names[l[i][j]] if class_names else None,
                          mask=binary_mask_to_base64(trim_mask(masks[j], b[i][j])) if encode_masks else None)
                for j in range(ni)]))
        return out


class BisenetFormerProcessor(MaskFormerProcessor):
    """Mirror of focoos/models/bisenetformer/processor.py:25-300 - the same processor as MaskFormerProcessor (the reference files
    are line-for-line copies) for the "semantic" / "instance" configurations; ``postprocess`` covers both the threshold branch and
    the predict_all_pixels branch (per-pixel argmax over queries, ``fx_seg_postprocess``).  The trainer-side ``eval_postprocess``
    (semantic_inference einsum / instance_inference, :95-157) is inherited from MaskFormerProcessor (the reference files are copies)."""

    _postprocessing_types = ("semantic", "instance")
    _export_output_names = ("logits", "masks")


class _MfDeviceResults:
    """Output buffers of fx_mf_postprocess (same attribute names as the engine plan)."""

    def __init__(self, B: int, Q: int, H: int, W: int, dev):
        self.mask_width = W
        self.det_count = torch.zeros(B, dtype=torch.int32, device=dev)
        self.det_query = torch.zeros(B, Q, dtype=torch.int32, device=dev)
        self.det_scores = torch.zeros(B, Q, dtype=torch.float32, device=dev)
        self.det_labels = torch.zeros(B, Q, dtype=torch.int32, device=dev)
        self.det_boxes = torch.zeros(B, Q, 4, dtype=torch.int32, device=dev)
        self.det_area = torch.zeros(B, Q, dtype=torch.int32, device=dev)
        self.mask_words = torch.zeros(B, Q, H, (W + 31) // 32, dtype=torch.int32, device=dev)   # rows padded to whole words
