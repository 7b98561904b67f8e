"""Drop-in registration into an installed FocoosAI/focoos (the reference) — SURVEY §8b seams B1/B2/B4.

    import focoos_amd.integration as fx; fx.register()
    model = focoos.ModelManager.get("fai-detr-l-obj365")   # unchanged user code
    model.infer(image)                                      # forward now runs on libfocoos_amd.so

``register()`` re-registers ``ModelFamily.DETR`` (and ``ModelFamily.MASKFORMER``) through the reference's own
``ModelManager.register_model`` (focoos/model_manager.py:93-105) with a subclass of the reference's ``FAIDetr`` whose
parameters / ``state_dict()`` / training path are untouched (it still IS the reference module, so checkpoints, EMA, DDP
wrapping, ``.export()`` keep working) and whose **eval-mode forward** is the HIP engine.  Weights are re-packed lazily
whenever the module's parameters changed (``load_state_dict``, training steps).
The function-pointer seam ``MSDeformableAttention.ms_deformable_attn_core`` (fai_detr/modelling.py:806) can be bound to
``fx_msda_bf16`` separately with ``bind_msda_core(model)`` for no-grad use of the stock module graph.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch


def _config_to_dict(cfg) -> dict:
    d = dict(cfg) if isinstance(cfg, dict) else {k: getattr(cfg, k) for k in cfg.__dataclass_fields__}
    bb = d.get("backbone_config")
    if bb is not None and not isinstance(bb, dict):
        d["backbone_config"] = {k: getattr(bb, k) for k in bb.__dataclass_fields__}
    return d


def make_engine_class():
    """Build the adapter class against the installed reference (import deferred: focoos is optional)."""
    from focoos.models.fai_detr.modelling import FAIDetr as RefFAIDetr
    from focoos.models.fai_detr.ports import DETRModelOutput

    from .engine import DetrEngine

    class EngineFAIDetr(RefFAIDetr):
        """Reference FAIDetr with its eval forward replaced by the gfx950 engine."""

        def __init__(self, config):
            super().__init__(config)
            self._fx_engine: Optional[DetrEngine] = None
            self._fx_version = None

        def _fx_sync(self):
            ver = tuple(p._version for p in self.parameters()) + tuple(b._version for b in self.buffers())
            if self._fx_engine is None:
                self._fx_engine = DetrEngine(_config_to_dict(self.config), self.state_dict(), str(self.device))
            elif ver != self._fx_version:
                self._fx_engine.load_state_dict(self.state_dict())
            self._fx_version = ver

        def forward(self, images, targets=[]):
            if self.training or (targets is not None and len(targets) > 0) or torch.is_grad_enabled() and images.requires_grad:
                return super().forward(images, targets)  # training path: the reference's own graph (out of scope here)
            self._fx_sync()
            x = images
            if x.dim() == 4 and x.shape[1] == 3 and x.shape[-1] != 3:
                x = x.permute(0, 2, 3, 1)
            x = (x if x.dtype == torch.uint8 else x.float()).contiguous()
            pl = self._fx_engine.forward(x)
            return DETRModelOutput(logits=pl.probs.clone(), boxes=pl.boxes.clone(), loss=None)

    return EngineFAIDetr


def make_mf_engine_class():
    """Adapter for the MaskFormer family: reference FAIMaskFormer whose eval forward is the gfx950 engine."""
    from focoos.models.fai_mf.modelling import FAIMaskFormer as RefFAIMaskFormer
    from focoos.models.fai_mf.ports import MaskFormerModelOutput

    from .engine_mf import MfEngine

    class EngineFAIMaskFormer(RefFAIMaskFormer):
        def __init__(self, config):
            super().__init__(config)
            self._fx_engine: Optional[MfEngine] = None
            self._fx_version = None

        def _fx_sync(self):
            ver = tuple(p._version for p in self.parameters()) + tuple(b._version for b in self.buffers())
            if self._fx_engine is None:
                self._fx_engine = MfEngine(_config_to_dict(self.config), self.state_dict(), str(self.device), full_masks=True)
            elif ver != self._fx_version:
                self._fx_engine.load_state_dict(self.state_dict())
            self._fx_version = ver

        def forward(self, images, targets=[]):
            if self.training or (targets is not None and len(targets) > 0) or torch.is_grad_enabled() and images.requires_grad:
                return super().forward(images, targets)  # training path: the reference's own graph
            self._fx_sync()
            x = images
            if x.dim() == 4 and x.shape[1] == 3 and x.shape[-1] != 3:
                x = x.permute(0, 2, 3, 1)
            x = (x if x.dtype == torch.uint8 else x.float()).contiguous()
            pl = self._fx_engine.forward(x, full_masks=True)
            return MaskFormerModelOutput(masks=pl.masks.clone(), logits=pl.probs.clone(), loss=None)

    return EngineFAIMaskFormer


def make_bf_engine_class():
    """Adapter for the BiSeNetFormer family: reference BisenetFormer whose eval forward is the gfx950 engine."""
    from focoos.models.bisenetformer.modelling import BisenetFormer as RefBisenetFormer
    from focoos.models.bisenetformer.ports import BisenetFormerOutput

    from .engine_bf import BfEngine

    class EngineBisenetFormer(RefBisenetFormer):
        def __init__(self, config):
            super().__init__(config)
            self._fx_engine: Optional[BfEngine] = None
            self._fx_version = None

        def _fx_sync(self):
            ver = tuple(p._version for p in self.parameters()) + tuple(b._version for b in self.buffers())
            if self._fx_engine is None:
                self._fx_engine = BfEngine(_config_to_dict(self.config), self.state_dict(), str(self.device), full_masks=True)
            elif ver != self._fx_version:
                self._fx_engine.load_state_dict(self.state_dict())
            self._fx_version = ver

        def forward(self, images, targets=[]):
            if self.training or (targets is not None and len(targets) > 0) or torch.is_grad_enabled() and images.requires_grad:
                return super().forward(images, targets)  # training path: the reference's own graph
            self._fx_sync()
            x = images
            if x.dim() == 4 and x.shape[1] == 3 and x.shape[-1] != 3:
                x = x.permute(0, 2, 3, 1)
            x = (x if x.dtype == torch.uint8 else x.float()).contiguous()
            pl = self._fx_engine.forward(x, full_masks=True)
            return BisenetFormerOutput(masks=pl.masks.clone(), logits=pl.probs.clone(), loss=None)

    return EngineBisenetFormer


def register() -> None:
    """Re-register the DETR, MaskFormer and BiSeNetFormer families with the engine-backed model classes (last registration wins,
    model_manager.py:93-105)."""
    from focoos.model_manager import ModelManager
    from focoos.ports import ModelFamily

    import focoos.models.bisenetformer as family_bf
    import focoos.models.fai_detr as family
    import focoos.models.fai_mf as family_mf

    for fam in (family, family_mf, family_bf):
        for name in dir(fam):  # the family's own _register(): config + processor (+ stock model) registries
            if name.startswith("_register") and callable(getattr(fam, name)):
                getattr(fam, name)()
    cls = make_engine_class()
    ModelManager.register_model(ModelFamily.DETR, lambda: cls)
    cls_mf = make_mf_engine_class()
    ModelManager.register_model(ModelFamily.MASKFORMER, lambda: cls_mf)
    cls_bf = make_bf_engine_class()
    ModelManager.register_model(ModelFamily.BISENETFORMER, lambda: cls_bf)


def bind_msda_core(module) -> int:
    """Bind every ``MSDeformableAttention.ms_deformable_attn_core`` slot under ``module`` to the HIP kernel (mode 0 =
    the seam's exact signature: value [B,S,M,D], shapes, sampling_locations [B,Q,M,L,P,2], weights [B,Q,M,L,P]).
    Inference only (no autograd).  Returns the number of slots bound."""
    from . import _lib

    lib = _lib.load()

    def core(value, value_spatial_shapes, sampling_locations, attention_weights):
        if torch.is_grad_enabled() and (value.requires_grad or sampling_locations.requires_grad):
            raise _lib.FocoosAmdError("fx_msda_bf16 is forward-only; use the reference core under autograd")
        B, S, M, D = value.shape
        Q, L, P = sampling_locations.shape[1], sampling_locations.shape[3], sampling_locations.shape[4]
        dev = value.device
        v = value.reshape(B, S, M * D).to(torch.bfloat16).contiguous()
        loc = sampling_locations.float().contiguous()
        aw = attention_weights.float().contiguous()
        shapes = torch.tensor(value_spatial_shapes, dtype=torch.int32, device=dev)
        starts = torch.tensor([0] + list(torch.tensor([h * w for h, w in value_spatial_shapes]).cumsum(0)[:-1]), dtype=torch.int32, device=dev)
        out = torch.empty(B, Q, M * D, dtype=torch.bfloat16, device=dev)
        _lib.check(lib.fx_msda_bf16(v.data_ptr(), M * D, shapes.data_ptr(), starts.data_ptr(), L, P, loc.data_ptr(), M * L * P * 2, aw.data_ptr(),
                                    M * L * P, None, 0, out.data_ptr(), M * D, B, S, Q, M, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                   "fx_msda_bf16")
        return out.to(value.dtype)

    n = 0
    for m in module.modules():
        if hasattr(m, "ms_deformable_attn_core"):
            m.ms_deformable_attn_core = core
            n += 1
    return n
