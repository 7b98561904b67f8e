"""Drop-in registration into an installed FocoosAI/focoos (the reference) — SURVEY §8b seams B1/B2/B4.

    import focoos_amd.integration as fx; fx.register()
    model = focoos.ModelManager.get("fai-detr-l-obj365")   # unchanged user code
    model.infer(image)                                      # forward now runs on libfocoos_amd.so

``register()`` re-registers ``ModelFamily.DETR`` (and ``ModelFamily.MASKFORMER``) through the reference's own
``ModelManager.register_model`` (focoos/model_manager.py:93-105) with a subclass of the reference's ``FAIDetr`` whose
parameters / ``state_dict()`` are untouched (it still IS the reference module, so checkpoints, EMA and DDP wrapping keep working)
and whose forward - eval AND train - is the HIP engine.  Weights are re-packed lazily whenever the module's parameters changed
(``load_state_dict``, training steps).
``.export()`` (models/focoos_model.py:418-573) needs a graph a tracer can record, and the ctypes engine is opaque to ``torch.jit.trace`` /
``torch.onnx.export`` (its outputs are fresh tensors with no graph edge to the input).  ``ExportableModel`` (focoos_model.py:40-85) deep-copies
the model and calls ``switch_to_export`` on the copy: the adapters override that hook to mark the COPY as an export model, and a marked copy
(or any forward running under a tracer) delegates to the reference's stock module graph - the one legitimate ``super().forward`` of this file
("export() may internally fall back to stock modules").  The live model the user keeps is never marked.
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


def _fx_version_key(module) -> tuple:
    """Changes whenever a parameter / buffer was written in place, replaced, or the module moved to another device."""
    return (str(module.device),) + tuple(p._version for p in module.parameters()) + tuple(b._version for b in module.buffers()) + \
        tuple(p.data_ptr() for p in module.parameters())


def _require_engine_input(images, size_multiple: int = 1) -> None:
    """The adapters NEVER run the reference's stock PyTorch graph (a result produced that way says nothing about the engine):
    inputs the engine has no plan for are refused loudly.  That is (i) H or W below 32 (all three families run at the image's own size
    with the reference's ceil(H/2) arithmetic at every level - tests/test_gpu_odd_sizes.py; RT-DETR since round 5: ragged training batches
    are padded to the batch maximum, not to a multiple of 32, and the reference accepts them) - and (ii) a gradient with respect to the
    input images.  CPU tensors are refused by the engine itself (FocoosAmdError: no CPU path)."""
    from ._lib import FocoosAmdError

    if images.dim() != 4:
        return    # let the engine raise on the malformed input
    h, w = (images.shape[2], images.shape[3]) if (images.shape[1] == 3 and images.shape[-1] != 3) else (images.shape[1], images.shape[2])
    if h % size_multiple or w % size_multiple or h < 32 or w < 32:
        raise FocoosAmdError(f"focoos_amd: input {h}x{w} is not a multiple of {size_multiple} (or smaller than 32); the HIP engine has no plan for it "
                             "and does not fall back to the reference's stock graph (resize or pad the image in the processor, or unregister "
                             "the engine for this model)")
    if torch.is_grad_enabled() and images.requires_grad:
        raise FocoosAmdError("focoos_amd: gradients with respect to the input images are not implemented by the HIP training graph "
                             "(and the adapter does not fall back to the reference's stock graph)")


def share_parameters(engine_graph, reference_module) -> int:
    """Make every parameter and buffer of ``engine_graph`` (train_detr.FAIDetrTrainable: the reference's key names) BE the tensor
    object of the same name in ``reference_module``: gradients of the engine's backward accumulate in the reference module's
    ``.grad`` fields, so the reference's optimizer, EMA hook, checkpointer and DDP wrapper keep working on the module they know.
    Returns the number of tensors shared; raises on a name or shape mismatch."""
    ref_p, ref_b = dict(reference_module.named_parameters()), dict(reference_module.named_buffers())
    n = 0
    for name, p in list(engine_graph.named_parameters()):
        if name not in ref_p or tuple(ref_p[name].shape) != tuple(p.shape):
            raise KeyError(f"engine parameter {name} {tuple(p.shape)} has no counterpart of that shape in the reference module")
        mod = engine_graph.get_submodule(name.rsplit(".", 1)[0]) if "." in name else engine_graph
        mod._parameters[name.rsplit(".", 1)[-1]] = ref_p[name]
        n += 1
    for name, b in list(engine_graph.named_buffers()):
        if name not in ref_b:
            continue   # engine-only scratch buffers
        if tuple(ref_b[name].shape) != tuple(b.shape):
            raise KeyError(f"engine buffer {name} {tuple(b.shape)} != reference {tuple(ref_b[name].shape)}")
        mod = engine_graph.get_submodule(name.rsplit(".", 1)[0]) if "." in name else engine_graph
        mod._buffers[name.rsplit(".", 1)[-1]] = ref_b[name]
        n += 1
    return n


class _FxAdapterState:
    """What the adapters add to the reference module lives in ``__dict__`` and is NOT state: the engine (``_fx_engine``: a ctypes.CDLL, plans with
    function pointers and hipGraph handles), the parameter-version key it was packed from and the HIP training graph that shares this module's
    parameters (``_fx_train``).  ``copy.deepcopy`` (FocoosModel.export models/focoos_model.py:465, EMAState trainer/solver/ema.py:49) and pickling
    into spawned ranks (utils/distributed/dist.py:78-91) therefore drop them; the copy re-builds each lazily on its first forward, from ITS parameters."""

    _FX_TRANSIENT = ("_fx_engine", "_fx_version", "_fx_train")

    def switch_to_export(self, test_cfg=None, device="cuda"):
        """``ExportableModel.__init__`` (models/focoos_model.py:72-74) calls this on its deep copy of the model right before tracing it: from here
        on THIS object is an export model and its forward is the reference's stock graph (see ``_fx_exporting``)."""
        self.__dict__["_fx_export"] = True
        self.__dict__["_fx_engine"] = None       # an export copy never runs the engine: do not keep plans / HBM buffers alive
        self.__dict__["_fx_version"] = None
        self.__dict__.pop("_fx_train", None)
        return super().switch_to_export(test_cfg=test_cfg, device=device)

    def _fx_exporting(self) -> bool:
        """True when the forward must be recordable by a tracer: the module was marked by ``switch_to_export`` (FocoosModel.export's deep
        copy) or a TorchScript / ONNX trace is running."""
        return bool(self.__dict__.get("_fx_export")) or torch.jit.is_tracing() or torch.onnx.is_in_onnx_export()

    def __getstate__(self):
        st = self.__dict__.copy()
        st["_fx_engine"] = None
        st["_fx_version"] = None
        st.pop("_fx_train", None)
        return st


_CLASSES: dict = {}


def _publish(cls):
    """Adapter classes are created against the installed reference at run time; pickle finds a class by ``module.qualname``, so each is given
    this module as its home and served by the module-level ``__getattr__`` below (also in a freshly spawned rank)."""
    cls.__module__ = __name__
    cls.__qualname__ = cls.__name__
    _CLASSES[cls.__name__] = cls
    return cls


_FACTORIES = {"EngineFAIDetr": "make_engine_class", "EngineFAIMaskFormer": "make_mf_engine_class", "EngineBisenetFormer": "make_bf_engine_class"}


def __getattr__(name):
    if name in _FACTORIES:
        return globals()[_FACTORIES[name]]()
    if name in ("EngineDETRProcessor", "EngineMaskFormerProcessor", "EngineBisenetFormerProcessor"):
        make_processor_classes()
        return _CLASSES[name]
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


def _mask_family_forward(self, images, targets, OutputCls, stock_forward):
    """Shared forward of the two mask-family adapters - every branch of FAIMaskFormer.forward (fai_mf/modelling.py:712-725) /
    BisenetFormer.forward (bisenetformer/modelling.py:594-621) on the engine, none on the reference's stock graph:
    train + targets -> losses of the HIP training graph + the last head's probabilities (not upsampled); train without targets -> the same
    forward, loss None; eval -> the inference engine (masks at input resolution); eval + targets -> the engine's outputs AND the losses
    of the training graph in eval mode (frozen statistics; MaskFormerHead.forward computes them whenever targets are given, :603-609)."""
    if self._fx_exporting():
        return stock_forward(images, targets)     # export copy / tracer running: the reference's own graph (module docstring)
    _require_engine_input(images, 1)   # any size >= 32 (ceil-size arithmetic); raises otherwise
    x = images
    if x.dim() == 4 and x.shape[1] == 3 and x.shape[-1] != 3:
        x = x.permute(0, 2, 3, 1)
    x = (x if x.dtype == torch.uint8 else x.float()).contiguous()
    has_targets = targets is not None and len(targets) > 0
    losses = None
    if self.training or has_targets:
        from . import train_nn

        train_nn.WEIGHTS_EPOCH[0] += 1     # an external optimizer may have stepped the parameters since the last forward
        net = self._fx_train_graph()
        if has_targets:
            losses = net(x, targets)
        else:
            net.forward_outputs(x)
        if self.training:
            with torch.no_grad():
                o = net.last_outputs
                logits = torch.softmax(o["pred_logits"].float(), -1)[..., :-1]
                masks = torch.sigmoid(o["pred_masks"])
            return OutputCls(masks=masks, logits=logits, loss=losses)
    self._fx_sync()
    pl = self._fx_engine.forward(x, full_masks=True)
    return OutputCls(masks=pl.masks.clone(), logits=pl.probs.clone(), loss=losses)


def make_engine_class():
    """Build the adapter class against the installed reference (import deferred: focoos is optional)."""
    if "EngineFAIDetr" in _CLASSES:
        return _CLASSES["EngineFAIDetr"]
    from focoos.models.fai_detr.modelling import FAIDetr as RefFAIDetr
    from focoos.models.fai_detr.ports import DETRModelOutput

    from .engine import DetrEngine

    class EngineFAIDetr(_FxAdapterState, RefFAIDetr):
        """Reference FAIDetr with its eval forward replaced by the gfx950 engine."""

        def __init__(self, config):
            super().__init__(config)
            self._fx_engine: Optional[DetrEngine] = None
            self._fx_version = None

        def _fx_sync(self):
            ver = _fx_version_key(self)
            if self._fx_engine is None or str(self._fx_engine.dev) != str(self.device):
                self._fx_engine = DetrEngine(_config_to_dict(self.config), self.state_dict(), str(self.device))
            elif ver != self._fx_version:
                self._fx_engine.load_state_dict(self.state_dict())
            self._fx_version = ver

        def _fx_train_graph(self):
            """The HIP autograd graph (train_detr.FAIDetrTrainable) over THIS module's parameters: built once, re-built when the
            module moved; BatchNorm mode follows the module (all norms frozen -> FrozenBN, SyncBatchNorm present -> SyncBN, else BN)."""
            g = self.__dict__.get("_fx_train")
            if g is None or g[1] != str(self.device):
                from .train_detr import FAIDetrTrainable

                mods = list(self.modules())
                sync = any(isinstance(m, torch.nn.SyncBatchNorm) for m in mods)
                live = any(isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.weight is not None and m.weight.requires_grad for m in mods)
                net = FAIDetrTrainable(_config_to_dict(self.config), norm="SyncBN" if sync else ("BN" if live else "FrozenBN")).to(self.device)
                share_parameters(net, self)
                g = (net, str(self.device))
                self.__dict__["_fx_train"] = g     # not a registered submodule: state_dict() / parameters() stay the reference's
            g[0].train(self.training)
            return g[0]

        def forward(self, images, targets=[]):
            if self._fx_exporting():
                return super().forward(images, targets)   # export copy / tracer running: the reference's own graph (module docstring)
            _require_engine_input(images)   # raises for < 32 / input gradients: no fallback to the reference's own graph
            x = images
            if x.dim() == 4 and x.shape[1] == 3 and x.shape[-1] != 3:
                x = x.permute(0, 2, 3, 1)
            x = (x if x.dtype == torch.uint8 else x.float()).contiguous()
            if self.training:
                # FAIDetr.forward in train mode (modelling.py:1344-1358): empty boxes / logits + the dict of weighted losses, computed by
                # the HIP training graph on this module's own parameters (TrainerLoop.run_step sums the dict and calls backward)
                from . import train_nn

                train_nn.WEIGHTS_EPOCH[0] += 1     # an external optimizer may have stepped the parameters since the last forward
                losses = self._fx_train_graph()(x, targets)
                z = torch.zeros(0, 0, 0, device=images.device)
                return DETRModelOutput(boxes=z, logits=z, loss=losses)
            # eval mode: FAIDetr.forward returns loss=None whether or not targets were passed (modelling.py:1352-1358: the head's losses
            # are dropped), so the outputs are the engine's either way
            self._fx_sync()
            pl = self._fx_engine.forward(x)
            return DETRModelOutput(logits=pl.probs.clone(), boxes=pl.boxes.clone(), loss=None)

    return _publish(EngineFAIDetr)


def make_mf_engine_class():
    """Adapter for the MaskFormer family: reference FAIMaskFormer whose eval forward is the gfx950 engine."""
    if "EngineFAIMaskFormer" in _CLASSES:
        return _CLASSES["EngineFAIMaskFormer"]
    from focoos.models.fai_mf.modelling import FAIMaskFormer as RefFAIMaskFormer
    from focoos.models.fai_mf.ports import MaskFormerModelOutput

    from .engine_mf import MfEngine

    class EngineFAIMaskFormer(_FxAdapterState, RefFAIMaskFormer):
        def __init__(self, config):
            super().__init__(config)
            self._fx_engine: Optional[MfEngine] = None
            self._fx_version = None

        def _fx_sync(self):
            ver = _fx_version_key(self)
            if self._fx_engine is None or str(self._fx_engine.dev) != str(self.device):
                self._fx_engine = MfEngine(_config_to_dict(self.config), self.state_dict(), str(self.device), full_masks=True)
            elif ver != self._fx_version:
                self._fx_engine.load_state_dict(self.state_dict())
            self._fx_version = ver

        def _fx_train_graph(self):
            """train_mf.FAIMaskFormerTrainable over THIS module's parameters and buffers (see EngineBisenetFormer._fx_train_graph)."""
            g = self.__dict__.get("_fx_train")
            if g is None or g[1] != str(self.device):
                from .train_mf import FAIMaskFormerTrainable

                mods = list(self.modules())
                sync = any(isinstance(m, torch.nn.SyncBatchNorm) for m in mods)
                live = any(isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.weight is not None and m.weight.requires_grad for m in mods)
                net = FAIMaskFormerTrainable(_config_to_dict(self.config), norm="SyncBN" if sync else ("BN" if live else "FrozenBN")).to(self.device)
                share_parameters(net, self)
                g = (net, str(self.device))
                self.__dict__["_fx_train"] = g
            g[0].train(self.training)
            return g[0]

        def forward(self, images, targets=[]):
            return _mask_family_forward(self, images, targets, MaskFormerModelOutput, super().forward)

    return _publish(EngineFAIMaskFormer)


def make_bf_engine_class():
    """Adapter for the BiSeNetFormer family: reference BisenetFormer whose eval forward is the gfx950 engine."""
    if "EngineBisenetFormer" in _CLASSES:
        return _CLASSES["EngineBisenetFormer"]
    from focoos.models.bisenetformer.modelling import BisenetFormer as RefBisenetFormer
    from focoos.models.bisenetformer.ports import BisenetFormerOutput

    from .engine_bf import BfEngine

    class EngineBisenetFormer(_FxAdapterState, RefBisenetFormer):
        def __init__(self, config):
            super().__init__(config)
            self._fx_engine: Optional[BfEngine] = None
            self._fx_version = None

        def _fx_sync(self):
            ver = _fx_version_key(self)
            if self._fx_engine is None or str(self._fx_engine.dev) != str(self.device):
                self._fx_engine = BfEngine(_config_to_dict(self.config), self.state_dict(), str(self.device), full_masks=True)
            elif ver != self._fx_version:
                self._fx_engine.load_state_dict(self.state_dict())
            self._fx_version = ver

        def _fx_train_graph(self):
            """The HIP autograd graph (train_bf.BisenetFormerTrainable) over THIS module's parameters and buffers; BatchNorm mode follows
            the module (SyncBatchNorm present -> SyncBN, trainable BatchNorm affine -> BN, else FrozenBN)."""
            g = self.__dict__.get("_fx_train")
            if g is None or g[1] != str(self.device):
                from .train_bf import BisenetFormerTrainable

                mods = list(self.modules())
                sync = any(isinstance(m, torch.nn.SyncBatchNorm) for m in mods)
                live = any(isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.weight is not None and m.weight.requires_grad for m in mods)
                net = BisenetFormerTrainable(_config_to_dict(self.config), norm="SyncBN" if sync else ("BN" if live else "FrozenBN")).to(self.device)
                share_parameters(net, self)
                g = (net, str(self.device))
                self.__dict__["_fx_train"] = g     # not a registered submodule: state_dict() / parameters() stay the reference's
            g[0].train(self.training)
            return g[0]

        def forward(self, images, targets=[]):
            return _mask_family_forward(self, images, targets, BisenetFormerOutput, super().forward)

    return _publish(EngineBisenetFormer)


def make_processor_classes():
    """Engine-backed processors for the reference's ProcessorManager (processor/processor_manager.py:13-18, seam B1/B3): subclasses of
    the reference processors (pre-process, training targets, export paths untouched) whose ``postprocess`` on GPU tensors runs the
    device kernels (fx_topk_rows_f32 + fx_detr_postprocess; fx_mf_postprocess / fx_seg_postprocess) and builds the reference's own
    ``FocoosDetections`` objects from ONE packed device->host copy."""
    if "EngineDETRProcessor" in _CLASSES:
        return _CLASSES["EngineDETRProcessor"], _CLASSES["EngineMaskFormerProcessor"], _CLASSES["EngineBisenetFormerProcessor"]
    from focoos.models.bisenetformer.processor import BisenetFormerProcessor as RefBF
    from focoos.models.fai_detr.processor import DETRProcessor as RefDETR
    from focoos.models.fai_mf.processor import MaskFormerProcessor as RefMF
    from focoos.ports import FocoosDet, FocoosDetections

    from . import processor as fxp

    def convert(dets):
        return [FocoosDetections(detections=[FocoosDet(bbox=d.bbox, conf=d.conf, cls_id=d.cls_id, label=d.label, mask=d.mask) for d in fd.detections])
                for fd in dets]

    def cfg_dict(self):
        return _config_to_dict(self.config)

    class EngineDETRProcessor(RefDETR):
        def postprocess(self, output, inputs, class_names=[], top_k=None, threshold=None):
            if output.logits.device.type != "cuda":
                return super().postprocess(output, inputs, class_names, top_k, threshold)
            mirror = fxp.DETRProcessor({"top_k": self.top_k, "threshold": self.threshold}, self.image_size)
            return convert(mirror.postprocess(output, inputs, class_names, top_k, threshold))

    class EngineMaskFormerProcessor(RefMF):
        def postprocess(self, output, inputs, class_names=[], threshold=None, **kw):
            if output.logits.device.type != "cuda" or kw:
                return super().postprocess(output, inputs, class_names, threshold=threshold, **kw)
            mirror = fxp.MaskFormerProcessor(cfg_dict(self), self.image_size)
            return convert(mirror.postprocess(output, inputs, class_names, threshold=threshold))

    class EngineBisenetFormerProcessor(RefBF):
        def postprocess(self, output, inputs, class_names=[], threshold=None, **kw):
            if output.logits.device.type != "cuda" or kw:
                return super().postprocess(output, inputs, class_names, threshold=threshold, **kw)
            mirror = fxp.BisenetFormerProcessor(cfg_dict(self), self.image_size)
            return convert(mirror.postprocess(output, inputs, class_names, threshold=threshold))

    return _publish(EngineDETRProcessor), _publish(EngineMaskFormerProcessor), _publish(EngineBisenetFormerProcessor)


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
    # processors: ``model.infer()`` of a dropped-in model then post-processes on the device too
    from focoos.processor.processor_manager import ProcessorManager

    p_detr, p_mf, p_bf = make_processor_classes()
    ProcessorManager.register_processor(ModelFamily.DETR, lambda: p_detr)
    ProcessorManager.register_processor(ModelFamily.MASKFORMER, lambda: p_mf)
    ProcessorManager.register_processor(ModelFamily.BISENETFORMER, lambda: p_bf)


def bind_msda_core(module) -> int:
    """Bind every ``MSDeformableAttention.ms_deformable_attn_core`` slot under ``module`` to the HIP kernel (mode 0 =
    the seam's exact signature: value [B,S,M,D], shapes, sampling_locations [B,Q,M,L,P,2], weights [B,Q,M,L,P]): fx_msda_bf16
    without autograd, the differentiable fx_msda_f32_fwd/bwd pair when a gradient is required.  Returns the number of slots bound."""
    from . import _lib

    lib = _lib.load()

    def core(value, value_spatial_shapes, sampling_locations, attention_weights):
        if torch.is_grad_enabled() and (value.requires_grad or sampling_locations.requires_grad or attention_weights.requires_grad):
            # training: the differentiable fp32 pair fx_msda_f32_fwd / fx_msda_f32_bwd behind a torch.autograd.Function (train.py)
            from .train import ms_deform_attn_core

            return ms_deform_attn_core(value.float(), value_spatial_shapes, sampling_locations.float(), attention_weights.float()).to(value.dtype)
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
