"""Model registry of the engine: the RT-DETR and MaskFormer-L entries of the reference's registry
(focoos/model_registry/fai-detr-l-{obj365,coco}.json, fai-mf-l-coco-ins.json — config values are the spec
for every shape on the hot path).  Weights URIs are remote in the reference;
offline the engine is constructed with seeded synthetic weights (``synth.py``)
or a local ``model_final.pth`` given by ``weights_uri``."""
from __future__ import annotations

import copy
import json
import os
from typing import Dict, List, Optional

_DETR_L_CONFIG = {
    "num_classes": 365,
    "backbone_config": {
        "use_pretrained": False, "backbone_url": None, "model_type": "resnet", "in_chans": 3, "depth": 50,
        "variant": "d", "freeze_at": -1, "num_stages": 4, "freeze_norm": False, "act": "relu", "pretrained": False,
    },
    "num_queries": 300, "resolution": 640,
    "pixel_mean": [123.675, 116.28, 103.53], "pixel_std": [58.395, 57.12, 57.375], "size_divisibility": 0,
    "pixel_decoder_out_dim": 256, "pixel_decoder_feat_dim": 256, "pixel_decoder_num_encoder_layers": 1,
    "pixel_decoder_expansion": 1.0, "pixel_decoder_dim_feedforward": 1024,
    "transformer_predictor_out_dim": 256, "transformer_predictor_hidden_dim": 256,
    "transformer_predictor_dec_layers": 6, "transformer_predictor_dim_feedforward": 1024,
    "head_out_dim": 256, "pixel_decoder_dropout": 0.0, "pixel_decoder_nhead": 8, "transformer_predictor_nhead": 8,
    "threshold": 0.5, "top_k": 300,
    "criterion_deep_supervision": True, "criterion_eos_coef": 0.1, "criterion_losses": ["vfl", "boxes"], "criterion_num_points": 0,
    "criterion_focal_alpha": 0.75, "criterion_focal_gamma": 2.0, "weight_dict_loss_vfl": 1, "weight_dict_loss_bbox": 5, "weight_dict_loss_giou": 2,
    "matcher_cost_class": 2, "matcher_cost_bbox": 5, "matcher_cost_giou": 2, "matcher_use_focal_loss": True, "matcher_alpha": 0.25, "matcher_gamma": 2.0,
}

# focoos/model_registry/fai-detr-m-coco.json: STDC-2 backbone, 128-channel hybrid encoder WITHOUT the AIFI layer, 3 decoder layers
_DETR_M_COCO_CONFIG = copy.deepcopy(_DETR_L_CONFIG)
_DETR_M_COCO_CONFIG.update({
    "num_classes": 80,
    "backbone_config": {"use_pretrained": False, "backbone_url": None, "model_type": "stdc", "in_chans": 3, "base": 64, "layers": [4, 5, 3],
                        "out_features": ["res2", "res3", "res4", "res5"], "block_num": 4, "block_type": "cat", "use_conv_last": False},
    "pixel_decoder_out_dim": 128, "pixel_decoder_feat_dim": 128, "pixel_decoder_num_encoder_layers": 0,
    "transformer_predictor_out_dim": 128, "transformer_predictor_dec_layers": 3, "head_out_dim": 128,
})


def _entry(name: str, num_classes: int, description: str, base: Optional[Dict] = None) -> Dict:
    cfg = copy.deepcopy(base if base is not None else _DETR_L_CONFIG)
    cfg["num_classes"] = num_classes
    return {
        "name": name, "model_family": "fai_detr", "task": "detection", "im_size": 640,
        "classes": [f"class_{i}" for i in range(num_classes)],  # class names are data of the reference registry
        "config": cfg, "weights_uri": None, "description": description,
    }


_MF_L_COCO_INS_CONFIG = {
    "num_classes": 80,
    "backbone_config": {
        "use_pretrained": False, "backbone_url": None, "model_type": "resnet", "in_chans": 3, "depth": 101,
        "variant": "d", "freeze_at": -1, "num_stages": 4, "freeze_norm": False, "act": "relu", "pretrained": False,
    },
    "num_queries": 100, "resolution": 1024,
    "pixel_mean": [123.675, 116.28, 103.53], "pixel_std": [58.395, 57.12, 57.375], "size_divisibility": 0,
    "pixel_decoder_out_dim": 256, "pixel_decoder_feat_dim": 256, "pixel_decoder_transformer_layers": 6,
    "pixel_decoder_transformer_dropout": 0.0, "pixel_decoder_transformer_nheads": 8,
    "pixel_decoder_transformer_dim_feedforward": 1024,
    "transformer_predictor_out_dim": 256, "transformer_predictor_hidden_dim": 256,
    "transformer_predictor_dec_layers": 9, "transformer_predictor_dim_feedforward": 2048,
    "head_out_dim": 256, "cls_sigmoid": False, "postprocessing_type": "instance", "mask_threshold": 0.5,
    "predict_all_pixels": False, "use_mask_score": True, "threshold": 0.5, "top_k": 100,
    "criterion_deep_supervision": True, "criterion_eos_coef": 0.1, "criterion_num_points": 12544,
    "weight_dict_loss_dice": 5, "weight_dict_loss_mask": 5, "weight_dict_loss_ce": 2,
    "matcher_cost_class": 2, "matcher_cost_mask": 5, "matcher_cost_dice": 5,
}


# focoos/model_registry/bisenetformer-l-ade.json (config block; the loss / matcher weights of the training criterion included)
_BF_L_ADE_CONFIG = {
    "num_classes": 150,
    "backbone_config": {
        "use_pretrained": False, "backbone_url": None, "model_type": "stdc", "in_chans": 3, "base": 64, "layers": [4, 5, 3],
        "out_features": ["res2", "res3", "res4", "res5"], "block_num": 4, "block_type": "cat", "use_conv_last": False,
    },
    "num_queries": 100, "resolution": 640,
    "pixel_mean": [123.675, 116.28, 103.53], "pixel_std": [58.395, 57.12, 57.375], "size_divisibility": 0,
    "pixel_decoder_out_dim": 128, "pixel_decoder_feat_dim": 128,
    "transformer_predictor_out_dim": 128, "transformer_predictor_hidden_dim": 256,
    "transformer_predictor_dec_layers": 6, "transformer_predictor_dim_feedforward": 1024,
    "head_out_dim": 128, "cls_sigmoid": False, "postprocessing_type": "semantic", "top_k": 100, "mask_threshold": 0.5,
    "predict_all_pixels": True, "use_mask_score": False, "threshold": 0.5,
    "criterion_deep_supervision": True, "criterion_eos_coef": 0.1, "criterion_num_points": 12544,
    "weight_dict_loss_dice": 5, "weight_dict_loss_mask": 5, "weight_dict_loss_ce": 2,
    "matcher_cost_class": 2, "matcher_cost_mask": 5, "matcher_cost_dice": 5,
}


# focoos/model_registry/bisenetformer-m-ade.json: STDC-2, 96-channel pixel decoder / mask dimension, 4 decoder layers, 512-wide FFN
_BF_M_ADE_CONFIG = copy.deepcopy(_BF_L_ADE_CONFIG)
_BF_M_ADE_CONFIG.update({"pixel_decoder_out_dim": 96, "pixel_decoder_feat_dim": 96, "transformer_predictor_out_dim": 96, "head_out_dim": 96,
                         "transformer_predictor_dec_layers": 4, "transformer_predictor_dim_feedforward": 512})


# focoos/model_registry/bisenetformer-s-ade.json: the same head on STDC-1 (two blocks per stage)
_BF_S_ADE_CONFIG = copy.deepcopy(_BF_L_ADE_CONFIG)
_BF_S_ADE_CONFIG["backbone_config"]["layers"] = [2, 2, 2]


# focoos/model_registry/fai-mf-l-ade.json: R101-vd, 128-channel FPN without a transformer encoder, 6 decoder layers, semantic post-processing
_MF_L_ADE_CONFIG = copy.deepcopy(_MF_L_COCO_INS_CONFIG)
_MF_L_ADE_CONFIG.update({
    "num_classes": 150, "resolution": 640, "pixel_decoder_out_dim": 128, "pixel_decoder_feat_dim": 128, "pixel_decoder_transformer_layers": 0,
    "transformer_predictor_out_dim": 128, "transformer_predictor_dec_layers": 6, "transformer_predictor_dim_feedforward": 1024, "head_out_dim": 128,
    "postprocessing_type": "semantic", "predict_all_pixels": True, "use_mask_score": False,
    "criterion_deep_supervision": True, "criterion_eos_coef": 0.1, "criterion_num_points": 12544,
    "weight_dict_loss_dice": 5, "weight_dict_loss_mask": 5, "weight_dict_loss_ce": 2,
    "matcher_cost_class": 2, "matcher_cost_mask": 5, "matcher_cost_dice": 5,
})

# focoos/model_registry/fai-mf-{m,s}-coco-ins.json: R101-vd / R50-vd, 128-channel pixel decoder WITH a 3-layer transformer encoder (8 heads of 16
# channels), 6 decoder layers, instance post-processing
_MF_M_COCO_INS_CONFIG = copy.deepcopy(_MF_L_COCO_INS_CONFIG)
_MF_M_COCO_INS_CONFIG.update({
    "pixel_decoder_out_dim": 128, "pixel_decoder_feat_dim": 128, "pixel_decoder_transformer_layers": 3,
    "transformer_predictor_out_dim": 128, "transformer_predictor_dec_layers": 6, "transformer_predictor_dim_feedforward": 1024, "head_out_dim": 128,
    "criterion_deep_supervision": True, "criterion_eos_coef": 0.1, "criterion_num_points": 12544,
    "weight_dict_loss_dice": 5, "weight_dict_loss_mask": 5, "weight_dict_loss_ce": 2,
    "matcher_cost_class": 2, "matcher_cost_mask": 5, "matcher_cost_dice": 5,
})
_MF_S_COCO_INS_CONFIG = copy.deepcopy(_MF_M_COCO_INS_CONFIG)
_MF_S_COCO_INS_CONFIG["backbone_config"] = dict(_MF_S_COCO_INS_CONFIG["backbone_config"], depth=50)

# focoos/model_registry/fai-mf-m-ade.json: STDC-2 backbone (the BiSeNetFormer-L one), 128-channel FPN, 3 decoder layers with a 512-wide FFN
_MF_M_ADE_CONFIG = copy.deepcopy(_MF_L_ADE_CONFIG)
_MF_M_ADE_CONFIG.update({
    "backbone_config": copy.deepcopy(_BF_L_ADE_CONFIG["backbone_config"]),
    "transformer_predictor_dec_layers": 3, "transformer_predictor_dim_feedforward": 512,
})
_MF_M_ADE_CONFIG["backbone_config"]["out_features"] = ["res2", "res3", "res4", "res5"]


def _bf_entry(name: str, cfg: Dict, description: str) -> Dict:
    cfg = copy.deepcopy(cfg)
    return {
        "name": name, "model_family": "bisenetformer", "task": "semseg", "im_size": int(cfg["resolution"]),
        "classes": [f"class_{i}" for i in range(int(cfg["num_classes"]))],
        "config": cfg, "weights_uri": None, "description": description,
    }


def _mf_entry(name: str, cfg: Dict, description: str) -> Dict:
    cfg = copy.deepcopy(cfg)
    return {
        "name": name, "model_family": "fai_mf", "task": "instseg", "im_size": int(cfg["resolution"]),
        "classes": [f"class_{i}" for i in range(int(cfg["num_classes"]))],
        "config": cfg, "weights_uri": None, "description": description,
    }


_REGISTRY = {
    "bisenetformer-l-ade": _bf_entry("bisenetformer-l-ade", _BF_L_ADE_CONFIG, "BiSeNetFormer large (STDC-2), ADE20K semantic segmentation"),
    "bisenetformer-m-ade": _bf_entry("bisenetformer-m-ade", _BF_M_ADE_CONFIG, "BiSeNetFormer medium (STDC-2, 96-channel pixel decoder), ADE20K semantic segmentation"),
    "bisenetformer-s-ade": _bf_entry("bisenetformer-s-ade", _BF_S_ADE_CONFIG, "BiSeNetFormer small (STDC-1), ADE20K semantic segmentation"),
    "fai-mf-l-coco-ins": _mf_entry("fai-mf-l-coco-ins", _MF_L_COCO_INS_CONFIG, "MaskFormer large (R101-vd), COCO instance segmentation"),
    "fai-mf-m-coco-ins": _mf_entry("fai-mf-m-coco-ins", _MF_M_COCO_INS_CONFIG, "MaskFormer medium (R101-vd, 128-channel pixel decoder), COCO instance segmentation"),
    "fai-mf-s-coco-ins": _mf_entry("fai-mf-s-coco-ins", _MF_S_COCO_INS_CONFIG, "MaskFormer small (R50-vd, 128-channel pixel decoder), COCO instance segmentation"),
    "fai-mf-l-ade": dict(_mf_entry("fai-mf-l-ade", _MF_L_ADE_CONFIG, "MaskFormer large (R101-vd), ADE20K semantic segmentation"), task="semseg"),
    "fai-mf-m-ade": dict(_mf_entry("fai-mf-m-ade", _MF_M_ADE_CONFIG, "MaskFormer medium (STDC-2), ADE20K semantic segmentation"), task="semseg"),
    "fai-detr-l-obj365": _entry("fai-detr-l-obj365", 365, "RT-DETR large (R50-vd), Objects365 head"),
    "fai-detr-l-coco": _entry("fai-detr-l-coco", 80, "RT-DETR large (R50-vd), COCO head"),
    "fai-detr-m-coco": _entry("fai-detr-m-coco", 80, "RT-DETR medium (STDC-2, 128-channel encoder, 3 decoder layers), COCO head", _DETR_M_COCO_CONFIG),
}


def _reference_class_names(name: str, n: int) -> Optional[List[str]]:
    """Label names of a registry model: they are DATA of the reference's registry (focoos/model_registry/<name>.json, "classes"),
    not shipped here - where an installed focoos is importable (the integration use) they are read from ITS file (located with find_spec:
    the package is not imported), so ``FocoosDet.label`` and the ``model_info.json`` a training run writes carry the reference's names; without
    it the entries keep ``class_<i>``.  A file whose list has another length than the head is ignored."""
    import importlib.util

    import sys

    roots: List[str] = []
    mod = sys.modules.get("focoos")
    if mod is not None:
        roots = [str(r) for r in getattr(mod, "__path__", [])]
    else:
        try:
            spec = importlib.util.find_spec("focoos")
            roots = list(spec.submodule_search_locations) if spec is not None and spec.submodule_search_locations else []
        except (ImportError, ValueError, AttributeError):
            roots = []
    for root in roots:
        path = os.path.join(root, "model_registry", f"{name}.json")
        if os.path.isfile(path):
            try:
                with open(path, encoding="utf-8") as f:
                    names = json.load(f).get("classes")
            except (OSError, ValueError):
                return None
            return [str(c) for c in names] if isinstance(names, list) and len(names) == n else None
    return None


class ModelRegistry:
    """Mirror of focoos/model_registry/model_registry.py for the engine's entries."""

    @classmethod
    def list_models(cls) -> List[str]:
        return sorted(_REGISTRY)

    @classmethod
    def exists(cls, name: str) -> bool:
        return name in _REGISTRY

    @classmethod
    def get_model_info(cls, name: str) -> Dict:
        """model_registry.py:50-71: a registry name, or the path of a ``model_info.json`` (a fine-tuned model's own description)."""
        if name in _REGISTRY:
            d = copy.deepcopy(_REGISTRY[name])
            names = _reference_class_names(name, len(d["classes"]))
            if names is not None:
                d["classes"] = names
            return d
        if not os.path.exists(name):
            raise ValueError(f"⚠️ Model {name} not found. Available models: {cls.list_models()}")
        with open(name, encoding="utf-8") as f:
            d = json.load(f)
        missing = [k for k in ("name", "model_family", "classes", "im_size", "task", "config") if k not in d]
        if missing:
            raise ValueError(f"⚠️ Model {name}: model info without the required field(s) {missing}")
        d.setdefault("weights_uri", None)
        d.setdefault("description", None)
        return d
