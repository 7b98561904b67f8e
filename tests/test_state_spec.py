"""The engine's state-dict layout equals the reference's (key names, order, shapes)."""
import json
import os

import pytest

from focoos_amd.registry import ModelRegistry
from focoos_amd.state_spec import detr_state_spec
from focoos_amd.synth import synth_state_dict


def test_keys_match_reference_dump(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "detr_l_state_keys.json")))
    spec = detr_state_spec(ModelRegistry.get_model_info("fai-detr-l-obj365")["config"])
    assert list(spec) == list(ref)
    assert {k: list(v[0]) for k, v in spec.items()} == ref


def test_synth_is_deterministic_and_complete():
    cfg = ModelRegistry.get_model_info("fai-detr-l-coco")["config"]
    a, b = synth_state_dict(cfg, 3), synth_state_dict(cfg, 3)
    c = synth_state_dict(cfg, 4)
    spec = detr_state_spec(cfg)
    assert list(a) == list(spec)
    k = "head.predictor.enc_score_classifier.weight"
    assert tuple(a[k].shape) == (80, 256)
    assert all((a[n] == b[n]).all() for n in a)
    assert not (a[k] == c[k]).all()
    assert a["pixel_decoder.backbone.conv1.conv1_1.norm.num_batches_tracked"].shape == ()


def test_unsupported_backbone_is_loud():
    cfg = ModelRegistry.get_model_info("fai-detr-l-coco")["config"]
    cfg["backbone_config"]["depth"] = 18
    with pytest.raises(ValueError):
        detr_state_spec(cfg)
