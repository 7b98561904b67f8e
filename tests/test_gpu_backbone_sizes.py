"""The reference's backbone test (tests/test_backbone.py::test_backbone_forward / ::test_output_shapes: every backbone type at 1x224^2, 1x384^2
and 2x224^2, the outputs a dict of res2..res5 with the batch preserved and the advertised strides / channels) mirrored for the two backbone
families on this path - ResNet-vd (depth 50 and 101; focoos/nn/backbone/resnet.py:252-266) and STDC (layers [2,2,2] and [4,5,3];
focoos/nn/backbone/stdc.py:313-320) - with what the reference's test cannot check: the VALUES of the four feature maps against the fp32 oracle
(pinned live to the reference: tests/test_oracle_vs_reference.py).  The backbones run inside their engines (there is no stand-alone backbone
entry point in the C ABI), so each case is a whole forward pass whose res2..res5 buffers are compared."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from focoos_amd.engine_bf import BfEngine  # noqa: E402
from focoos_amd.engine_mf import MfEngine  # noqa: E402
from focoos_amd.registry import ModelRegistry  # noqa: E402
from focoos_amd.synth import synth_image_structured, synth_state_dict  # noqa: E402
from oracle import bf_oracle as BF  # noqa: E402
from oracle import mf_oracle as M  # noqa: E402
from oracle.detr_oracle import get_torch_batch  # noqa: E402
from tests.helpers import rel_l2  # noqa: E402

DEV = "cuda:0"
INPUT_SIZES = [(1, 224, 224), (1, 384, 384), (2, 224, 224)]      # the reference's INPUT_SIZES (tests/test_backbone.py:60-64)


def _case(kind):
    if kind.startswith("resnet"):
        cfg = dict(ModelRegistry.get_model_info("fai-mf-l-coco-ins")["config"])
        cfg["backbone_config"] = dict(cfg["backbone_config"], depth=int(kind[6:]))
        return cfg, "fai_mf", MfEngine, (256, 512, 1024, 2048)
    cfg = ModelRegistry.get_model_info("bisenetformer-s-ade" if kind == "stdc222" else "bisenetformer-l-ade")["config"]
    return cfg, "bisenetformer", BfEngine, (64, 256, 512, 1024)


@pytest.mark.parametrize("kind", ["resnet50", "resnet101", "stdc222", "stdc453"])
def test_backbone_forward_sizes_vs_oracle(kind):
    cfg, family, Engine, channels = _case(kind)
    sd = synth_state_dict(cfg, 3, family=family)
    eng = Engine(cfg, sd, device=DEV, full_masks=False)
    mean = torch.tensor(cfg.get("pixel_mean", [123.675, 116.28, 103.53]), dtype=torch.float32).view(-1, 1, 1)
    std = torch.tensor(cfg.get("pixel_std", [58.395, 57.12, 57.375]), dtype=torch.float32).view(-1, 1, 1)
    for B, H, W in INPUT_SIZES:
        images = [synth_image_structured(7 * B + i, H, W) for i in range(B)]
        with torch.no_grad():
            x = (get_torch_batch(images, None) - mean) / std
            want = M.backbone_features(sd, cfg, x) if family == "fai_mf" else BF.stdc(sd, "pixel_decoder.backbone", x, tuple(cfg["backbone_config"]["layers"]))
        pl = eng.forward(torch.from_numpy(np.stack(images)).to(DEV))
        torch.cuda.synchronize()
        for lvl, (name, ch) in enumerate(zip(("res2", "res3", "res4", "res5"), channels)):
            got = pl.bufs[name].torch_view().float().cpu().permute(0, 3, 1, 2)
            stride = 4 << lvl
            assert tuple(got.shape) == (B, ch, H // stride, W // stride) == tuple(want[name].shape), (kind, name, got.shape)
            e = rel_l2(got, want[name])
            assert e <= 2.5e-2, (kind, (B, H, W), name, e)
