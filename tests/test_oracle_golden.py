"""Pin oracle/detr_oracle.py (the CPU restatement) against the committed golden fixtures, which
scripts/make_golden.py produced by running the REAL reference in the build container."""
import numpy as np
import pytest
import torch

from focoos_amd.registry import ModelRegistry
from focoos_amd.synth import synth_image, synth_image_structured, synth_state_dict
from oracle import detr_oracle as O
from tests._cases import MSDA_SHAPES, msda_case_inputs
from tests.helpers import load_golden, strided_sample

CASES = {
    "detr_l_obj365_b2": ("fai-detr-l-obj365", lambda: [synth_image(0), synth_image_structured(1)]),
    "detr_l_coco_resize": ("fai-detr-l-coco", lambda: [synth_image_structured(2, 480, 600)]),
}


@pytest.fixture(scope="module", params=sorted(CASES))
def case(request):
    name, mk = CASES[request.param]
    g = load_golden(request.param + ".npz")
    cfg = ModelRegistry.get_model_info(name)["config"]
    sd = synth_state_dict(cfg, int(g["seed"]))
    images = mk()
    torch.set_num_threads(8)
    x = O.get_torch_batch(images, (640, 640))
    col = {}
    with torch.no_grad():
        # teacher-force the reference's query order: torch.topk's order among near-ties (gap ~3e-6)
        # depends on fp summation order; the SET is checked separately below.
        probs, boxes = O.detr_forward(sd, cfg, x, forced_topk=torch.from_numpy(g["enc_topk"]).long(), collect=col)
        col2 = {}
        O.detr_forward(sd, cfg, x[:1], collect=col2)
    return g, cfg, images, x, probs, boxes, col, col2


def test_preprocess(case):
    g, _, _, x, *_ = case
    np.testing.assert_allclose(strided_sample(x, 4096), g["pre_sample"], rtol=0, atol=1e-4)


def test_stages(case):
    g, _, _, _, _, _, col, _ = case
    for k in ("res3", "res4", "res5", "enc_s32", "enc_s16", "enc_s8"):
        ref = g[f"{k}_sample"]
        got = strided_sample(col[k], 4096)
        assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max(), k
    ref = g["memory_sample"]
    assert np.abs(strided_sample(col["memory"], 8192) - ref).max() <= 2e-5 * np.abs(ref).max()
    np.testing.assert_allclose(strided_sample(col["aifi"], 4096), g["aifi_sample"], atol=5e-5)
    np.testing.assert_allclose(strided_sample(col["target"], 4096), g["target_sample"], atol=5e-5)
    np.testing.assert_allclose(col["ref_unact"].numpy(), g["ref_unact"], atol=5e-5)
    for i in range(6):
        np.testing.assert_allclose(strided_sample(col[f"dec{i}_out"], 2048), g[f"dec{i}_sample"], atol=1e-4)


def test_free_running_topk_set(case):
    g, *_, col2 = case
    assert set(col2["topk_ind"][0].tolist()) == set(g["enc_topk"][0].tolist())


def test_outputs(case):
    g, _, _, _, probs, boxes, _, _ = case
    np.testing.assert_allclose(boxes.numpy(), g["boxes"], atol=2e-5)
    np.testing.assert_allclose(probs.max(-1).values.numpy(), g["probs_max"], atol=2e-5)
    assert (probs.argmax(-1).numpy() == g["probs_argmax"]).mean() > 0.999


def test_postprocess(case):
    g, _, images, _, probs, boxes, _, _ = case
    sizes = [im.shape[:2] for im in images]
    res = O.postprocess(probs, boxes, sizes, 300, float(g["threshold"]))
    for i, (s, labels, q, bp) in enumerate(res):
        n = int(g["det_count"][i])
        assert len(s) == n
        np.testing.assert_allclose(s.numpy(), g["det_scores"][i, :n], atol=2e-5)
        # class ids / integer boxes are bit-exact wherever the ranking is not a near-tie
        gap = np.abs(np.diff(g["det_scores"][i, :n], append=0.0))
        safe = np.ones(n, bool)
        safe[:-1] &= gap[:-1] > 1e-4
        safe[1:] &= gap[:-1] > 1e-4
        assert (labels.numpy()[safe] == g["det_labels"][i, :n][safe]).all()
        d = np.abs(bp.numpy()[safe] - g["det_boxes"][i, :n][safe])
        assert d.max() <= 1 and (d > 0).mean() < 0.01  # round() of x.5 +- 1e-5 may flip one pixel


def test_msda_core_golden():
    g = load_golden("msda_core.npz")
    value, loc, w = (torch.from_numpy(a) for a in msda_case_inputs())
    out = O.ms_deform_attn_core(value, MSDA_SHAPES, loc, w)
    np.testing.assert_allclose(out.numpy(), g["out"], atol=1e-5)


REGISTRY_VARIANTS = ["fai-detr-m-coco", "fai-mf-m-coco-ins", "fai-mf-s-coco-ins", "fai-mf-l-ade", "fai-mf-m-ade", "bisenetformer-m-ade", "bisenetformer-s-ade"]


@pytest.mark.parametrize("name", REGISTRY_VARIANTS)
def test_oracle_reproduces_registry_variant_golden(name):
    """tests/golden/registry_variants.npz (scripts/make_golden.py::variants_case: the REAL reference on the round-4 registry variants): the
    restatements, teacher-forced with the reference's top-k / attention masks, reproduce its class probabilities, boxes and mask logits
    to the fixture's fp16 storage - the link that lets the GPU box (no /root/reference there) trust the oracle on these models."""
    g = load_golden("registry_variants.npz")
    key = name.replace("-", "_")
    h, w = (int(v) for v in g["hw"])
    images = [synth_image_structured(40 + i, h, w) for i in range(2)]
    info = ModelRegistry.get_model_info(name)
    cfg, fam = info["config"], info["model_family"]
    sd = synth_state_dict(cfg, int(g["seed"]), family=fam)
    torch.set_num_threads(8)
    if fam == "fai_detr":
        x = O.get_torch_batch([np.ascontiguousarray(im[:h, :h]) for im in images], None)
        with torch.no_grad():
            probs, boxes = O.detr_forward(sd, dict(cfg, resolution=h), x, forced_topk=torch.from_numpy(g[f"{key}.enc_topk"]).long())
        np.testing.assert_allclose(probs.numpy(), g[f"{key}.probs"].astype(np.float32), atol=1e-3)
        np.testing.assert_allclose(boxes.numpy(), g[f"{key}.boxes"], atol=1e-4)
        return
    n = int(g[f"{key}.n_masks"])
    forced = [torch.from_numpy(np.unpackbits(g[f"{key}.attn_mask{i}"], axis=-1)[..., : int(g[f"{key}.attn_mask{i}_len"])].astype(bool)) for i in range(n)]
    x = O.get_torch_batch(images, None)
    col = {}
    with torch.no_grad():
        if fam == "fai_mf":
            from oracle import mf_oracle as M

            probs, _ = M.mf_forward(sd, cfg, x, forced_attn=forced, collect=col, upsample=False)
        else:
            from oracle import bf_oracle as BF

            probs, _ = BF.bf_forward(sd, cfg, x, forced_attn=forced, collect=col, upsample=False)
    np.testing.assert_allclose(probs.numpy(), g[f"{key}.probs"].astype(np.float32), atol=1e-3)
    ml = col["mask_logits"][..., ::2, ::2].numpy()
    ref = g[f"{key}.mask_logits"].astype(np.float32)
    assert np.abs(ml - ref).max() <= 2e-3 * max(1.0, np.abs(ref).max())       # fp16 storage of the fixture
