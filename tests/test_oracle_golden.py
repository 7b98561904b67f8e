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
