"""The registry variants added in round 4 against the REAL reference's outputs (tests/golden/registry_variants.npz, written by
scripts/make_golden.py::variants_case in the build container from /root/reference): fai-detr-m-coco, fai-mf-{m,s}-coco-ins,
fai-mf-{l,m}-ade, bisenetformer-{m,s}-ade.  The engine is teacher-forced with the reference's discrete choices (encoder top-k / boolean
attention masks) and compared with its class probabilities, boxes and quarter-resolution mask logits; gates = the absolute gates of the
large models, or 2.5x what the reference itself moves by when nothing but its weights are rounded to bf16 (stored in the fixture)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from focoos_amd.engine_bf import BfEngine  # noqa: E402
from focoos_amd.engine_mf import MfEngine  # noqa: E402
from focoos_amd.model import FAIDetr  # noqa: E402
from focoos_amd.registry import ModelRegistry  # noqa: E402
from focoos_amd.synth import synth_image_structured, synth_state_dict  # noqa: E402
from tests.helpers import load_golden  # noqa: E402

DEV = "cuda:0"
MASK_VARIANTS = ["fai-mf-m-coco-ins", "fai-mf-s-coco-ins", "fai-mf-l-ade", "fai-mf-m-ade", "bisenetformer-m-ade", "bisenetformer-s-ade"]


@pytest.fixture(scope="module")
def golden():
    return load_golden("registry_variants.npz")


def _images(g):
    h, w = (int(v) for v in g["hw"])
    return [synth_image_structured(40 + i, h, w) for i in range(2)], h, w


@pytest.mark.parametrize("name", MASK_VARIANTS)
def test_mask_variant_vs_reference_golden(golden, name):
    g, key = golden, name.replace("-", "_")
    images, h, w = _images(g)
    info = ModelRegistry.get_model_info(name)
    cfg, fam = info["config"], info["model_family"]
    sd = synth_state_dict(cfg, int(g["seed"]), family=fam)
    eng = (MfEngine if fam == "fai_mf" else BfEngine)(cfg, sd, device=DEV, full_masks=False)
    n = int(g[f"{key}.n_masks"])
    forced = [torch.from_numpy(np.unpackbits(g[f"{key}.attn_mask{i}"], axis=-1)[..., : int(g[f"{key}.attn_mask{i}_len"])].astype(bool)) for i in range(n)]
    pl = eng.forward(torch.from_numpy(np.stack(images)).to(DEV), forced_attn=forced)
    torch.cuda.synchronize()
    probs_r, probs_w = g[f"{key}.probs"].astype(np.float32), g[f"{key}.probs_w16"].astype(np.float32)
    ml_r, ml_w = g[f"{key}.mask_logits"].astype(np.float32), g[f"{key}.mask_logits_w16"].astype(np.float32)
    sig = lambda a: 1.0 / (1.0 + np.exp(-a))   # noqa: E731
    dp, dp_w = np.abs(pl.probs.cpu().numpy() - probs_r).max(), np.abs(probs_w - probs_r).max()
    mp = pl.mask_probs.cpu().numpy()[..., ::2, ::2]
    assert mp.shape == ml_r.shape
    dm, dm_w = np.abs(mp - sig(ml_r)).mean(), np.abs(sig(ml_w) - sig(ml_r)).mean()
    clear = np.abs(ml_r) > 0.5          # binary agreement where the reference's logit is not within bf16 reach of zero
    ag = ((mp >= 0.5) == (ml_r >= 0))[clear].mean()
    print(f"{name}: |dprob| {dp:.4f} (weights-only {dp_w:.4f}), mean |dmask| {dm:.4f} ({dm_w:.4f}), binary agreement on clear pixels {ag:.4f}")
    assert dp <= max(3e-2, 2.5 * dp_w), (dp, dp_w)
    assert dm <= max(1e-2, 2.5 * dm_w), (dm, dm_w)
    assert ag >= 0.99


def test_detr_m_vs_reference_golden(golden):
    g, key = golden, "fai_detr_m_coco"
    images, h, w = _images(g)
    cfg = dict(ModelRegistry.get_model_info("fai-detr-m-coco")["config"], resolution=h)
    sq = [np.ascontiguousarray(im[:h, :h]) for im in images]          # the fixture's inputs: the h x h corner (the processor resizes to resolution^2)
    model = FAIDetr(cfg, device=DEV, seed=int(g["seed"]))
    model.engine.load_state_dict(synth_state_dict(cfg, int(g["seed"])))
    forced = torch.from_numpy(g[f"{key}.enc_topk"]).long()
    out = model.forward(torch.from_numpy(np.stack(sq)).to(DEV), forced_topk=forced, use_graph=False)
    torch.cuda.synchronize()
    probs_r, probs_w = g[f"{key}.probs"].astype(np.float32), g[f"{key}.probs_w16"].astype(np.float32)
    boxes_r, boxes_w = g[f"{key}.boxes"], g[f"{key}.boxes_w16"]
    dp, dp_w = np.abs(out.logits.cpu().numpy() - probs_r).max(), np.abs(probs_w - probs_r).max()
    db, db_w = np.abs(out.boxes.cpu().numpy() - boxes_r).max(), np.abs(boxes_w - boxes_r).max()
    print(f"fai-detr-m-coco: |dprob| {dp:.4f} (weights-only {dp_w:.4f}), |dbox| {db:.4f} ({db_w:.4f})")
    assert dp <= max(2.5e-2, 2.5 * dp_w) and db <= max(7e-3, 2.5 * db_w), (dp, dp_w, db, db_w)
