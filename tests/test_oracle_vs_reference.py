"""Live check of the oracle restatement against the reference imported from /root/reference
(build container only; skipped on the GPU box where the reference does not exist)."""
import numpy as np
import pytest
import torch

from oracle import ref_import

pytestmark = pytest.mark.skipif(not ref_import.reference_available(), reason="/root/reference not present")


def test_oracle_matches_reference_small_input():
    from focoos_amd.registry import ModelRegistry
    from focoos_amd.synth import synth_image_structured, synth_state_dict
    from oracle import detr_oracle as O

    cfg = ModelRegistry.get_model_info("fai-detr-l-coco")["config"]
    cfg["resolution"] = 320
    model, proc, _ = ref_import.build_reference_detr(cfg)
    sd = synth_state_dict(cfg, seed=5)
    model.load_state_dict(sd, strict=True)
    imgs = [synth_image_structured(9, 200, 260)]
    x, _ = proc.preprocess(imgs, device=torch.device("cpu"), dtype=torch.float32)
    assert tuple(x.shape) == (1, 3, 320, 320)
    np.testing.assert_allclose(O.get_torch_batch(imgs, (320, 320)).numpy(), x.numpy(), atol=1e-4)
    with torch.no_grad():
        out = model(x)
        probs, boxes = O.detr_forward(sd, cfg, x)
    # rows may be permuted among near-tied queries: compare as multisets through a sort on the boxes
    def canon(p, b):
        key = (b * 1e4).round().to(torch.int64)
        order = np.lexsort(key[0].numpy().T[::-1])
        return p[0][order], b[0][order]
    pr, br = canon(out.logits, out.boxes)
    po, bo = canon(probs, boxes)
    np.testing.assert_allclose(bo.numpy(), br.numpy(), atol=1e-4)
    np.testing.assert_allclose(po.numpy(), pr.numpy(), atol=1e-4)
    dets = proc.postprocess(out, imgs, threshold=0.3)[0].detections
    res = O.postprocess(probs, boxes, [(200, 260)], 300, 0.3)[0]
    assert len(dets) == len(res[0])
    assert sorted(d.cls_id for d in dets) == sorted(res[1].tolist())


def test_reference_state_keys_match_spec():
    from focoos_amd.registry import ModelRegistry
    from focoos_amd.state_spec import detr_state_spec

    cfg = ModelRegistry.get_model_info("fai-detr-l-coco")["config"]
    model, _, _ = ref_import.build_reference_detr(cfg)
    ref = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    mine = {k: tuple(v[0]) for k, v in detr_state_spec(cfg).items()}
    assert list(ref) == list(mine) and ref == mine


def test_mf_oracle_matches_reference_live():
    """MaskFormer (A11/A12): forward + batch-1 postprocess of the restatement vs the real reference, another seed and
    size than the committed golden."""
    from focoos_amd.registry import ModelRegistry
    from focoos_amd.synth import synth_image_structured, synth_state_dict
    from oracle import mf_oracle as M
    import focoos.models.fai_mf.processor as fp

    cfg = ModelRegistry.get_model_info("fai-mf-l-coco-ins")["config"]
    model, proc, _ = ref_import.build_reference_mf(cfg)
    fp.binary_mask_to_base64 = lambda m: ""  # cv2/PNG tail is not installed and outside the path
    sd = synth_state_dict(cfg, seed=7, family="fai_mf")
    model.load_state_dict(sd, strict=True)
    assert list(model.state_dict()) == list(sd)
    imgs = [synth_image_structured(21, 96, 128)]
    x, _ = proc.preprocess(imgs, device=torch.device("cpu"), dtype=torch.float32)
    with torch.no_grad():
        out = model(x)
        probs, masks = M.mf_forward(sd, cfg, x)
    np.testing.assert_allclose(probs.numpy(), out.logits.numpy(), atol=1e-4)
    assert (masks - out.masks).abs().max().item() < 5e-3
    dets = proc.postprocess(out, imgs)[0].detections
    s, l, q, boxes, bm = M.postprocess(probs, masks, [(96, 128)], cfg["mask_threshold"], cfg["threshold"], cfg["use_mask_score"])[0]
    assert len(dets) == len(s)
    np.testing.assert_allclose([d.conf for d in dets], s.numpy(), atol=5e-4)
    assert [d.cls_id for d in dets] == l.tolist()
    assert [list(d.bbox) for d in dets] == boxes.tolist()
