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
