"""Live check of the oracle restatement against the reference imported from /root/reference
(build container only; skipped on the GPU box where the reference does not exist)."""
import numpy as np
import pytest
import torch

from oracle import ref_import

pytestmark = pytest.mark.skipif(not ref_import.reference_available(), reason="/root/reference not present")


@pytest.mark.parametrize("name", ["fai-detr-l-coco", "fai-detr-m-coco"])
def test_oracle_matches_reference_small_input(name):
    """fai-detr-m-coco (focoos/model_registry/fai-detr-m-coco.json): STDC-2 backbone, 128-channel hybrid encoder without the AIFI layer, three
    decoder layers fed through 128 -> 256 input projections; the registry config equals the reference's file."""
    import json
    import os

    from focoos_amd.registry import ModelRegistry
    from focoos_amd.synth import synth_image_structured, synth_state_dict
    from oracle import detr_oracle as O

    cfg = ModelRegistry.get_model_info(name)["config"]
    ref_cfg = json.load(open(os.path.join(ref_import.REFERENCE_ROOT, f"focoos/model_registry/{name}.json")))["config"]
    assert {k: v for k, v in cfg.items() if k in ref_cfg} == ref_cfg
    cfg["resolution"] = 320
    model, proc, _ = ref_import.build_reference_detr(cfg)
    sd = synth_state_dict(cfg, seed=5)
    model.load_state_dict(sd, strict=True)
    assert list(model.state_dict()) == list(sd)
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


@pytest.mark.parametrize("name", ["fai-mf-l-ade", "fai-mf-m-ade", "fai-mf-m-coco-ins", "fai-mf-s-coco-ins", "fai-mf-l-coco-ins"])
def test_mf_ade_variant_oracle_matches_reference_live(name):
    """fai-mf-l-ade (focoos/model_registry/fai-mf-l-ade.json: R101-vd, 128-channel FPN without a transformer encoder, 6 decoder layers,
    semantic post-processing with predict_all_pixels) and fai-mf-m-ade (fai-mf-m-ade.json: the same head, 3 decoder layers with a 512-wide
    FFN, on the STDC-2 backbone): registry config = the reference's file, state-dict keys = the reference model's, forward and the
    per-pixel-argmax post-process of the restatement vs the real reference.  fai-mf-{m,s}-coco-ins (R101-vd / R50-vd, 128-channel pixel
    decoder WITH a three-layer transformer encoder of 8 heads x 16 channels, instance post-processing) through the same check."""
    import json
    import os

    from focoos_amd.registry import ModelRegistry
    from focoos_amd.synth import synth_image_structured, synth_state_dict
    from oracle import mf_oracle as M

    ref_import.install()
    import focoos.models.fai_mf.processor as fp

    cfg = ModelRegistry.get_model_info(name)["config"]
    ref_cfg = json.load(open(os.path.join(ref_import.REFERENCE_ROOT, f"focoos/model_registry/{name}.json")))["config"]
    assert {k: v for k, v in cfg.items() if k in ref_cfg} == ref_cfg
    model, proc, _ = ref_import.build_reference_mf(ref_cfg)
    fp.binary_mask_to_base64 = lambda m: ""
    sd = synth_state_dict(cfg, seed=17, family="fai_mf")
    model.load_state_dict(sd, strict=True)
    assert list(model.state_dict()) == list(sd)
    imgs = [synth_image_structured(27, 96, 128)]
    x, _ = proc.preprocess(imgs, device=torch.device("cpu"), dtype=torch.float32)
    with torch.no_grad():
        out = model(x)
        probs, masks = M.mf_forward(sd, cfg, x)
    np.testing.assert_allclose(probs.numpy(), out.logits.numpy(), atol=1e-4)
    assert (masks - out.masks).abs().max().item() < 5e-3
    dets = proc.postprocess(out, imgs)[0].detections
    s, l, q, boxes, bm = M.postprocess(out.logits, out.masks, [(96, 128)], cfg["mask_threshold"], cfg["threshold"], cfg["use_mask_score"],
                                       predict_all_pixels=bool(cfg["predict_all_pixels"]))[0]
    assert len(dets) == len(s) and len(s) >= 1
    np.testing.assert_allclose([d.conf for d in dets], s.numpy(), atol=1e-6)
    assert [d.cls_id for d in dets] == l.tolist()
    assert [list(d.bbox) for d in dets] == boxes.tolist()


@pytest.mark.parametrize("name", ["fai-mf-l-coco-ins", "fai-mf-m-ade", "bisenetformer-l-ade"])
def test_odd_sizes_oracle_matches_reference_live(name):
    """Inputs that are not multiples of 32 (the mask families' processors do not resize or pad: size_divisibility 0): the restatements
    against the real reference at 100x150, 75x94 and 130x97 - ceil(H/2) at every stride-2 layer, partial AvgPool2d(ceil_mode) windows,
    F.interpolate(size=...) with non-integer ratios.  What tests/test_gpu_odd_sizes.py compares the engine with."""
    import json
    import os

    from focoos_amd.registry import ModelRegistry
    from focoos_amd.synth import synth_image_structured, synth_state_dict

    ref_import.install()
    info = ModelRegistry.get_model_info(name)
    cfg, family = info["config"], info["model_family"]
    ref_cfg = json.load(open(os.path.join(ref_import.REFERENCE_ROOT, f"focoos/model_registry/{name}.json")))["config"]
    if family == "fai_mf":
        from oracle import mf_oracle as M

        model, proc, _ = ref_import.build_reference_mf(ref_cfg)
        fwd = lambda sd, x: M.mf_forward(sd, cfg, x)   # noqa: E731
    else:
        from oracle import bf_oracle as BF

        model, proc, _ = ref_import.build_reference_bf(ref_cfg)
        fwd = lambda sd, x: BF.bf_forward(sd, cfg, x)   # noqa: E731
    sd = synth_state_dict(cfg, seed=17, family=family)
    model.load_state_dict(sd, strict=True)
    for h, w in ((100, 150), (75, 94), (130, 97)):
        imgs = [synth_image_structured(27, h, w)]
        x, _ = proc.preprocess(imgs, device=torch.device("cpu"), dtype=torch.float32)
        assert tuple(x.shape) == (1, 3, h, w)
        with torch.no_grad():
            out = model(x)
            probs, masks = fwd(sd, x)
        assert tuple(out.masks.shape[-2:]) == (h, w)
        np.testing.assert_allclose(probs.numpy(), out.logits.numpy(), atol=1e-4)
        assert (masks - out.masks).abs().max().item() < 5e-3


def test_detr_odd_sizes_oracle_matches_reference_live():
    """RT-DETR at inputs that are not multiples of 32 (ragged training batches are padded to the batch maximum; the reference's encoder
    resizes with F.interpolate(size=...) in both directions, modelling.py:334,342): the encoder levels of the restatement equal the real
    reference's at 200x232, 250x188 and 208x272, and with the reference's own query selection so do the outputs (free-running the two
    may pick different tokens among the INVALID anchors, which all carry the same masked score - torch.topk's order among ties is
    unspecified).  What tests/test_gpu_odd_sizes.py::test_detr_odd_size_matches_oracle compares the engine with."""
    from focoos_amd.registry import ModelRegistry
    from focoos_amd.synth import synth_image_structured, synth_state_dict
    from oracle import detr_oracle as O

    ref_import.install()
    from focoos.model_manager import ConfigManager
    from focoos.models.fai_detr.modelling import FAIDetr
    from focoos.ports import ModelFamily

    cfg = ModelRegistry.get_model_info("fai-detr-l-coco")["config"]
    model = FAIDetr(ConfigManager.from_dict(ModelFamily.DETR, dict(cfg))).eval()
    sd = synth_state_dict(cfg, 17)
    model.load_state_dict(sd, strict=True)
    for hw in ((200, 232), (250, 188), (208, 272)):
        x = O.get_torch_batch([synth_image_structured(5 + i, *hw) for i in range(2)], hw)
        picked, orig = [], torch.topk

        def recording_topk(*a, **k):
            r = orig(*a, **k)
            picked.append(r.indices.clone())
            return r

        torch.topk = recording_topk
        try:
            with torch.no_grad():
                out = model(x)
                feats = model.pixel_decoder((x - model.pixel_mean) / model.pixel_std)
        finally:
            torch.topk = orig
        col = {}
        with torch.no_grad():
            probs, boxes = O.detr_forward(sd, cfg, x, forced_topk=picked[0], collect=col)
        by_shape = {}

        def walk(o):
            if isinstance(o, torch.Tensor):
                by_shape[tuple(o.shape)] = o
            elif isinstance(o, dict):
                [walk(v) for v in o.values()]
            elif isinstance(o, (list, tuple)):
                [walk(v) for v in o]

        walk(feats)
        ceil2 = lambda n: -(-n // 2)   # noqa: E731
        h, w = ceil2(ceil2(ceil2(hw[0]))), ceil2(ceil2(ceil2(hw[1])))
        for k in ("enc_s8", "enc_s16", "enc_s32"):   # ceil(H/2) at every stride-2 layer
            assert tuple(col[k].shape[-2:]) == (h, w), (hw, k, col[k].shape)
            ref_level = by_shape[tuple(col[k].shape)]
            assert float((ref_level - col[k]).norm() / ref_level.norm()) < 1e-5, (hw, k)
            h, w = ceil2(h), ceil2(w)
        np.testing.assert_allclose(probs.numpy(), out.logits.numpy(), atol=1e-4)
        np.testing.assert_allclose(boxes.numpy(), out.boxes.numpy(), atol=1e-4)


def test_bf_oracle_matches_reference_live():
    """BiSeNetFormer (A13): forward + batch-1 postprocess (predict_all_pixels) of the restatement vs the real reference, another
    seed and size than the committed golden; the registry config equals the reference's own registry file."""
    _bf_forward_vs_reference("bisenetformer-l-ade")


@pytest.mark.parametrize("name", ["bisenetformer-s-ade", "bisenetformer-m-ade"])
def test_bf_small_variant_forward_matches_reference(name):
    """bisenetformer-s-ade (STDC-1: two blocks per stage) and bisenetformer-m-ade (96-channel pixel decoder / mask dimension, four decoder
    layers): the same restatement against the reference built from ITS registry file."""
    _bf_forward_vs_reference(name)


def _bf_forward_vs_reference(name):
    import json
    import os

    from focoos_amd.registry import ModelRegistry
    from focoos_amd.synth import synth_image_structured, synth_state_dict
    from oracle import bf_oracle as BF

    ref_import.install()
    import focoos.models.bisenetformer.processor as bp

    cfg = ModelRegistry.get_model_info(name)["config"]
    ref_cfg = json.load(open(os.path.join(ref_import.REFERENCE_ROOT, f"focoos/model_registry/{name}.json")))["config"]
    assert {k: v for k, v in cfg.items() if k != "resolution"} == ref_cfg
    model, proc, _ = ref_import.build_reference_bf(ref_cfg)
    bp.binary_mask_to_base64 = lambda m: ""  # cv2/PNG tail is not installed and outside the path
    sd = synth_state_dict(cfg, seed=9, family="bisenetformer")
    model.load_state_dict(sd, strict=True)
    assert list(model.state_dict()) == list(sd)
    imgs = [synth_image_structured(23, 96, 128)]
    x, _ = proc.preprocess(imgs, device=torch.device("cpu"), dtype=torch.float32)
    with torch.no_grad():
        out = model(x)
        probs, masks = BF.bf_forward(sd, cfg, x)
    np.testing.assert_allclose(probs.numpy(), out.logits.numpy(), atol=1e-4)
    assert (masks - out.masks).abs().max().item() < 5e-3
    dets = proc.postprocess(out, imgs)[0].detections
    # the post-process on the REFERENCE's own tensors (isolates the restated post-process from forward float noise)
    s, l, q, boxes, bm = BF.postprocess(out.logits, out.masks, [(96, 128)], cfg)[0]
    assert len(dets) == len(s) and len(s) >= 1
    np.testing.assert_allclose([d.conf for d in dets], s.numpy(), atol=1e-6)
    assert [d.cls_id for d in dets] == l.tolist()
    assert [list(d.bbox) for d in dets] == boxes.tolist()


def test_train_oracle_matches_reference_losses_and_gradients():
    """oracle/train_oracle.py (training forward + 7-set criterion, BatchNorm frozen) vs the REAL reference in train mode with
    its BatchNorm modules switched to eval: the 21 weighted losses and the gradients of parameters spread over the model."""
    ref_import.install()
    from focoos.models.fai_detr.ports import DETRTargets

    from focoos_amd.registry import ModelRegistry
    from focoos_amd.synth import synth_image_structured, synth_state_dict
    from oracle import detr_oracle as O
    from oracle import train_oracle as T

    cfg = ModelRegistry.get_model_info("fai-detr-l-coco")["config"]
    model, proc, _ = ref_import.build_reference_detr(cfg)
    sd = synth_state_dict(cfg, seed=6)
    model.load_state_dict(sd, strict=True)
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.eval()
    imgs = [synth_image_structured(30 + i, 128, 160) for i in range(2)]
    x = O.get_torch_batch(imgs, None)
    labels, boxes = T.synth_targets(1, 2, 80, counts=(3, 5))
    out = model(x, [DETRTargets(labels=l, boxes=b) for l, b in zip(labels, boxes)])
    ref_losses = out.loss
    assert len(ref_losses) == 21
    sum(ref_losses.values()).backward()
    sdg = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and v.dim() > 0 and "running" not in k and "empty_weight" not in k else v)
           for k, v in sd.items()}
    outs = T.detr_train_outputs(sdg, cfg, x)
    losses, _ = T.criterion(outs, labels, boxes)
    assert sorted(losses) == sorted(ref_losses)
    for k in ref_losses:
        np.testing.assert_allclose(float(losses[k]), float(ref_losses[k]), rtol=2e-4, atol=1e-5, err_msg=k)
    sum(losses.values()).backward()
    named = dict(model.named_parameters())
    for k in ("pixel_decoder.backbone.res_layers.1.blocks.0.branch2b.conv.weight", "pixel_decoder.encoder.0.layers.0.self_attn.in_proj_weight",
              "pixel_decoder.fpn_blocks.1.bottlenecks.2.conv1.conv.weight", "head.predictor.decoder.layers.3.cross_attn.sampling_offsets.weight",
              "head.predictor.enc_score_classifier.weight", "head.predictor.dec_bbox_classifier.5.layers.2.weight", "head.predictor.query_pos_head.layers.0.weight"):
        g_ref, g_mine = named[k].grad, sdg[k].grad
        assert g_ref is not None and g_mine is not None, k
        assert (g_mine - g_ref).abs().max() <= 2e-3 * g_ref.abs().max() + 1e-7, k


def test_train_oracle_batch_stat_batchnorm_matches_reference():
    """Same check with the reference fully in .train() (BatchNorm on batch statistics, the reference's default when
    freeze_bn is off): losses, gradients including the BN affine parameters, and the running-statistics update."""
    ref_import.install()
    from focoos.models.fai_detr.ports import DETRTargets

    from focoos_amd.registry import ModelRegistry
    from focoos_amd.synth import synth_image_structured, synth_state_dict
    from oracle import detr_oracle as O
    from oracle import train_oracle as T

    cfg = ModelRegistry.get_model_info("fai-detr-l-coco")["config"]
    model, proc, _ = ref_import.build_reference_detr(cfg)
    sd = synth_state_dict(cfg, seed=6)
    model.load_state_dict(sd, strict=True)
    model.train()
    imgs = [synth_image_structured(30 + i, 128, 160) for i in range(2)]
    x = O.get_torch_batch(imgs, None)
    labels, boxes = T.synth_targets(1, 2, 80, counts=(3, 5))
    out = model(x, [DETRTargets(labels=l, boxes=b) for l, b in zip(labels, boxes)])
    ref_losses = out.loss
    sum(ref_losses.values()).backward()
    sdg = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and v.dim() > 0 and "running" not in k and "empty_weight" not in k
               else v.clone()) for k, v in sd.items()}
    O.BN_TRAINING[0] = True
    try:
        outs = T.detr_train_outputs(sdg, cfg, x)
    finally:
        O.BN_TRAINING[0] = False
    losses, _ = T.criterion(outs, labels, boxes)
    for k in ref_losses:
        np.testing.assert_allclose(float(losses[k]), float(ref_losses[k]), rtol=5e-4, atol=1e-5, err_msg=k)
    sum(losses.values()).backward()
    named = dict(model.named_parameters())
    for k in ("pixel_decoder.backbone.res_layers.1.blocks.0.branch2b.conv.weight", "pixel_decoder.backbone.res_layers.2.blocks.1.branch2c.norm.weight",
              "pixel_decoder.backbone.conv1.conv1_1.norm.bias", "pixel_decoder.input_proj.1.1.weight",
              "pixel_decoder.fpn_blocks.1.bottlenecks.2.conv1.conv.weight", "pixel_decoder.fpn_blocks.0.bottlenecks.0.conv2.norm.weight",
              "head.predictor.input_proj.0.norm.bias", "head.predictor.enc_score_classifier.weight"):
        g_ref, g_mine = named[k].grad, sdg[k].grad
        assert g_ref is not None and g_mine is not None, k
        assert (g_mine - g_ref).abs().max() <= 5e-3 * g_ref.abs().max() + 1e-7, k
    ref_sd = model.state_dict()
    for k in ("pixel_decoder.backbone.conv1.conv1_2.norm.running_mean", "pixel_decoder.backbone.res_layers.3.blocks.2.branch2b.norm.running_var",
              "pixel_decoder.pan_blocks.0.conv2.norm.running_var", "head.predictor.input_proj.2.norm.running_mean"):
        np.testing.assert_allclose(sdg[k].numpy(), ref_sd[k].numpy(), rtol=1e-4, atol=1e-5, err_msg=k)
        assert not torch.equal(sdg[k], sd[k]), k


def test_bf_train_oracle_matches_reference_losses_and_gradients():
    """oracle/train_oracle.bf_train_outputs + bf_criterion (BiSeNetFormer training forward, 7 supervised prediction heads, point-sampled
    Hungarian criterion) vs the REAL reference fully in .train() (BatchNorm on batch statistics): the 21 weighted losses on the
    reference's own recorded torch.rand draws, gradients of parameters spread over the model incl. BatchNorm affine and the learned
    queries, and the running-statistics update."""
    ref_import.install()
    import json
    import os

    from focoos.models.bisenetformer.ports import BisenetFormerTargets

    from focoos_amd.registry import ModelRegistry
    from focoos_amd.synth import synth_image_structured, synth_state_dict
    from oracle import detr_oracle as O
    from oracle import mask_criterion_oracle as MC
    from oracle import train_oracle as T

    cfg = ModelRegistry.get_model_info("bisenetformer-l-ade")["config"]
    cfg = dict(cfg, criterion_num_points=1024)   # 12544 points per pair x 21 point-sampling calls is minutes on CPU; same code path
    ref_cfg = json.load(open(os.path.join(ref_import.REFERENCE_ROOT, "focoos/model_registry/bisenetformer-l-ade.json")))["config"]
    ref_cfg = dict(ref_cfg, criterion_num_points=1024)
    model, proc, _ = ref_import.build_reference_bf(ref_cfg)
    sd = synth_state_dict(cfg, seed=12, family="bisenetformer")
    model.load_state_dict(sd, strict=True)
    model.train()
    imgs = [synth_image_structured(40 + i, 128, 160) for i in range(2)]
    x = O.get_torch_batch(imgs, None)
    labels, masks = T.synth_mask_targets(3, 2, int(cfg["num_classes"]), (128, 160), counts=(3, 5))
    rec, orig_rand = [], torch.rand

    def spy_rand(*a, **k):
        t = orig_rand(*a, **k)
        rec.append(t.clone())
        return t

    torch.rand = spy_rand
    try:
        out = model(x, [BisenetFormerTargets(labels=l, masks=m) for l, m in zip(labels, masks)])
    finally:
        torch.rand = orig_rand
    ref_losses = out.loss
    assert len(ref_losses) == 21
    sum(ref_losses.values()).backward()
    sdg = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and v.dim() > 0 and "running" not in k and "empty_weight" not in k
               else v.clone()) for k, v in sd.items()}
    O.BN_TRAINING[0] = True
    try:
        outs = T.bf_train_outputs(sdg, cfg, x)
    finally:
        O.BN_TRAINING[0] = False
    losses, _ = T.bf_criterion(outs, labels, masks, MC.RandStream(rec), cfg)
    assert sorted(losses) == sorted(ref_losses)
    for k in ref_losses:
        np.testing.assert_allclose(float(losses[k]), float(ref_losses[k]), rtol=5e-4, atol=1e-5, err_msg=k)
    sum(losses.values()).backward()
    named = dict(model.named_parameters())
    for k in ("pixel_decoder.backbone.features.1.conv.weight", "pixel_decoder.backbone.features.6.avd_layer.0.weight",
              "pixel_decoder.backbone.features.6.avd_layer.1.weight", "pixel_decoder.backbone.features.9.conv_list.2.bn.bias",
              "pixel_decoder.cp.arm32.conv_atten.weight", "pixel_decoder.cp.arm16.bn_atten.weight", "pixel_decoder.cp.conv_avg.conv.weight",
              "pixel_decoder.ffm.proj1.weight", "pixel_decoder.ffm.conv2.weight", "pixel_decoder.conv_out.conv.weight",
              "head.predictor.query_embed.weight", "head.predictor.query_feat.weight", "head.predictor.input_proj.1.bias",
              "head.predictor.transformer_cross_attention_layers.2.multihead_attn.in_proj_weight",
              "head.predictor.forward_prediction_heads.mask_classifier.layers.2.weight", "head.predictor.forward_prediction_heads.classifier.bias"):
        g_ref, g_mine = named[k].grad, sdg[k].grad
        assert g_ref is not None and g_mine is not None, k
        assert (g_mine - g_ref).abs().max() <= 5e-3 * g_ref.abs().max() + 1e-7, (k, float((g_mine - g_ref).abs().max()), float(g_ref.abs().max()))
    ref_sd = model.state_dict()
    for k in ("pixel_decoder.backbone.features.3.conv_list.1.bn.running_mean", "pixel_decoder.backbone.features.11.avd_layer.1.running_var",
              "pixel_decoder.cp.arm32.bn_atten.running_mean", "pixel_decoder.conv_out.bn.running_var"):
        np.testing.assert_allclose(sdg[k].numpy(), ref_sd[k].numpy(), rtol=1e-4, atol=1e-5, err_msg=k)
        assert not torch.equal(sdg[k], sd[k]), k


def test_reference_fp16_amp_losses_vs_fp32_and_bf16_autocast():
    """BASELINE config 5 names fp16: the reference trains under ``torch.autocast(dtype=float16)`` + GradScaler (trainer/trainer.py:645,
    735-773).  The engine computes in bf16 with fp32 masters and needs no loss scaling (bf16 has fp32's exponent range; GradScaler exists
    for fp16's 6e-8 underflow).  This pins the chain that justifies it, on the REAL reference's training forward (BiSeNetFormer, identical
    sample points in the three runs): fp16-AMP losses deviate from fp32 by <= 1 %, bf16-autocast losses by <= 2.5 % (measured 0.66 % /
    1.7 %) - i.e. bf16 is within ~2 % of what the reference's fp16 AMP computes - and tests/test_gpu_train_bf.py holds the engine's bf16
    losses within 3 % of the fp32 oracle that is pinned to the same reference."""
    ref_import.install()
    import json
    import os

    from focoos.models.bisenetformer.ports import BisenetFormerTargets

    from focoos_amd.registry import ModelRegistry
    from focoos_amd.synth import synth_image_structured, synth_state_dict
    from oracle import detr_oracle as O
    from oracle import train_oracle as T

    cfg = ModelRegistry.get_model_info("bisenetformer-l-ade")["config"]
    ref_cfg = dict(json.load(open(os.path.join(ref_import.REFERENCE_ROOT, "focoos/model_registry/bisenetformer-l-ade.json")))["config"], criterion_num_points=1024)
    model, _, _ = ref_import.build_reference_bf(ref_cfg)
    model.load_state_dict(synth_state_dict(cfg, seed=12, family="bisenetformer"), strict=True)
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.eval()
    x = O.get_torch_batch([synth_image_structured(40 + i, 128, 160) for i in range(2)], None)
    labels, masks = T.synth_mask_targets(3, 2, int(cfg["num_classes"]), (128, 160), counts=(3, 5))
    tg = [BisenetFormerTargets(labels=l, masks=m) for l, m in zip(labels, masks)]
    res = {}
    for name, dt in (("fp32", None), ("fp16", torch.float16), ("bf16", torch.bfloat16)):
        torch.manual_seed(0)   # the criterion's torch.rand draws: the same points in every run
        with torch.no_grad():
            if dt is None:
                out = model(x, tg)
            else:
                with torch.autocast("cpu", dtype=dt):
                    out = model(x, tg)
        res[name] = {k: float(v) for k, v in out.loss.items()}
    dev = {n: max(abs(res[n][k] - res["fp32"][k]) / (abs(res["fp32"][k]) + 1e-3) for k in res["fp32"]) for n in ("fp16", "bf16")}
    print("max relative loss deviation vs fp32:", dev)
    assert dev["fp16"] <= 1e-2 and dev["bf16"] <= 2.5e-2, dev


def test_mf_train_oracle_matches_reference_losses_and_gradients():
    """oracle/train_oracle.mf_train_outputs + bf_criterion (MaskFormer training forward: ResNet-vd, TransformerFPN with its pre-norm encoder,
    10 supervised prediction heads, the shared point-sampled criterion) vs the REAL reference fully in .train(): the 30 weighted losses on
    the reference's recorded torch.rand draws, gradients of parameters spread over the model, running statistics."""
    ref_import.install()
    import json
    import os

    from focoos.models.fai_mf.ports import MaskFormerTargets

    from focoos_amd.registry import ModelRegistry
    from focoos_amd.synth import synth_image_structured, synth_state_dict
    from oracle import detr_oracle as O
    from oracle import mask_criterion_oracle as MC
    from oracle import train_oracle as T

    cfg = dict(ModelRegistry.get_model_info("fai-mf-l-coco-ins")["config"], criterion_num_points=1024)
    cfg["backbone_config"] = dict(cfg["backbone_config"], depth=50)    # R50 instead of R101: the same code path, a third of the CPU time
    ref_cfg = dict(json.load(open(os.path.join(ref_import.REFERENCE_ROOT, "focoos/model_registry/fai-mf-l-coco-ins.json")))["config"], criterion_num_points=1024)
    ref_cfg["backbone_config"] = dict(ref_cfg["backbone_config"], depth=50)
    model, proc, _ = ref_import.build_reference_mf(ref_cfg)
    sd = synth_state_dict(cfg, seed=14, family="fai_mf")
    model.load_state_dict(sd, strict=True)
    model.train()
    imgs = [synth_image_structured(44 + i, 128, 160) for i in range(2)]
    x = O.get_torch_batch(imgs, None)
    labels, masks = T.synth_mask_targets(4, 2, int(cfg["num_classes"]), (128, 160), counts=(3, 4))
    rec, orig_rand = [], torch.rand

    def spy_rand(*a, **k):
        t = orig_rand(*a, **k)
        rec.append(t.clone())
        return t

    torch.rand = spy_rand
    try:
        out = model(x, [MaskFormerTargets(labels=l, masks=m) for l, m in zip(labels, masks)])
    finally:
        torch.rand = orig_rand
    ref_losses = out.loss
    assert len(ref_losses) == 30
    sum(ref_losses.values()).backward()
    sdg = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and v.dim() > 0 and "running" not in k and "empty_weight" not in k
               else v.clone()) for k, v in sd.items()}
    O.BN_TRAINING[0] = True
    try:
        outs = T.mf_train_outputs(sdg, cfg, x)
    finally:
        O.BN_TRAINING[0] = False
    losses, _ = T.bf_criterion(outs, labels, masks, MC.RandStream(rec), cfg)
    assert sorted(losses) == sorted(ref_losses)
    for k in ref_losses:
        np.testing.assert_allclose(float(losses[k]), float(ref_losses[k]), rtol=5e-4, atol=1e-5, err_msg=k)
    sum(losses.values()).backward()
    named = dict(model.named_parameters())
    for k in ("pixel_decoder.backbone.res_layers.1.blocks.0.branch2b.conv.weight", "pixel_decoder.input_proj.bias",
              "pixel_decoder.transformer.encoder.layers.3.self_attn.in_proj_weight", "pixel_decoder.transformer.encoder.norm.weight",
              "pixel_decoder.adapter_2.weight", "pixel_decoder.layer_1.norm.weight", "pixel_decoder.layer_4.weight", "pixel_decoder.mask_features.bias",
              "head.predictor.query_embed.weight", "head.predictor.input_proj.2.weight",
              "head.predictor.transformer_cross_attention_layers.7.multihead_attn.in_proj_weight",
              "head.predictor.forward_prediction_heads.mask_classifier.layers.0.weight"):
        g_ref, g_mine = named[k].grad, sdg[k].grad
        assert g_ref is not None and g_mine is not None, k
        assert (g_mine - g_ref).abs().max() <= 5e-3 * g_ref.abs().max() + 1e-7, (k, float((g_mine - g_ref).abs().max()), float(g_ref.abs().max()))
    ref_sd = model.state_dict()
    for k in ("pixel_decoder.adapter_3.norm.running_mean", "pixel_decoder.layer_2.norm.running_var"):
        np.testing.assert_allclose(sdg[k].numpy(), ref_sd[k].numpy(), rtol=1e-4, atol=1e-5, err_msg=k)
        assert not torch.equal(sdg[k], sd[k]), k


@pytest.mark.parametrize("counts", [(0, 0), (0, 3)])
def test_train_oracle_matches_reference_with_empty_targets(counts):
    """Images without a single ground-truth box (a whole batch of them, or one of two): the training oracle against the REAL reference -
    `num_boxes` clamped to 1, empty index pairs from the matcher, box losses that sum over nothing (fai_detr/modelling.py:553-612,
    693-758).  This is what tests/test_gpu_detr_variants.py::test_detr_train_step_with_empty_targets holds the engine against."""
    ref_import.install()
    from focoos.models.fai_detr.ports import DETRTargets

    from focoos_amd.registry import ModelRegistry
    from focoos_amd.synth import synth_image_structured, synth_state_dict
    from oracle import detr_oracle as O
    from oracle import train_oracle as T

    cfg = ModelRegistry.get_model_info("fai-detr-m-coco")["config"]
    model, proc, _ = ref_import.build_reference_detr(cfg)
    sd = synth_state_dict(cfg, seed=33)
    model.load_state_dict(sd, strict=True)
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.eval()
    imgs = [synth_image_structured(90 + i, 128, 160) for i in range(2)]
    x = O.get_torch_batch(imgs, None)
    g = torch.Generator().manual_seed(4)
    labels = [torch.randint(0, 80, (n,), generator=g) for n in counts]
    boxes = [torch.cat([torch.rand(n, 2, generator=g) * 0.5 + 0.25, torch.rand(n, 2, generator=g) * 0.3 + 0.1], 1) for n in counts]
    with torch.no_grad():
        ref_losses = model(x, [DETRTargets(labels=l, boxes=b) for l, b in zip(labels, boxes)]).loss
        outs = T.detr_train_outputs(sd, cfg, x)
        losses, matches = T.criterion(outs, labels, boxes)
    assert sorted(losses) == sorted(ref_losses) and len(losses) == 3 * (3 + 1)
    for k in ref_losses:
        a, b = float(losses[k]), float(ref_losses[k])
        assert np.isfinite(b)
        np.testing.assert_allclose(a, b, rtol=2e-4, atol=1e-5, err_msg=k)
        if sum(counts) == 0 and ("bbox" in k or "giou" in k):
            assert b == 0.0, (k, b)
    for per_set in matches:
        assert [len(i) for i, _ in per_set] == list(counts)


def test_mask_train_oracle_matches_reference_with_an_empty_image():
    """One of two images without a ground-truth mask: the mask-family training oracle (BiSeNetFormer-S, 1 024 sample points) against the
    REAL reference - `num_masks` counts the other image only, the empty image contributes its no-object class loss and nothing to the
    mask / dice sums (fai_mf/loss.py:345-607, 661-723).  tests/test_gpu_train_bf.py::test_mask_train_step_with_empty_targets holds the
    engine against the oracle on the same kind of input."""
    ref_import.install()
    import json
    import os

    from focoos.models.bisenetformer.ports import BisenetFormerTargets

    from focoos_amd.registry import ModelRegistry
    from focoos_amd.synth import synth_image_structured, synth_state_dict
    from oracle import detr_oracle as O
    from oracle import mask_criterion_oracle as MC
    from oracle import train_oracle as T

    name = "bisenetformer-s-ade"
    cfg = dict(ModelRegistry.get_model_info(name)["config"], criterion_num_points=1024)
    ref_cfg = dict(json.load(open(os.path.join(ref_import.REFERENCE_ROOT, f"focoos/model_registry/{name}.json")))["config"], criterion_num_points=1024)
    model, proc, _ = ref_import.build_reference_bf(ref_cfg)
    sd = synth_state_dict(cfg, seed=17, family="bisenetformer")
    model.load_state_dict(sd, strict=True)
    model.train()
    imgs = [synth_image_structured(60 + i, 128, 160) for i in range(2)]
    x = O.get_torch_batch(imgs, None)
    labels, masks = T.synth_mask_targets(5, 2, int(cfg["num_classes"]), (128, 160), counts=(0, 4))
    assert labels[0].numel() == 0 and masks[0].shape[0] == 0
    rec, orig_rand = [], torch.rand

    def spy_rand(*a, **k):
        t = orig_rand(*a, **k)
        rec.append(t.clone())
        return t

    torch.rand = spy_rand
    try:
        with torch.no_grad():
            ref_losses = model(x, [BisenetFormerTargets(labels=l, masks=m) for l, m in zip(labels, masks)]).loss
    finally:
        torch.rand = orig_rand
    O.BN_TRAINING[0] = True
    try:
        with torch.no_grad():
            outs = T.bf_train_outputs({k: v.clone() for k, v in sd.items()}, cfg, x)
    finally:
        O.BN_TRAINING[0] = False
    losses, _ = T.bf_criterion(outs, labels, masks, MC.RandStream([r for r in rec if r.numel() > 0]), cfg)
    assert sorted(losses) == sorted(ref_losses)
    for k in ref_losses:
        assert np.isfinite(float(ref_losses[k]))
        np.testing.assert_allclose(float(losses[k]), float(ref_losses[k]), rtol=5e-4, atol=1e-5, err_msg=k)


@pytest.mark.parametrize("case", ["default", "none_above_threshold", "top_k_10", "threshold_zero_means_default", "ragged_sizes"])
def test_detr_postprocess_oracle_equals_reference_on_random_outputs(case):
    """oracle.detr_oracle.postprocess against the REAL DETRProcessor.postprocess (fai_detr/processor.py:146-217) on the same random model
    outputs - no model in the loop, so exact: scores, class ids and the rounded integer boxes detection by detection, for the edge cases
    the GPU post-process tests hold fx_detr_postprocess to: nothing above the threshold (empty detection lists), a small top_k,
    `threshold=0.0` (the reference's `threshold or self.threshold`: 0.0 selects the configured default), a batch of differently sized
    non-square originals."""
    ref_import.install()
    from focoos.models.fai_detr.ports import DETRModelOutput as RefOut
    from focoos.models.fai_detr.processor import DETRProcessor as RefProc
    from focoos.model_manager import ConfigManager
    from focoos.ports import ModelFamily

    from focoos_amd.registry import ModelRegistry
    from oracle import detr_oracle as O

    g = torch.Generator().manual_seed({"default": 1, "none_above_threshold": 2, "top_k_10": 3, "threshold_zero_means_default": 4, "ragged_sizes": 5}[case])
    B, Q, K = 3, 300, 80
    probs = torch.rand(B, Q, K, generator=g) ** 6            # a few confident entries per image, most near zero
    cxcy, wh = torch.rand(B, Q, 2, generator=g) * 0.6 + 0.2, torch.rand(B, Q, 2, generator=g) * 0.3 + 0.02
    boxes = torch.cat([cxcy - wh / 2, cxcy + wh / 2], -1)
    sizes = [(480, 640), (333, 517), (1080, 1920)] if case == "ragged_sizes" else [(480, 640)] * B
    imgs = [np.zeros((h, w, 3), np.uint8) for h, w in sizes]
    cfg = ConfigManager.from_dict(ModelFamily.DETR, dict(ModelRegistry.get_model_info("fai-detr-l-coco")["config"], top_k=10 if case == "top_k_10" else 300,
                                                         threshold=0.5))
    proc = RefProc(cfg, image_size=640).eval()
    if case == "none_above_threshold":
        probs = probs * 0.4
    thr_arg = {"default": 0.5, "none_above_threshold": 0.5, "top_k_10": 0.3, "threshold_zero_means_default": 0.0, "ragged_sizes": 0.25}[case]
    thr_eff = thr_arg or cfg.threshold
    dets = proc.postprocess(RefOut(boxes=boxes, logits=probs, loss=None), imgs, threshold=thr_arg)
    res = O.postprocess(probs, boxes, sizes, cfg.top_k, thr_eff)
    assert len(dets) == B
    total = 0
    for d, (s, l, q, bx) in zip(dets, res):
        assert len(d.detections) == len(s)
        total += len(s)
        assert [x.cls_id for x in d.detections] == l.tolist()
        assert [list(x.bbox) for x in d.detections] == bx.tolist()
        np.testing.assert_allclose([x.conf for x in d.detections], s.numpy(), rtol=0, atol=1e-7)
    assert (total == 0) == (case == "none_above_threshold")
    if case == "top_k_10":
        assert all(len(d.detections) <= 10 for d in dets)


@pytest.mark.parametrize("family", ["fai_mf", "bisenetformer"])
@pytest.mark.parametrize("case", ["instances", "instances_no_mask_score", "nothing_kept", "tiny_masks_dropped", "all_pixels", "resized_original"])
def test_mask_postprocess_oracle_equals_reference_on_random_outputs(family, case):
    """oracle.mf_oracle.postprocess against the REAL MaskFormerProcessor / BisenetFormerProcessor.postprocess (fai_mf/processor.py:168-306,
    bisenetformer/processor.py:176-300) on the same random model outputs, batch 1 (what the reference's gather indexing supports) - no
    model in the loop: scores, class ids and integer boxes detection by detection, the decoded masks through their areas.  Cases: the
    threshold branch with and without the mask score (the x1e-3 scaling quirk :247-255), scores all below the threshold (empty list),
    masks of 0 / 1 pixels (dropped: strictly more than one pixel), the predict_all_pixels branch, an original size other than the mask's."""
    ref_import.install()
    import base64
    import json
    import os

    from oracle import mf_oracle as M

    if family == "fai_mf":
        import focoos.models.fai_mf.processor as mod
        from focoos.models.fai_mf.ports import MaskFormerModelOutput as RefOut

        name, build = "fai-mf-l-coco-ins", ref_import.build_reference_mf
    else:
        import focoos.models.bisenetformer.processor as mod
        from focoos.models.bisenetformer.ports import BisenetFormerOutput as RefOut

        name, build = "bisenetformer-l-ade", ref_import.build_reference_bf
    ref_cfg = json.load(open(os.path.join(ref_import.REFERENCE_ROOT, f"focoos/model_registry/{name}.json")))["config"]
    ref_cfg = dict(ref_cfg, predict_all_pixels=(case == "all_pixels"), use_mask_score=(case != "instances_no_mask_score"), threshold=0.3, mask_threshold=0.5)
    ref_cfg["backbone_config"] = dict(ref_cfg["backbone_config"], **({"depth": 50} if family == "fai_mf" else {}))
    _, proc, cfg = build(ref_cfg)
    areas = []
    mod.binary_mask_to_base64 = lambda m: areas.append(int(np.asarray(m).sum())) or base64.b64encode(b"x").decode()   # cv2 tail: not installed
    g = torch.Generator().manual_seed(sum(map(ord, case)))
    Q, K, h, w = 12, int(ref_cfg["num_classes"]), 48, 64
    probs = torch.softmax(torch.randn(1, Q, K + 1, generator=g) * 3.0, -1)[..., :K]
    yy, xx = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing="ij")
    masks = torch.zeros(1, Q, h, w)
    for q in range(Q):
        cy, cx, r = float(torch.rand(1, generator=g)) * h, float(torch.rand(1, generator=g)) * w, 4.0 + 8.0 * float(torch.rand(1, generator=g))
        masks[0, q] = torch.sigmoid((r - ((yy - cy) ** 2 + (xx - cx) ** 2).sqrt()) * 0.8)
    if case == "nothing_kept":
        probs = probs * 0.2
    if case == "tiny_masks_dropped":
        masks[0, :6] = 0.01
        masks[0, 0, 5, 5] = 0.9                      # one pixel: dropped
        masks[0, 1, 7, 7:9] = 0.9                    # two pixels: kept (if its score passes)
        probs[0, 1] = 0.0
        probs[0, 1, 3] = 0.95
    size = (96, 120) if case == "resized_original" else (h, w)
    img = np.zeros(size + (3,), np.uint8)
    dets = proc.postprocess(RefOut(masks=masks, logits=probs, loss=None), [img])[0].detections
    s, l, q, boxes, bm = M.postprocess(probs, masks, [size], 0.5, 0.3, case != "instances_no_mask_score", predict_all_pixels=(case == "all_pixels"))[0]
    assert len(dets) == len(s), (len(dets), len(s))
    assert (len(s) == 0) == (case == "nothing_kept")
    print(f"{family} {case}: {len(s)} detections, areas {areas}")
    np.testing.assert_allclose([d.conf for d in dets], s.numpy(), rtol=0, atol=1e-6)
    assert [d.cls_id for d in dets] == l.tolist()
    assert [list(d.bbox) for d in dets] == np.asarray(boxes).tolist()
    # the reference encodes the mask cropped to its box with an exclusive slice end (utils/vision.py:264-267): the same crop of the oracle's mask
    want = [int(m[b[1]:min(b[3], m.shape[0]), b[0]:min(b[2], m.shape[1])].sum()) for m, b in zip(bm, np.asarray(boxes).tolist())]
    assert areas == want
    if case == "tiny_masks_dropped":
        assert 0 not in q.tolist() and 1 in q.tolist()
