#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REAL reference (FocoosAI/focoos, imported
from /root/reference through oracle/ref_import.py) on seeded synthetic weights/inputs.

Run in the build container only (the GPU box has no /root/reference):
    python scripts/make_golden.py
The fixtures pin ``oracle/detr_oracle.py`` (tests/test_oracle_golden.py) and, through
it, the HIP engine.  Everything is seeded: weights = focoos_amd.synth.synth_state_dict(cfg, seed),
images = synth_image / synth_image_structured.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from focoos_amd.registry import ModelRegistry  # noqa: E402
from focoos_amd.synth import synth_image, synth_image_structured, synth_state_dict  # noqa: E402
from oracle import ref_import  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def strided_sample(t: torch.Tensor, n: int = 2048) -> np.ndarray:
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].to(torch.float32).numpy().copy()


def stats(t: torch.Tensor) -> np.ndarray:
    t = t.detach().to(torch.float64)
    return np.array([t.mean().item(), t.abs().mean().item(), t.abs().max().item(), (t * t).mean().sqrt().item()])


def run_case(model_name: str, seed: int, images, tag: str, threshold: float = 0.3):
    info = ModelRegistry.get_model_info(model_name)
    cfg = info["config"]
    model, proc, _ = ref_import.build_reference_detr(cfg)
    sd = synth_state_dict(cfg, seed=seed)
    res = model.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys

    cap = {}

    def hook(name):
        def f(m, i, o):
            cap[name] = o
        return f

    model.pixel_decoder.backbone.register_forward_hook(hook("bb"))
    model.pixel_decoder.register_forward_hook(hook("enc"))
    model.pixel_decoder.encoder[0].register_forward_hook(hook("aifi"))
    pred = model.head.predictor
    orig = pred._get_decoder_input

    def wrap(memory, shapes):
        cap["memory"] = memory
        r = orig(memory, shapes)
        cap["dec_in"] = r
        return r

    pred._get_decoder_input = wrap
    for i, layer in enumerate(pred.decoder.layers):
        layer.register_forward_hook(hook(f"dec{i}"))
    orig_topk = torch.topk
    topk_calls = []

    def my_topk(*a, **k):
        r = orig_topk(*a, **k)
        topk_calls.append(r)
        return r

    x, _ = proc.preprocess(images, device=torch.device("cpu"), dtype=torch.float32)
    torch.topk = my_topk
    try:
        with torch.no_grad():
            out = model(x)
    finally:
        torch.topk = orig_topk
    enc_topk = topk_calls[0][1]  # modelling.py:1214

    dets = proc.postprocess(out, images, threshold=threshold)
    g = {
        "seed": np.array(seed), "threshold": np.array(threshold),
        "image_sizes": np.array([im.shape[:2] for im in images], dtype=np.int32),
        "pre_sample": strided_sample(x, 4096), "pre_stats": stats(x),
        "enc_topk": enc_topk.numpy().astype(np.int32),
        "probs_max": out.logits.max(-1).values.numpy(), "probs_argmax": out.logits.argmax(-1).numpy().astype(np.int32),
        "boxes": out.boxes.numpy(),
        "memory_sample": strided_sample(cap["memory"], 8192), "memory_stats": stats(cap["memory"]),
        "target_sample": strided_sample(cap["dec_in"][0], 4096),
        "ref_unact": cap["dec_in"][1].numpy(),
        "aifi_sample": strided_sample(cap["aifi"], 4096), "aifi_stats": stats(cap["aifi"]),
    }
    for k in ("res3", "res4", "res5"):
        g[f"{k}_sample"] = strided_sample(cap["bb"][k], 4096)
        g[f"{k}_stats"] = stats(cap["bb"][k])
    for n, e in zip(("enc_s32", "enc_s16", "enc_s8"), cap["enc"][1]):
        g[f"{n}_sample"] = strided_sample(e, 4096)
        g[f"{n}_stats"] = stats(e)
    for i in range(len(pred.decoder.layers)):
        g[f"dec{i}_sample"] = strided_sample(cap[f"dec{i}"], 2048)
    # flat top-k of the final scores (what processor.py:147 selects)
    flat = out.logits.flatten(1)
    v, idx = orig_topk(flat, 300, dim=-1)
    g["post_topk_val"] = v.numpy()
    g["post_topk_idx"] = idx.numpy().astype(np.int32)
    n = max(len(d.detections) for d in dets)
    db = np.full((len(dets), n, 4), -1, np.int32)
    dl = np.full((len(dets), n), -1, np.int32)
    ds = np.zeros((len(dets), n), np.float32)
    dn = np.array([len(d.detections) for d in dets], np.int32)
    for i, d in enumerate(dets):
        for j, det in enumerate(d.detections):
            db[i, j] = det.bbox
            dl[i, j] = det.cls_id
            ds[i, j] = det.conf
    g.update(det_boxes=db, det_labels=dl, det_scores=ds, det_count=dn)
    path = os.path.join(GOLDEN, f"{tag}.npz")
    np.savez_compressed(path, **g)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB); detections/img: {dn.tolist()}")


def deform_core_case():
    """Golden vectors for the B4 seam (ms_deform_attn_core_pytorch, deformable.py:10-35)."""
    ref_import.install()
    from focoos.nn.layers.deformable import ms_deform_attn_core_pytorch

    from tests._cases import MSDA_SHAPES, msda_case_inputs

    shapes = MSDA_SHAPES
    value, loc, w = (torch.from_numpy(a) for a in msda_case_inputs())
    out = ms_deform_attn_core_pytorch(value, shapes, loc, w)
    path = os.path.join(GOLDEN, "msda_core.npz")
    np.savez_compressed(path, shapes=np.array(shapes, np.int32), out=out.numpy())
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def criterion_case():
    """Golden vectors of the REAL reference's BoxHungarianMatcher + SetCriterion (modelling.py:409-758) on seeded synthetic
    predictions/targets (incl. an image without targets): the cost blocks handed to SciPy, the matched indices, the losses."""
    ref_import.install()
    import focoos.models.fai_detr.modelling as M
    from focoos.models.fai_detr.ports import DETRTargets

    from oracle.criterion_oracle import synth_predictions_and_targets

    logits, boxes, labels, tboxes = synth_predictions_and_targets(0)
    targets = [DETRTargets(labels=l, boxes=b) for l, b in zip(labels, tboxes)]
    captured = []
    orig = M.linear_sum_assignment

    def spy(c):
        captured.append(np.array(c, dtype=np.float32))
        return orig(c)

    M.linear_sum_assignment = spy
    try:
        matcher = M.BoxHungarianMatcher(cost_class=2, cost_bbox=5, cost_giou=2, use_focal_loss=True, alpha=0.25, gamma=2.0)
        crit = M.SetCriterion(num_classes=logits.shape[-1], matcher=matcher, weight_dict={"loss_vfl": 1, "loss_bbox": 5, "loss_giou": 2},
                              losses=["vfl", "boxes"], focal_alpha=0.75, focal_gamma=2.0)
        losses = crit({"pred_logits": logits, "pred_boxes": boxes}, targets)
        captured_first = list(captured)
        ind = matcher({"pred_logits": logits, "pred_boxes": boxes}, targets)
    finally:
        M.linear_sum_assignment = orig
    g = {"loss": np.array([float(losses["loss_vfl"]), float(losses["loss_bbox"]), float(losses["loss_giou"])], np.float64)}
    for b, ((i, j), c) in enumerate(zip(ind, captured_first)):
        g[f"pred_idx_{b}"] = i.numpy().astype(np.int32)
        g[f"tgt_idx_{b}"] = j.numpy().astype(np.int32)
        g[f"cost_{b}"] = c
    path = os.path.join(GOLDEN, "detr_criterion.npz")
    np.savez_compressed(path, **g)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB); losses {g['loss']}")


def mf_case(tag: str = "mf_l_coco_ins_b2", seed: int = 3, hw=(160, 192)):
    """Golden vectors of the REAL reference's FAIMaskFormer + MaskFormerProcessor (fai_mf/modelling.py, fai_mf/processor.py)
    on seeded synthetic weights / images: stage samples, the boolean attention masks each decoder layer used (the discrete
    step a bf16 engine is teacher-forced with), low-res mask logits, class probabilities and the batch-1 post-process."""
    from focoos_amd.synth import synth_image_structured as sis
    info = ModelRegistry.get_model_info("fai-mf-l-coco-ins")
    cfg = info["config"]
    model, proc, _ = ref_import.build_reference_mf(cfg)
    import focoos.models.fai_mf.processor as fp
    fp.binary_mask_to_base64 = lambda m: ""  # the cv2/PNG tail is absent here and outside the path
    sd = synth_state_dict(cfg, seed=seed, family="fai_mf")
    res = model.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    cap = {}
    model.pixel_decoder.backbone.register_forward_hook(lambda m, i, o: cap.__setitem__("bb", o))
    model.pixel_decoder.transformer.register_forward_hook(lambda m, i, o: cap.__setitem__("enc", o))
    model.pixel_decoder.register_forward_hook(lambda m, i, o: cap.__setitem__("pd", o))
    model.head.predictor.register_forward_hook(lambda m, i, o: cap.__setitem__("pred", o))
    masks_used = []
    for lyr in model.head.predictor.transformer_cross_attention_layers:
        lyr.register_forward_pre_hook(lambda m, a, kw: masks_used.append(kw["memory_mask"]), with_kwargs=True)
    dec_out = []
    for lyr in model.head.predictor.transformer_ffn_layers:
        lyr.register_forward_hook(lambda m, i, o: dec_out.append(o))
    images = [sis(i, *hw) for i in range(2)]
    x, _ = proc.preprocess(images, device=torch.device("cpu"), dtype=torch.float32)
    with torch.no_grad():
        out = model(x)
    B = x.shape[0]
    g = {"seed": np.int64(seed), "hw": np.array(hw), "pre_sample": strided_sample(x, 4096)}
    for k in ("res2", "res3", "res4", "res5"):
        g[f"{k}_sample"] = strided_sample(cap["bb"][k], 4096)
    g["enc_sample"] = strided_sample(cap["enc"], 4096)
    mf, msf = cap["pd"]
    g["mask_features_sample"] = strided_sample(mf, 8192)
    for i in range(3):
        g[f"msf{i}_sample"] = strided_sample(msf[i], 4096)
    for i, m in enumerate(masks_used):  # [B*heads, Q, Lk], identical across heads
        mm = m.view(B, 8, m.shape[1], m.shape[2])
        assert bool((mm == mm[:, :1]).all())
        g[f"attn_mask{i}"] = np.packbits(mm[:, 0].numpy(), axis=-1)
        g[f"attn_mask{i}_len"] = np.int64(m.shape[2])
    for i, o in enumerate(dec_out):  # [Q, B, C]
        g[f"dec{i}_sample"] = strided_sample(o.permute(1, 0, 2).contiguous(), 2048)
    g["cls_logits"] = cap["pred"]["pred_logits"].numpy()
    g["mask_logits_f16"] = cap["pred"]["pred_masks"].numpy().astype(np.float16)
    g["probs"] = out.logits.numpy()
    g["masks_sample"] = strided_sample(out.masks, 16384)
    for i in range(B):
        o1 = type(out)(masks=out.masks[i:i + 1], logits=out.logits[i:i + 1], loss=None)
        d = proc.postprocess(o1, [images[i]])[0].detections
        g[f"det{i}_conf"] = np.array([a.conf for a in d], np.float32)
        g[f"det{i}_cls"] = np.array([a.cls_id for a in d], np.int64)
        g[f"det{i}_bbox"] = np.array([a.bbox for a in d], np.int64).reshape(-1, 4)
    np.savez_compressed(os.path.join(GOLDEN, tag + ".npz"), **g)
    print(tag, {k: v.shape for k, v in g.items() if hasattr(v, "shape") and v.ndim}, [len(g[f"det{i}_conf"]) for i in range(B)])


def bf_case(tag: str = "bf_l_ade_b2", seed: int = 4, hw=(160, 192)):
    """Golden vectors of the REAL reference's BisenetFormer + BisenetFormerProcessor (bisenetformer/modelling.py, processor.py)
    on seeded synthetic weights / images: stage samples (STDC features, context path, fusion, mask features), the boolean
    attention masks each decoder layer used, low-res mask logits, class probabilities and the batch-1 post-process
    (predict_all_pixels=True: per-pixel argmax over queries), including each detection's mask area."""
    from focoos_amd.synth import synth_image_structured as sis
    info = ModelRegistry.get_model_info("bisenetformer-l-ade")
    cfg = info["config"]
    rc = {k: v for k, v in cfg.items() if k != "resolution"}
    model, proc, _ = ref_import.build_reference_bf(rc)
    import focoos.models.bisenetformer.processor as bp
    bp.binary_mask_to_base64 = lambda m: ""  # the cv2/PNG tail is absent here and outside the path
    areas = []
    orig_trim = bp.trim_mask
    bp.trim_mask = lambda m, b: (areas.append(int(np.asarray(m).sum())), orig_trim(m, b))[1]
    sd = synth_state_dict(cfg, seed=seed, family="bisenetformer")
    res = model.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    cap = {}
    model.pixel_decoder.backbone.register_forward_hook(lambda m, i, o: cap.__setitem__("bb", o))
    model.pixel_decoder.cp.register_forward_hook(lambda m, i, o: cap.__setitem__("cp", o))
    model.pixel_decoder.ffm.register_forward_hook(lambda m, i, o: cap.__setitem__("ffm", o))
    model.pixel_decoder.register_forward_hook(lambda m, i, o: cap.__setitem__("pd", o))
    model.head.predictor.register_forward_hook(lambda m, i, o: cap.__setitem__("pred", o))
    masks_used = []
    for lyr in model.head.predictor.transformer_cross_attention_layers:
        lyr.register_forward_pre_hook(lambda m, a, kw: masks_used.append(kw["memory_mask"]), with_kwargs=True)
    dec_out = []
    for lyr in model.head.predictor.transformer_ffn_layers:
        lyr.register_forward_hook(lambda m, i, o: dec_out.append(o))
    images = [sis(i, *hw) for i in range(2)]
    x, _ = proc.preprocess(images, device=torch.device("cpu"), dtype=torch.float32)
    with torch.no_grad():
        out = model(x)
    B = x.shape[0]
    g = {"seed": np.int64(seed), "hw": np.array(hw), "pre_sample": strided_sample(x, 4096)}
    for k in ("res2", "res3", "res4", "res5"):
        g[f"{k}_sample"] = strided_sample(cap["bb"][k], 4096)
    _, cp8, cp16, cp32 = cap["cp"]
    for k, v in (("cp8", cp8), ("cp16", cp16), ("cp32", cp32), ("ffm", cap["ffm"])):
        g[f"{k}_sample"] = strided_sample(v, 4096)
    g["mask_features_sample"] = strided_sample(cap["pd"][0], 8192)
    for i, m in enumerate(masks_used):  # [B*heads, Q, Lk], identical across heads
        mm = m.view(B, 8, m.shape[1], m.shape[2])
        assert bool((mm == mm[:, :1]).all())
        g[f"attn_mask{i}"] = np.packbits(mm[:, 0].numpy(), axis=-1)
        g[f"attn_mask{i}_len"] = np.int64(m.shape[2])
    for i, o in enumerate(dec_out):  # [Q, B, C]
        g[f"dec{i}_sample"] = strided_sample(o.permute(1, 0, 2).contiguous(), 2048)
    g["cls_logits"] = cap["pred"]["pred_logits"].numpy()
    g["mask_logits_f16"] = cap["pred"]["pred_masks"].numpy().astype(np.float16)
    g["probs"] = out.logits.numpy()
    g["masks_sample"] = strided_sample(out.masks, 16384)
    for i in range(B):
        o1 = type(out)(masks=out.masks[i:i + 1], logits=out.logits[i:i + 1], loss=None)
        del areas[:]
        d = proc.postprocess(o1, [images[i]])[0].detections
        g[f"det{i}_conf"] = np.array([a.conf for a in d], np.float32)
        g[f"det{i}_cls"] = np.array([a.cls_id for a in d], np.int64)
        g[f"det{i}_bbox"] = np.array([a.bbox for a in d], np.int64).reshape(-1, 4)
        g[f"det{i}_area"] = np.array(areas, np.int64)
    np.savez_compressed(os.path.join(GOLDEN, tag + ".npz"), **g)
    print(tag, {k: v.shape for k, v in g.items() if hasattr(v, "shape") and v.ndim}, [len(g[f"det{i}_conf"]) for i in range(B)])


def mask_criterion_case():
    """Golden vectors of the REAL reference's MaskHungarianMatcher + mask SetCriterion (fai_mf/loss.py:345-723; bisenetformer/loss.py
    is identical) on seeded synthetic predictions / targets with deep supervision (main + 2 aux sets), num_points = 256: the
    reference's torch.rand draws (recorded in order - they are inputs of the restatement and of the kernels), the cost blocks
    handed to SciPy, the matched indices and the 9 weighted losses."""
    ref_import.install()
    import focoos.models.fai_mf.loss as L
    from focoos.models.fai_mf.ports import MaskFormerTargets

    from oracle.mask_criterion_oracle import synth_mask_predictions_and_targets

    out, labels, masks = synth_mask_predictions_and_targets(0)
    P = 256
    matcher = L.MaskHungarianMatcher(cost_class=2, cost_mask=5, cost_dice=5, num_points=P)
    crit = L.SetCriterion(num_classes=80, matcher=matcher, weight_dict={"loss_ce": 2, "loss_mask": 5, "loss_dice": 5}, losses=["labels", "masks"],
                          eos_coef=0.1, num_points=P, oversample_ratio=3.0, importance_sample_ratio=0.75)
    targets = [MaskFormerTargets(labels=l, masks=m) for l, m in zip(labels, masks)]
    rec, costs, idxs = [], [], []
    orig_rand, orig_lsa = torch.rand, L.linear_sum_assignment

    def spy_rand(*a, **k):
        t = orig_rand(*a, **k)
        rec.append(t.clone())
        return t

    def spy_lsa(c):
        costs.append(np.array(c, dtype=np.float32))
        r = orig_lsa(c)
        idxs.append((np.asarray(r[0]).copy(), np.asarray(r[1]).copy()))
        return r

    torch.manual_seed(5)
    torch.rand, L.linear_sum_assignment = spy_rand, spy_lsa
    try:
        ref = crit(out, targets)
    finally:
        torch.rand, L.linear_sum_assignment = orig_rand, orig_lsa
    g = {"num_points": np.int64(P), "n_rand": np.int64(len(rec)), "loss_names": np.array(sorted(ref)),
         "losses": np.array([float(ref[k]) for k in sorted(ref)], np.float64)}
    for i, t in enumerate(rec):
        g[f"rand_{i}"] = t.numpy()
    for i, (c, (a, b)) in enumerate(zip(costs, idxs)):   # order: set 0 image 0, set 0 image 1, set 1 image 0, ...
        g[f"cost_{i}"] = c
        g[f"pred_idx_{i}"] = a.astype(np.int32)
        g[f"tgt_idx_{i}"] = b.astype(np.int32)
    path = os.path.join(GOLDEN, "mask_criterion.npz")
    np.savez_compressed(path, **g)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB); losses {dict(zip(g['loss_names'], np.round(g['losses'], 4)))}")


def masks_to_xyxy_case():
    """The reference's own known-answer vectors for this path: tests/utils/test_vision.py:185-205 (test_masks_to_xyxy),
    evaluated with the reference's masks_to_xyxy (utils/vision.py:344-370) on extra seeded random masks as well."""
    ref_import.install()
    import importlib.util
    spec = importlib.util.spec_from_file_location("_ref_vision_m2x", os.path.join(ref_import.REFERENCE_ROOT, "focoos/utils/vision.py"))
    rs = np.random.RandomState(11)
    masks = np.zeros((16, 24, 40), bool)
    for m in masks:  # a random rectangle plus a few stray pixels
        y0, x0 = rs.randint(0, 20), rs.randint(0, 36)
        m[y0:y0 + rs.randint(1, 12), x0:x0 + rs.randint(1, 20)] = True
        m[rs.randint(0, 24, 2), rs.randint(0, 40, 2)] = True
    masks[3] = False
    try:
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        fn = mod.masks_to_xyxy
    except Exception:
        from focoos.utils.vision import masks_to_xyxy as fn
    np.savez_compressed(os.path.join(GOLDEN, "masks_to_xyxy.npz"), masks=np.packbits(masks, axis=-1), shape=np.array(masks.shape),
                        xyxy=fn(masks).astype(np.int64))
    print("masks_to_xyxy", fn(masks)[:4].tolist())


VARIANTS = ["fai-detr-m-coco", "fai-mf-m-coco-ins", "fai-mf-s-coco-ins", "fai-mf-l-ade", "fai-mf-m-ade", "bisenetformer-m-ade", "bisenetformer-s-ade"]


def variants_case(tag: str = "registry_variants", seed: int = 6, hw=(128, 160)):
    """The registry variants added in round 4, one compact fixture: per model the REAL reference's outputs on two seeded images - the
    discrete choices a bf16 engine is teacher-forced with (encoder top-k of RT-DETR / the boolean attention masks of every masked decoder
    layer), class probabilities, boxes or quarter-resolution mask logits (fp16 sample) - and the same outputs with nothing but the weights
    rounded to bf16 (the configuration's own sensitivity: what the GPU test gates the engine against)."""
    from focoos_amd.synth import synth_image_structured as sis

    g = {"seed": np.int64(seed), "hw": np.array(hw)}
    images = [sis(40 + i, *hw) for i in range(2)]
    for name in VARIANTS:
        info = ModelRegistry.get_model_info(name)
        cfg, fam = info["config"], info["model_family"]
        sd = synth_state_dict(cfg, seed=seed, family=fam)
        sdb = {k: (v.bfloat16().float() if v.dtype == torch.float32 and v.dim() >= 2 else v) for k, v in sd.items()}
        key = name.replace("-", "_")
        if fam == "fai_detr":
            model, proc, _ = ref_import.build_reference_detr(dict(cfg, resolution=hw[0]))
            outs = {}
            forced = None
            for wtag, weights in (("", sd), ("_w16", sdb)):
                model.load_state_dict(weights, strict=True)
                x, _ = proc.preprocess([np.ascontiguousarray(im[:hw[0], :hw[0]]) for im in images], device=torch.device("cpu"), dtype=torch.float32)
                orig_topk, calls = torch.topk, []

                def my_topk(*a, **k):
                    r = orig_topk(*a, **k)
                    calls.append(r)
                    if forced is not None and len(calls) == 1:       # second pass: the first pass's selection
                        return type(r)((a[0].gather(1, forced), forced)) if isinstance(r, tuple) else r
                    return r

                torch.topk = my_topk
                try:
                    with torch.no_grad():
                        out = model(x)
                finally:
                    torch.topk = orig_topk
                if forced is None:
                    forced = calls[0][1]
                    g[f"{key}.enc_topk"] = forced.numpy().astype(np.int32)
                g[f"{key}.probs{wtag}"] = out.logits.numpy().astype(np.float16)
                g[f"{key}.boxes{wtag}"] = out.boxes.numpy()
            continue
        import json

        build = ref_import.build_reference_mf if fam == "fai_mf" else ref_import.build_reference_bf
        ref_cfg = json.load(open(os.path.join(ref_import.REFERENCE_ROOT, f"focoos/model_registry/{name}.json")))["config"]
        assert {k: v for k, v in cfg.items() if k in ref_cfg} == ref_cfg
        model, proc, _ = build(ref_cfg)
        masks_used = []
        for lyr in model.head.predictor.transformer_cross_attention_layers:
            lyr.register_forward_pre_hook(lambda m, a, kw: masks_used.append(kw["memory_mask"]), with_kwargs=True)
        cap = {}
        model.head.predictor.register_forward_hook(lambda m, i, o: cap.__setitem__("pred", o))
        x, _ = proc.preprocess(images, device=torch.device("cpu"), dtype=torch.float32)
        model.load_state_dict(sd, strict=True)
        with torch.no_grad():
            out = model(x)
        B = x.shape[0]
        for i, m in enumerate(masks_used):
            mm = m.view(B, 8, m.shape[1], m.shape[2])
            g[f"{key}.attn_mask{i}"] = np.packbits(mm[:, 0].numpy(), axis=-1)
            g[f"{key}.attn_mask{i}_len"] = np.int64(m.shape[2])
        g[f"{key}.n_masks"] = np.int64(len(masks_used))
        g[f"{key}.probs"] = out.logits.numpy().astype(np.float16)
        g[f"{key}.mask_logits"] = cap["pred"]["pred_masks"][..., ::2, ::2].numpy().astype(np.float16)     # every other row / column
        # weights-only sensitivity: the same reference with bf16-rounded weights, ITS attention masks replaced by the fp32 run's
        forced_masks = list(masks_used)
        del masks_used[:]
        hooks = []
        for i, lyr in enumerate(model.head.predictor.transformer_cross_attention_layers):
            def pre(m, a, kw, i=i):
                kw = dict(kw)
                kw["memory_mask"] = forced_masks[i]
                return a, kw
            hooks.append(lyr.register_forward_pre_hook(pre, with_kwargs=True))
        model.load_state_dict(sdb, strict=True)
        with torch.no_grad():
            out_w = model(x)
        g[f"{key}.probs_w16"] = out_w.logits.numpy().astype(np.float16)
        g[f"{key}.mask_logits_w16"] = cap["pred"]["pred_masks"][..., ::2, ::2].numpy().astype(np.float16)
        for h in hooks:
            h.remove()
    np.savez_compressed(os.path.join(GOLDEN, tag + ".npz"), **g)
    print(tag, os.path.getsize(os.path.join(GOLDEN, tag + ".npz")) // 1024, "KiB", len(g), "arrays")


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    os.makedirs(GOLDEN, exist_ok=True)
    run_case("fai-detr-l-obj365", 0, [synth_image(0), synth_image_structured(1)], "detr_l_obj365_b2", threshold=0.5)
    # non-square inputs that are resized by the processor (base_processor.py:285-288)
    run_case("fai-detr-l-coco", 1, [synth_image_structured(2, 480, 600)], "detr_l_coco_resize")
    deform_core_case()
    criterion_case()
    mf_case()
    bf_case()
    mask_criterion_case()
    masks_to_xyxy_case()
    variants_case()


if __name__ == "__main__" and len(sys.argv) > 1:
    torch.set_num_threads(os.cpu_count() or 1)
    for name in sys.argv[1:]:  # e.g. `python scripts/make_golden.py mf_case masks_to_xyxy_case`
        globals()[name]()
elif __name__ == "__main__":
    main()
