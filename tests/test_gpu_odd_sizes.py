"""Input sizes that are NOT multiples of 32 through the engines (mask families: VERDICT r3 missing #5 / next #10; RT-DETR: round 5).

The reference hands such images over at their own size (``size_divisibility`` 0; fai_mf/processor.py:96, bisenetformer/processor.py:96)
and every stride-2 layer produces ceil(H/2): 3x3/s2/p1 convolutions and the 3x3/s2/p1 max-pool (nn/backbone/resnet.py:184-196,254),
``AvgPool2d(2, 2, ceil_mode=True)`` of the variant-d shortcuts with partial windows averaged over their valid taps (resnet.py:89-100),
STDC's ``AvgPool2d(3, 2, 1)`` skip (stdc.py:128), ``F.interpolate(size=...)`` to the lateral's size in the top-down paths with
non-integer ratios (nearest: fai_mf/modelling.py:364; bilinear: bisenetformer/modelling.py:186-212), attention masks resized to the
levels' sizes (modelling.py:104), the final bilinear up-sampling to (H, W) and the mask post-process on rows that are not whole
32-bit words.  The oracle is pinned against the real reference at such sizes in tests/test_oracle_vs_reference.py
(test_odd_sizes_oracle_matches_reference_live).  Same gates as the multiple-of-32 stage-parity tests."""
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


def nchw(nt):
    return nt.torch_view().float().cpu().permute(0, 3, 1, 2)


def _unpack(words, H, W):
    w = words.cpu().numpy().view(np.uint8)
    return np.unpackbits(w, axis=-1, bitorder="little").reshape(words.shape[0], H, -1)[..., :W].astype(bool)


# 150x200: H odd at stride 4 (38 -> 19 -> 10 -> 5), W a multiple of 8 only; 250x188: both ragged at every level; 97x130: odd from the start
@pytest.mark.parametrize("hw", [(150, 200), (250, 188), (97, 130)])
def test_mf_odd_size_matches_oracle(hw):
    """fai-mf-l-coco-ins (R101-vd, six pixel-decoder encoder layers on the ceil(H/32) x ceil(W/32) tokens, nine decoder layers)."""
    h, w = hw
    cfg = ModelRegistry.get_model_info("fai-mf-l-coco-ins")["config"]
    sd = synth_state_dict(cfg, 3, family="fai_mf")     # the seed of tests/golden/mf_l_coco_ins_b2.npz
    eng = MfEngine(cfg, sd, device=DEV, full_masks=True)
    images = [synth_image_structured(70 + i, h, w) for i in range(2)]
    col = {}
    with torch.no_grad():
        probs_o, masks_o = M.mf_forward(sd, cfg, get_torch_batch(images, None), collect=col)
    pl = eng.forward(torch.from_numpy(np.stack(images)).to(DEV), forced_attn=col["attn_masks"])
    torch.cuda.synchronize()
    for name in ("res2", "res3", "res4", "res5", "msf0", "msf1", "msf2", "fpn_s4", "mask_features"):
        assert tuple(nchw(pl.bufs[name]).shape) == tuple(col[name].shape), (name, tuple(nchw(pl.bufs[name]).shape), tuple(col[name].shape))
        assert rel_l2(nchw(pl.bufs[name]), col[name]) <= 2.5e-2, name
    B, L, Cc = col["enc_tokens"].shape
    assert rel_l2(pl.bufs["enc_tokens"].torch_view().float().cpu().reshape(B, L, Cc), col["enc_tokens"]) <= 2.5e-2
    for i in range(9):
        assert rel_l2(pl.bufs[f"dec{i}.out"].torch_view().float().cpu().reshape(B, -1, 256), col[f"dec{i}_out"]) <= 3e-2, i
    lo_o = torch.sigmoid(col["mask_logits"])
    assert tuple(pl.mask_probs.shape) == tuple(lo_o.shape)
    # the absolute gate of the multiple-of-32 test, or 2.5x what rounding nothing but the weights to bf16 does to the fp32 oracle at THIS
    # size (a 250x188 image leaves 8 x 6 tokens at stride 32: fewer keys per query, mask logits closer to 0)
    sdb = {k: (v.bfloat16().float() if v.dtype == torch.float32 and v.dim() >= 2 else v) for k, v in sd.items()}
    colb = {}
    with torch.no_grad():
        probs_w, _ = M.mf_forward(sdb, cfg, get_torch_batch(images, None), forced_attn=col["attn_masks"], collect=colb, upsample=False)
    dm_w, dp_w = float((torch.sigmoid(colb["mask_logits"]) - lo_o).abs().mean()), float((probs_w - probs_o).abs().max())
    dm, dp = float((pl.mask_probs.cpu() - lo_o).abs().mean()), float((pl.probs.cpu() - probs_o).abs().max())
    print(f"{hw}: engine mean |dmask| {dm:.4f} max |dprob| {dp:.4f}; bf16-weights-only oracle {dm_w:.4f} {dp_w:.4f}")
    # measured: (150, 200) 0.016 / 0.009, (250, 188) 0.0301 / 0.0105, (97, 130) 0.020 / 0.010 - a maximum over 2 x 100 x 80 probabilities
    assert dm <= max(1e-2, 2.5 * dm_w) and dp <= max(3e-2, 3.5 * dp_w), (dm, dm_w, dp, dp_w)
    assert ((pl.mask_probs.cpu() >= 0.5) == (lo_o >= 0.5)).float().mean() >= 0.99
    assert tuple(pl.masks.shape) == (2, 100, h, w) and (pl.masks.cpu() - masks_o).abs().mean() <= max(1e-2, 2.5 * dm_w)
    # device post-process (generic bilinear taps, rows of ceil(W/32) words) vs the oracle's restatement fed the ENGINE's outputs
    up = pl.masks.cpu()
    for b in range(2):
        s, l, q, boxes, bm = M.postprocess(pl.probs[b:b + 1].cpu(), up[b:b + 1], [(h, w)], cfg["mask_threshold"], cfg["threshold"],
                                           cfg["use_mask_score"])[0]
        n = int(pl.det_count[b])
        assert n >= 1 and abs(n - len(s)) <= 1      # a score exactly at the threshold may go either way
        if n == len(s):
            assert pl.det_query[b, :n].cpu().tolist() == q.tolist() and pl.det_labels[b, :n].cpu().tolist() == l.tolist()
            np.testing.assert_allclose(pl.det_scores[b, :n].cpu().numpy(), s.numpy(), atol=2e-4)
            got = _unpack(pl.mask_words[b, :n], h, w)
            # pixels whose up-sampled probability sits within 1e-5 of the threshold may differ between the fused tap and F.interpolate
            near = (up[b][q] - cfg["mask_threshold"]).abs().numpy() <= 1e-5
            assert ((got == np.asarray(bm).astype(bool)) | near).all()
            assert (np.abs(pl.det_boxes[b, :n].cpu().numpy() - np.asarray(boxes)) <= 1).all()


def _variants(pl):
    out = {}
    for m in pl.meta.values():
        if m.get("kind") == "conv":
            out[m["variant"]] = out.get(m["variant"], 0) + 1
    return dict(sorted(out.items(), key=lambda kv: -kv[1]))


@pytest.mark.parametrize("hw", [(600, 800), (500, 750)])
def test_mf_odd_size_production_routing(hw):
    """The sizes VERDICT r3 names (800x600 and 750x500, neither a multiple of 32), one image: large enough that the production kernels run
    (halo / k-plane 3x3, pointwise k-plane, pw_chain) on maps whose widths are odd (600x800: 25 columns at stride 32; 500x750: 47 at stride
    16, 24 x 16 at stride 32 from 63 x 47 rows) - the small cases above mostly route to the implicit-GEMM tiles."""
    h, w = hw
    cfg = ModelRegistry.get_model_info("fai-mf-l-coco-ins")["config"]
    sd = synth_state_dict(cfg, 3, family="fai_mf")
    eng = MfEngine(cfg, sd, device=DEV, full_masks=False)
    images = [synth_image_structured(75, h, w)]
    col = {}
    with torch.no_grad():
        probs_o, masks_o = M.mf_forward(sd, cfg, get_torch_batch(images, None), collect=col, upsample=False)
    pl = eng.forward(torch.from_numpy(np.stack(images)).to(DEV), forced_attn=col["attn_masks"])
    torch.cuda.synchronize()
    var = _variants(pl)
    print(f"{hw}: kernel variants under the oracle (launches per step):", var)
    assert any(k.startswith("conv3x3_kplane") or k.startswith("conv3x3_flat") for k in var) and any(k.startswith("pw_") for k in var), var
    for name in ("res2", "res3", "res4", "res5", "msf0", "msf1", "msf2", "fpn_s4", "mask_features"):
        assert tuple(nchw(pl.bufs[name]).shape) == tuple(col[name].shape), name
        assert rel_l2(nchw(pl.bufs[name]), col[name]) <= 2.5e-2, name
    for i in range(9):
        assert rel_l2(pl.bufs[f"dec{i}.out"].torch_view().float().cpu().reshape(1, -1, 256), col[f"dec{i}_out"]) <= 3e-2, i
    assert (pl.probs.cpu() - probs_o).abs().max() <= 3e-2
    assert (pl.mask_probs.cpu() - masks_o).abs().mean() <= 1e-2
    assert ((pl.mask_probs.cpu() >= 0.5) == (masks_o >= 0.5)).float().mean() >= 0.99
    assert int(pl.det_count[0]) >= 1 and tuple(pl.mask_words.shape[-2:]) == (h, (w + 31) // 32)


@pytest.mark.parametrize("hw", [(150, 200), (250, 188)])
def test_mf_stdc_variant_odd_size_matches_oracle(hw):
    """fai-mf-m-ade: STDC-2 backbone (3x3/s2 depthwise `avd_layer` and AvgPool2d(3,2,1) skips at ceil sizes), semantic post-process."""
    h, w = hw
    cfg = ModelRegistry.get_model_info("fai-mf-m-ade")["config"]
    sd = synth_state_dict(cfg, 13, family="fai_mf")
    eng = MfEngine(cfg, sd, device=DEV, full_masks=False)
    images = [synth_image_structured(80 + i, h, w) for i in range(2)]
    col = {}
    with torch.no_grad():
        probs_o, masks_o = M.mf_forward(sd, cfg, get_torch_batch(images, None), collect=col, upsample=False)
    pl = eng.forward(torch.from_numpy(np.stack(images)).to(DEV), forced_attn=col["attn_masks"])
    torch.cuda.synchronize()
    for name in ("res2", "res3", "res4", "res5", "msf0", "msf1", "msf2", "fpn_s4", "mask_features"):
        assert tuple(nchw(pl.bufs[name]).shape) == tuple(col[name].shape), name
        assert rel_l2(nchw(pl.bufs[name]), col[name]) <= 2.5e-2, name
    # decoder gates relative to this configuration's measured bf16-weights-only sensitivity (see tests/test_gpu_mf.py, the ADE variants)
    sdb = {k: (v.bfloat16().float() if v.dtype == torch.float32 and v.dim() >= 2 else v) for k, v in sd.items()}
    colb = {}
    with torch.no_grad():
        probs_w, masks_w = M.mf_forward(sdb, cfg, get_torch_batch(images, None), forced_attn=col["attn_masks"], collect=colb, upsample=False)
    for i in range(3):
        e_w = rel_l2(colb[f"dec{i}_out"], col[f"dec{i}_out"])
        assert rel_l2(pl.bufs[f"dec{i}.out"].torch_view().float().cpu().reshape(2, -1, 256), col[f"dec{i}_out"]) <= max(3e-2, 2.5 * e_w), i
    dp, dp_w = float((pl.probs.cpu() - probs_o).abs().max()), float((probs_w - probs_o).abs().max())
    dm, dm_w = float((pl.mask_probs.cpu() - masks_o).abs().mean()), float((masks_w - masks_o).abs().mean())
    assert dp <= max(3e-2, 2.5 * dp_w) and dm <= max(1e-2, 2.5 * dm_w), (dp, dp_w, dm, dm_w)
    up = torch.nn.functional.interpolate(pl.mask_probs.cpu(), size=(h, w), mode="bilinear", align_corners=False)
    for b in range(2):
        s, l, q, boxes, bm = M.postprocess(pl.probs[b:b + 1].cpu(), up[b:b + 1], [(h, w)], cfg["mask_threshold"], cfg["threshold"],
                                           cfg["use_mask_score"], predict_all_pixels=True)[0]
        n = int(pl.det_count[b])
        assert n >= 1 and abs(n - len(s)) <= 1
        if n == len(s):
            assert pl.det_labels[b, :n].cpu().tolist() == l.tolist()
            np.testing.assert_allclose(pl.det_scores[b, :n].cpu().numpy(), s.numpy(), atol=2e-5)
            got = _unpack(pl.mask_words[b, :n], h, w)
            win = pl.winner[b].cpu().numpy()
            assert win.shape == (h, w)
            for j, qq in enumerate(pl.det_query[b, :n].cpu().tolist()):      # the bit-packed masks ARE the winner map's level sets
                assert (got[j] == (win == qq)).all()
            assert float((got == np.asarray(bm).astype(bool)).mean()) >= 0.999   # exact ties between two queries may go either way


@pytest.mark.parametrize("hw", [(150, 200), (250, 188), (500, 750)])
def test_bf_odd_size_matches_oracle(hw):
    """bisenetformer-l-ade: bilinear F.interpolate to the (ceil) size of the next finer level in the ContextPath."""
    h, w = hw
    cfg = ModelRegistry.get_model_info("bisenetformer-l-ade")["config"]
    sd = synth_state_dict(cfg, 11, family="bisenetformer")
    eng = BfEngine(cfg, sd, device=DEV, full_masks=False)
    images = [synth_image_structured(90 + i, h, w) for i in range(2)]
    col = {}
    with torch.no_grad():
        probs_o, masks_o = BF.bf_forward(sd, cfg, get_torch_batch(images, None), collect=col, upsample=False)
    pl = eng.forward(torch.from_numpy(np.stack(images)).to(DEV), forced_attn=col["attn_masks"])
    torch.cuda.synchronize()
    for name in ("res2", "res3", "res4", "res5", "cp32", "cp16", "cp8", "ffm", "mask_features"):
        assert tuple(nchw(pl.bufs[name]).shape) == tuple(col[name].shape), name
        assert rel_l2(nchw(pl.bufs[name]), col[name]) <= 2.5e-2, name
    for i in range(6):
        assert rel_l2(pl.bufs[f"dec{i}.out"].torch_view().float().cpu().reshape(2, -1, 256), col[f"dec{i}_out"]) <= 3e-2, i
    assert (pl.probs.cpu() - probs_o).abs().max() <= 3e-2
    assert (pl.mask_probs.cpu() - masks_o).abs().mean() <= 1e-2
    assert ((pl.mask_probs.cpu() >= 0.5) == (masks_o >= 0.5)).float().mean() >= 0.99
    assert tuple(pl.winner.shape) == (2, h, w)
    up = torch.nn.functional.interpolate(pl.mask_probs.cpu(), size=(h, w), mode="bilinear", align_corners=False)
    win_o = (pl.probs.cpu().max(-1).values.view(2, -1, 1, 1) * up).argmax(dim=1)
    assert float((pl.winner.cpu().long() == win_o).float().mean()) >= 0.999       # fed the engine's own outputs: ties only


@pytest.mark.parametrize("name", ["fai-mf-l-coco-ins", "bisenetformer-l-ade"])
def test_model_manager_surface_at_odd_size(name):
    """The standalone surface (ModelManager.get -> FocoosModel.infer_batch: fused detect path, packed D2H, host tail) on 150x200 images:
    boxes inside the image, one bit-packed mask row of ceil(W/32) words per image row."""
    from focoos_amd.model import ModelManager

    fm = ModelManager.get(name, seed=3)
    h, w = 150, 200
    images = [synth_image_structured(50 + i, h, w) for i in range(2)]
    dets = fm.infer_batch(images)
    assert len(dets) == 2 and all(len(d) >= 1 for d in dets)
    for d in dets:
        for det in d.detections:
            x0, y0, x1, y1 = det.bbox
            assert 0 <= x0 <= x1 < w and 0 <= y0 <= y1 < h and det.mask is not None


@pytest.mark.parametrize("hw", [(200, 232), (250, 188), (330, 270)])
def test_detr_odd_size_matches_oracle(hw):
    """RT-DETR at sizes that are not multiples of 32 (round 5, ADVICE r4: ragged training batches are padded to the batch maximum and the
    reference accepts them - its encoder resizes with F.interpolate(size=...) in both directions, modelling.py:334,342): levels of
    ceil(H/8) / ceil(H/16) / ceil(H/32) rows, anchors / valid mask / sampling shapes from those, teacher-forced on the oracle's query
    selection (at these sizes many anchors are invalid and tie at the masked score; the oracle equals the real reference here when given
    the reference's selection: tests/test_oracle_vs_reference.py::test_detr_odd_sizes_oracle_matches_reference_live).  Gates of
    tests/test_gpu_e2e.py::test_stage_parity_teacher_forced."""
    from focoos_amd.engine import DetrEngine
    from oracle import detr_oracle as O

    cfg = ModelRegistry.get_model_info("fai-detr-l-coco")["config"]
    sd = synth_state_dict(cfg, 17)
    imgs = [synth_image_structured(5 + i, *hw) for i in range(2)]
    col = {}
    with torch.no_grad():
        probs_o, boxes_o = O.detr_forward(sd, cfg, get_torch_batch(imgs, hw), collect=col)
    eng = DetrEngine(cfg, sd, device=DEV)
    x = torch.from_numpy(np.stack(imgs)).to(DEV)
    pl = eng.forward(x, forced_topk=col["topk_ind"].long(), use_graph=False)
    torch.cuda.synchronize()
    for k in ("res3", "res4", "res5", "enc_s32", "enc_s16", "enc_s8"):
        assert tuple(nchw(pl.bufs[k]).shape) == tuple(col[k].shape), (k, nchw(pl.bufs[k]).shape, col[k].shape)
        e = rel_l2(nchw(pl.bufs[k]), col[k])
        assert e < 2e-2, (k, e)
    assert rel_l2(pl.bufs["memory"].t.float().cpu().view(2, -1, 256), col["memory"]) < 2e-2
    for i in range(6):
        e = rel_l2(pl.bufs[f"dec{i}.out"].t.float().cpu().view(2, 300, 256), col[f"dec{i}_out"])
        assert e < 4e-2, (i, e)
    assert (pl.probs.cpu() - probs_o).abs().max().item() <= 2.5e-2
    assert (pl.boxes.cpu() - boxes_o).abs().max().item() <= 7e-3
    # free-running (graph replay, no teacher forcing): the same plan, finite outputs, selection overlapping the oracle's among the tokens
    # whose score is off the masked value (ties among invalid anchors are broken arbitrarily by both)
    pl2 = eng.forward(x)
    torch.cuda.synchronize()
    assert torch.isfinite(pl2.probs).all() and torch.isfinite(pl2.boxes).all()
    sc = col["enc_scores"]
    for b in range(2):
        thr = sc[b].topk(300).values[-1]
        sure = set(torch.nonzero(sc[b] > thr + 0.2).flatten().tolist())
        assert sure <= set(pl2.enc_topk[b].cpu().tolist())


def test_detr_refuses_inputs_with_fewer_tokens_than_queries():
    """64 x 64 gives 8*8 + 4*4 + 2*2 = 84 encoder tokens for 300 queries: the reference's torch.topk raises ("selected index k out of
    range", modelling.py:1219); the engine raises its own error instead of selecting out of range."""
    from focoos_amd import _lib
    from focoos_amd.engine import DetrEngine

    cfg = ModelRegistry.get_model_info("fai-detr-l-coco")["config"]
    eng = DetrEngine(cfg, synth_state_dict(cfg, 3), device=DEV)
    with pytest.raises(_lib.FocoosAmdError, match="fewer than the 300 queries"):
        eng.forward(torch.zeros(1, 64, 64, 3, dtype=torch.uint8, device=DEV))
    with pytest.raises(_lib.FocoosAmdError):
        eng.forward(torch.zeros(1, 20, 200, 3, dtype=torch.uint8, device=DEV))


@pytest.mark.parametrize("name", ["fai-mf-l-coco-ins", "bisenetformer-l-ade", "bisenetformer-m-ade", "fai-mf-m-ade"])
def test_masked_decoder_row_chains_match_per_op_launches(monkeypatch, name):
    """The decoder's row-local layers as fx_row_chain programs (engine_maskdec._masked_decoder_row_chains: from 1 000 rows on in production, i.e.
    BiSeNetFormer bs >= 20 per part) against the one-launch-per-layer form AND the oracle at the test's small batch: MaskFormer-L (FFN of 2 048
    hidden channels = two 1 024-channel halves), BiSeNetFormer-L, BiSeNetFormer-M (96-wide mask embedding padded to 128), fai-mf-m-ade (512-wide
    FFN, three layers).  Attention masks teacher-forced, so both forms see the same discrete inputs."""
    info = ModelRegistry.get_model_info(name)
    cfg, fam = info["config"], info["model_family"]
    sd = synth_state_dict(cfg, 9, family=fam)
    h, w = 160, 192
    images = [synth_image_structured(30 + i, h, w) for i in range(2)]
    col = {}
    with torch.no_grad():
        if fam == "fai_mf":
            probs_o, masks_o = M.mf_forward(sd, cfg, get_torch_batch(images, None), collect=col, upsample=False)
        else:
            probs_o, masks_o = BF.bf_forward(sd, cfg, get_torch_batch(images, None), collect=col, upsample=False)
    x = torch.from_numpy(np.stack(images)).to(DEV)
    outs = {}
    for mode, min_rows in (("chains", "1"), ("per_op", "0")):
        monkeypatch.setenv("FX_MASKDEC_ROW_CHAIN_MIN_ROWS", min_rows)
        eng = (MfEngine if fam == "fai_mf" else BfEngine)(cfg, sd, device=DEV, full_masks=False)
        pl = eng.forward(x, forced_attn=col["attn_masks"])
        torch.cuda.synchronize()
        n_chain = sum(1 for m in pl.meta.values() if m.get("variant") == "row_chain")
        assert (n_chain == 2 * eng.nl + 1) if mode == "chains" else (n_chain == 0), (mode, n_chain)
        outs[mode] = (pl.probs.cpu().clone(), pl.mask_probs.cpu().clone(),
                      [pl.bufs[f"dec{i}.out"].torch_view().float().cpu().reshape(2, -1, 256).clone() for i in range(eng.nl)], int(pl.det_count.sum()))
    (pc, mc, dc, nc_), (pp, mp, dp_, np_) = outs["chains"], outs["per_op"]
    for i, (a, b) in enumerate(zip(dc, dp_)):
        assert rel_l2(a, b) <= 1e-2, (i, rel_l2(a, b))                      # same arithmetic up to the order of fp32 sums and one bf16 rounding
        assert rel_l2(a, col[f"dec{i}_out"]) <= max(3e-2, 1.5 * rel_l2(b, col[f"dec{i}_out"])), i
    assert (pc - pp).abs().max() <= 2e-2 and (mc - mp).abs().mean() <= 2e-3
    assert (mc - masks_o).abs().mean() <= max(1e-2, 1.5 * float((mp - masks_o).abs().mean()))
    assert abs(nc_ - np_) <= 2
