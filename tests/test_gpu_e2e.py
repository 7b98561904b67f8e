"""End-to-end parity of the HIP engine (bf16 MFMA path) against the CPU fp32 oracle and the golden
fixtures produced by the real reference, on a real MI355X.

Tolerances (bf16 activations/weights, fp32 accumulate, fp32 scores/boxes; stated per north_star):
  * intermediate feature maps: relative L2 error <= 2e-2 vs the fp32 oracle;
  * with the encoder top-k query set teacher-forced to the reference's: |dprob| <= 3e-2,
    |dbox| <= 8e-3 (normalised units), per-query argmax class identical wherever the top-2 margin > 6e-2;
  * free-running: encoder class scores within 1e-1; the top-300 token set and the post-process (query, class) pairs are
    EXACTLY the reference's for every candidate further than 2 x tolerance from the selection boundary, and the set
    overlaps the real reference's by >= 90 % (the boundary tokens themselves are ill-conditioned under any rounding
    change: the reference under bf16 autocast changes the set too — SURVEY §0.8);
  * integer outputs (class ids, query ids, int32 pixel boxes): bit-exact given equal float inputs
    (tests/test_gpu_kernels.py::test_head_out_and_postprocess_vs_oracle), and equal to the reference's for every
    detection whose score margin to its neighbours and to the threshold exceeds the stated prob tolerance.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from focoos_amd.model import FAIDetr, ModelManager  # noqa: E402
from focoos_amd.registry import ModelRegistry  # noqa: E402
from focoos_amd.synth import synth_image, synth_image_structured, synth_state_dict  # noqa: E402
from oracle import detr_oracle as O  # noqa: E402
from tests.helpers import load_golden, rel_l2, strided_sample  # noqa: E402

DEV = "cuda:0"


@pytest.fixture(scope="module")
def setup():
    assert torch.cuda.is_available()
    g = load_golden("detr_l_obj365_b2.npz")
    cfg = ModelRegistry.get_model_info("fai-detr-l-obj365")["config"]
    sd = synth_state_dict(cfg, int(g["seed"]))
    model = FAIDetr(cfg, device=DEV, seed=int(g["seed"]))
    images = [synth_image(0), synth_image_structured(1)]
    x_u8 = torch.from_numpy(np.stack(images)).to(DEV)
    forced = torch.from_numpy(g["enc_topk"]).long()
    torch.set_num_threads(max(1, torch.get_num_threads()))
    col = {}
    with torch.no_grad():
        xo = O.get_torch_batch(images, (640, 640))
        probs_o, boxes_o = O.detr_forward(sd, cfg, xo, forced_topk=forced, collect=col)
    return g, cfg, sd, model, images, x_u8, forced, probs_o, boxes_o, col


def nchw(nt):
    return nt.torch_view().float().cpu().permute(0, 3, 1, 2)


def test_stage_parity_teacher_forced(setup):
    g, cfg, sd, model, images, x_u8, forced, probs_o, boxes_o, col = setup
    out = model.forward(x_u8, forced_topk=forced, use_graph=False)
    torch.cuda.synchronize()
    pl = model.last_plan
    for k in ("res3", "res4", "res5", "enc_s32", "enc_s16", "enc_s8"):
        e = rel_l2(nchw(pl.bufs[k]), col[k])
        assert e < 2e-2, (k, e)
    mem = pl.bufs["memory"].t.float().cpu().view(2, -1, 256)
    assert rel_l2(mem, col["memory"]) < 2e-2
    # golden (real reference) samples of the same stages
    for k in ("res5", "enc_s8"):
        got = strided_sample(nchw(pl.bufs[k]), 4096)
        assert np.linalg.norm(got - g[f"{k}_sample"]) / np.linalg.norm(g[f"{k}_sample"]) < 2.5e-2, k
    assert rel_l2(pl.bufs["target"].t.float().cpu().view(2, 300, 256), col["target"]) < 3e-2
    assert (pl.ref_unact.cpu().view(2, 300, 4) - col["ref_unact"]).abs().max() < 5e-2
    for i in range(6):
        e = rel_l2(pl.bufs[f"dec{i}.out"].t.float().cpu().view(2, 300, 256), col[f"dec{i}_out"])
        assert e < 4e-2, (i, e)
        assert (pl.refs[i + 1].cpu().view(2, 300, 4) - col[f"dec{i}_ref"]).abs().max() < 1e-2, i
    dp = (out.logits.cpu() - probs_o).abs().max().item()
    db = (out.boxes.cpu() - boxes_o).abs().max().item()
    assert dp <= TOL_PROB and db <= TOL_BOX, (dp, db)
    # vs the reference's golden outputs
    assert np.abs(out.boxes.cpu().numpy() - g["boxes"]).max() <= TOL_BOX
    assert np.abs(out.logits.cpu().max(-1).values.numpy() - g["probs_max"]).max() <= TOL_PROB
    top2 = probs_o.topk(2, -1).values
    safe = (top2[..., 0] - top2[..., 1]) > 2 * TOL_PROB
    assert (out.logits.cpu().argmax(-1)[safe].numpy() == g["probs_argmax"][safe.numpy()]).all()


# Stated tolerances of the free-running (no teacher forcing) comparison, bf16 engine vs fp32 oracle:
# Measured on MI355X (scripts/dev/parity_probe.py, profiles/r02_parity_probe.txt): stage rel-L2 0.6 % (backbone) -> 0.85 % (encoder) -> 1.0 %
# (decoder); encoder scores max |d| 0.067-0.069 (mean 0.013, score std 0.53); final logits max |d| 0.09-0.11 at std 1.9, i.e.
# probabilities max |d| 0.016-0.024 (mean 2e-4) and boxes max |d| 0.0046-0.0055 - the maxima over 219 000 values move by +-30 % between
# equivalent kernel selections (fused / unfused layers), so the gates are the measured maxima x 1.25-1.45, not x 1.0.
# Round 3: gates at measured x 1.25 (profiles/r03_parity_probe.txt: scores 0.068, probabilities 0.014-0.024, boxes
# 0.0049-0.0055).  The score tolerance cannot go to 2e-2 by keeping `memory` in fp32: rounding the oracle's fp32 memory to bf16 moves the
# scores by 0.011 at most, while a 0.3-0.9 % relative perturbation of memory - the error the bf16 backbone + encoder arrive with - moves them
# by 0.10-0.27 (scripts/dev/score_sensitivity.py, profiles/r03_score_sensitivity.txt): the score error is upstream of the head's input.
TOL_SCORE = 8.5e-2  # encoder class-score logits feeding the top-300 selection
TOL_PROB = 2.5e-2   # final class probabilities
TOL_BOX = 7e-3      # final boxes, normalised units (= 4.5 px at 640)


def test_free_running_index_parity_outside_margins(setup):
    """north_star: "bit-exact class/box indices".  Both top-k selections are discontinuous, so the claim is made where it can hold:
    every candidate whose fp32 score is further than 2 x tolerance from the selection boundary must be selected (or rejected) exactly as
    the reference does - for the encoder's top-300 of 8400 tokens and for the post-process' (class, query) pairs above the threshold."""
    g, cfg, sd, model, images, x_u8, forced, probs_o, boxes_o, col = setup
    thr = float(g["threshold"])
    pl = model.detect(x_u8, threshold=thr)
    torch.cuda.synchronize()
    sc_e, sc_o = pl.enc_scores.cpu(), col["enc_scores"]
    ds = (sc_e - sc_o).abs().max().item()
    assert ds <= TOL_SCORE, f"encoder score error {ds:.4f}"
    mine_all = pl.enc_topk.cpu().long()
    overlap = []
    for i in range(2):
        mine = set(mine_all[i].tolist())
        assert len(mine) == 300
        cutoff = sc_o[i].topk(300).values[-1].item()
        must_in = set(torch.nonzero(sc_o[i] > cutoff + 2 * TOL_SCORE).flatten().tolist())
        must_out = set(torch.nonzero(sc_o[i] < cutoff - 2 * TOL_SCORE).flatten().tolist())
        assert len(must_in) >= 75 and len(must_out) >= 6500, (len(must_in), len(must_out))   # the margin test is not vacuous
        assert must_in <= mine, sorted(must_in - mine)[:8]
        assert not (mine & must_out), sorted(mine & must_out)[:8]
        ref = set(g["enc_topk"][i].tolist())      # the REAL reference's set (golden): same statement, plus the overlap as a number
        assert (must_in <= ref) and not (ref & must_out)
        overlap.append(len(mine & ref))
        print(f"free-running top-300 overlap with the real reference, image {i}: {overlap[-1]}/300; outside the +-{2 * TOL_SCORE:.2f} band: "
              f"{len(must_in)} must-select, {len(must_out)} must-reject, all honoured")
        assert overlap[-1] >= 275, overlap
        # within the set the engine's ORDER follows its own scores exactly (descending, ties to the lower index)
        v = sc_e[i][mine_all[i]]
        assert (v[:-1] >= v[1:]).all()
    # ---- post-process indices: oracle on the engine's own query set (the selection itself was checked above)
    with torch.no_grad():
        p_o, b_o = O.detr_forward(sd, cfg, O.get_torch_batch(images, (640, 640)), forced_topk=mine_all)
    dp = (pl.probs.cpu() - p_o).abs().max().item()
    db = (pl.boxes.cpu() - b_o).abs().max().item()
    assert dp <= TOL_PROB and db <= TOL_BOX, (dp, db)
    K = p_o.shape[-1]
    for i in range(2):
        n = int(pl.det_count[i])
        s = pl.det_scores[i, :n].cpu()
        assert (s[:-1] >= s[1:]).all() and (s > thr).all()
        pairs = {(int(q), int(c)): j for j, (q, c) in enumerate(zip(pl.det_queries[i, :n].cpu().tolist(), pl.det_labels[i, :n].cpu().tolist()))}
        assert len(pairs) == n
        flat = p_o[i].flatten()
        sure = torch.nonzero(flat > thr + 2 * TOL_PROB).flatten().tolist()
        never = set(torch.nonzero(flat < thr - 2 * TOL_PROB).flatten().tolist())
        assert len(sure) >= 10
        for f in sure:                                    # every confident reference detection: same (query, class), score and box within tolerance
            q, c = divmod(f, K)
            assert (q, c) in pairs, (i, q, c, float(flat[f]))
            j = pairs[(q, c)]
            assert abs(float(s[j]) - float(flat[f])) <= TOL_PROB
            ref_box = torch.round(b_o[i, q] * 640.0)
            assert (pl.det_boxes[i, j].cpu().float() - ref_box).abs().max().item() <= 640 * TOL_BOX + 1
        assert not any((q * K + c) in never for (q, c) in pairs)   # and nothing the reference puts clearly below the threshold
        # top_k = 300 truncation (processor.py:147): applies only if > 300 pairs pass - not the case for these inputs
        assert n < 300


def test_free_running_detections_vs_reference_golden(setup):
    """Against the REAL reference's detections (golden, its own free-running query set): every confident reference detection is found with
    the same class, score within tolerance and box within a few pixels; counts agree within the band the score tolerance allows."""
    g, cfg, sd, model, images, x_u8, forced, probs_o, boxes_o, col = setup
    thr = float(g["threshold"])
    pl = model.detect(x_u8, threshold=thr)
    torch.cuda.synchronize()
    tol = 2 * TOL_PROB
    for i in range(2):
        n_ref = int(g["det_count"][i])
        n = int(pl.det_count[i])
        s = pl.det_scores[i, :n].cpu().numpy()
        lab = pl.det_labels[i, :n].cpu().numpy()
        box = pl.det_boxes[i, :n].cpu().numpy()
        rs, rl, rb = g["det_scores"][i, :n_ref], g["det_labels"][i, :n_ref], g["det_boxes"][i, :n_ref]
        confident = np.where(rs > thr + 2 * tol)[0]
        assert len(confident) > 10
        hit = 0
        for j in confident:
            cand = np.where((lab == rl[j]) & (np.abs(s - rs[j]) < tol))[0]
            if any(np.abs(box[c] - rb[j]).max() <= 640 * TOL_BOX + 1 for c in cand):
                hit += 1
        # a reference detection can only be missed when its QUERY is one of the boundary tokens the two top-300 sets disagree on
        assert hit >= len(confident) - 4, (hit, len(confident))
        assert abs(n - n_ref) <= 0.15 * n_ref + 5


def test_graph_replay_equals_eager_and_is_deterministic(setup):
    g, cfg, sd, model, images, x_u8, *_ = setup
    a = model.forward(x_u8, use_graph=False)
    b = model.forward(x_u8, use_graph=True)
    c = model.forward(x_u8, use_graph=True)
    torch.cuda.synchronize()
    assert torch.equal(a.logits, b.logits) and torch.equal(a.boxes, b.boxes)
    assert torch.equal(b.logits, c.logits)


def test_reference_input_contract_nchw_float(setup):
    """BaseModelNN.forward(images[B,3,H,W] float 0..255) (modelling.py:1344-1349) gives the same result as the fused u8 path."""
    g, cfg, sd, model, images, x_u8, *_ = setup
    a = model.forward(x_u8, use_graph=False)
    b = model.forward(x_u8.permute(0, 3, 1, 2).float(), use_graph=False)
    torch.cuda.synchronize()
    assert torch.equal(a.logits, b.logits)


def test_processor_postprocess_and_export_postprocess_agree(setup):
    """Processor.postprocess on the model output and Processor.export_postprocess on the raw [boxes, logits] arrays an exported
    runtime would return (numpy, host memory; fai_detr/processor.py:219-240) give the same detections."""
    g, cfg, sd, model, images, x_u8, *_ = setup
    fm = ModelManager.get("fai-detr-l-obj365", device=DEV, seed=int(g["seed"]))
    out = fm.model.forward(x_u8, use_graph=False)
    torch.cuda.synchronize()
    a = fm.processor.postprocess(out, list(images), threshold=0.3)
    b = fm.processor.export_postprocess([out.boxes.cpu().numpy(), out.logits.cpu().numpy()], list(images), threshold=0.3)
    assert len(a) == len(b) == len(images) and sum(len(d) for d in a) > 0
    for da, db in zip(a, b):
        assert [(d.cls_id, d.bbox, d.conf) for d in da.detections] == [(d.cls_id, d.bbox, d.conf) for d in db.detections]


def test_resize_case_against_reference_golden():
    g = load_golden("detr_l_coco_resize.npz")
    fm = ModelManager.get("fai-detr-l-coco", device=DEV, seed=int(g["seed"]))
    img = synth_image_structured(2, 480, 600)
    x, _ = fm.processor.preprocess([img], device=fm.model.device)
    assert x.dtype == torch.float32 and tuple(x.shape) == (1, 640, 640, 3)
    got = strided_sample(x.permute(0, 3, 1, 2), 4096)
    assert np.abs(got - g["pre_sample"]).max() < 2e-3
    forced = torch.from_numpy(g["enc_topk"]).long()
    out = fm.model.forward(x, forced_topk=forced, use_graph=False)
    torch.cuda.synchronize()
    assert np.abs(out.boxes.cpu().numpy() - g["boxes"]).max() <= TOL_BOX
    assert np.abs(out.logits.cpu().max(-1).values.numpy() - g["probs_max"]).max() <= TOL_PROB
    dets = fm.infer_batch([img], threshold=float(g["threshold"]))[0]
    n_ref = int(g["det_count"][0])
    assert abs(len(dets) - n_ref) <= 4
    if len(dets):
        d0 = dets.detections[0]
        assert isinstance(d0.bbox, list) and len(d0.bbox) == 4 and isinstance(d0.cls_id, int) and isinstance(d0.conf, float)
        assert dets.latency is not None


def test_state_dict_roundtrip_and_loud_failures(setup):
    g, cfg, sd, model, *_ = setup
    st = model.state_dict()
    assert list(st) == list(sd) and all(torch.equal(st[k], sd[k]) for k in sd)
    res = model.load_state_dict({("module." + k): v for k, v in st.items()}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    bad = dict(st)
    bad.pop("head.predictor.enc_score_classifier.bias")
    with pytest.raises(RuntimeError):
        model.load_state_dict(bad, strict=True)
    model.train(True)  # nn.Module.train only flips the flag; the training graph lives behind FocoosModel.train (trainer.run_train)
    try:
        with pytest.raises(NotImplementedError):
            model.forward(setup[5][:1])
    finally:
        model.eval()


def test_batch_parts_equal_single_plan(setup, flat_small_shapes):
    """_MultiPlan (experimental, off by default): the batch cut into two parts with their own buffers, writing contiguous batch
    slices of the same output tensors, gives bit-identical results to the single plan when the parts run one after the other
    (use_graph=False).  Their CONCURRENT replay is not asserted: it is the unsafe experiment documented in engine._MultiPlan."""
    g, cfg, sd, model, images, x_u8, *_ = setup
    x8 = torch.cat([x_u8] * 4, 0)  # 8 images -> 2 parts of 4
    eng = model.engine
    outs = []
    for ns in (1, 2):
        pl = eng.plan(8, 640, 640, False, ns)
        assert getattr(pl, "n", 1) == ns
        with torch.cuda.stream(eng.stream):
            pl.input.copy_(x8)
            pl.sizes.copy_(torch.tensor([[640, 640]] * 8, dtype=torch.int32))
            pl.run(eng.stream.cuda_stream, 0.3, None, ns == 1)
            pl.run(eng.stream.cuda_stream, 0.3, None, ns == 1)
        eng.stream.synchronize()
        outs.append([getattr(pl, k).clone() for k in ("probs", "boxes", "det_scores", "det_labels", "det_boxes", "det_count")])
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert torch.equal(outs[0][0][:2], outs[0][0][2:4])  # the repeated images give repeated rows


def test_full_size_batch_properties(setup, flat_small_shapes):
    """BASELINE configs[1] at full size (bs=32, 640x640): size-independent properties of the whole path.
    (i) every image is computed independently of its batch position / neighbours: a permuted batch gives the permuted
    result bit-for-bit, and image i of the bs=32 step equals the same image run alone; (ii) replay is idempotent;
    (iii) post-process invariants: scores sorted descending, counts = #scores > threshold, labels in range, x2>=x1, y2>=y1.
    Kernel routing is pinned (flat_small_shapes): in production a layer's kernel is chosen by M = B*H*W, so a bs=1 and a bs=32
    run of the same image agree to bf16 rounding, not bit-for-bit; with one routing the batch position must not matter at all."""
    from focoos_amd.synth import synth_image_structured as sis

    g, cfg, sd, model, *_ = setup
    eng = model.engine
    imgs = torch.from_numpy(np.stack([sis(100 + i) for i in range(32)])).to(DEV)
    perm = torch.randperm(32, generator=torch.Generator().manual_seed(0))

    def run(x, thr=0.3):
        pl = eng.forward(x, threshold=thr)
        torch.cuda.synchronize()
        return {k: getattr(pl, k).clone() for k in ("probs", "boxes", "det_scores", "det_labels", "det_boxes", "det_count")}

    a = run(imgs)
    b = run(imgs[perm.to(DEV)].contiguous())
    for k in a:
        assert torch.equal(a[k][perm.to(DEV)], b[k]), k
    a2 = run(imgs)
    for k in a:
        assert torch.equal(a[k], a2[k]), k
    one = run(imgs[5:6].contiguous())
    assert torch.equal(one["probs"][0], a["probs"][5]) and torch.equal(one["det_boxes"][0], a["det_boxes"][5])
    n = a["det_count"].cpu()
    s = a["det_scores"].cpu()
    assert (s[:, :-1] >= s[:, 1:]).all()
    assert ((s > 0.3).sum(1) == n).all() and int(n.max()) > 0
    K = a["probs"].shape[-1]
    for i in range(32):
        ni = int(n[i])
        assert (a["det_labels"][i, :ni] >= 0).all() and (a["det_labels"][i, :ni] < K).all()
        bx = a["det_boxes"][i, :ni].cpu()
        assert (bx[:, 2] >= bx[:, 0]).all() and (bx[:, 3] >= bx[:, 1]).all()  # (the reference does not clip boxes to the image)


def test_checkpoint_file_roundtrip(setup, tmp_path):
    """N4 (checkpoint bridge): a reference-format checkpoint file ({"model": state_dict} with the reference's key names, as
    written by the reference trainer) loads through ModelInfo.weights_uri and reproduces the outputs of the in-memory weights."""
    from focoos_amd.ports import ModelInfo

    g, cfg, sd, model, images, x_u8, *_ = setup
    path = tmp_path / "model_final.pth"
    torch.save({"model": {("module." + k): v for k, v in sd.items()}, "iteration": 1}, path)  # DDP-prefixed, like a trainer dump
    d = ModelRegistry.get_model_info("fai-detr-l-obj365")
    info = ModelInfo(**{k: d[k] for k in ("name", "model_family", "classes", "im_size", "task", "config", "description")}, weights_uri=str(path))
    fm = ModelManager.get("fai-detr-l-obj365", model_info=info, seed=12345)  # seed differs: the file must win
    out_a = model.forward(x_u8)
    out_b = fm.model.forward(x_u8)
    assert torch.equal(out_a.logits, out_b.logits) and torch.equal(out_a.boxes, out_b.boxes)


@pytest.mark.parametrize("name,size,nb", [("fai-detr-l-obj365", 640, 4), ("bisenetformer-l-ade", 256, 4)])
def test_infer_stream_equals_infer_batch(name, size, nb):
    """FocoosModel.infer_stream (round 6: the engine's throughput mode behind the public surface - up to three batches in flight) yields, batch
    for batch and in order, exactly what infer_batch returns for the same images (class ids, integer boxes, scores, mask strings)."""
    fm = ModelManager.get(name, device=DEV, seed=3)
    batches = [[synth_image_structured(700 + 10 * j + i, size, size) for i in range(nb)] for j in range(5)]
    want = [fm.infer_batch(b, threshold=0.3) for b in batches]
    got = list(fm.infer_stream(iter(batches), threshold=0.3))
    assert len(got) == len(want) == 5 and sum(len(d.detections) for w in want for d in w) > 0
    for w, g_ in zip(want, got):
        assert len(w) == len(g_) == nb
        for dw, dg in zip(w, g_):
            assert [(d.cls_id, d.bbox, d.conf, d.mask) for d in dw.detections] == [(d.cls_id, d.bbox, d.conf, d.mask) for d in dg.detections]
