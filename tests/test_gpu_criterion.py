"""GPU parity of the set-criterion kernels (matching cost, Hungarian/LSA, VFL+L1+GIoU losses) through the C ABI against
SciPy, the CPU oracle and the real reference's golden vectors.  Matched indices are int and must be bit-exact."""
import ctypes as C

import numpy as np
import pytest
import torch
from scipy.optimize import linear_sum_assignment

pytestmark = pytest.mark.gpu

from focoos_amd import _lib  # noqa: E402
from focoos_amd._lib import check  # noqa: E402
from oracle import criterion_oracle as CO  # noqa: E402
from tests.helpers import load_golden  # noqa: E402

DEV = "cuda:0"


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def pack_targets(labels, tboxes):
    off = np.zeros(len(labels) + 1, np.int32)
    off[1:] = np.cumsum([len(l) for l in labels])
    lab = torch.cat([l.to(torch.int32) for l in labels]) if off[-1] else torch.zeros(0, dtype=torch.int32)
    tb = torch.cat(tboxes) if off[-1] else torch.zeros(0, 4)
    return torch.from_numpy(off).to(DEV), lab.to(DEV), tb.contiguous().to(DEV), off


def run_lsa(lib, cost_np_list, Q):
    """cost blocks [Q,T_b] (float32) -> list of (pred_idx, tgt_idx) through fx_lsa_f32."""
    B = len(cost_np_list)
    Ts = [c.shape[1] for c in cost_np_list]
    Tmax = max(Ts + [0])
    off = np.zeros(B + 1, np.int32)
    off[1:] = np.cumsum(Ts)
    cost = np.zeros((B, Q, max(Tmax, 1)), np.float32)
    for b, c in enumerate(cost_np_list):
        cost[b, :, : c.shape[1]] = c
    cd = torch.from_numpy(cost).to(DEV)
    od = torch.from_numpy(off).to(DEV)
    n = int(off[-1])
    pi = torch.full((max(n, 1),), -7, dtype=torch.int32, device=DEV)
    ti = torch.full((max(n, 1),), -7, dtype=torch.int32, device=DEV)
    check(_lib.load().fx_lsa_f32(cd.data_ptr(), B, Q, Tmax, od.data_ptr(), pi.data_ptr(), ti.data_ptr(), stream()), "fx_lsa_f32")
    torch.cuda.synchronize()
    pi, ti = pi.cpu().numpy(), ti.cpu().numpy()
    return [(pi[off[b]:off[b + 1]], ti[off[b]:off[b + 1]]) for b in range(B)]


def test_lsa_bit_exact_vs_scipy():
    lib = _lib.load()
    rs = np.random.RandomState(1)
    Q = 300
    blocks = [rs.rand(Q, t).astype(np.float32) for t in (1, 7, 20, 64, 100, 0, 300)]
    blocks.append(rs.randint(0, 3, (Q, 40)).astype(np.float32))          # integer costs: ties everywhere
    blocks.append(np.zeros((Q, 17), np.float32))                          # all equal
    blocks.append((rs.rand(Q, 33) * 1e-3 + 5.0).astype(np.float32))      # tiny gaps on a large offset
    got = run_lsa(lib, blocks, Q)
    for c, (pi, ti) in zip(blocks, got):
        r, cc = linear_sum_assignment(c)
        assert np.array_equal(pi, r) and np.array_equal(ti, cc), c.shape
    # other Q (not a multiple of 64) and T == Q
    blocks = [rs.rand(77, 77).astype(np.float32), rs.rand(77, 5).astype(np.float32)]
    for c, (pi, ti) in zip(blocks, run_lsa(lib, blocks, 77)):
        r, cc = linear_sum_assignment(c)
        assert np.array_equal(pi, r) and np.array_equal(ti, cc)
    assert lib.fx_lsa_f32(None, 1, 2000, 5, None, None, None, stream()) == -1


def test_lsa_infeasible_costs_raise_like_scipy():
    """NaN / inf costs (a diverged step): SciPy's linear_sum_assignment raises "cost matrix is infeasible" inside the reference matcher
    (fai_detr/modelling.py:749-750).  The kernel cannot raise: it sets the sticky status word of fx_lsa_status_f32, and the host mirror raises
    the same ValueError at its next synchronisation point - the matcher's public forward does so itself."""
    from focoos_amd.criterion import BoxHungarianMatcher, lsa_status, raise_if_infeasible
    from types import SimpleNamespace

    lib = _lib.load()
    # 1. the C ABI: status bit set for the block with a NaN column, indices of the feasible block still exact
    rs = np.random.RandomState(3)
    Q = 50
    good, bad = rs.rand(Q, 6).astype(np.float32), rs.rand(Q, 4).astype(np.float32)
    bad[:, 2] = np.nan
    with pytest.raises(ValueError):
        linear_sum_assignment(bad)
    cost = np.zeros((2, Q, 6), np.float32)
    cost[0], cost[1, :, :4] = good, bad
    off = torch.tensor([0, 6, 10], dtype=torch.int32, device=DEV)
    cd = torch.from_numpy(cost).to(DEV)
    pi = torch.zeros(10, dtype=torch.int32, device=DEV)
    ti = torch.zeros(10, dtype=torch.int32, device=DEV)
    st = torch.zeros(1, dtype=torch.int32, device=DEV)
    check(lib.fx_lsa_status_f32(cd.data_ptr(), 2, Q, 6, off.data_ptr(), pi.data_ptr(), ti.data_ptr(), st.data_ptr(), stream()), "fx_lsa_status_f32")
    torch.cuda.synchronize()
    assert int(st.item()) == 2          # NaN entries: SciPy's "matrix contains invalid numeric entries"
    r, c = linear_sum_assignment(good)
    assert np.array_equal(pi[:6].cpu().numpy(), r) and np.array_equal(ti[:6].cpu().numpy(), c)
    assert int(pi[6:].abs().sum()) == 0   # the infeasible image's slots are untouched
    # +inf column: every assignment of that target costs inf -> SciPy's "cost matrix is infeasible", bit 0
    bad2 = rs.rand(Q, 4).astype(np.float32)
    bad2[:, 1] = np.inf
    with pytest.raises(ValueError, match="infeasible"):
        linear_sum_assignment(bad2)
    cost[1, :, :4] = bad2
    cd = torch.from_numpy(cost).to(DEV)
    st.zero_()
    pi.zero_()
    check(lib.fx_lsa_status_f32(cd.data_ptr(), 2, Q, 6, off.data_ptr(), pi.data_ptr(), ti.data_ptr(), st.data_ptr(), stream()), "fx_lsa_status_f32")
    torch.cuda.synchronize()
    assert int(st.item()) == 1 and np.array_equal(pi[:6].cpu().numpy(), r) and int(pi[6:].abs().sum()) == 0
    # 2. the host mirror: NaN logits -> NaN costs -> ValueError from the matcher, as from the reference's; the flag is cleared by raising
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(2, Q, 20, generator=g).to(DEV)
    boxes = (torch.rand(2, Q, 4, generator=g) * 0.5 + 0.25).to(DEV)
    tgts = [SimpleNamespace(labels=torch.tensor([1, 2, 3]), boxes=torch.rand(3, 4, generator=g) * 0.5 + 0.25),
            SimpleNamespace(labels=torch.tensor([4]), boxes=torch.rand(1, 4, generator=g) * 0.5 + 0.25)]
    m = BoxHungarianMatcher()
    ok = m({"pred_logits": logits, "pred_boxes": boxes}, tgts)
    assert [len(p) for p, _ in ok] == [3, 1]
    logits_nan = logits.clone()
    logits_nan[1] = float("nan")
    with pytest.raises(ValueError, match="invalid numeric entries"):
        m({"pred_logits": logits_nan, "pred_boxes": boxes}, tgts)
    assert int(lsa_status(DEV).item()) == 0
    raise_if_infeasible(DEV)   # nothing pending
    again = m({"pred_logits": logits, "pred_boxes": boxes}, tgts)
    assert all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(ok, again))


def test_matcher_and_losses_vs_reference_golden():
    lib = _lib.load()
    g = load_golden("detr_criterion.npz")
    logits, boxes, labels, tboxes = CO.synth_predictions_and_targets(0)
    B, Q, K = logits.shape
    od, lab, tb, off = pack_targets(labels, tboxes)
    Tmax = int(max(np.diff(off)))
    ld, bd = logits.to(DEV), boxes.to(DEV)
    cost = torch.full((B, Q, Tmax), float("nan"), device=DEV)
    check(lib.fx_detr_match_cost_f32(ld.data_ptr(), K, bd.data_ptr(), lab.data_ptr(), tb.data_ptr(), od.data_ptr(), B, Q, K, Tmax, 2.0, 5.0, 2.0, 0.25,
                                     2.0, cost.data_ptr(), stream()), "cost")
    n = int(off[-1])
    pi = torch.empty(n, dtype=torch.int32, device=DEV)
    ti = torch.empty(n, dtype=torch.int32, device=DEV)
    check(lib.fx_lsa_f32(cost.data_ptr(), B, Q, Tmax, od.data_ptr(), pi.data_ptr(), ti.data_ptr(), stream()), "lsa")
    ws = torch.empty(lib.fx_detr_set_loss_workspace_bytes(B, Q, n) // 8 + 1, dtype=torch.float64, device=DEV)
    out3 = torch.empty(3, device=DEV)
    nb = float(max(n, 1))
    check(lib.fx_detr_set_loss_f32(ld.data_ptr(), K, bd.data_ptr(), lab.data_ptr(), tb.data_ptr(), od.data_ptr(), pi.data_ptr(), ti.data_ptr(), B, Q, K, n,
                                   nb, 0.75, 2.0, 1.0, 5.0, 2.0, ws.data_ptr(), out3.data_ptr(), stream()), "loss")
    torch.cuda.synchronize()
    cc = cost.cpu().numpy()
    pin, tin = pi.cpu().numpy(), ti.cpu().numpy()
    for b in range(B):
        T = off[b + 1] - off[b]
        np.testing.assert_allclose(cc[b, :, :T], g[f"cost_{b}"], rtol=0, atol=5e-6)      # fp32 cost vs the reference's
        assert (cc[b, :, T:] == 0).all()
        assert np.array_equal(pin[off[b]:off[b + 1]], g[f"pred_idx_{b}"])                 # indices: bit-exact vs reference+SciPy
        assert np.array_equal(tin[off[b]:off[b + 1]], g[f"tgt_idx_{b}"])
    np.testing.assert_allclose(out3.cpu().numpy().astype(np.float64), g["loss"], rtol=2e-5)
    # a second, larger random case against the oracle (B=8, K=365, up to 60 targets, empty images)
    logits, boxes, labels, tboxes = CO.synth_predictions_and_targets(5, B=8, Q=300, K=365, counts=(60, 3, 0, 31, 12, 0, 1, 45))
    costs = CO.matcher_cost(logits, boxes, labels, tboxes)
    ref_idx = CO.hungarian(costs)
    od, lab, tb, off = pack_targets(labels, tboxes)
    B, Q, K = logits.shape
    Tmax, n = int(max(np.diff(off))), int(off[-1])
    ld, bd = logits.to(DEV), boxes.to(DEV)
    cost = torch.empty(B, Q, Tmax, device=DEV)
    pi = torch.empty(n, dtype=torch.int32, device=DEV)
    ti = torch.empty(n, dtype=torch.int32, device=DEV)
    check(lib.fx_detr_match_cost_f32(ld.data_ptr(), K, bd.data_ptr(), lab.data_ptr(), tb.data_ptr(), od.data_ptr(), B, Q, K, Tmax, 2.0, 5.0, 2.0, 0.25,
                                     2.0, cost.data_ptr(), stream()))
    check(lib.fx_lsa_f32(cost.data_ptr(), B, Q, Tmax, od.data_ptr(), pi.data_ptr(), ti.data_ptr(), stream()))
    ws = torch.empty(lib.fx_detr_set_loss_workspace_bytes(B, Q, n) // 8 + 1, dtype=torch.float64, device=DEV)
    check(lib.fx_detr_set_loss_f32(ld.data_ptr(), K, bd.data_ptr(), lab.data_ptr(), tb.data_ptr(), od.data_ptr(), pi.data_ptr(), ti.data_ptr(), B, Q, K, n,
                                   float(n), 0.75, 2.0, 1.0, 5.0, 2.0, ws.data_ptr(), out3.data_ptr(), stream()))
    torch.cuda.synchronize()
    pin, tin = pi.cpu().numpy(), ti.cpu().numpy()
    for b in range(B):
        assert np.array_equal(pin[off[b]:off[b + 1]], ref_idx[b][0]) and np.array_equal(tin[off[b]:off[b + 1]], ref_idx[b][1]), b
    ref_l = CO.set_criterion_losses(logits, boxes, labels, tboxes, ref_idx, float(n))
    np.testing.assert_allclose(out3.cpu().numpy(), [float(ref_l[k]) for k in ("loss_vfl", "loss_bbox", "loss_giou")], rtol=2e-5)


def test_host_mirrors_match_oracle_structure():
    """focoos_amd.criterion.{BoxHungarianMatcher, SetCriterion}: same call / return structure as the reference classes."""
    from types import SimpleNamespace

    from focoos_amd.criterion import BoxHungarianMatcher, SetCriterion

    logits, boxes, labels, tboxes = CO.synth_predictions_and_targets(9, B=3, Q=300, K=80, counts=(5, 0, 12))
    aux_l, aux_b, _, _ = CO.synth_predictions_and_targets(10, B=3, Q=300, K=80, counts=(5, 0, 12))
    targets = [SimpleNamespace(labels=l.to(DEV), boxes=b.to(DEV)) for l, b in zip(labels, tboxes)]
    m = BoxHungarianMatcher()
    outputs = {"pred_logits": logits.to(DEV), "pred_boxes": boxes.to(DEV), "aux_outputs": [{"pred_logits": aux_l.to(DEV), "pred_boxes": aux_b.to(DEV)}]}
    ind = m(outputs, targets)
    ref = CO.hungarian(CO.matcher_cost(logits, boxes, labels, tboxes))
    assert all(i.dtype == torch.int64 and np.array_equal(i.numpy(), r[0]) and np.array_equal(j.numpy(), r[1]) for (i, j), r in zip(ind, ref))
    crit = SetCriterion(80, m, {"loss_vfl": 1, "loss_bbox": 5, "loss_giou": 2})
    losses = crit(outputs, targets)
    assert sorted(losses) == sorted(["loss_vfl", "loss_bbox", "loss_giou", "loss_vfl_0", "loss_bbox_0", "loss_giou_0"])
    nb = float(sum(len(l) for l in labels))
    exp = CO.set_criterion_losses(logits, boxes, labels, tboxes, ref, nb)
    exp_aux = CO.set_criterion_losses(aux_l, aux_b, labels, tboxes, CO.hungarian(CO.matcher_cost(aux_l, aux_b, labels, tboxes)), nb)
    for k in ("loss_vfl", "loss_bbox", "loss_giou"):
        assert abs(float(losses[k]) - float(exp[k])) <= 2e-5 * abs(float(exp[k]))
        assert abs(float(losses[k + "_0"]) - float(exp_aux[k])) <= 2e-5 * abs(float(exp_aux[k]))


def _torch_box_losses(boxes, tboxes_cat, pi, ti, off, slot_b):
    """SetCriterion.loss_boxes on the matched pairs in plain torch autograd (focoos/utils/box.py:14-64 restricted to the diagonal) - the
    formulation the training graph used before fx_detr_box_loss_f32 replaced its ~150 elementwise launches per prediction set."""
    def xyxy(b):
        cx, cy, w, h = b.unbind(-1)
        return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], -1)

    src = boxes[slot_b, pi.long()]
    tb = tboxes_cat[torch.from_numpy(off[:-1]).long()[slot_b] + ti.long()]
    a, b = xyxy(src), xyxy(tb)
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt, rb = torch.max(a[:, :2], b[:, :2]), torch.min(a[:, 2:], b[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, 0] * wh[:, 1]
    union = area_a + area_b - inter
    iou = inter / union
    lt2, rb2 = torch.min(a[:, :2], b[:, :2]), torch.max(a[:, 2:], b[:, 2:])
    wh2 = (rb2 - lt2).clamp(min=0)
    area = wh2[:, 0] * wh2[:, 1]
    giou = iou - (area - union) / (area + 1e-5)
    return (src - tb).abs().sum(), (1 - giou).sum(), iou.detach()


@pytest.mark.parametrize("counts", [(3, 0, 7, 20), (0, 0), (1,), (17, 5, 0, 9, 20, 2, 11, 1)])
def test_box_loss_with_gradient_vs_torch_autograd(counts):
    """fx_detr_box_loss_f32 / fx_detr_box_loss_bwd_f32 (L1 + GIoU of the matched pairs, their gradient, and the per-query VFL targets)
    vs torch CPU fp32 autograd of the reference formulation: losses within 1e-5 relative, gradients within 1e-4 of their max
    (fp32 both sides, different operation order); labels bit-exact, IoU targets within 1e-6.  Includes disjoint pairs (zero
    intersection: the clamp branch), containing pairs (min/max pick the same box for both corners) and exact ties."""
    lib = _lib.load()
    rs = np.random.RandomState(sum(counts) + len(counts))
    B, Q, K = len(counts), 50, 80
    boxes = torch.from_numpy(np.concatenate([rs.uniform(0.2, 0.8, (B, Q, 2)), rs.uniform(0.02, 0.5, (B, Q, 2))], -1).astype(np.float32))
    labels = [torch.from_numpy(rs.randint(0, K, (t,))) for t in counts]
    tboxes = [torch.from_numpy(np.concatenate([rs.uniform(0.2, 0.8, (t, 2)), rs.uniform(0.02, 0.5, (t, 2))], -1).astype(np.float32)) for t in counts]
    pi_l, ti_l = [], []
    for t in counts:
        pi_l.append(np.sort(rs.permutation(Q)[:t]))
        ti_l.append(rs.permutation(t))
    n = int(sum(counts))
    pi = torch.from_numpy(np.concatenate(pi_l).astype(np.int32)) if n else torch.zeros(1, dtype=torch.int32)
    ti = torch.from_numpy(np.concatenate(ti_l).astype(np.int32)) if n else torch.zeros(1, dtype=torch.int32)
    if n >= 3:   # a predicted box identical to its target (every max / min is a tie, L1 = 0), and one containing its target
        b0 = next(b for b, t in enumerate(counts) if t)
        boxes[b0, pi_l[b0][0]] = tboxes[b0][ti_l[b0][0]]
        b1 = max(b for b, t in enumerate(counts) if t)
        boxes[b1, pi_l[b1][-1]] = tboxes[b1][ti_l[b1][-1]] * torch.tensor([1.0, 1.0, 1.5, 1.5])
    off_d, lab_d, tb_d, off = pack_targets(labels, tboxes)
    if not n:
        lab_d, tb_d = torch.zeros(1, dtype=torch.int32, device=DEV), torch.zeros(1, 4, device=DEV)
    sb, sg = 5.0 / max(n, 1), 2.0 / max(n, 1)
    bd = boxes.to(DEV)
    cls = torch.full((B * Q,), -3, dtype=torch.int32, device=DEV)
    score = torch.full((B * Q,), -3.0, device=DEV)
    loss2 = torch.full((2,), -3.0, device=DEV)
    pg = torch.zeros(max(n, 1), 8, device=DEV)
    pid, tid = pi.to(DEV), ti.to(DEV)
    check(lib.fx_detr_box_loss_f32(bd.data_ptr(), lab_d.data_ptr(), tb_d.data_ptr(), off_d.data_ptr(), pid.data_ptr(), tid.data_ptr(), B, Q, K, n,
                                   sb, sg, cls.data_ptr(), score.data_ptr(), loss2.data_ptr(), pg.data_ptr(), stream()), "fx_detr_box_loss_f32")
    g = torch.tensor([1.3, 0.7], device=DEV)
    dboxes = torch.full((B, Q, 4), 9.0, device=DEV)
    check(lib.fx_detr_box_loss_bwd_f32(pg.data_ptr(), off_d.data_ptr(), pid.data_ptr(), B, Q, n, g[0:1].data_ptr(), g[1:2].data_ptr(),
                                       dboxes.data_ptr(), stream()), "fx_detr_box_loss_bwd_f32")
    torch.cuda.synchronize()
    # reference
    bt = boxes.clone().requires_grad_(True)
    cls_ref = torch.full((B * Q,), K, dtype=torch.int32)
    score_ref = torch.zeros(B * Q)
    if n:
        slot_b = torch.from_numpy(np.repeat(np.arange(B), np.diff(off))).long()
        l1, gi, iou = _torch_box_losses(bt, torch.cat(tboxes), pi[:n], ti[:n], off, slot_b)
        (1.3 * sb * l1 + 0.7 * sg * gi).backward()
        flat = slot_b * Q + pi[:n].long()
        cls_ref[flat] = torch.cat(labels).to(torch.int32)[torch.from_numpy(off[:-1]).long()[slot_b] + ti[:n].long()]
        score_ref[flat] = iou
        ref = torch.stack([sb * l1.detach(), sg * gi.detach()])
        gref = bt.grad
    else:
        ref, gref = torch.zeros(2), torch.zeros(B, Q, 4)
    if n:   # the same through the oracle's SetCriterion restatement (pinned to the reference's golden losses): full [n, n] IoU matrices, diagonal taken
        bo = boxes.clone().requires_grad_(True)
        indices = [(torch.as_tensor(a, dtype=torch.int64), torch.as_tensor(b_, dtype=torch.int64)) for a, b_ in zip(pi_l, ti_l)]
        lo = CO.set_criterion_losses(torch.zeros(B, Q, K), bo, labels, tboxes, indices, float(n))
        (1.3 * lo["loss_bbox"] + 0.7 * lo["loss_giou"]).backward()
        assert abs(float(lo["loss_bbox"]) - float(ref[0])) <= 1e-5 * max(float(ref[0]), 1.0) and abs(float(lo["loss_giou"]) - float(ref[1])) <= 1e-5 * max(float(ref[1]), 1.0)
        assert (bo.grad - gref).abs().max() <= 1e-5 * max(float(gref.abs().max()), 1e-3)
    assert torch.equal(cls.cpu(), cls_ref)
    assert (score.cpu() - score_ref).abs().max() <= 1e-6
    assert (loss2.cpu() - ref).abs().max() <= 1e-5 * max(float(ref.abs().max()), 1.0)
    assert (dboxes.cpu() - gref).abs().max() <= 1e-4 * max(float(gref.abs().max()), 1e-3), (dboxes.cpu() - gref).abs().max()
    # NULL upstream gradients count as zero
    check(lib.fx_detr_box_loss_bwd_f32(pg.data_ptr(), off_d.data_ptr(), pid.data_ptr(), B, Q, n, None, None, dboxes.data_ptr(), stream()))
    torch.cuda.synchronize()
    assert float(dboxes.abs().max()) == 0.0


def test_batched_matching_of_all_prediction_sets_equals_per_set_matching():
    """BoxHungarianMatcher.match_packed_sets (all prediction sets of a step against the same targets as ONE cost launch + ONE assignment
    launch over S*B virtual images) returns exactly the S per-set results of match_packed; images without targets and a set count of 1
    included."""
    from focoos_amd.criterion import BoxHungarianMatcher, _Targets
    from focoos_amd.ports import DETRTargets

    g = torch.Generator().manual_seed(21)
    B, Q, K, S = 5, 60, 17, 4
    sizes = [7, 0, 13, 1, 20]
    targets = [DETRTargets(labels=torch.randint(0, K, (t,), generator=g).to(DEV),
                           boxes=torch.cat([torch.rand(t, 2, generator=g) * 0.6 + 0.2, torch.rand(t, 2, generator=g) * 0.3 + 0.05], -1).to(DEV)) for t in sizes]
    tg = _Targets(targets, DEV)
    logits = [torch.randn(B, Q, K, generator=g).bfloat16().to(DEV) for _ in range(S)]
    boxes = [torch.cat([torch.rand(B, Q, 2, generator=g) * 0.6 + 0.2, torch.rand(B, Q, 2, generator=g) * 0.3 + 0.05], -1).to(DEV) for _ in range(S)]
    m = BoxHungarianMatcher()
    single = [m.match_packed(l, b, tg) for l, b in zip(logits, boxes)]
    batched = m.match_packed_sets(logits, boxes, tg)
    torch.cuda.synchronize()
    assert len(batched) == S
    for (p0, t0), (p1, t1) in zip(single, batched):
        assert torch.equal(p0, p1) and torch.equal(t0, t1)
    one = m.match_packed_sets(logits[:1], boxes[:1], tg)
    assert torch.equal(one[0][0], single[0][0]) and torch.equal(one[0][1], single[0][1])
