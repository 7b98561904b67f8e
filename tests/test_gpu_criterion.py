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
