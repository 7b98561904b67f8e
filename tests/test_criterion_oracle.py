"""Pin oracle/criterion_oracle.py: the LSA restatement against SciPy itself (the reference's third-party dependency) and the
matcher / losses against golden vectors produced by the REAL reference (scripts/make_golden.py::criterion_case)."""
import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

from oracle import criterion_oracle as CO
from tests.helpers import load_golden


def test_lsa_restatement_equals_scipy_including_ties_and_shapes():
    rs = np.random.RandomState(0)
    for trial in range(200):
        nr, nc = rs.randint(1, 40), rs.randint(1, 40)
        c = rs.rand(nr, nc) if trial % 3 else rs.randint(0, 4, (nr, nc)).astype(float)  # integer costs: massive ties
        a, b = CO.lsa_crouse(c), linear_sum_assignment(c)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (trial, nr, nc)
    assert CO.lsa_crouse(np.zeros((5, 0)))[0].size == 0
    c = rs.rand(300, 23).astype(np.float32)
    a, b = CO.lsa_crouse(c), linear_sum_assignment(c)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_matcher_and_losses_match_reference_golden():
    g = load_golden("detr_criterion.npz")
    logits, boxes, labels, tboxes = CO.synth_predictions_and_targets(0)
    costs = CO.matcher_cost(logits, boxes, labels, tboxes)
    idx = CO.hungarian(costs)
    for b, c in enumerate(costs):
        np.testing.assert_allclose(c.numpy(), g[f"cost_{b}"], rtol=0, atol=1e-6)
        assert np.array_equal(idx[b][0], g[f"pred_idx_{b}"]) and np.array_equal(idx[b][1], g[f"tgt_idx_{b}"])
        a = CO.lsa_crouse(c.numpy())
        assert np.array_equal(a[0], g[f"pred_idx_{b}"]) and np.array_equal(a[1], g[f"tgt_idx_{b}"])
    nb = float(max(sum(len(l) for l in labels), 1))
    got = CO.set_criterion_losses(logits, boxes, labels, tboxes, idx, nb)
    np.testing.assert_allclose([float(got[k]) for k in ("loss_vfl", "loss_bbox", "loss_giou")], g["loss"], rtol=1e-6)
