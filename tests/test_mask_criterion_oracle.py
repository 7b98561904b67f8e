"""Pin oracle/mask_criterion_oracle.py (mask matcher + point-sampled criterion) against the golden vectors the REAL
reference produced (tests/golden/mask_criterion.npz, scripts/make_golden.py::mask_criterion_case): cost blocks, SciPy's matches and
the 9 weighted losses, on the reference's own recorded torch.rand draws."""
import numpy as np
import torch

from oracle import mask_criterion_oracle as MC
from oracle.criterion_oracle import hungarian, lsa_crouse
from tests.helpers import load_golden


def _case():
    g = load_golden("mask_criterion.npz")
    out, labels, masks = MC.synth_mask_predictions_and_targets(0)
    rand = [torch.from_numpy(g[f"rand_{i}"]) for i in range(int(g["n_rand"]))]
    return g, out, labels, masks, rand


def test_matcher_costs_and_matches_equal_reference():
    g, out, labels, masks, rand = _case()
    P = int(g["num_points"])
    sets = [out] + out["aux_outputs"]
    ri, ci = 0, 0
    for o in sets:
        for b in range(2):
            coords = rand[ri + b]
            c = MC.matcher_cost(o["pred_logits"][b], o["pred_masks"][b], labels[b], masks[b], coords)
            np.testing.assert_allclose(c.numpy(), g[f"cost_{ci}"], rtol=1e-5, atol=1e-5)
            i, j = lsa_crouse(c.numpy().astype(np.float64))
            assert i.tolist() == g[f"pred_idx_{ci}"].tolist() and j.tolist() == g[f"tgt_idx_{ci}"].tolist()
            ci += 1
        ri += 2 + 2   # two matcher draws, then the loss's two draws
    assert rand[0].shape == (1, P, 2)


def test_criterion_losses_equal_reference():
    g, out, labels, masks, rand = _case()
    losses, matches = MC.criterion(out, labels, masks, MC.RandStream(rand), 80, int(g["num_points"]))
    assert sorted(losses) == g["loss_names"].tolist()
    np.testing.assert_allclose([float(losses[k]) for k in sorted(losses)], g["losses"], rtol=1e-6, atol=1e-7)
    for si, m in enumerate(matches):
        for b, (i, j) in enumerate(m):
            assert np.asarray(i).tolist() == g[f"pred_idx_{2 * si + b}"].tolist()


def test_point_sample_is_grid_sample_on_unit_square():
    x = torch.arange(12, dtype=torch.float32).view(1, 1, 3, 4)
    # pixel centres map to themselves: (col + 0.5) / W, (row + 0.5) / H
    c = torch.tensor([[[(2 + 0.5) / 4, (1 + 0.5) / 3], [0.5 / 4, 0.5 / 3], [1.0, 1.0]]])
    v = MC.point_sample(x, c)[0, 0]
    assert abs(v[0].item() - 6.0) < 1e-5 and abs(v[1].item()) < 1e-5   # 2*c - 1 is not exact in fp32
    assert abs(v[2].item() - 11.0 * 0.25) < 1e-6      # the corner: three of four taps fall outside (zero padding)


def test_importance_points_keep_most_uncertain():
    g = torch.Generator().manual_seed(0)
    src = torch.randn(3, 1, 8, 8, generator=g) * 3
    over, extra = torch.rand(3, 48, 2, generator=g), torch.rand(3, 4, 2, generator=g)
    pts = MC.importance_points(src, over, extra, 16, 0.75)
    assert pts.shape == (3, 16, 2) and torch.equal(pts[:, 12:], extra)
    unc_all = MC.point_sample(src, over)[:, 0].abs()
    unc_sel = MC.point_sample(src, pts[:, :12])[:, 0].abs()
    assert (unc_sel.max(1).values <= unc_all.sort(1).values[:, 11] + 1e-6).all()
