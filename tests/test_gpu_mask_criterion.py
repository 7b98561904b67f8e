"""Mask-classification criterion on a real MI355X: fx_point_sample_f32, fx_mask_match_cost_f32 (+ fx_lsa_f32) and
fx_mask_set_loss_f32 against oracle/mask_criterion_oracle.py (bit-equal to the real reference on its recorded torch.rand draws:
tests/test_mask_criterion_oracle.py) and against the reference's own golden costs / matches / losses.
Tolerances: sampled values 1e-5 abs (fp32 bilinear), cost blocks rtol 2e-5, losses rtol 2e-5; Hungarian indices identical."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from focoos_amd import _lib  # noqa: E402
from focoos_amd._lib import check  # noqa: E402
from focoos_amd.mask_criterion import MaskHungarianMatcher, SetCriterion  # noqa: E402
from focoos_amd.ports import MaskFormerTargets  # noqa: E402
from oracle import mask_criterion_oracle as MC  # noqa: E402
from tests.helpers import load_golden  # noqa: E402

DEV = "cuda:0"


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return _lib.load()


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _Replay:
    """rand callable that replays recorded draws (shape-checked) on the device."""

    def __init__(self, tensors):
        self.t, self.i = list(tensors), 0

    def __call__(self, *shape, device):
        t = self.t[self.i]
        assert tuple(t.shape) == tuple(shape), (self.i, tuple(t.shape), shape)
        self.i += 1
        return t.to(device)


@pytest.mark.parametrize("u8", [False, True])
def test_point_sample_vs_grid_sample(lib, u8):
    g = torch.Generator().manual_seed(3)
    R, H, W, P = 5, 13, 17, 777
    src = (torch.rand(R, H, W, generator=g) > 0.5).to(torch.uint8) if u8 else torch.randn(R, H, W, generator=g)
    coords = torch.rand(2, P, 2, generator=g)
    coords[0, :4] = torch.tensor([[0.0, 0.0], [1.0, 1.0], [0.5 / W, 0.5 / H], [1.0, 0.0]])   # corners / exact pixel centre
    cidx = torch.tensor([0, 1, 1, 0, 1], dtype=torch.int32)
    sidx = torch.tensor([4, 0, 2, 2, 1], dtype=torch.int32)
    sd, cd, ci, si = src.to(DEV), coords.to(DEV), cidx.to(DEV), sidx.to(DEV)
    out = torch.empty(R, P, device=DEV)
    check(lib.fx_point_sample_f32(sd.data_ptr(), int(u8), H, W, si.data_ptr(), cd.data_ptr(), ci.data_ptr(), out.data_ptr(), R, P, stream()))
    torch.cuda.synchronize()
    ref = MC.point_sample(src[sidx.long()][:, None].float(), coords[cidx.long()])[:, 0]
    assert (out.cpu() - ref).abs().max() <= 1e-5


def _targets(labels, masks, as_bool):
    return [MaskFormerTargets(labels=l.to(DEV), masks=(m.bool() if as_bool else m).to(DEV)) for l, m in zip(labels, masks)]


def test_matcher_and_criterion_vs_reference_golden():
    """The reference's own recorded draws replayed: cost blocks, SciPy's matches and the 9 weighted losses of the golden file."""
    g = load_golden("mask_criterion.npz")
    out, labels, masks = MC.synth_mask_predictions_and_targets(0)
    P = int(g["num_points"])
    rand = _Replay([torch.from_numpy(g[f"rand_{i}"]) for i in range(int(g["n_rand"]))])
    matcher = MaskHungarianMatcher(cost_class=2, cost_mask=5, cost_dice=5, num_points=P, rand=rand)
    crit = SetCriterion(80, matcher, {"loss_ce": 2, "loss_mask": 5, "loss_dice": 5}, eos_coef=0.1, num_points=P, oversample_ratio=3.0,
                        importance_sample_ratio=0.75, rand=rand)
    dev_out = {"pred_logits": out["pred_logits"].to(DEV), "pred_masks": out["pred_masks"].to(DEV),
               "aux_outputs": [{k: v.to(DEV) for k, v in a.items()} for a in out["aux_outputs"]]}
    costs, matches = [], []
    orig = crit._one_set

    def spy(o, tg, nm, fixed=None):
        r = orig(o, tg, nm, fixed)
        costs.append(matcher.last_cost.cpu())
        matches.append((crit.last_matches[0].cpu(), crit.last_matches[1].cpu(), tg.off_host.copy()))
        return r

    crit._one_set = spy
    losses = crit(dev_out, _targets(labels, masks, as_bool=True))
    torch.cuda.synchronize()
    assert rand.i == int(g["n_rand"])
    for si in range(3):
        pi, ti, off = matches[si]
        for b in range(2):
            T = off[b + 1] - off[b]
            np.testing.assert_allclose(costs[si][b, :, :T].numpy(), g[f"cost_{2 * si + b}"], rtol=2e-5, atol=2e-5)
            assert pi[off[b]:off[b + 1]].tolist() == g[f"pred_idx_{2 * si + b}"].tolist()
            assert ti[off[b]:off[b + 1]].tolist() == g[f"tgt_idx_{2 * si + b}"].tolist()
    got = np.array([float(losses[k]) for k in sorted(losses)])
    assert sorted(losses) == g["loss_names"].tolist()
    np.testing.assert_allclose(got, g["losses"], rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("cfg", [(3, 100, 150, (80, 80), 8, 12544, (5, 0, 17)), (2, 100, 80, (50, 64), 4, 1000, (1, 30))])
def test_criterion_vs_oracle_full_size(cfg):
    """Registry-size problem (BiSeNetFormer: 100 queries, 150 classes, 1/8-resolution masks of a 640x640 image, 12544 points; an
    image without targets) against the oracle on identical seeded draws, float (not bool) target masks on the second case."""
    B, Q, K, hw, scale, P, counts = cfg
    out, labels, masks = MC.synth_mask_predictions_and_targets(11, B=B, Q=Q, K=K, hw=hw, scale=scale, counts=counts, n_aux=1)
    gen = torch.Generator().manual_seed(P)
    n = sum(counts)
    draws = []
    for _ in range(2):   # two prediction sets
        draws += [torch.rand(1, P, 2, generator=gen) for _ in range(B)]
        draws += [torch.rand(n, int(P * 3.0), 2, generator=gen), torch.rand(n, P - int(0.75 * P), 2, generator=gen)]
    ref, ref_matches = MC.criterion(out, labels, masks, MC.RandStream(draws), K, P)
    rand = _Replay(draws)
    matcher = MaskHungarianMatcher(cost_class=2, cost_mask=5, cost_dice=5, num_points=P, rand=rand)
    crit = SetCriterion(K, matcher, {"loss_ce": 2, "loss_mask": 5, "loss_dice": 5}, eos_coef=0.1, num_points=P, oversample_ratio=3.0,
                        importance_sample_ratio=0.75, rand=rand)
    dev_out = {"pred_logits": out["pred_logits"].to(DEV), "pred_masks": out["pred_masks"].to(DEV),
               "aux_outputs": [{k: v.to(DEV) for k, v in a.items()} for a in out["aux_outputs"]]}
    tg = _targets(labels, masks, as_bool=(scale == 8))
    losses = crit(dev_out, tg)
    torch.cuda.synchronize()
    for k in ref:
        np.testing.assert_allclose(float(losses[k]), float(ref[k]), rtol=3e-5, atol=1e-6, err_msg=k)
    # the matcher mirror alone returns the reference's structure: list of (index_i, index_j) int64 tensors per image
    rand.i = 0
    m = MaskHungarianMatcher(cost_class=2, cost_mask=5, cost_dice=5, num_points=P, rand=rand)(dev_out, tg)
    for b, (i, j) in enumerate(m):
        assert i.dtype == torch.int64 and i.tolist() == np.asarray(ref_matches[0][b][0]).tolist() and j.tolist() == np.asarray(ref_matches[0][b][1]).tolist()


def test_default_random_points_and_determinism():
    """Without injected draws the mirror samples torch.rand on the device: losses are finite, close to the oracle's value on other
    draws of the same distribution (the loss is an average over 12544 points), and a fixed torch seed reproduces them bit for bit."""
    out, labels, masks = MC.synth_mask_predictions_and_targets(3, B=2, Q=100, K=80, hw=(40, 48), scale=4, counts=(6, 3), n_aux=0)
    dev_out = {"pred_logits": out["pred_logits"].to(DEV), "pred_masks": out["pred_masks"].to(DEV)}
    P = 12544
    crit = SetCriterion(80, MaskHungarianMatcher(2, 5, 5, num_points=P), {"loss_ce": 2, "loss_mask": 5, "loss_dice": 5}, num_points=P,
                        importance_sample_ratio=0.75)
    tg = _targets(labels, masks, True)
    torch.manual_seed(0)
    a = {k: float(v) for k, v in crit(dev_out, tg).items()}
    torch.manual_seed(0)
    b = {k: float(v) for k, v in crit(dev_out, tg).items()}
    assert a == b and all(np.isfinite(v) for v in a.values())
    gen = torch.Generator().manual_seed(1)
    draws = [torch.rand(1, P, 2, generator=gen) for _ in range(2)] + [torch.rand(9, 3 * P, 2, generator=gen), torch.rand(9, P - int(0.75 * P), 2, generator=gen)]
    ref, _ = MC.criterion(out, labels, masks, MC.RandStream(draws), 80, P)
    for k in ref:
        assert abs(a[k] - float(ref[k])) <= 0.05 * abs(float(ref[k])) + 1e-3, (k, a[k], float(ref[k]))


@pytest.mark.parametrize("shape", [(2, 100, 1000, (3, 5)), (3, 37, 777, (0, 21, 40)), (1, 100, 12544, (64,)), (2, 20, 500, (70, 2))])
def test_match_cost_tiled_equals_untiled(shape):
    """fx_mask_match_cost_ws_f32 (16 queries x 16 targets x point slices per workgroup, slices reduced in a fixed order) against
    fx_mask_match_cost_f32 (one workgroup per query) on ragged shapes: an image without targets, more than 16 and more than 32 targets (several
    target blocks in registers), point counts that are not multiples of the 64-point tile, and more than 64 targets in one image (the ws
    variant then reports a zero workspace and runs the untiled kernel).  Agreement to the order of fp32 sums."""
    lib = _lib.load()
    B, Q, P, counts = shape
    K = 20
    g = torch.Generator().manual_seed(B * 1000 + Q)
    logits = torch.randn(B, Q, K + 1, generator=g).to(DEV)
    pp = (torch.randn(B * Q, P, generator=g) * 3).to(DEV)
    n = sum(counts)
    tp = (torch.rand(n, P, generator=g) > 0.6).float().to(DEV)
    labels = torch.randint(0, K, (n,), generator=g, dtype=torch.int32).to(DEV)
    off = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32, device=DEV)
    tmax = max(max(counts), 1)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    c0 = torch.full((B, Q, tmax), float("nan"), device=DEV)
    c1 = torch.full((B, Q, tmax), float("nan"), device=DEV)
    check(lib.fx_mask_match_cost_f32(logits.data_ptr(), K + 1, pp.data_ptr(), tp.data_ptr(), labels.data_ptr(), off.data_ptr(), B, Q, K, P, tmax, 2.0, 5.0, 5.0, 0,
                                     c0.data_ptr(), st))
    nws = int(lib.fx_mask_match_cost_workspace_bytes(B, Q, tmax))
    assert (nws == 0) == (tmax > 64)
    ws = torch.full((max(nws, 16),), 0x7F, dtype=torch.uint8, device=DEV)
    check(lib.fx_mask_match_cost_ws_f32(logits.data_ptr(), K + 1, pp.data_ptr(), tp.data_ptr(), labels.data_ptr(), off.data_ptr(), B, Q, K, P, tmax, 2.0, 5.0, 5.0,
                                        0, c1.data_ptr(), ws.data_ptr(), nws, st))
    torch.cuda.synchronize()
    assert torch.isfinite(c1).all()
    np.testing.assert_allclose(c1.cpu().numpy(), c0.cpu().numpy(), rtol=2e-5, atol=2e-5)
