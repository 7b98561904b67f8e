"""Seeded test-case inputs shared by scripts/make_golden.py and the tests (numpy only)."""
import numpy as np

MSDA_SHAPES = [[5, 7], [10, 14], [20, 28]]


def msda_case_inputs():
    """Inputs of the ms_deform_attn_core golden case (value, sampling locations incl. out-of-range
    points to exercise zero padding, softmaxed attention weights)."""
    rs = np.random.RandomState(7)
    S = sum(h * w for h, w in MSDA_SHAPES)
    N, M, D, Lq, L, P = 2, 8, 32, 19, 3, 4
    value = rs.standard_normal((N, S, M, D)).astype(np.float32)
    loc = rs.uniform(-0.15, 1.15, (N, Lq, M, L, P, 2)).astype(np.float32)
    logits = rs.standard_normal((N, Lq, M, L * P)).astype(np.float32)
    e = np.exp(logits - logits.max(-1, keepdims=True))
    w = (e / e.sum(-1, keepdims=True)).astype(np.float32).reshape(N, Lq, M, L, P)
    return value, loc, w
