"""Shared helpers for the parity tests (numpy/torch only; no product imports)."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def strided_sample(t: torch.Tensor, n: int = 2048) -> np.ndarray:
    """Same sampler as scripts/make_golden.py."""
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].to(torch.float32).cpu().numpy().copy()


def rel_err(a, b) -> float:
    """max|a-b| / max|b| (b = reference)."""
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def rel_l2(a, b) -> float:
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def load_golden(name: str):
    return np.load(os.path.join(GOLDEN, name))
