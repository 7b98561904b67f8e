import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


@pytest.fixture
def flat_small_shapes(monkeypatch):
    """The halo 3x3 / flat pointwise kernels are routed to from M >= 5000 pixels in production (smaller layers do not fill the
    chip).  Kernel-level parity tests run small shapes through them by lowering the thresholds (re-read by the library per call);
    end-to-end tests keep the production routing, i.e. they check the numerics a user gets at that size."""
    monkeypatch.setenv("FX_CONV3_MIN_M", "0")
    monkeypatch.setenv("FX_PW_MIN_M", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
