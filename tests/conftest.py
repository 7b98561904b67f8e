import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The halo 3x3 kernel is routed to from M >= 40000 pixels in production (smaller layers do not fill the chip); the parity
# tests run small shapes through it too.  Read once by the library at its first convolution call.
os.environ.setdefault("FX_CONV3_MIN_M", "0")
os.environ.setdefault("FX_PW_MIN_M", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
