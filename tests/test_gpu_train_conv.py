"""Training-path convolution kernels on a real MI355X vs torch CPU fp32 autograd of the same op
(F.conv2d on bf16-rounded operands): weight gradient (transpose-read MFMA kernel), input gradient, activation / pooling
backward.  Tolerance: products of bf16 operands accumulated in fp32 -> relative error <= 2e-3 of the gradient's max."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from focoos_amd import _lib  # noqa: E402
from focoos_amd._lib import check  # noqa: E402

DEV = "cuda:0"


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available()
    return _lib.load()


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


WGRAD_CASES = [
    # B, H, W, C, N, k, stride
    (2, 20, 24, 64, 64, 3, 1),      # C = 64: a 128-column tile spans two filter taps
    (2, 17, 19, 128, 256, 1, 1),    # odd sizes: pixel tail
    (1, 32, 32, 32, 64, 3, 1),      # C = 32 (stem conv1_3): four taps per tile, Ktot = 288 (tile tail)
    (3, 16, 16, 128, 128, 3, 2),    # stride 2
    (2, 40, 40, 256, 64, 1, 1),     # N = 64: half-empty n tile
    (1, 28, 28, 512, 2048, 1, 1),   # many tiles
    (4, 64, 64, 64, 256, 1, 1),     # M = 16384: several pixel chunks accumulate atomically
    (2, 12, 12, 3 * 8, 40, 3, 1),   # C = 24, N = 40: both below a tile, C not a power of two
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_conv_wgrad(lib, case):
    B, H, W, Cc, N, k, stride = case
    pad = (k - 1) // 2
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, H, W, Cc, generator=g).bfloat16()
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    dz = torch.randn(B, Ho, Wo, N, generator=g).bfloat16()
    w = torch.zeros(N, Cc, k, k, requires_grad=True)
    y = F.conv2d(x.float().permute(0, 3, 1, 2), w, None, stride=stride, padding=pad)
    y.backward(dz.float().permute(0, 3, 1, 2))
    ref = w.grad.permute(0, 2, 3, 1).contiguous()  # [N][KH][KW][C]
    xd, dzd = x.to(DEV), dz.to(DEV)
    dw = torch.zeros(N, k, k, Cc, dtype=torch.float32, device=DEV)
    check(lib.fx_conv2d_wgrad_nhwc_bf16(xd.data_ptr(), Cc, dzd.data_ptr(), N, dw.data_ptr(), B, H, W, Cc, Ho, Wo, N, k, k, stride, pad, stream()))
    torch.cuda.synchronize()
    err = (dw.cpu() - ref).abs().max() / ref.abs().max()
    assert err < 2e-3, f"rel err {err}"
    # accumulation semantics: a second call doubles the result
    check(lib.fx_conv2d_wgrad_nhwc_bf16(xd.data_ptr(), Cc, dzd.data_ptr(), N, dw.data_ptr(), B, H, W, Cc, Ho, Wo, N, k, k, stride, pad, stream()))
    torch.cuda.synchronize()
    assert (dw.cpu() - 2 * ref).abs().max() / ref.abs().max() < 4e-3


def test_conv_wgrad_asymmetric_and_strided_views(lib):
    """Transpose-detecting check (one-hot pixels / channels) and channel-slice views (ldx > C, lddz > N)."""
    B, H, W, Cc, N = 1, 4, 8, 16, 24
    x = torch.zeros(B, H, W, 32)
    dz = torch.zeros(B, H, W, 40)
    x[0, 1, 2, 5] = 3.0       # pixel (1,2), input channel 5
    dz[0, 1, 2, 7] = 2.0      # same pixel, output channel 7
    dz[0, 3, 7, 11] = 1.0     # another pixel with no activation
    xd, dzd = x.bfloat16().to(DEV), dz.bfloat16().to(DEV)
    dw = torch.zeros(N, 1, 1, Cc, dtype=torch.float32, device=DEV)
    check(lib.fx_conv2d_wgrad_nhwc_bf16(xd.data_ptr(), 32, dzd.data_ptr(), 40, dw.data_ptr(), B, H, W, Cc, H, W, N, 1, 1, 1, 0, stream()))
    torch.cuda.synchronize()
    want = torch.zeros(N, 1, 1, Cc)
    want[7, 0, 0, 5] = 6.0
    assert torch.equal(dw.cpu(), want)
