"""Bandwidth of the slab-sum / re-layout pass of the weight gradients (fx_unpack_conv_wgrad_sum_f32) on typical layer shapes.  (dev tool; GPU)"""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from focoos_amd import _lib  # noqa: E402
from focoos_amd._lib import check  # noqa: E402

lib = _lib.load()
dev = "cuda:0"
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (N, Cc, k, S) in [(256, 256, 3, 18), (256, 256, 3, 8), (512, 512, 3, 9), (256, 1024, 1, 40), (1024, 256, 1, 40), (256, 64, 1, 64), (64, 64, 3, 205), (2048, 512, 1, 10),
                      (256, 256, 1, 50)]:
    slab = N * k * k * Cc
    ws = torch.randn(S * slab, device=dev)
    scale = torch.rand(N, device=dev)
    out = torch.zeros(N, Cc, k, k, device=dev)
    for _ in range(3):
        check(lib.fx_unpack_conv_wgrad_sum_f32(ws.data_ptr(), slab, S, scale.data_ptr(), out.data_ptr(), N, Cc, k, k, Cc, 1, st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        check(lib.fx_unpack_conv_wgrad_sum_f32(ws.data_ptr(), slab, S, scale.data_ptr(), out.data_ptr(), N, Cc, k, k, Cc, 1, st))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    by = (S + 2) * slab * 4
    ref = (ws.view(S, N, k, k, Cc).sum(0).permute(0, 3, 1, 2) * scale.view(-1, 1, 1, 1))
    out.zero_()
    check(lib.fx_unpack_conv_wgrad_sum_f32(ws.data_ptr(), slab, S, scale.data_ptr(), out.data_ptr(), N, Cc, k, k, Cc, 1, st))
    torch.cuda.synchronize()
    err = float((out - ref).abs().max() / ref.abs().max())
    print(f"N={N:5d} C={Cc:5d} k={k} S={S:4d}: {us:7.1f} us, {by / 1e6:7.1f} MB -> {by / us / 1e6:6.2f} TB/s   (max rel err {err:.1e})")
