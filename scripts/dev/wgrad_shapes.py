"""Dev: per-shape throughput of fx_conv2d_wgrad_nhwc_bf16 and of the forward conv on the RT-DETR training shapes (bs=16, 640^2)."""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from focoos_amd import _lib  # noqa: E402
from focoos_amd._lib import FxConvDesc, check  # noqa: E402

lib = _lib.load()
DEV = "cuda:0"
B = 16
shapes = [  # (name, H, C, N, k, stride)
    ("conv1_2", 320, 32, 32, 3, 1), ("conv1_3", 320, 32, 64, 3, 1),
    ("res2.a", 160, 64, 64, 1, 1), ("res2.a'", 160, 256, 64, 1, 1), ("res2.b", 160, 64, 64, 3, 1), ("res2.c", 160, 64, 256, 1, 1),
    ("res3.a0", 160, 256, 128, 1, 1), ("res3.b0", 160, 128, 128, 3, 2), ("res3.a", 80, 512, 128, 1, 1), ("res3.b", 80, 128, 128, 3, 1), ("res3.c", 80, 128, 512, 1, 1),
    ("res4.a0", 80, 512, 256, 1, 1), ("res4.b0", 80, 256, 256, 3, 2), ("res4.a", 40, 1024, 256, 1, 1), ("res4.b", 40, 256, 256, 3, 1), ("res4.c", 40, 256, 1024, 1, 1),
    ("res5.a0", 40, 1024, 512, 1, 1), ("res5.b0", 40, 512, 512, 3, 2), ("res5.a", 20, 2048, 512, 1, 1), ("res5.b", 20, 512, 512, 3, 1), ("res5.c", 20, 512, 2048, 1, 1),
    ("enc.3x3@80", 80, 256, 256, 3, 1), ("enc.3x3@40", 40, 256, 256, 3, 1), ("enc.1x1@80", 80, 512, 256, 1, 1), ("enc.1x1@80b", 80, 256, 256, 1, 1),
]
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
tot_w = tot_f = 0.0
for name, H, Cc, N, k, s in shapes:
    pad = (k - 1) // 2
    Ho = (H + 2 * pad - k) // s + 1
    x = torch.randn(B, H, H, Cc, device=DEV).bfloat16()
    dz = torch.randn(B, Ho, Ho, N, device=DEV).bfloat16()
    dw = torch.zeros(N, k, k, Cc, device=DEV)
    w = torch.randn((N + 127) // 128 * 128, k, k, Cc, device=DEV).bfloat16()
    y = torch.empty(B, Ho, Ho, N, device=DEV, dtype=torch.bfloat16)
    d = FxConvDesc()
    d.x, d.w, d.y, d.bias, d.residual = x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None
    d.B, d.H, d.W, d.C, d.ldx = B, H, H, Cc, Cc
    d.Ho, d.Wo, d.N, d.ldy, d.ldr = Ho, Ho, N, N, 0
    d.KH, d.KW, d.stride, d.pad = k, k, s, pad
    d.pool2, d.act, d.out_f32, d.residual_after_act, d.y_batch_stride = 0, 0, 0, 0, 0

    def wg():
        check(lib.fx_conv2d_wgrad_nhwc_bf16(x.data_ptr(), Cc, dz.data_ptr(), N, dw.data_ptr(), B, H, H, Cc, Ho, Ho, N, k, k, s, pad, st))

    def fw():
        check(lib.fx_conv2d_nhwc_bf16(C.byref(d), st))

    res = []
    for fn in (wg, fw):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 10)
    fl = 2.0 * B * Ho * Ho * N * k * k * Cc
    tot_w += res[0]
    tot_f += res[1]
    print(f"{name:12s} M={B * Ho * Ho:7d} N={N:4d} K={k * k * Cc:5d}  wgrad {res[0] * 1e3:7.1f} us {fl / res[0] / 1e9:6.0f} TF/s   fwd {res[1] * 1e3:7.1f} us {fl / res[1] / 1e9:6.0f} TF/s")
print(f"sum wgrad {tot_w:.3f} ms, fwd {tot_f:.3f} ms (one of each shape)")
