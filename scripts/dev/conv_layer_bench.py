"""Single-layer timing of fx_conv2d_nhwc_bf16 (HIP events over back-to-back launches on rotating buffers, so that the input is not
L2-resident from the previous launch).  usage: python scripts/dev/conv_layer_bench.py B,H,W,C,N,k,stride[,act] ...   (env knobs apply)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from focoos_amd import _lib
from focoos_amd._lib import FX_ACT, FxConvDesc, check

lib = _lib.load()
DEV = "cuda:0"


def frag_pack(W2):
    N, K = W2.shape
    w = W2.float().reshape(N // 32, 32, K // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous()
    return w.reshape(N // 32, K // 16, 64, 8).to(DEV, torch.bfloat16)


def bench(spec):
    f = spec.split(",")
    B, H, W, Cc, N, k, stride = map(int, f[:7])
    act = f[7] if len(f) > 7 else "relu"
    pad = (k - 1) // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    NB = 6
    xs = [torch.randn(B, H, W, Cc, device=DEV).bfloat16() for _ in range(NB)]
    ys = [torch.empty(B, Ho, Wo, N, device=DEV, dtype=torch.bfloat16) for _ in range(NB)]
    W4 = torch.randn(N, Cc, k, k) / (Cc * k * k) ** 0.5
    Np = (N + 127) // 128 * 128
    w = torch.zeros(Np, k, k, Cc)
    w[:N] = W4.permute(0, 2, 3, 1)
    wd = w.to(DEV, torch.bfloat16)
    bd = torch.zeros(Np, device=DEV)
    wf = frag_pack(W4.permute(0, 2, 3, 1).reshape(N, k * k * Cc))
    descs = []
    for x, y in zip(xs, ys):
        d = FxConvDesc()
        d.x, d.w, d.bias, d.y = x.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr()
        d.B, d.H, d.W, d.C, d.ldx = B, H, W, Cc, Cc
        d.Ho, d.Wo, d.N, d.ldy = Ho, Wo, N, N
        d.KH, d.KW, d.stride, d.pad, d.act = k, k, stride, pad, FX_ACT[act if act != "none" else None]
        d.w_frag = wf.data_ptr()
        descs.append(d)
    label = C.create_string_buffer(64)
    check(lib.fx_conv2d_variant(C.byref(descs[0]), label, 64), "variant")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for d in descs:
        check(lib.fx_conv2d_nhwc_bf16(C.byref(d), st), "conv")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        for d in descs:
            check(lib.fx_conv2d_nhwc_bf16(C.byref(d), st), "conv")
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * NB)
    fl = 2.0 * B * Ho * Wo * N * k * k * Cc
    by = 2.0 * (B * H * W * Cc + B * Ho * Wo * N)
    print(f"{spec:36s} {label.value.decode():28s} {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s  {by / us / 1e3:7.0f} GB/s")


for s in sys.argv[1:]:
    bench(s)
