"""Debug: ConvNormLayer in BN (batch statistics) mode vs torch fp32 autograd, layer by layer."""
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from focoos_amd import _lib  # noqa: E402
from focoos_amd.train_nn import ConvNormLayer, set_norm_mode  # noqa: E402

DEV = "cuda:0"
lib = _lib.load()


def rel(a, b):
    return float((a.float().cpu() - b.float().cpu()).norm() / (b.float().cpu().norm() + 1e-12))


def one(cin, cout, k, stride, act, with_res, B=4, H=20, W=24, seed=0):
    g = torch.Generator().manual_seed(seed)
    layer = ConvNormLayer(lib, cin, cout, k, stride, act)
    set_norm_mode(layer, "BN")
    with torch.no_grad():
        layer._conv_h.weight.copy_(torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5)
        layer._norm_h.weight.copy_(torch.rand(cout, generator=g) + 0.5)
        layer._norm_h.bias.copy_(torch.randn(cout, generator=g) * 0.3)
    layer = layer.to(DEV)
    x = torch.randn(B, H, W, cin, generator=g).bfloat16()
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    res = torch.randn(B, Ho, Wo, cout, generator=g).bfloat16() if with_res else None
    cot = torch.randn(B, Ho, Wo, cout, generator=g).bfloat16()
    xd = x.to(DEV).requires_grad_(True)
    rd = res.to(DEV).requires_grad_(True) if with_res else None
    y = layer(xd, residual=rd)
    y.backward(cot.to(DEV))
    torch.cuda.synchronize()
    # reference
    xt = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    w = layer._conv_h.weight.detach().cpu().clone().requires_grad_(True)
    ga = layer._norm_h.weight.detach().cpu().clone().requires_grad_(True)
    be = layer._norm_h.bias.detach().cpu().clone().requires_grad_(True)
    z = F.conv2d(xt, w.bfloat16().float() + (w - w.detach()), None, stride, k // 2)
    a = F.batch_norm(z, torch.zeros(cout), torch.ones(cout), ga, be, True, 0.1, 1e-5)
    rt = None
    if with_res:
        rt = res.float().permute(0, 3, 1, 2).requires_grad_(True)
        a = a + rt
    yt = {"relu": F.relu, "silu": F.silu, None: lambda t: t}[act](a)
    yt.backward(cot.float().permute(0, 3, 1, 2))
    print(f"cin={cin} cout={cout} k={k} s={stride} act={act} res={with_res}: y {rel(y.detach().permute(0,3,1,2), yt.detach()):.4f} "
          f"dx {rel(xd.grad.permute(0,3,1,2), xt.grad):.4f} dw {rel(layer._conv_h.weight.grad, w.grad):.4f} "
          f"dgamma {rel(layer._norm_h.weight.grad, ga.grad):.4f} dbeta {rel(layer._norm_h.bias.grad, be.grad):.4f}"
          + (f" dres {rel(rd.grad.permute(0,3,1,2), rt.grad):.4f}" if with_res else ""))


one(64, 64, 3, 1, "relu", False)
one(64, 128, 3, 2, "relu", False)
one(64, 256, 1, 1, "relu", True)
one(256, 64, 1, 1, "relu", False)
one(256, 256, 1, 1, "silu", True)
one(32, 32, 3, 1, "relu", False, B=2, H=64, W=80)
