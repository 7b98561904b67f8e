#!/usr/bin/env python
"""Times the grouped MSDA backward (6 layers, 16 x 300 queries, 80/40/20 levels) in both value-gradient forms; run under
rocprofv3 --kernel-trace --stats for the per-kernel split."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from focoos_amd.train import ValueGradSink, ms_deform_attn_grouped

DEV = "cuda:0"
G, B, Q = 6, 16, 300
shapes = [[80, 80], [40, 40], [20, 20]]
S = sum(h * w for h, w in shapes)
g = torch.Generator().manual_seed(0)
value_all = torch.randn(B, S, G * 256, generator=g).bfloat16().to(DEV).requires_grad_()
locs = [(torch.rand(B, Q, 8, 3, 4, 2, generator=g)).to(DEV).requires_grad_() for _ in range(G)]
aws = [torch.softmax(torch.randn(B, Q, 8, 12, generator=g), -1).view(B, Q, 8, 3, 4).to(DEV).requires_grad_() for _ in range(G)]
gos = [torch.randn(B, Q, 256, generator=g).bfloat16().to(DEV) for _ in range(G)]
for mode in ("0", "1"):
    os.environ["FX_MSDA_BWD_SLAB"] = mode
    for it in range(6):
        if it == 2:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        sink = ValueGradSink(G)
        outs = [ms_deform_attn_grouped(value_all, sink, i, shapes, locs[i], aws[i]) for i in range(G)]
        torch.autograd.backward(outs, gos)
        value_all.grad = None
    torch.cuda.synchronize()
    print(f"slab={mode}: {(time.perf_counter() - t0) / 4 * 1e3:.3f} ms per fwd+bwd of {G} layers")
