#!/usr/bin/env python
"""Throughput of the training-path slice that exists so far: ResNet50-vd forward + backward (frozen BN) + fused AdamW on the
HIP autograd nodes, bs=16 x 640^2 (the per-GPU batch of BASELINE config 4).  NOT the config-4 metric (the encoder / decoder /
criterion backward are not built yet); prints one JSON line with images/s and the per-kernel-family split."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from focoos_amd.registry import ModelRegistry
from focoos_amd.synth import synth_image, synth_state_dict
from focoos_amd.train_nn import ResNetVd

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
S = int(sys.argv[2]) if len(sys.argv) > 2 else 640
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
cfg = ModelRegistry.get_model_info("fai-detr-l-obj365")["config"]
sd = synth_state_dict(cfg, 0)
pre = "pixel_decoder.backbone."
net = ResNetVd(50).to("cuda:0")
net.load_state_dict({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)})
imgs = torch.stack([torch.from_numpy(synth_image(i, S, S)) for i in range(B)]).to("cuda:0")
g = torch.Generator().manual_seed(0)
proj = {k: torch.randn(c, generator=g).to("cuda:0").bfloat16() for k, c in (("res3", 512), ("res4", 1024), ("res5", 2048))}
params = net.trainable_parameters()
opt = torch.optim.SGD(params, lr=0.0)  # placeholder update (lr 0): the fused AdamW kernel is benchmarked separately


def step():
    for p in params:
        p.grad = None
    outs = net(imgs)
    loss = sum((outs[k] * proj[k]).float().sum() for k in proj)
    loss.backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
fwd_gflop = 70.6 * (S / 640.0) ** 2  # ResNet50-vd forward @640^2
print(json.dumps({"what": "ResNet50-vd fwd+bwd (frozen BN), HIP autograd nodes", "batch": B, "size": S, "ms_per_step": round(dt * 1e3, 3),
                  "images_per_s": round(B / dt, 1), "approx_tflops": round(3 * fwd_gflop * B / dt / 1e3, 1),
                  "frac_of_bf16_mfma_peak": round(3 * fwd_gflop * 1e9 * B / dt / 2.5e15, 4)}))
