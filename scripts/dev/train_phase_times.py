#!/usr/bin/env python
"""Where the wall time of an RT-DETR training step goes, by phase (events on the main stream at the phase boundaries; the weight gradients
run beside the backward phases on the side stream): backbone fwd, encoder fwd, decoder fwd, criterion, backward of criterion + decoder,
backward of the encoder, backward of the backbone, optimizer."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from focoos_amd.ports import DETRTargets
from focoos_amd.registry import ModelRegistry
from focoos_amd.synth import synth_image, synth_state_dict
from focoos_amd.train_detr import FAIDetrTrainable, TrainStep

dev = "cuda:0"
cfg = ModelRegistry.get_model_info("fai-detr-l-obj365")["config"]
K, B, S = int(cfg["num_classes"]), 16, 640
model = FAIDetrTrainable(cfg, norm="FrozenBN").to(dev)
model.load_state_dict(synth_state_dict(cfg, 0, family="fai_detr"), strict=True)
model.train()
stepper = TrainStep(model)
imgs = torch.stack([torch.from_numpy(synth_image(i, S, S)) for i in range(B)]).to(dev)
rs = np.random.RandomState(0)
tg = []
for _ in range(B):
    t = rs.randint(1, 21)
    bx = np.concatenate([rs.uniform(0.2, 0.8, (t, 2)), rs.uniform(0.05, 0.35, (t, 2))], -1).astype(np.float32)
    tg.append(DETRTargets(labels=torch.from_numpy(rs.randint(0, K, (t,))).to(dev), boxes=torch.from_numpy(bx).to(dev)))

marks = []


import time


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((name, e, time.perf_counter()))


def wrap(mod, before, after):
    orig = mod.forward

    def f(*a, **k):
        if before:
            mark(before)
        out = orig(*a, **k)
        mark(after)
        return out

    mod.forward = f


wrap(model.pixel_decoder.backbone, "start", "backbone fwd")
wrap(model.pixel_decoder, None, "encoder fwd")
wrap(model.head.predictor, None, "decoder fwd")
wrap(model.head.criterion, None, "criterion fwd")
orig_nw = model._notify_when_all_grads


def nw(tensors, name):
    live = [t for t in tensors if t.requires_grad]
    left = [len(live)]

    def hook(_g):
        left[0] -= 1
        if left[0] == 0:
            mark({"head": "criterion + decoder bwd", "encoder": "encoder bwd"}[name])

    for t in live:
        t.register_hook(hook)
    orig_nw(tensors, name)


model._notify_when_all_grads = nw
orig_join = stepper._join_wgrads


def join():
    mark("backbone bwd")
    orig_join()
    mark("wgrad join")


stepper._join_wgrads = join
for _ in range(4):
    marks.clear()
    stepper.step(imgs, tg)
acc, hacc = {}, {}
N = 6
allm = []
torch.cuda.synchronize()
for _ in range(N):   # back to back, no sync between the steps: the host may run ahead of the GPU as in the bench loop
    marks.clear()
    stepper.step(imgs, tg)
    mark("optimizer")
    allm.append(list(marks))
torch.cuda.synchronize()
for ms_ in allm[1:]:
    for (n0, e0, h0), (n1, e1, h1) in zip(ms_[:-1], ms_[1:]):
        acc[n1] = acc.get(n1, 0.0) + e0.elapsed_time(e1)
        hacc[n1] = hacc.get(n1, 0.0) + (h1 - h0) * 1e3
n_ = len(allm) - 1
print(f"{'phase':28s} {'GPU ms':>8s} {'host ms':>8s}   (host = time the Python threads spent issuing the phase)")
for k, v in acc.items():
    print(f"{k:28s} {v / n_:8.3f} {hacc[k] / n_:8.3f}")
print(f"{'total':28s} {sum(acc.values()) / n_:8.3f} {sum(hacc.values()) / n_:8.3f}")
first, last = allm[1][0], allm[-1][-1]
print(f"wall per step over {n_} back-to-back steps: GPU {first[1].elapsed_time(last[1]) / n_:.3f} ms, host {(last[2] - first[2]) * 1e3 / n_:.3f} ms")
