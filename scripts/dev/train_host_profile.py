#!/usr/bin/env python
"""cProfile of the host side of the RT-DETR training step (8 steps after warm-up): where the ~20 ms of host issue time per step go."""
import cProfile
import os
import pstats
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
for _ in range(4):
    stepper.step(imgs, tg)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(8):
    stepper.step(imgs, tg)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(60)
