"""Probe for the captured training step: two eager steps, then TrainStep's graph capture + 3 replays (run under rocgdb to get the native
backtrace of a crash inside the HIP runtime).  usage: python scripts/dev/graph_capture_probe.py [model] [norm]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from focoos_amd.ports import DETRTargets
from focoos_amd.registry import ModelRegistry
from focoos_amd.synth import synth_image_structured, synth_state_dict
from focoos_amd.train_detr import FAIDetrTrainable, TrainStep
from oracle import train_oracle as T

DEV = "cuda:0"
cfg = ModelRegistry.get_model_info("fai-detr-l-coco")["config"]
model = FAIDetrTrainable(cfg, norm=sys.argv[2] if len(sys.argv) > 2 else "FrozenBN").to(DEV)
model.load_state_dict(synth_state_dict(cfg, 8), strict=True)
ts = TrainStep(model, lr=1e-4, graphs=True)
B = 2
side = torch.cuda.current_stream()   # TrainStep runs on its own stream now
side.wait_stream(torch.cuda.current_stream())
torch.cuda.set_stream(side)
for it in range(5):
    imgs = torch.from_numpy(np.stack([synth_image_structured(300 + it * B + i, 128, 160) for i in range(B)])).to(DEV)
    labels, boxes = T.synth_targets(60 + it, B, 80, counts=(3, 5))
    targets = [DETRTargets(labels=l.to(DEV), boxes=b.to(DEV)) for l, b in zip(labels, boxes)]
    losses = ts.step(imgs, targets)
    torch.cuda.synchronize()
    print("step", it, "graphed" if ts._graph_state is not None else "eager", float(sum(v.detach().float() for v in losses.values())), flush=True)
