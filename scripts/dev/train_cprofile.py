"""Dev: cProfile of the host side of the training step (the step is host-bound: ~5000 eager launches)."""
import cProfile
import pstats
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from focoos_amd.ports import DETRTargets  # noqa: E402
from focoos_amd.registry import ModelRegistry  # noqa: E402
from focoos_amd.synth import synth_image, synth_state_dict  # noqa: E402
from focoos_amd.train_detr import FAIDetrTrainable, TrainStep  # noqa: E402

dev = "cuda:0"
cfg = ModelRegistry.get_model_info("fai-detr-l-obj365")["config"]
K, B, S = int(cfg["num_classes"]), 16, 640
model = FAIDetrTrainable(cfg).to(dev)
model.load_state_dict(synth_state_dict(cfg, 0), strict=True)
stepper = TrainStep(model)
imgs = torch.stack([torch.from_numpy(synth_image(i, S, S)) for i in range(B)]).to(dev)
rs = np.random.RandomState(0)
tg = []
for _ in range(B):
    t = rs.randint(1, 21)
    bx = np.concatenate([rs.uniform(0.2, 0.8, (t, 2)), rs.uniform(0.05, 0.35, (t, 2))], -1).astype(np.float32)
    tg.append(DETRTargets(labels=torch.from_numpy(rs.randint(0, K, (t,))).to(dev), boxes=torch.from_numpy(bx).to(dev)))
for _ in range(3):
    stepper.step(imgs, tg)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    stepper.step(imgs, tg)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(38)
