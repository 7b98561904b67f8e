"""Dev: torch.profiler view of one training step (which aten ops launch the glue kernels) + step timing."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from focoos_amd.ports import DETRTargets  # noqa: E402
from focoos_amd.registry import ModelRegistry  # noqa: E402
from focoos_amd.synth import synth_image, synth_state_dict  # noqa: E402
from focoos_amd.train_detr import FAIDetrTrainable, TrainStep  # noqa: E402

dev = "cuda:0"
norm = sys.argv[1] if len(sys.argv) > 1 else "FrozenBN"
cfg = ModelRegistry.get_model_info("fai-detr-l-obj365")["config"]
K, B, S = int(cfg["num_classes"]), 16, 640
model = FAIDetrTrainable(cfg, norm=norm).to(dev)
model.load_state_dict(synth_state_dict(cfg, 0), strict=True)
stepper = TrainStep(model)
imgs = torch.stack([torch.from_numpy(synth_image(i, S, S)) for i in range(B)]).to(dev)


def targets(it):
    rs = np.random.RandomState(it)
    out = []
    for _ in range(B):
        t = rs.randint(1, 21)
        bx = np.concatenate([rs.uniform(0.2, 0.8, (t, 2)), rs.uniform(0.05, 0.35, (t, 2))], -1).astype(np.float32)
        out.append(DETRTargets(labels=torch.from_numpy(rs.randint(0, K, (t,))).to(dev), boxes=torch.from_numpy(bx).to(dev)))
    return out


for it in range(3):
    stepper.step(imgs, targets(it))
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(5):
    stepper.step(imgs, targets(3 + it))
torch.cuda.synchronize()
print(f"{norm}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms/step")
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    stepper.step(imgs, targets(9))
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=70))
