"""Which Python lines launch the small aten kernels of an RT-DETR training step?  (dev tool; GPU)
torch.profiler with stacks over one eager step; aten ops grouped by the innermost focoos_amd frame."""
import collections
import sys

import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, ".")
from focoos_amd.ports import DETRTargets  # noqa: E402
from focoos_amd.registry import ModelRegistry  # noqa: E402
from focoos_amd.synth import synth_image, synth_state_dict  # noqa: E402
from focoos_amd.train_detr import FAIDetrTrainable, TrainStep  # noqa: E402

dev = "cuda:0"
cfg = ModelRegistry.get_model_info("fai-detr-l-obj365")["config"]
B, S = 16, 640
model = FAIDetrTrainable(cfg, norm="FrozenBN").to(dev)
model.load_state_dict(synth_state_dict(cfg, 0), strict=True)
stepper = TrainStep(model, lr=1e-4)
imgs = torch.stack([torch.from_numpy(synth_image(i, S, S)) for i in range(B)]).to(dev)
rs = np.random.RandomState(0)
targets = []
for i in range(B):
    t = int(rs.randint(1, 21))
    cxcy, wh = rs.uniform(0.2, 0.8, (t, 2)), rs.uniform(0.05, 0.3, (t, 2))
    targets.append(DETRTargets(labels=torch.from_numpy(rs.randint(0, 365, t)).to(dev), boxes=torch.from_numpy(np.concatenate([cxcy, wh], 1)).float().to(dev)))
for _ in range(4):
    stepper.step(imgs, targets)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    stepper.step(imgs, targets)
    torch.cuda.synchronize()
by = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0:
        continue
    if ev.cpu_children:      # count leaves only (the op that launched the kernel)
        if any(c.name.startswith("aten::") and c.device_time_total > 0 for c in ev.cpu_children):
            continue
    site = next((f for f in (ev.stack or []) if "focoos_amd/" in f), (ev.stack[0] if ev.stack else "(autograd engine / no Python frame)"))
    site = site.split("focoos_amd/")[-1][:70]
    k = (ev.name, site)
    by[k][0] += 1
    by[k][1] += ev.device_time_total
rows = sorted(by.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
print(f"aten kernels of one step: {sum(v[0] for _, v in rows)} launches, {tot / 1e3:.2f} ms device time")
for (name, site), (n, t) in rows[:45]:
    print(f"{t / 1e3:7.3f} ms {n:4d} x  {name:28s} {site}")
