"""Call sites (focoos_amd file:line) of the small torch ops inside one RT-DETR training step: .contiguous() that copies, torch.cat, .to(),
zeros / zero_, elementwise arithmetic on tensors - Python-level wrappers around the calls made from our autograd Functions.  (dev tool; GPU)"""
import collections
import sys
import traceback

import numpy as np
import torch

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
for _ in range(3):
    stepper.step(imgs, targets)
torch.cuda.synchronize()

log = collections.defaultdict(lambda: [0, 0])


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "focoos_amd/" in fr.filename:
            return f"{fr.filename.split('focoos_amd/')[-1]}:{fr.lineno} {fr.line[:60] if fr.line else ''}"
    return "?"


def wrap_method(name, pred):
    orig = getattr(torch.Tensor, name)

    def f(self, *a, **k):
        if self.is_cuda and pred(self, a, k):
            e = log[(name, site())]
            e[0] += 1
            e[1] += self.numel() * self.element_size()
        return orig(self, *a, **k)

    setattr(torch.Tensor, name, f)


wrap_method("contiguous", lambda t, a, k: not t.is_contiguous())
wrap_method("to", lambda t, a, k: True)
wrap_method("float", lambda t, a, k: t.dtype != torch.float32)
wrap_method("zero_", lambda t, a, k: True)
wrap_method("clone", lambda t, a, k: True)
for nm in ("__add__", "__mul__", "__sub__", "__truediv__", "__iadd__", "__imul__"):
    wrap_method(nm, lambda t, a, k: True)
for fn in ("cat", "zeros", "zeros_like", "stack"):
    orig = getattr(torch, fn)

    def mk(orig, fn):
        def f(*a, **k):
            out = orig(*a, **k)
            if isinstance(out, torch.Tensor) and out.is_cuda:
                e = log[(fn, site())]
                e[0] += 1
                e[1] += out.numel() * out.element_size()
            return out
        return f

    setattr(torch, fn, mk(orig, fn))

stepper.step(imgs, targets)
torch.cuda.synchronize()
rows = sorted(log.items(), key=lambda kv: -kv[1][1])
print(f"{sum(v[0] for _, v in rows)} calls, {sum(v[1] for _, v in rows) / 1e6:.0f} MB touched")
for (name, s), (n, b) in rows[:50]:
    print(f"{b / 1e6:9.1f} MB {n:4d} x {name:12s} {s}")
