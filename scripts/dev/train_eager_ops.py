"""Which source lines of the training step launch the torch eager kernels?  (rocprofv3: ~2500 of the ~3700 launches per RT-DETR training
step are at::native / rocclr copies.)  Profiles one step with the torch profiler (python stacks on) and attributes every kernel launch
to the innermost focoos_amd frame of the op that issued it.  Also prints the host time to ISSUE a step against its GPU time."""
import collections, copy, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
import bench

sys.argv = ["bench.py", "--train"] + sys.argv[1:]
args = bench.parse()
from focoos_amd.ports import DETRTargets
from focoos_amd.registry import ModelRegistry
from focoos_amd.synth import synth_image, synth_state_dict
from focoos_amd.train_detr import FAIDetrTrainable, TrainStep

dev = "cuda:0"
cfg = ModelRegistry.get_model_info(args.model)["config"]
K, B, S = int(cfg["num_classes"]), args.batch, args.size
model = FAIDetrTrainable(cfg, norm=args.norm).to(dev)
model.load_state_dict(synth_state_dict(cfg, 0, family="fai_detr"), strict=True)
model.train()
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


tg = [targets(i) for i in range(8)]
for i in range(3):
    stepper.step(imgs, tg[i])
torch.cuda.synchronize()
# host issue time vs GPU time
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
t0 = time.perf_counter(); e0.record()
for i in range(3, 6):
    stepper.step(imgs, tg[i])
e1.record(); t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"3 steps: host issue {1e3 * t_issue / 3:.2f} ms/step, GPU span {e0.elapsed_time(e1) / 3:.2f} ms/step", flush=True)

import traceback
from torch.utils._python_dispatch import TorchDispatchMode

SKIP = {"aten::view", "aten::_unsafe_view", "aten::reshape", "aten::select", "aten::slice", "aten::unsqueeze", "aten::squeeze", "aten::expand", "aten::permute",
        "aten::transpose", "aten::t", "aten::detach", "aten::alias", "aten::as_strided", "aten::unbind", "aten::split", "aten::empty", "aten::empty_like",
        "aten::empty_strided", "aten::new_empty", "aten::_local_scalar_dense", "aten::lift_fresh", "aten::is_nonzero", "aten::unflatten", "aten::flatten", "aten::narrow", "aten::new_empty_strided", "aten::view_as"}


class Count(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.by = collections.Counter(); self.ops = collections.defaultdict(collections.Counter); self.bw = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.name().split(".")[0]
        if name not in SKIP:
            where = "<backward / no focoos frame>"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if "focoos_amd/" in fr.filename and not fr.filename.endswith("_lib.py"):
                    where = f"{fr.filename.split('focoos_amd/')[-1]}:{fr.lineno} {fr.name}"
                    break
            self.by[where] += 1; self.ops[where][name] += 1
            if where.startswith("<backward"):
                shp = tuple(tuple(a.shape) for a in args if isinstance(a, torch.Tensor))
                self.bw[(name, shp, str(args[0].dtype) if args and isinstance(args[0], torch.Tensor) else "")] += 1
        return func(*args, **(kwargs or {}))


with Count() as cnt:
    stepper.step(imgs, tg[6])
torch.cuda.synchronize()
print("aten ops (non-view) per step:", sum(cnt.by.values()))
for where, n in cnt.by.most_common(90):
    ops = ", ".join(f"{o.replace('aten::', '')}x{c}" for o, c in cnt.ops[where].most_common(8))
    print(f"{n:5d}  {where[:70]:70s} {ops[:150]}")

print("backward-thread aten ops by (op, shapes):")
for (name, shp, dt), n in cnt.bw.most_common(45):
    print(f"{n:5d}  {name:28s} {dt:16s} {shp}")

# ---- per-layer weight-gradient table (events around _conv_param_grads: wgrad + slab sum + unpack)
from focoos_amd import train_nn
orig = train_nn._conv_param_grads
rec = []


def timed(layer, x, dz, scale):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = orig(layer, x, dz, scale); e1.record()
    rec.append((e0, e1, tuple(x.shape), tuple(dz.shape), layer.k, layer.stride))
    return out


train_nn._conv_param_grads = timed
stepper.wgrad_stream = None   # events are recorded on the main stream
stepper.step(imgs, tg[7])
train_nn._conv_param_grads = orig
torch.cuda.synchronize()
rows = collections.defaultdict(lambda: [0, 0.0])
for e0, e1, xs, zs, k, st in rec:
    r = rows[(xs, zs, k, st)]; r[0] += 1; r[1] += e0.elapsed_time(e1)
print(f"conv weight gradients: {len(rec)} calls, {sum(r[1] for r in rows.values()):.3f} ms")
for (xs, zs, k, st), (n, ms) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    fl = 2.0 * zs[0] * zs[1] * zs[2] * zs[3] * xs[3] * k * k
    by = 2.0 * (np.prod(xs) + np.prod(zs))
    print(f"  x{xs} dz{zs} k{k} s{st}: {n} calls {ms:7.3f} ms  {ms / n * 1e3:7.1f} us each  {fl / (ms / n * 1e-3) / 1e12:6.1f} TF/s  {by / (ms / n * 1e-3) / 1e9:7.1f} GB/s alg")
