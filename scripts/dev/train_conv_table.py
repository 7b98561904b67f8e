"""Forward / input-gradient convolution launches of one RT-DETR training step by kernel variant and shape (the weight gradients have their
own table: FX_WGRAD_TABLE in bench.py).  Each train_nn._conv_call is bracketed by events on the step's stream (weight gradients on the
same stream for this step, so nothing overlaps).  usage: python scripts/dev/train_conv_table.py [model] [norm]"""
import collections
import ctypes as C
import os
import sys

import numpy as np
import torch

os.environ.setdefault("FX_WGRAD_STREAM", "0")
sys.path.insert(0, ".")
from focoos_amd import train_nn as NN  # noqa: E402
from focoos_amd.ports import DETRTargets  # noqa: E402
from focoos_amd.registry import ModelRegistry  # noqa: E402
from focoos_amd.synth import synth_image, synth_state_dict  # noqa: E402
from focoos_amd.train_detr import FAIDetrTrainable, TrainStep  # noqa: E402

dev = "cuda:0"
name = sys.argv[1] if len(sys.argv) > 1 else "fai-detr-l-obj365"
cfg = ModelRegistry.get_model_info(name)["config"]
B, S = 16, 640
model = FAIDetrTrainable(cfg, norm=sys.argv[2] if len(sys.argv) > 2 else "FrozenBN").to(dev)
model.load_state_dict(synth_state_dict(cfg, 0), strict=True)
stepper = TrainStep(model, lr=1e-4)
imgs = torch.stack([torch.from_numpy(synth_image(i, S, S)) for i in range(B)]).to(dev)
rs = np.random.RandomState(0)
targets = []
for i in range(B):
    t = int(rs.randint(1, 21))
    cxcy, wh = rs.uniform(0.2, 0.8, (t, 2)), rs.uniform(0.05, 0.3, (t, 2))
    targets.append(DETRTargets(labels=torch.from_numpy(rs.randint(0, cfg["num_classes"], t)).to(dev),
                               boxes=torch.from_numpy(np.concatenate([cxcy, wh], 1)).float().to(dev)))
for _ in range(3):
    stepper.step(imgs, targets)
torch.cuda.synchronize()

rec = []
orig = NN._conv_call


def wrapped(lib, x, w, bias, N, KH, KW, stride, pad, act, residual, res_mode=0, out_f32=False, w_frag=None, mask=None):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    y = orig(lib, x, w, bias, N, KH, KW, stride, pad, act, residual, res_mode, out_f32, w_frag, mask)
    e1.record()
    Bx, H, W_, Cc = x.shape
    key = (w.data_ptr(), bias.data_ptr() if bias is not None else 0, Bx, H, W_, Cc, N, KH, KW, stride, pad, act, residual is not None, res_mode, out_f32,
           w_frag.data_ptr() if w_frag is not None else 0, mask is not None)
    d = NN._DESC_CACHE[key][0]
    buf = C.create_string_buffer(128)
    lib.fx_conv2d_variant(C.byref(d), buf, 128)
    rec.append((buf.value.decode(), y.shape[0] * y.shape[1] * y.shape[2], N, KH * KW * Cc, stride, act, residual is not None, mask is not None, e0, e1))
    return y


NN._conv_call = wrapped
for modname in ("focoos_amd.train_detr", "focoos_amd.train", "focoos_amd.train_mf", "focoos_amd.train_bf"):
    m = sys.modules.get(modname)
    if m is not None and hasattr(m, "_conv_call"):
        m._conv_call = wrapped
stepper.step(imgs, targets)
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for v, M, N, K, s, act, res, mask, e0, e1 in rec:
    a = agg[(v, M, N, K, s, act, res, mask)]
    a[0] += 1
    a[1] += e0.elapsed_time(e1)
tot = sum(a[1] for a in agg.values())
print(f"{len(rec)} conv launches, {tot:.2f} ms (events around each launch, one stream)")
byv = collections.defaultdict(lambda: [0, 0.0])
for (v, *_), (n, t) in agg.items():
    byv[v][0] += n
    byv[v][1] += t
for v, (n, t) in sorted(byv.items(), key=lambda kv: -kv[1][1]):
    print(f"  {t:7.3f} ms {n:4d} x  {v}")
print()
for (v, M, N, K, s, act, res, mask), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    fl = 2.0 * M * N * K * n
    print(f"{t:7.3f} ms {n:3d} x {t / n * 1e3:7.1f} us {fl / t / 1e9:7.1f} TF/s  {v:34s} M={M} N={N} K={K} s={s} act={act} res={int(res)} mask={int(mask)}")
