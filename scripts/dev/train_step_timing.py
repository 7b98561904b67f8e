import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from focoos_amd.ports import DETRTargets
from focoos_amd.registry import ModelRegistry
from focoos_amd.synth import synth_image, synth_state_dict
from focoos_amd.train_detr import FAIDetrTrainable, TrainStep
from focoos_amd import train_nn as nn_
dev = "cuda:0"
cfg = ModelRegistry.get_model_info("fai-detr-l-obj365")["config"]
K, B, S = 365, 16, 640
model = FAIDetrTrainable(cfg).to(dev)
model.load_state_dict(synth_state_dict(cfg, 0), strict=True)
st = TrainStep(model)
imgs = torch.stack([torch.from_numpy(synth_image(i, S, S)) for i in range(B)]).to(dev)
def targets(it):
    rs = np.random.RandomState(it); out = []
    for _ in range(B):
        t = rs.randint(1, 21)
        bx = np.concatenate([rs.uniform(0.2, 0.8, (t, 2)), rs.uniform(0.05, 0.35, (t, 2))], -1).astype(np.float32)
        out.append(DETRTargets(labels=torch.from_numpy(rs.randint(0, K, (t,))).to(dev), boxes=torch.from_numpy(bx).to(dev)))
    return out
for it in range(3): st.step(imgs, targets(it))
torch.cuda.synchronize()
T = {k: 0.0 for k in ("zero", "backbone", "encoder", "predictor", "criterion", "backward", "opt")}
Tg = dict(T)
def lap(name, t0, sync):
    t1 = time.perf_counter(); T[name] += t1 - t0
    if sync:
        torch.cuda.synchronize(); t2 = time.perf_counter(); Tg[name] += t2 - t0; return t2
    return t1
N = 5
for sync in (False, True):
    for k in T: T[k] = 0.0; Tg[k] = 0.0
    for it in range(N):
        tg = targets(10 + it)
        torch.cuda.synchronize(); t = time.perf_counter()
        st.opt.zero_grad()
        for n, p in st.named: p.grad = st.opt.grads[n]
        nn_.ARENA.arm(st.opt.numel + (8 << 20), st.opt.dev); nn_.DIRECT_GRAD[0] = True
        t = lap("zero", t, sync)
        f = model.pixel_decoder.backbone(imgs); t = lap("backbone", t, sync)
        enc = model.pixel_decoder([f["res3"], f["res4"], f["res5"]]); t = lap("encoder", t, sync)
        out = model.head.predictor(enc); t = lap("predictor", t, sync)
        losses = model.head.criterion(out, tg); total = sum(losses.values()); t = lap("criterion", t, sync)
        total.backward(); t = lap("backward", t, sync)
        nn_.DIRECT_GRAD[0] = False
        st.opt.step(); nn_.WEIGHTS_EPOCH[0] += 1; t = lap("opt", t, sync)
        torch.cuda.synchronize()
    print("sync" if sync else "async (host enqueue time)", {k: round(1e3 * (Tg[k] if sync else T[k]) / N, 2) for k in T})
