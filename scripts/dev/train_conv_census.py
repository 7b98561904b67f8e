#!/usr/bin/env python
"""Census of the fx_conv2d_nhwc_bf16 calls of ONE training step: which kernel the library routes each call to (fx_conv2d_variant),
its shape, how many times it runs and its event-bracketed time (events on the stream the step runs on; the weight gradients stay
on the side stream and are not in this table).

    python scripts/dev/train_conv_census.py [--model fai-detr-l-obj365|bisenetformer-l-ade] [--norm FrozenBN|BN]
"""
import argparse
import ctypes as C
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="fai-detr-l-obj365")
    ap.add_argument("--norm", default="FrozenBN")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--size", type=int, default=None)
    a = ap.parse_args()
    from focoos_amd import _lib, train_nn
    from focoos_amd.ports import DETRTargets, MaskFormerTargets
    from focoos_amd.registry import ModelRegistry
    from focoos_amd.synth import synth_image, synth_state_dict
    from focoos_amd.train_detr import FAIDetrTrainable, TrainStep

    bf = a.model.startswith("bisenetformer")
    B = a.batch or (8 if bf else 16)
    S = a.size or (1024 if bf else 640)
    dev = "cuda:0"
    cfg = ModelRegistry.get_model_info(a.model)["config"]
    K = int(cfg["num_classes"])
    if bf:
        from focoos_amd.train_bf import BisenetFormerTrainable
        model = BisenetFormerTrainable(cfg, norm=a.norm).to(dev)
    else:
        model = FAIDetrTrainable(cfg, norm=a.norm).to(dev)
    model.load_state_dict(synth_state_dict(cfg, 0, family="bisenetformer" if bf else "fai_detr"), strict=True)
    model.train()
    stepper = TrainStep(model)
    imgs = torch.stack([torch.from_numpy(synth_image(i, S, S)) for i in range(B)]).to(dev)
    rs = np.random.RandomState(0)
    tg = []
    for _ in range(B):
        if bf:
            t = rs.randint(5, 16)
            m = np.zeros((t, S, S), bool)
            for i in range(t):
                y0, x0 = rs.randint(0, S - 32), rs.randint(0, S - 32)
                m[i, y0:y0 + rs.randint(32, S // 2), x0:x0 + rs.randint(32, S // 2)] = True
            tg.append(MaskFormerTargets(labels=torch.from_numpy(rs.randint(0, K, (t,))).to(dev), masks=torch.from_numpy(m).to(dev)))
        else:
            t = rs.randint(1, 21)
            bx = np.concatenate([rs.uniform(0.2, 0.8, (t, 2)), rs.uniform(0.05, 0.35, (t, 2))], -1).astype(np.float32)
            tg.append(DETRTargets(labels=torch.from_numpy(rs.randint(0, K, (t,))).to(dev), boxes=torch.from_numpy(bx).to(dev)))
    for _ in range(3):
        stepper.step(imgs, tg)
    torch.cuda.synchronize()

    lib = _lib.load()
    real = lib.fx_conv2d_nhwc_bf16
    rec = []

    def wrapped(ref, st):
        d = ref._obj
        label = C.create_string_buffer(64)
        lib.fx_conv2d_variant(ref, label, 64)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = real(ref, st)
        e1.record()
        rec.append((label.value.decode(), (d.B, d.H, d.W, d.C, d.N, d.KH, d.stride, int(bool(d.residual)), d.act), e0, e1))
        return rc

    lib.fx_conv2d_nhwc_bf16 = wrapped
    try:
        stepper.step(imgs, tg)
        torch.cuda.synchronize()
    finally:
        lib.fx_conv2d_nhwc_bf16 = real
    agg = defaultdict(lambda: [0, 0.0])
    for lab, shp, e0, e1 in rec:
        k = (lab, shp)
        agg[k][0] += 1
        agg[k][1] += e0.elapsed_time(e1)
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for v in agg.values())
    print(f"{len(rec)} conv calls, {tot:.3f} ms (event-bracketed, wgrad on the side stream beside them)")
    print(f"{'variant':34s} {'B,H,W,C,N,k,s,res,act':36s} {'n':>3s} {'ms':>8s} {'us/call':>8s} {'TF/s':>7s}")
    for (lab, shp), (n, ms) in rows:
        Bn, H, W, Cc, N, k, s, _, _ = shp
        fl = 2.0 * Bn * (H // s) * (W // s) * Cc * N * k * k * n
        print(f"{lab:34s} {str(shp):36s} {n:3d} {ms:8.3f} {ms / n * 1e3:8.1f} {fl / ms / 1e9:7.0f}")
    by = defaultdict(float)
    for (lab, _), (n, ms) in agg.items():
        by[lab] += ms
    print("-- by variant")
    for lab, ms in sorted(by.items(), key=lambda kv: -kv[1]):
        print(f"{lab:34s} {ms:8.3f}")


if __name__ == "__main__":
    main()
