"""EXPERIMENT (round 6): throughput of the RT-DETR step with the lanes PINNED to disjoint CU sets (hipExtStreamCreateWithCUMask) instead of
sharing all 256 CUs dynamically.  usage: python scripts/dev/cumask_probe.py [nlanes] [mode]   mode: none | split (contiguous bit ranges) |
interleave (bit i -> lane i % n)"""
import ctypes
import sys
import time

import numpy as np
import torch

from focoos_amd.model import FAIDetr
from focoos_amd.registry import ModelRegistry
from focoos_amd.synth import synth_image
from focoos_amd.engine import _Plan

nl = int(sys.argv[1]) if len(sys.argv) > 1 else 2
mode = sys.argv[2] if len(sys.argv) > 2 else "split"
dev = torch.device("cuda:0")
cfg = ModelRegistry.get_model_info("fai-detr-l-obj365")["config"]
eng = FAIDetr(cfg, device="cuda:0", seed=0).engine
B = 32
imgs = torch.stack([torch.from_numpy(synth_image(i, 640, 640)) for i in range(B)]).to(dev)
sizes = torch.tensor([[640, 640]] * B, dtype=torch.int32, device=dev)
hip = ctypes.CDLL("libamdhip64.so")
NCU = torch.cuda.get_device_properties(0).multi_processor_count
streams = []
for j in range(nl):
    if mode == "none":
        streams.append(torch.cuda.Stream(dev))
        continue
    bits = np.zeros(NCU, dtype=np.uint8)
    if mode == "split":
        bits[j * NCU // nl:(j + 1) * NCU // nl] = 1
    else:
        bits[j::nl] = 1
    words = np.packbits(bits.reshape(-1, 32)[:, ::-1], axis=1).view(">u4").astype(np.uint32).ravel()
    arr = (ctypes.c_uint32 * len(words))(*[int(w) for w in words])
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(len(words)), arr)
    assert rc == 0, rc
    streams.append(torch.cuda.ExternalStream(st.value, device=dev))
plans = [_Plan(eng, B, 640, 640, False) for _ in range(nl)]


def step(i):
    pl, st = plans[i % nl], streams[i % nl]
    with torch.cuda.stream(st):
        pl.input.copy_(imgs, non_blocking=True)
        pl.sizes.copy_(sizes, non_blocking=True)
        pl.run(st.cuda_stream, 0.5, None, True)


for i in range(2 * nl):
    step(i)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(30):
        step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"lanes {nl} mode {mode}: {B * 30 / dt:8.1f} img/s  {dt / 30 * 1e3:.3f} ms/step")
