"""Part 0 (16 images) on stream s0 while an unrelated torch matmul loop runs on s1: does foreign concurrency alone break it?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from focoos_amd.model import FAIDetr
from focoos_amd.registry import ModelRegistry
from focoos_amd.synth import synth_image_structured as sis
cfg = ModelRegistry.get_model_info("fai-detr-l-obj365")["config"]
model = FAIDetr(cfg, device="cuda:0", seed=0)
eng = model.engine
B = 16
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
mode = sys.argv[2] if len(sys.argv) > 2 else "matmul"
imgs = torch.from_numpy(np.stack([sis(100 + i) for i in range(B)])).to("cuda:0")
pl = eng.plan(B, 640, 640, False, 1)
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
def run(x, foreign):
    pl.input.copy_(x); pl.sizes.copy_(torch.tensor([[640, 640]] * B, dtype=torch.int32).cuda())
    torch.cuda.synchronize()
    if foreign:
        with torch.cuda.stream(s1):
            if mode == "matmul":
                a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
                for _ in range(60):
                    a = (a @ a).clamp_(-1, 1)
            else:  # memory-bound elementwise traffic
                a = torch.randn(64 * 1024 * 1024, device="cuda")
                for _ in range(60):
                    a = a * 1.0001 + 0.5
    pl._launch(pl.ops, s0.cuda_stream, 0.3)
    torch.cuda.synchronize()
    return pl.probs.clone(), pl.boxes.clone()
ref = run(imgs, False)
g = torch.Generator().manual_seed(1)
for foreign in (False, True):
    bad = 0
    for it in range(N):
        perm = torch.randperm(B, generator=g).to("cuda:0")
        p, b = run(imgs[perm].contiguous(), foreign)
        bad += not (torch.equal(p, ref[0][perm]) and torch.equal(b, ref[1][perm]))
    print(f"foreign={foreign} ({mode}) replays={N} mismatching={bad}")
