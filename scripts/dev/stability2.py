"""Concurrency hazard probe: the two batch parts launched EAGERLY (no hipGraph) on two streams."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from focoos_amd.model import FAIDetr
from focoos_amd.registry import ModelRegistry
from focoos_amd.synth import synth_image_structured as sis
cfg = ModelRegistry.get_model_info("fai-detr-l-obj365")["config"]
model = FAIDetr(cfg, device="cuda:0", seed=0)
eng = model.engine
B = 32
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
imgs = torch.from_numpy(np.stack([sis(100 + i) for i in range(B)])).to("cuda:0")
pl1 = eng.plan(B, 640, 640, False, 1)
with torch.cuda.stream(eng.stream):
    pl1.input.copy_(imgs); pl1.sizes.copy_(torch.tensor([[640, 640]] * B, dtype=torch.int32))
    pl1.run(eng.stream.cuda_stream, 0.3, None, True)
eng.stream.synchronize()
ref = (pl1.probs.clone(), pl1.boxes.clone())
pl = eng.plan(B, 640, 640, False, 2)
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
bad = 0
g = torch.Generator().manual_seed(1)
for it in range(N):
    perm = torch.randperm(B, generator=g).to("cuda:0")
    pl.input.copy_(imgs[perm]); pl.sizes.copy_(torch.tensor([[640, 640]] * B, dtype=torch.int32).cuda())
    torch.cuda.synchronize()
    pl.parts[0]._launch(pl.parts[0].ops, s0.cuda_stream, 0.3)
    pl.parts[1]._launch(pl.parts[1].ops, s1.cuda_stream, 0.3)
    torch.cuda.synchronize()
    ok = torch.equal(pl.probs, ref[0][perm]) and torch.equal(pl.boxes, ref[1][perm])
    bad += (not ok)
print(f"eager two-stream replays={N} mismatching={bad}")
