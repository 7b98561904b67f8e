"""Replay stability of the multi-part plan: N replays on fresh permutations must equal the single-part result bit-for-bit."""
import sys, os
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
def run(x, ns):
    pl = eng.plan(B, 640, 640, False, ns)
    with torch.cuda.stream(eng.stream):
        pl.input.copy_(x); pl.sizes.copy_(torch.tensor([[640, 640]] * B, dtype=torch.int32))
        pl.run(eng.stream.cuda_stream, 0.3, None, True)
    eng.stream.synchronize()
    return pl.probs.clone(), pl.boxes.clone()
ref = run(imgs, 1)
bad = 0
g = torch.Generator().manual_seed(1)
for it in range(N):
    perm = torch.randperm(B, generator=g).to("cuda:0")
    p, b = run(imgs[perm].contiguous(), 2)
    ok = torch.equal(p, ref[0][perm]) and torch.equal(b, ref[1][perm])
    bad += (not ok)
print(f"mode={os.environ.get('FX_MULTI_MODE','graphs')} replays={N} mismatching={bad}")
