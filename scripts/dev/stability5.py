"""Which FRONT ops of part 0 disturb the DECODER of part 1 (or are disturbed by it) when they overlap?"""
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
N = int(sys.argv[1])
imgs = torch.from_numpy(np.stack([sis(100 + i) for i in range(B)])).to("cuda:0")
pl1 = eng.plan(B, 640, 640, False, 1)
with torch.cuda.stream(eng.stream):
    pl1.input.copy_(imgs); pl1.sizes.copy_(torch.tensor([[640, 640]] * B, dtype=torch.int32))
    pl1.run(eng.stream.cuda_stream, 0.3, None, True)
eng.stream.synchronize()
ref = (pl1.probs.clone(), pl1.boxes.clone())
pl = eng.plan(B, 640, 640, False, 2)
p0, p1 = pl.parts
nops, split = len(p0.ops), p0.split_at
names = [fn.__name__ for fn, _ in p0.ops]
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
hb = B // 2
def trial(a, b, reps=1):
    bad0 = bad1 = 0
    g = torch.Generator().manual_seed(1)
    for it in range(N):
        perm = torch.randperm(B, generator=g).to("cuda:0")
        pl.input.copy_(imgs[perm]); pl.sizes.copy_(torch.tensor([[640, 640]] * B, dtype=torch.int32).cuda())
        torch.cuda.synchronize()
        p1._launch(p1.ops[:split], s0.cuda_stream, 0.3)          # victim's front, alone
        p0._launch(p0.ops[:a], s0.cuda_stream, 0.3)
        torch.cuda.synchronize()
        for _ in range(reps):
            p0._launch(p0.ops[a:b], s0.cuda_stream, 0.3)          # aggressor candidate range (front ops of part 0)
        p1._launch(p1.ops[split:], s1.cuda_stream, 0.3)           # decoder of part 1, concurrently
        torch.cuda.synchronize()
        p0._launch(p0.ops[b:], s0.cuda_stream, 0.3)
        torch.cuda.synchronize()
        ok0 = torch.equal(pl.probs[:hb], ref[0][perm][:hb]) and torch.equal(pl.boxes[:hb], ref[1][perm][:hb])
        ok1 = torch.equal(pl.probs[hb:], ref[0][perm][hb:]) and torch.equal(pl.boxes[hb:], ref[1][perm][hb:])
        bad0 += not ok0; bad1 += not ok1
    return bad0, bad1
print("names front:", sorted(set(names[:split])))
for a, b in ((0, split), (0, 5), (5, 30), (30, 59), (59, 80), (80, split)):
    print(f"front ops [{a}:{b}] ({names[a]}..{names[b-1]}) overlapping part-1 decoder: bad(part0, part1) =", trial(a, b))
