"""Where does the wall time of the two-part inference step go?  rocprofv3's kernel trace serialises the two hardware queues, so the
concurrent schedule is measured by truncation instead: the step is cut after op k of BOTH parts (graphs of the prefixes, replayed
concurrently exactly like the full step) and timed; the increments are the marginal wall cost of each stage under concurrency, printed
next to the stage's serial kernel-time sum (per-op HIP events).  usage: python scripts/dev/prefix_wall.py [--streams N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from focoos_amd.model import FAIDetr
from focoos_amd.registry import ModelRegistry
from focoos_amd.synth import synth_image

B, S = int(os.environ.get("PW_B", "32")), 640
cfg = ModelRegistry.get_model_info("fai-detr-l-obj365")["config"]
model = FAIDetr(cfg, device="cuda:0", seed=0)
eng = model.engine
imgs = torch.stack([torch.from_numpy(synth_image(i, S, S)) for i in range(B)]).to("cuda:0")
pl = eng.plan(B, S, S, False)
parts = getattr(pl, "parts", [pl])
st = eng.stream
full = [list(p.ops) for p in parts]
n = len(full[0])
# stage boundaries from the op metadata of part 0
cuts, names = [], []
meta = parts[0].meta
def mark(i, name):
    cuts.append(i); names.append(name)
for i in range(n):
    m = meta.get(i)
    d = (m["name"] if m else "")
    fn = full[0][i][0].__name__
    if d.endswith("res_layers.0.blocks.0.a"): mark(i, "stem")
    elif d.endswith("res_layers.1.blocks.0.b"): mark(i, "res2")
    elif d.endswith("res_layers.2.blocks.0.b"): mark(i, "res3")
    elif d.endswith("res_layers.3.blocks.0.a"): mark(i, "res4")
    elif m and m["variant"].startswith("score_head"): mark(i, "encoder")
    elif d == "value_all": mark(i, "select")
    elif d == "dec0.pre": mark(i, "value_all+bbox")
    elif d.endswith(".post_msda"): mark(i + 1, d.split(".")[0])
    elif fn == "fx_detr_head_out": pass
res5_end = [i for i in range(n) if (meta.get(i) or {}).get("name", "").endswith("res_layers.3.blocks.2.c")][0] + 1
cuts.append(res5_end); names.append("res5")
order = sorted(range(len(cuts)), key=lambda j: cuts[j])
cuts = [cuts[j] for j in order] + [n]; names = [names[j] for j in order] + ["post"]

def wall(k, iters=30):
    for p, ops in zip(parts, full):
        p.ops = ops[:k]
        if hasattr(p, "graph") and p.graph is not None:
            p.graph = None
    pl.graph = None
    def step():
        with torch.cuda.stream(st):
            pl.run(st.cuda_stream, 0.5, None, True)
    for _ in range(5):
        step()
    st.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    st.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3

with torch.cuda.stream(st):
    pl.input.copy_(imgs)
    pl.sizes.copy_(torch.tensor([[S, S]] * B, dtype=torch.int32))
prev, rows = 0.0, []
for k, nm in zip(cuts, names):
    w = wall(k)
    rows.append((nm, k, w, w - prev))
    prev = w
print(f"{'stage':16s} {'ops':>5s} {'cum wall ms':>12s} {'marginal ms':>12s}")
for nm, k, w, d in rows:
    print(f"{nm:16s} {k:5d} {w:12.3f} {d:12.3f}")
