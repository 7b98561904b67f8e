"""Which autograd node runs in two stages of TrainStep._staged_backward?  (debug aid, GPU)"""
import collections
import torch
from focoos_amd.registry import ModelRegistry
from focoos_amd.train_detr import FAIDetrTrainable, TrainStep
from focoos_amd.synth import synth_state_dict
import bench

dev = "cuda:0"
cfg = ModelRegistry.get_model_info("fai-detr-l-coco")["config"]
model = FAIDetrTrainable(cfg, norm="FrozenBN").to(dev)
model.load_state_dict(synth_state_dict(cfg, 3), strict=True)
model.train()
st = TrainStep(model, staged=True, graphs=False)
imgs = torch.randint(0, 255, (2, 128, 160, 3), dtype=torch.uint8, device=dev)
targets = bench.synth_train_targets("fai_detr", 0, 0, 2, 128, 80, dev)
seen = collections.defaultdict(list)
stage = ["?"]
orig = st._staged_backward


def walk(roots):
    out, stack, vis = [], [t.grad_fn for t in roots if t.grad_fn is not None], set()
    while stack:
        n = stack.pop()
        if n is None or id(n) in vis:
            continue
        vis.add(id(n))
        out.append(n)
        for nx, _ in n.next_functions:
            stack.append(nx)
    return out


def patched(roots, grads, between=None, wrap=None):
    nodes = walk(roots)
    print("nodes below the prediction sets:", len(nodes))
    for n in nodes:
        n.register_prehook(lambda g, n=n: seen[id(n)].append((stage[0], type(n).__name__)))

    def wr(name, body):
        stage[0] = name
        try:
            body()
        except RuntimeError as e:
            print("FAILED in stage", name, str(e)[:80])
            dup = [(k, v) for k, v in seen.items() if len(v) > 1]
            print("nodes run more than once:", [v for _, v in dup][:10])
            raise
    return orig(roots, grads, between, wr)


st._staged_backward = patched
try:
    st.step(imgs, targets)
    print("ok")
except RuntimeError:
    bnd = model.segment_boundaries
    for k, ts in bnd.items():
        print(k, [(type(t.grad_fn).__name__, t.grad_fn is not None and len(seen[id(t.grad_fn)])) for t in ts])
