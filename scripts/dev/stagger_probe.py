"""Would a phase shift between the two batch parts help?  (dev probe; GPU)

The production step forks / joins the two 16-image parts every step, so both are in the byte-bound backbone at the same time and in the
MFMA / latency-bound encoder + decoder at the same time.  Here the two per-part hipGraphs free-run on their own streams for K steps
(no join between steps) with part 1 started `offset` ms late: what the throughput of a software-pipelined schedule would be.

  python scripts/dev/stagger_probe.py [K]
"""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, ".")
from focoos_amd._lib import check  # noqa: E402
from focoos_amd.model import FAIDetr  # noqa: E402
from focoos_amd.registry import ModelRegistry  # noqa: E402
from focoos_amd.synth import synth_image  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = "cuda:0"
cfg = ModelRegistry.get_model_info("fai-detr-l-obj365")["config"]
model = FAIDetr(cfg, device=dev, seed=0)
eng = model.engine
B, S = 32, 640
imgs = torch.stack([torch.from_numpy(synth_image(i, S, S)) for i in range(B)]).to(dev)
sizes = torch.tensor([[S, S]] * B, dtype=torch.int32, device=dev)
pl = eng.plan(B, S, S, False)
st = eng.stream
lib = eng.lib


def step():
    with torch.cuda.stream(st):
        pl.input.copy_(imgs, non_blocking=True)
        pl.sizes.copy_(sizes, non_blocking=True)
        pl.run(st.cuda_stream, 0.5, None, True)


for _ in range(5):
    step()
st.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    step()
st.synchronize()
base = (time.perf_counter() - t0) / K * 1e3
print(f"lockstep (fork / join every step): {base:.3f} ms/step = {B / base * 1e3:.0f} img/s")
ref_scores = pl.det_scores.clone()

g0, g1 = pl.graph
side = pl.side[0]
# calibrate torch.cuda._sleep
with torch.cuda.stream(side):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(side)
    torch.cuda._sleep(10_000_000)
    e1.record(side)
side.synchronize()
cyc_per_ms = 10_000_000 / e0.elapsed_time(e1)
for off in (0.0, 1.0, 2.0, 3.0, 3.6, 4.5, 5.5):
    res = []
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if off > 0:
            with torch.cuda.stream(side):
                torch.cuda._sleep(int(off * cyc_per_ms))
        for _ in range(K):
            check(lib.fx_graph_launch(g0, C.c_void_p(st.cuda_stream)), "launch")
            check(lib.fx_graph_launch(g1, C.c_void_p(side.cuda_stream)), "launch")
        st.synchronize()
        side.synchronize()
        res.append((time.perf_counter() - t0) / K * 1e3)
    ok = torch.equal(pl.det_scores, ref_scores)
    print(f"free-running parts, part 1 started {off:.1f} ms late: {min(res):.3f} ms/step (runs {[round(r, 3) for r in res]}) = {B / min(res) * 1e3:.0f} img/s"
          f" (incl. the one-time offset; steady state ~{B / ((min(res) * K - off) / K) * 1e3:.0f}); outputs equal: {ok}")
