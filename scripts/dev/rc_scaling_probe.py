"""How the time of ONE decoder row-chain launch depends on the number of 32-row workgroups (round 5): the post-MSDA chain of decoder layer 2 of the
bench plan re-launched with rows = 32 (one workgroup on an idle chip) ... 4800 (150 workgroups), HIP-event timed, + workgroup 0's s_memtime stage
stamps.  If one workgroup alone is as slow as 150 together, the chain is bound inside a CU (latency / the CU's own L2 port); if it is much faster
alone, by what the workgroups share (L2 channels, fabric)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
dbg = torch.zeros(64, dtype=torch.int64, device="cuda:0")
os.environ["FX_RC_DBG"] = hex(dbg.data_ptr())
from focoos_amd.model import FAIDetr
from focoos_amd.registry import ModelRegistry
from focoos_amd.synth import synth_image_structured as sis
cfg = ModelRegistry.get_model_info("fai-detr-l-obj365")["config"]
eng = FAIDetr(cfg, device="cuda:0", seed=0).engine
B = 16
imgs = torch.from_numpy(np.stack([sis(100 + i) for i in range(B)])).to("cuda:0")
pl = eng.plan(B, 640, 640, False, 1)
st = eng.stream
with torch.cuda.stream(st):
    pl.input.copy_(imgs)
    pl.sizes.copy_(torch.tensor([[640, 640]] * B, dtype=torch.int32))
    for _ in range(2):
        pl._launch(pl.ops, st.cuda_stream, 0.3)
st.synchronize()
for label in ("dec2.post_msda", "dec2.post_attn"):
    idx = [i for i in range(len(pl.ops)) if label in str(pl.meta.get(i, ""))][0]
    fn, args = pl.ops[idx][0], list(pl.ops[idx][1])
    print(label, "stages", args[1], "rows", args[2], "lds", args[3])
    for rows in (32, 64, 256, 1024, 2048, 4096, 4800):
        a = list(args)
        a[2] = rows
        op = (fn, tuple(a))
        with torch.cuda.stream(st):
            for _ in range(3):
                pl._launch([op], st.cuda_stream, 0.3)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(50):
                pl._launch([op], st.cuda_stream, 0.3)
            e1.record(st)
        st.synchronize()
        dbg.zero_()
        with torch.cuda.stream(st):
            pl._launch([op], st.cuda_stream, 0.3)
        st.synchronize()
        h = dbg.cpu().tolist()
        n = max(k for k in range(64) if h[k] != 0)
        print(f"  rows {rows:5d} ({(rows + 31) // 32:3d} workgroups): {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us per launch; workgroup 0: {h[n] - h[0]} cycles, per stage {[h[k + 1] - h[k] for k in range(1, n)]}")
