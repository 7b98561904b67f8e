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
names = {0: "LOAD", 1: "GEMM", 2: "GEMM_LN", 3: "ADD", 4: "K4", 5: "BBOX"}
with torch.cuda.stream(st):
    pl.input.copy_(imgs)
    pl.sizes.copy_(torch.tensor([[640, 640]] * B, dtype=torch.int32))
    for _ in range(2):
        pl._launch(pl.ops, st.cuda_stream, 0.3)
st.synchronize()
# launch row-chain ops one by one and dump stamps
import ctypes as C
for i, op in enumerate(pl.ops):
    fn = op[0]
    if getattr(fn, "__name__", "") != "fx_row_chain" and "row_chain" not in str(pl.meta.get(i, "")):
        continue
    dbg.zero_()
    with torch.cuda.stream(st):
        pl._launch([op], st.cuda_stream, 0.3)
    st.synchronize()
    h = dbg.cpu().tolist()
    n = max(k for k in range(64) if h[k] != 0)
    d = [h[k + 1] - h[k] for k in range(n)]
    print(pl.meta.get(i), "total cycles", h[n] - h[0], "per stage:", d[1:])
    if "dec2.post_msda" in str(pl.meta.get(i)):
        break
