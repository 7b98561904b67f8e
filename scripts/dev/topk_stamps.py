"""s_memtime stamps of the exact top-k kernel's phases (workgroup 0): FX_TOPK_DBG diagnostic of csrc/select_ops.hip."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
dbg = torch.zeros(8, dtype=torch.int64, device="cuda:0")
os.environ["FX_TOPK_DBG"] = hex(dbg.data_ptr())
from focoos_amd import _lib  # noqa: E402

lib = _lib.load()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
names = ["load keys", "threshold search", "collect", "sort", "write"]
for B, n, k, two in ((16, 8400, 300, False), (16, 109500, 300, True), (16, 4200, 300, False)):
    s = torch.sigmoid(torch.randn(B, n, device="cuda:0") * 2)
    val = torch.empty(B, k, dtype=torch.float32, device="cuda:0")
    idx = torch.empty(B, k, dtype=torch.int32, device="cuda:0")
    nws = lib.fx_topk_rows_workspace_bytes(B, n, k) if two else 0
    ws = torch.empty(max(nws, 8), dtype=torch.uint8, device="cuda:0")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(5):
        dbg.zero_()
        e0.record()
        if two:
            lib.fx_topk_rows_ws_f32(s.data_ptr(), n, B, n, k, val.data_ptr(), idx.data_ptr(), ws.data_ptr(), C.c_size_t(nws), st)
        else:
            lib.fx_topk_rows_f32(s.data_ptr(), n, B, n, k, val.data_ptr(), idx.data_ptr(), st)
        e1.record()
        torch.cuda.synchronize()
    h = dbg.cpu().tolist()
    d = [h[i + 1] - h[i] for i in range(5)]
    # s_memtime counts at 100 MHz on gfx950 (constant clock): ticks x 10 ns
    print(f"B={B} n={n} k={k} two_level={two}: launch {e0.elapsed_time(e1) * 1e3:.1f} us; last launch's workgroup 0 ticks {dict(zip(names, d))} total {h[5] - h[0]}")
