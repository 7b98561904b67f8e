#!/usr/bin/env python
"""Known-byte kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on this box (MI355X_MICROARCH.md §HBM: "calibrate on a known byte
count in your own access pattern").  Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes):
  * fx_cast_f32_bf16 over 512 MiB of fp32: 16 B/lane streaming reads (the conv kernels' pattern), 8 B/lane writes: reads 512 MiB, writes 256 MiB;
  * fx_add_rows_bf16 (x + y, same shape) over 256 MiB bf16 tensors: reads 512 MiB, writes 256 MiB with 16 B/lane on both sides.
Buffers are far larger than the 256 MiB Infinity Cache.  scripts/pmc_summary.py --calib reads the resulting counter CSVs and derives
bytes-per-count factors, which it then applies instead of assumed ones."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from focoos_amd import _lib  # noqa: E402
from focoos_amd._lib import check  # noqa: E402

lib = _lib.load()
dev = "cuda:0"
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
n = 128 << 20
x = torch.randn(n, device=dev)                      # 512 MiB fp32
y = torch.empty(n, dtype=torch.bfloat16, device=dev)
a = torch.randn(n // 256, 256, device=dev).bfloat16()   # 256 MiB bf16
b = torch.randn(n // 256, 256, device=dev).bfloat16()
o = torch.empty_like(a)
for _ in range(5):
    check(lib.fx_cast_f32_bf16(x.data_ptr(), y.data_ptr(), n, st))
    check(lib.fx_add_rows_bf16(a.data_ptr(), 256, b.data_ptr(), 256, a.shape[0], o.data_ptr(), 256, a.shape[0], 256, st))
torch.cuda.synchronize()
print("calibration kernels done: cast_f32_bf16_kernel reads", n * 4, "writes", n * 2, "; add_rows_kernel reads", 2 * a.numel() * 2, "writes", a.numel() * 2)
