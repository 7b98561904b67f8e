"""Build libfocoos_amd.so (hand-written HIP kernels for gfx950) in-tree with hipcc.

The .so is git-ignored but travels to the GPU box with the gpurun snapshot."""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libfocoos_amd.so")
ARCH = "gfx950"
# The kernels are compiled WITHOUT packed-fp32 VALU instructions (v_pk_mul/add/fma_f32, v_pk_mov_b32): on MI355X / ROCm 7.2 a wave that
# executes them while waves of a second hardware queue are resident on its CU computes wrong values in lanes 48-63 of one operand
# (DESIGN.md §5 "two-queue hazard": bisected to fx_bbox_head, per-lane dumps, 56-58 of 60 concurrent replays wrong with the feature on, 0 of
# 60 with it off; single-queue execution is unaffected).  FX_PK_F32=1 re-enables them (the reproducer).
EXTRA_FLAGS = [] if os.environ.get("FX_PK_F32", "0") == "1" else ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain is required to build the gfx950 kernels)")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(PKG, "..", "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    objs = []
    hipcc = _hipcc()
    for src in sources():
        obj = os.path.join(LIB_DIR, os.path.basename(src).replace(".hip", ".o"))
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC"] + EXTRA_FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
