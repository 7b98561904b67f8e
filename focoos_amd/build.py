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
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", obj]
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
