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
# the same sources with the 16-bit storage element = IEEE fp16 (csrc/common.h FX_FP16): the training library of the fp16 + loss-scale step
LIB_PATH_FP16 = os.path.join(LIB_DIR, "libfocoos_amd_fp16.so")
ARCH = "gfx950"
# The kernels are compiled WITHOUT packed-fp32 VALU instructions (v_pk_mul/add/fma_f32, v_pk_mov_b32): on MI355X / ROCm 7.2 a wave that
# executes them while waves of a second hardware queue are resident on its CU computes wrong values in lanes 48-63 of one operand
# (DESIGN.md §5 "two-queue hazard": bisected to fx_bbox_head, per-lane dumps, 56-58 of 60 concurrent replays wrong with the feature on, 0 of
# 60 with it off; single-queue execution is unaffected).  FX_PK_F32=1 re-enables them (the reproducer).
NO_PK_FLAGS = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
STAMP_PATH = os.path.join(LIB_DIR, "build_stamp.txt")
EXTRA_FLAGS: list = []


def _pk_units():
    """Translation units compiled WITH packed-fp32 instructions: none (default), all (FX_PK_F32=1) or a comma-separated list of
    file names (FX_PK_F32=select_ops.hip,token_ops.hip - the per-unit bisection of scripts/dev/pk_bisect.sh)."""
    v = os.environ.get("FX_PK_F32", "0")
    if v in ("", "0"):
        return set()
    if v == "1":
        return {os.path.basename(s) for s in sources()}
    return {u.strip() for u in v.split(",") if u.strip()}


def _stamp() -> str:
    return "arch=%s pk_units=%s extra=%s" % (ARCH, ",".join(sorted(_pk_units())) or "-", " ".join(EXTRA_FLAGS) or "-")


def _flags_key(flags) -> str:
    """What an object was compiled with (everything of its command line but the paths) + this recipe's mtime: kept in `<obj>.flags` so that an
    object left behind by a failed / interrupted build under OTHER flags (FX_PK_F32=1 ...) is never linked into a default library - the
    stamp of the last COMPLETED build says nothing about such objects; an edit of this recipe recompiles everything."""
    import hashlib

    with open(os.path.abspath(__file__), "rb") as f:     # content, not mtime: the snapshot on a GPU box need not keep file times
        return " ".join(flags) + " recipe=" + hashlib.sha1(f.read()).hexdigest()[:12]


def _obj_fresh(obj: str, src: str, hdr_t: float, key: str) -> bool:
    side = obj + ".flags"
    if not (os.path.exists(obj) and os.path.exists(side)) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
        return False
    with open(side) as f:
        return f.read().strip() == key


def _compile_jobs(hipcc, objdir, pk, force, hdr_t, fp16):
    """(objects, jobs, sidecars): one compile job per stale unit; the sidecar of a unit is written only after ITS compile succeeded."""
    objs, jobs, sides = [], [], []
    for src in sources():
        base = os.path.basename(src)
        obj = os.path.join(objdir, base.replace(".hip", ".o"))
        objs.append(obj)
        flags = ([] if base in pk else NO_PK_FLAGS) + (["-DFX_FP16=1"] if fp16 else [])
        if base == "runtime.hip":
            flags = flags + [f"-DFX_BUILD_FLAGS={1 if pk else 0}"]
        flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC"] + flags + EXTRA_FLAGS
        key = _flags_key(flags)
        if not force and _obj_fresh(obj, src, hdr_t, key):
            continue
        if os.path.exists(obj + ".flags"):
            os.remove(obj + ".flags")
        jobs.append([hipcc] + flags + ["-c", src, "-o", obj])
        sides.append((obj + ".flags", key))
    return objs, jobs, sides


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain is required to build the gfx950 kernels)")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build() -> bool:
    """Sources, headers or this recipe newer than the library, or a library built with other flags (the stamp file: a library once
    built with FX_PK_F32=1 must not be silently reused by a default run - ADVICE r2)."""
    if not os.path.exists(LIB_PATH) or not os.path.exists(STAMP_PATH):
        return True
    with open(STAMP_PATH) as f:
        if f.read().strip() != _stamp():
            return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(PKG, "..", "include", "*.h")) + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, out_path: str = None, fp16: bool = False) -> str:
    """Compile every csrc/*.hip for gfx950 and link the shared library (`out_path`: a variant library next to the product one, used by
    the packed-fp32 bisection; the product path keeps its stamp file).  ``fp16``: the fp16-element library (objects under lib/fp16/)."""
    if fp16 and out_path is None:
        return _build_fp16(force, verbose)
    if out_path is None and not force and not needs_build():
        return LIB_PATH
    target = out_path or LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    objdir = LIB_DIR if out_path is None else (target + ".obj")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    pk = _pk_units()
    # per-unit incremental build: a unit is recompiled when its object is missing, older than its source / any header, or was compiled
    # with other flags (its `.flags` sidecar); stale units compile in parallel (hipcc takes 5-60 s per unit).  The stamp of the product
    # library is removed BEFORE compiling and written only after a successful link.
    headers = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(PKG, "..", "include", "*.h"))
    hdr_t = max(os.path.getmtime(h) for h in headers)
    if out_path is None and os.path.exists(STAMP_PATH):
        os.remove(STAMP_PATH)
    objs, jobs, sides = _compile_jobs(hipcc, objdir, pk, force, hdr_t, False)
    _run_jobs(jobs, verbose, sides)
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", target] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    if out_path is None:
        with open(STAMP_PATH, "w") as f:
            f.write(_stamp() + "\n")
    else:
        shutil.rmtree(objdir, ignore_errors=True)
    return target


def _build_fp16(force: bool, verbose: bool) -> str:
    """libfocoos_amd_fp16.so: same recipe with -DFX_FP16=1, own object directory and stamp; rebuilt when a source / header is newer."""
    objdir = os.path.join(LIB_DIR, "fp16")
    os.makedirs(objdir, exist_ok=True)
    stamp = os.path.join(objdir, "build_stamp.txt")
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(PKG, "..", "include", "*.h"))
    fresh = (not force and os.path.exists(LIB_PATH_FP16) and os.path.exists(stamp) and open(stamp).read().strip() == _stamp() + " fp16"
             and all(os.path.getmtime(d) <= os.path.getmtime(LIB_PATH_FP16) for d in deps))
    if fresh:
        return LIB_PATH_FP16
    hipcc, pk = _hipcc(), _pk_units()
    hdr_t = max(os.path.getmtime(h) for h in deps if h.endswith(".h"))
    if os.path.exists(stamp):
        os.remove(stamp)
    objs, jobs, sides = _compile_jobs(hipcc, objdir, pk, force, hdr_t, True)
    _run_jobs(jobs, verbose, sides)
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH_FP16] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(_stamp() + " fp16\n")
    return LIB_PATH_FP16


def _run_jobs(jobs, verbose, sides=()):
    sides = dict(zip((tuple(j) for j in jobs), sides))
    running = []
    nproc = max(1, min(int(os.environ.get("FX_BUILD_JOBS", os.cpu_count() or 4)), 16))
    failed = None
    while (jobs or running) and failed is None:
        while jobs and len(running) < nproc:
            cmd = jobs.pop(0)
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            running.append((cmd, subprocess.Popen(cmd)))
        cmd, pr = running.pop(0)
        if pr.wait() != 0:
            failed = cmd
        elif tuple(cmd) in sides:
            path, key = sides[tuple(cmd)]
            with open(path, "w") as f:
                f.write(key + "\n")
    for cmd, pr in running:
        if pr.wait() == 0 and tuple(cmd) in sides:
            path, key = sides[tuple(cmd)]
            with open(path, "w") as f:
                f.write(key + "\n")
    if failed is not None:
        raise subprocess.CalledProcessError(1, failed)


if __name__ == "__main__":
    out = None
    for a in sys.argv[1:]:
        if a.startswith("--out="):
            out = os.path.abspath(a[len("--out="):])
    print(build(force="--force" in sys.argv, out_path=out))
    if out is None:
        print(build(force="--force" in sys.argv, fp16=True))
