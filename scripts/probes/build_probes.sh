#!/bin/bash
# Builds the stand-alone probes into scripts/probes/bin/ (git-ignored; travels to the GPU box with the gpurun snapshot).
#   scripts/probes/build_probes.sh [file.hip ...]     default: every probe
set -e
cd "$(dirname "$0")"
mkdir -p bin
BASE="--offload-arch=gfx950 -O3 -std=c++17"
NOPK="-Xclang -target-feature -Xclang -packed-fp32-ops"    # the library's own flags (focoos_amd/build.py)
[ $# -eq 0 ] && set -- c3_probe.hip ../../tests/probes/pk_f32_two_queue.hip
for src in "$@"; do
  out=bin/$(basename "$src" .hip)
  flags="$BASE $NOPK"
  case "$src" in *pk_f32*) flags="$BASE" ;; esac      # this probe IS the packed-fp32 instruction class
  echo "hipcc $src -> $out" >&2
  hipcc $flags $EXTRA "$src" -o "$out" 2>&1 | grep -v "not a recognized feature" >&2 || true
done
