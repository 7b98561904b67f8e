#!/bin/bash
# Builds the stand-alone probes into scripts/probes/bin/ (git-ignored; travels to the GPU box with the gpurun snapshot).
set -e
cd "$(dirname "$0")"
mkdir -p bin
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -Xclang -target-feature -Xclang -packed-fp32-ops"
for src in "${@:-c3_probe.hip}"; do
  out=bin/$(basename "$src" .hip)
  echo "hipcc $src -> $out" >&2
  hipcc $FLAGS $EXTRA "$src" -o "$out" 2>&1 | grep -v "not a recognized feature" >&2 || true
done
