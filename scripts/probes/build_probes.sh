#!/bin/bash
# Builds the stand-alone probes into scripts/probes/bin/ (git-ignored; travels to the GPU box with the gpurun snapshot).
#   scripts/probes/build_probes.sh [file.hip ...]     default: every probe
set -e
cd "$(dirname "$0")"
mkdir -p bin
BASE="--offload-arch=gfx950 -O3 -std=c++17"
NOPK="-Xclang -target-feature -Xclang -packed-fp32-ops"    # the library's own flags (focoos_amd/build.py)
[ $# -eq 0 ] && set -- c3_probe.hip ../../tests/probes/pk_f32_two_queue.hip pk_bbox
for src in "$@"; do
  if [ "$src" = pk_bbox ]; then    # one kernel source in two builds (with / without packed fp32) + the driver
    P=../../tests/probes
    echo "hipcc pk_bbox_{pk,nopk,two_queue}.hip -> bin/pk_bbox_two_queue" >&2
    hipcc $BASE -c $P/pk_bbox_pk.hip -o bin/pk_bbox_pk.o
    hipcc $BASE $NOPK -c $P/pk_bbox_nopk.hip -o bin/pk_bbox_nopk.o 2>&1 | grep -v "not a recognized feature" >&2 || true
    hipcc $BASE $NOPK -c $P/pk_bbox_two_queue.hip -o bin/pk_bbox_main.o 2>&1 | grep -v "not a recognized feature" >&2 || true
    hipcc --offload-arch=gfx950 bin/pk_bbox_pk.o bin/pk_bbox_nopk.o bin/pk_bbox_main.o -o bin/pk_bbox_two_queue
    rm -f bin/pk_bbox_*.o
    continue
  fi
  out=bin/$(basename "$src" .hip)
  flags="$BASE $NOPK"
  case "$src" in *pk_f32*) flags="$BASE" ;; esac      # this probe IS the packed-fp32 instruction class
  echo "hipcc $src -> $out" >&2
  hipcc $flags $EXTRA "$src" -o "$out" 2>&1 | grep -v "not a recognized feature" >&2 || true
done
