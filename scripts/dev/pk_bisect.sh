#!/bin/bash
# Per-translation-unit bisection of the two-queue failure: libraries in which only SOME units are compiled with
# packed-fp32 instructions, each run through the two-concurrent-parts test (60 replays must equal the serial result).
#   build (CPU container):  scripts/dev/pk_bisect.sh build
#   run   (GPU box):        scripts/dev/pk_bisect.sh run > gpurun_out/pk_bisect.txt
set -u
cd "$(dirname "$0")/../.."
VARIANTS=("all:1" "select_only:select_ops.hip" "token_only:token_ops.hip" "all_but_select:ALLBUT:select_ops.hip" "all_but_select_token:ALLBUT:select_ops.hip,token_ops.hip")
units() { ls focoos_amd/csrc/*.hip | xargs -n1 basename; }
[[ ${1:-} == build || ${1:-} == run ]] && for v in "${VARIANTS[@]}"; do
  name=${v%%:*}; spec=${v#*:}
  if [[ $spec == ALLBUT:* ]]; then
    excl=${spec#ALLBUT:}
    spec=$(units | grep -v -F -x -f <(echo "$excl" | tr , '\n') | paste -sd, -)
  fi
  lib=focoos_amd/lib/variants/libfocoos_amd_pk_$name.so
  if [[ ${1:-} == build ]]; then
    mkdir -p focoos_amd/lib/variants
    FX_PK_F32=$spec python -m focoos_amd.build --out=$lib >/dev/null 2>&1 && echo "built $lib (packed fp32 in: $spec)"
  else
    echo "== $name: packed fp32 in [$spec]"
    FOCOOS_AMD_LIB=$lib FX_ALLOW_PK_TWO_QUEUES=1 timeout 300 python -m pytest tests/test_gpu_two_streams.py -q -k "equal_serial_parts and not mask" 2>&1 | grep -E "passed|failed|concurrent replays differ" | head -3
  fi
done

# ---- second level: kernels of select_ops.hip (the only unit whose packed-fp32 build fails).  select_ops.hip is compiled WITH packed fp32
# and -DFX_SELECT_PK_MASK=<bit of one kernel> (every other kernel gets __attribute__((target("no-packed-fp32-ops")))), linked with the
# product's other objects.    scripts/dev/pk_bisect.sh kbuild | krun
KERNELS=("linear_k4_relu:16" "bbox_head:32" "detr_postprocess:128" "bbox_head+linear_k4:48")
if [[ ${1:-} == kbuild || ${1:-} == krun ]]; then
  for kv in "${KERNELS[@]}"; do
    name=${kv%%:*}; mask=${kv#*:}
    lib=focoos_amd/lib/variants/libfocoos_amd_pksel_$name.so
    if [[ $1 == kbuild ]]; then
      mkdir -p focoos_amd/lib/variants
      hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DFX_SELECT_PK_MASK=$mask -c focoos_amd/csrc/select_ops.hip -o /tmp/select_ops_pk.o &&
        hipcc --offload-arch=gfx950 -shared -fPIC -o $lib $(ls focoos_amd/lib/*.o | grep -v select_ops.o) /tmp/select_ops_pk.o && echo "built $lib (mask $mask)"
    else
      echo "== select_ops.hip: packed fp32 only in $name"
      FOCOOS_AMD_LIB=$lib timeout 300 python -m pytest tests/test_gpu_two_streams.py -q -k "equal_serial_parts and not mask" 2>&1 | grep -E "passed|failed|concurrent replays differ" | head -3
    fi
  done
fi
