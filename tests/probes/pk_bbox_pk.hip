// compiled WITH packed-fp32 instructions (scripts/probes/build_probes.sh)
#include <hip/hip_runtime.h>
#define KNAME partial_pk
#include "pk_bbox_kernel.inc"
void launch_partial_pk(const unsigned short* h, int ldh, const float* w, float* out, int rows, int K, hipStream_t s) {
  partial_pk<<<(rows + 3) / 4, 256, 0, s>>>(h, ldh, w, out, rows, K);
}
