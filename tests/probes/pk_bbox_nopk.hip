// compiled WITHOUT packed-fp32 instructions (-target-feature -packed-fp32-ops, the library's flags)
#include <hip/hip_runtime.h>
#define KNAME partial_nopk
#include "pk_bbox_kernel.inc"
void launch_partial_nopk(const unsigned short* h, int ldh, const float* w, float* out, int rows, int K, hipStream_t s) {
  partial_nopk<<<(rows + 3) / 4, 256, 0, s>>>(h, ldh, w, out, rows, K);
}
