// Stand-alone reproducer of the two-queue failure, reduced to the kernel the per-unit bisection points at (scripts/dev/pk_bisect.sh:
// only select_ops.hip compiled with packed-fp32 instructions fails, 59 of 60 replays; every other unit may keep them).
// Victim (stream A): the dot-product loop of bbox_head_kernel, per-lane partial sums written out, in two builds of the SAME source -
// with packed-fp32 instructions (partial_pk) and without (partial_nopk).  Aggressor (stream B): a streaming copy.  Every victim
// launch is compared bit for bit with the result of the SAME build run alone (the two builds associate the fp32 sums differently, so
// they are not bit-identical to each other); mismatches are counted per (lane, component).
//   build: scripts/probes/build_probes.sh pk_bbox      run: scripts/probes/bin/pk_bbox_two_queue [launches]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
void launch_partial_pk(const unsigned short*, int, const float*, float*, int, int, hipStream_t);
void launch_partial_nopk(const unsigned short*, int, const float*, float*, int, int, hipStream_t);

__global__ void compare(const unsigned* got, const unsigned* ref, long n, unsigned* hist /* [64][4] */, unsigned* launch_flag) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    if (got[i] != ref[i]) { atomicAdd(&hist[((i >> 2) & 63) * 4 + (i & 3)], 1u); *launch_flag = 1u; }
}
__global__ void aggressor(const float4* src, float4* dst, size_t n, int reps) {
  for (int r = 0; r < reps; ++r)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = src[i]; v.x += 1.0f; dst[i] = v; }
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 300;
  const int rows = 4800, K = 256;
  std::vector<unsigned short> hh((size_t)rows * K);
  std::vector<float> hw(4 * K);
  srand(11);
  for (auto& v : hh) { float f = (float)rand() / (float)RAND_MAX * 2.f - 1.f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  for (auto& v : hw) v = (float)rand() / (float)RAND_MAX - 0.5f;
  unsigned short* dh; float *dw, *dref, *dgot; unsigned *dhist, *dbad;
  const long n = (long)rows * 64 * 4;
  CK(hipMalloc(&dh, hh.size() * 2)); CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMalloc(&dref, n * 4)); CK(hipMalloc(&dgot, n * 4));
  CK(hipMalloc(&dhist, 256 * 4)); CK(hipMalloc(&dbad, 4 * 4096));
  CK(hipMemcpy(dh, hh.data(), hh.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  const size_t n4 = (size_t)1 << 25;
  float4 *src, *dst;
  CK(hipMalloc(&src, n4 * 16)); CK(hipMalloc(&dst, n4 * 16)); CK(hipMemset(src, 0, n4 * 16));
  hipStream_t sa, sb;
  CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
  for (int build = 0; build < 2; ++build) {
    if (build == 0) launch_partial_pk(dh, K, dw, dref, rows, K, sa); else launch_partial_nopk(dh, K, dw, dref, rows, K, sa);
    CK(hipStreamSynchronize(sa));
    for (int mode = 0; mode < 2; ++mode) {   // alone / beside the streaming copy
      CK(hipMemset(dhist, 0, 1024)); CK(hipMemset(dbad, 0, 4 * 4096));
      if (mode == 1) aggressor<<<4096, 256, 0, sb>>>(src, dst, n4, 30);
      for (int l = 0; l < launches; ++l) {
        CK(hipMemsetAsync(dgot, 0xff, n * 4, sa));
        if (build == 0) launch_partial_pk(dh, K, dw, dgot, rows, K, sa); else launch_partial_nopk(dh, K, dw, dgot, rows, K, sa);
        compare<<<64, 256, 0, sa>>>((const unsigned*)dgot, (const unsigned*)dref, n, dhist, dbad + (l & 4095));
      }
      CK(hipStreamSynchronize(sa));
      const bool overlapped = mode == 0 || hipStreamQuery(sb) == hipErrorNotReady;
      CK(hipDeviceSynchronize());
      unsigned hist[256], bad = 0;
      std::vector<unsigned> flags(4096);
      CK(hipMemcpy(hist, dhist, 1024, hipMemcpyDeviceToHost)); CK(hipMemcpy(flags.data(), dbad, 4 * 4096, hipMemcpyDeviceToHost));
      for (unsigned f : flags) bad += f;
      unsigned long long tot = 0, hi = 0, comp[4] = {0, 0, 0, 0};
      for (int i = 0; i < 256; ++i) { tot += hist[i]; comp[i & 3] += hist[i]; if (i / 4 >= 48) hi += hist[i]; }
      printf("%s, %s: %u of %d launches differ from the reference; %llu wrong values (%llu in lanes 48-63; by component %llu %llu %llu %llu)%s\n",
             build == 0 ? "WITH packed fp32" : "without packed fp32", mode == 0 ? "alone" : "beside a streaming copy", bad, launches, tot, hi,
             comp[0], comp[1], comp[2], comp[3], overlapped ? "" : "  [aggressor finished early]");
      if (tot) { printf("  wrong values per lane:"); for (int l = 0; l < 64; ++l) printf(" %u", hist[l * 4] + hist[l * 4 + 1] + hist[l * 4 + 2] + hist[l * 4 + 3]); printf("\n"); }
    }
  }
  return 0;
}
