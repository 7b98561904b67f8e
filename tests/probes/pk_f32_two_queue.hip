// Stand-alone reproducer attempt for the two-queue hazard of DESIGN.md section 5: "waves that execute packed-fp32
// VALU instructions compute wrong values in lanes 48-63 while waves of a second hardware queue share their CU".
//   victim  (stream A): one wave per workgroup; every lane runs a chain of v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 (float2 vector
//                       arithmetic, compiled WITH packed-fp32 ops - the build script leaves the library's -packed-fp32-ops flag out for
//                       this file; the compiler, not an asm string, places the instructions, so its hazard padding applies) on inputs
//                       loaded from memory, and the SAME chain with scalar v_fma_f32 / v_mul_f32 / v_add_f32; both are IEEE-exact, so
//                       any difference is a hardware event.  Mismatching lanes are counted per lane index.
//   aggressor (stream B): a bandwidth-bound float4 copy (the class of kernel the bisection named: pw_chain, conv_igemm, ...), optionally
//                       with an LDS-using MFMA-free variant, running concurrently.
// Output: mismatch counts alone / under the aggressor, per lane index.   build: scripts/probes/build_probes.sh ../../tests/probes/pk_f32_two_queue.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ __launch_bounds__(64) void victim(const float* in, unsigned* lane_hist, unsigned long long* checked, int iters) {
  const int lane = threadIdx.x;
  const float* p = in + ((size_t)blockIdx.x * 64 + lane) * 8;
  f2 a = {p[0], p[1]}, b = {p[2], p[3]}, c = {p[4], p[5]}, d = {p[6], p[7]};
  f2 acc = c;
  float s0 = c.x, s1 = c.y;
  for (int i = 0; i < iters; ++i) {
    // packed: acc = acc * a + b; acc = acc * d; acc = acc + b      (three packed-fp32 instructions per round; the empty asm
    // statements keep hipcc from contracting the mul + add into a second fma)
    acc = __builtin_elementwise_fma(acc, a, b);
    asm volatile("" : "+v"(acc));
    acc = acc * d;
    asm volatile("" : "+v"(acc));
    acc = acc + b;
    asm volatile("" : "+v"(acc));
    // scalar reference of the same arithmetic (fma / mul / add, single rounding each)
    // (an empty asm statement between the mul and the add: hipcc contracts a * b + c - also __fadd_rn(__fmul_rn()) - into an fma,
    // which rounds once where the packed mul + add round twice; the first version of this probe "found" 55 % mismatches that way)
    s0 = __builtin_fmaf(s0, a.x, b.x);
    s1 = __builtin_fmaf(s1, a.y, b.y);
    asm volatile("" : "+v"(s0), "+v"(s1));
    s0 = s0 * d.x;
    s1 = s1 * d.y;
    asm volatile("" : "+v"(s0), "+v"(s1));
    s0 = s0 + b.x;
    s1 = s1 + b.y;
    asm volatile("" : "+v"(s0), "+v"(s1));   // keep the scalar chain scalar (no SLP packing) and in step with the packed one
  }
  const bool bad = __float_as_uint(acc.x) != __float_as_uint(s0) || __float_as_uint(acc.y) != __float_as_uint(s1);
  if (bad) atomicAdd(&lane_hist[lane], 1u);
  if (lane == 0) atomicAdd(checked, 64ull);
}

__global__ void aggressor(const float4* src, float4* dst, size_t n, int reps) {
  for (int r = 0; r < reps; ++r)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
      float4 v = src[i];
      v.x += 1.0f;
      dst[i] = v;
    }
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 400;
  const int nblk = 4800, iters = 64;   // 4800 one-wave workgroups: the grid of fx_bbox_head at 16 x 300 queries
  std::vector<float> h((size_t)nblk * 64 * 8);
  srand(7);
  for (auto& v : h) v = 0.5f + (float)rand() / (float)RAND_MAX * 0.01f;   // |x| stays O(1) over the chain
  float* din; unsigned* dhist; unsigned long long* dchk;
  CK(hipMalloc(&din, h.size() * 4)); CK(hipMalloc(&dhist, 64 * 4)); CK(hipMalloc(&dchk, 8));
  CK(hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  const size_t n4 = (size_t)1 << 25;   // 512 MiB source + destination
  float4 *src, *dst;
  CK(hipMalloc(&src, n4 * 16)); CK(hipMalloc(&dst, n4 * 16)); CK(hipMemset(src, 0, n4 * 16));
  hipStream_t sa, sb;
  CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
  for (int mode = 0; mode < 3; ++mode) {   // 0: victim alone, 1: + streaming copy at full occupancy, 2: + streaming copy with 256-thread blocks at 1 block / CU
    CK(hipMemset(dhist, 0, 256)); CK(hipMemset(dchk, 0, 8));
    if (mode == 1) aggressor<<<4096, 256, 0, sb>>>(src, dst, n4, 40);
    if (mode == 2) aggressor<<<256, 256, 0, sb>>>(src, dst, n4, 12);
    for (int l = 0; l < launches; ++l) victim<<<nblk, 64, 0, sa>>>(din, dhist, dchk, iters);
    CK(hipStreamSynchronize(sa));
    const bool overlapped = mode == 0 || hipStreamQuery(sb) == hipErrorNotReady;   // the aggressor outlived the victims
    CK(hipDeviceSynchronize());
    unsigned hist[64]; unsigned long long chk;
    CK(hipMemcpy(hist, dhist, 256, hipMemcpyDeviceToHost)); CK(hipMemcpy(&chk, dchk, 8, hipMemcpyDeviceToHost));
    unsigned long long bad = 0, bad_hi = 0;
    for (int i = 0; i < 64; ++i) { bad += hist[i]; if (i >= 48) bad_hi += hist[i]; }
    printf("mode %d (%s): %llu lane results checked, %llu mismatches (%llu in lanes 48-63)%s\n", mode,
           mode == 0 ? "victim alone" : (mode == 1 ? "beside a streaming copy, 4096 x 256 threads" : "beside a streaming copy, 256 x 256 threads"), chk, bad, bad_hi,
           overlapped ? "" : "  [aggressor finished early: overlap not guaranteed]");
    if (bad) { printf("  per lane:"); for (int i = 0; i < 64; ++i) printf(" %u", hist[i]); printf("\n"); }
  }
  return 0;
}
