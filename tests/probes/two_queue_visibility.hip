// Minimal reproducer for the cross-queue visibility hazard of DESIGN.md §5 (two batch parts on two HIP streams: small kernels of
// one stream read stale outputs of their own predecessor while a large kernel runs on the other stream).
//   stream A:  for k = 1..N:  write_k(X)  ->  copy(X -> Y)  ->  verify_k(Y, errors)      (each kernel depends on the previous one)
//   stream B:  a long streaming kernel, relaunched back to back
// Prints the number of wrong elements seen by verify for several variants.  Build: hipcc --offload-arch=gfx950 -O3 -o two_queue_visibility two_queue_visibility.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned f(unsigned i, unsigned k) { return i * 2654435761u + k * 40503u; }

template <int FENCE>
__global__ void write_k(unsigned* x, unsigned n, unsigned k) {
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) x[i] = f(i, k);
  if (FENCE) __threadfence();
}
// mode 0: plain loads; 1: agent-scope acquire fence first; 2: relaxed agent-scope atomic loads (sc1: bypass L1)
template <int MODE>
__global__ void copy_k(const unsigned* x, unsigned* y, unsigned n) {
  if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    // read a PERMUTED element so that a block reads what other blocks (other CUs / XCDs) wrote
    unsigned j = (i * 97u + 13u) % n;
    unsigned v = MODE == 2 ? __hip_atomic_load(x + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : x[j];
    y[i] = v;
  }
}
__global__ void verify_k(const unsigned* y, unsigned n, unsigned k, unsigned* errors) {
  unsigned bad = 0;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    unsigned j = (i * 97u + 13u) % n;
    if (y[i] != f(j, k)) ++bad;
  }
  if (bad) atomicAdd(errors, bad);
}
__global__ void heavy(const float4* a, float4* b, size_t n, int reps) {
  for (int r = 0; r < reps; ++r)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
      float4 v = a[i];
      v.x += 1.0f;
      b[i] = v;
    }
}

// LDS variants: the aggressor owns a large dynamic LDS allocation and rewrites it all the time; the victim stages its copy through LDS
__global__ void heavy_lds(float* sink, int reps, int lds_words) {
  extern __shared__ unsigned lds_big[];
  for (int r = 0; r < reps; ++r) {
    for (int i = threadIdx.x; i < lds_words; i += blockDim.x) lds_big[i] = 0xDEAD0000u + (unsigned)r + i;
    __syncthreads();
    unsigned acc = 0;
    for (int i = threadIdx.x; i < lds_words; i += blockDim.x) acc += lds_big[i];
    if (acc == 0x12345678u) sink[blockIdx.x] = (float)acc;
    __syncthreads();
  }
}
__global__ void copy_lds(const unsigned* x, unsigned* y, unsigned n, int lds_words) {
  extern __shared__ unsigned lds_small[];
  for (unsigned base = blockIdx.x * (unsigned)lds_words; base < n; base += gridDim.x * (unsigned)lds_words) {
    for (int i = threadIdx.x; i < lds_words && base + i < n; i += blockDim.x) lds_small[i] = x[((base + i) * 97u + 13u) % n];
    __syncthreads();
    for (int k = 0; k < 64; ++k) __builtin_amdgcn_s_sleep(8);   // keep the tile resident in LDS for a while
    __syncthreads();
    for (int i = threadIdx.x; i < lds_words && base + i < n; i += blockDim.x) y[base + i] = lds_small[lds_words - 1 - ((lds_words - 1 - i))];
    __syncthreads();
  }
}
// wave-level butterfly reduction (ds_bpermute / DPP as hipcc chooses) of known values beside an LDS-heavy kernel of another queue
__global__ void shfl_reduce_check(unsigned* errors, unsigned seed, int rounds) {
  const unsigned lane = threadIdx.x & 63, gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  unsigned bad = 0;
  for (int r = 0; r < rounds; ++r) {
    float v[4];
    float expect[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < 4; ++j) {
      v[j] = (float)((lane * 7u + j * 13u + gw + seed + r) & 255u);       // small integers: every partial sum is exact in fp32
      for (unsigned l = 0; l < 64; ++l) expect[j] += (float)((l * 7u + j * 13u + gw + seed + r) & 255u);
    }
    for (int j = 0; j < 4; ++j)
      for (int o = 32; o > 0; o >>= 1) v[j] += __shfl_xor(v[j], o, 64);
    for (int j = 0; j < 4; ++j) bad += (v[j] != expect[j]);
  }
  if (bad) atomicAdd(errors, bad);
}
static unsigned run_shfl(int iters, int big_kb, const char* label) {
  unsigned* err; float* sink;
  CK(hipMalloc(&err, 4)); CK(hipMalloc(&sink, 4096 * 4));
  CK(hipMemset(err, 0, 4));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(heavy_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  CK(hipDeviceSynchronize());
  for (int k = 1; k <= iters; ++k) {
    if (big_kb && (k % 4) == 1) hipLaunchKernelGGL(heavy_lds, dim3(1024), dim3(256), big_kb * 1024, sb, sink, 40, big_kb * 256);
    hipLaunchKernelGGL(shfl_reduce_check, dim3(1200), dim3(256), 0, sa, err, (unsigned)k, 8);
  }
  CK(hipDeviceSynchronize());
  unsigned h = 0;
  CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
  printf("%-58s big=%d KB iters=%d  wrong sums: %u\n", label, big_kb, iters, h);
  CK(hipFree(err)); CK(hipFree(sink)); CK(hipStreamDestroy(sa)); CK(hipStreamDestroy(sb));
  return h;
}

static unsigned run_lds(unsigned n, int iters, int big_kb, int small_kb, const char* label) {
  unsigned *x, *y, *err; float* sink;
  CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&sink, 4096 * 4));
  CK(hipMemset(err, 0, 4));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(heavy_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(copy_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  CK(hipDeviceSynchronize());
  for (int k = 1; k <= iters; ++k) {
    if (big_kb && (k % 4) == 1) hipLaunchKernelGGL(heavy_lds, dim3(1024), dim3(256), big_kb * 1024, sb, sink, 40, big_kb * 256);
    hipLaunchKernelGGL((write_k<0>), dim3(300), dim3(256), 0, sa, x, n, (unsigned)k);
    hipLaunchKernelGGL(copy_lds, dim3(300), dim3(256), small_kb * 1024, sa, x, y, n, small_kb * 256);
    hipLaunchKernelGGL(verify_k, dim3(300), dim3(256), 0, sa, y, n, (unsigned)k, err);
  }
  CK(hipDeviceSynchronize());
  unsigned h = 0;
  CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
  printf("%-58s big=%d KB small=%d KB iters=%d  wrong elements: %u\n", label, big_kb, small_kb, iters, h);
  CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(err)); CK(hipFree(sink));
  CK(hipStreamDestroy(sa)); CK(hipStreamDestroy(sb));
  return h;
}

template <int WF, int MODE>
static unsigned run(unsigned n, int iters, bool concurrent, int small_grid, const char* label) {
  unsigned *x, *y, *err;
  CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4)); CK(hipMalloc(&err, 4));
  CK(hipMemset(err, 0, 4)); CK(hipMemset(x, 0, n * 4));
  float4 *ha, *hb;
  const size_t hn = (size_t)64 << 20;  // 1 GiB each
  CK(hipMalloc(&ha, hn * 16)); CK(hipMalloc(&hb, hn * 16));
  CK(hipMemset(ha, 0, hn * 16));
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  CK(hipDeviceSynchronize());
  for (int k = 1; k <= iters; ++k) {
    if (concurrent && (k % 8) == 1) hipLaunchKernelGGL(heavy, dim3(2048), dim3(256), 0, sb, ha, hb, hn, 2);
    hipLaunchKernelGGL((write_k<WF>), dim3(small_grid), dim3(256), 0, sa, x, n, (unsigned)k);
    hipLaunchKernelGGL((copy_k<MODE>), dim3(small_grid), dim3(256), 0, sa, x, y, n);
    hipLaunchKernelGGL(verify_k, dim3(small_grid), dim3(256), 0, sa, y, n, (unsigned)k, err);
  }
  CK(hipDeviceSynchronize());
  unsigned h = 0;
  CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
  printf("%-58s n=%u grid=%d iters=%d concurrent=%d  wrong elements: %u\n", label, n, small_grid, iters, (int)concurrent, h);
  CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(err)); CK(hipFree(ha)); CK(hipFree(hb));
  CK(hipStreamDestroy(sa)); CK(hipStreamDestroy(sb));
  return h;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 400;
  unsigned total = 0;
  for (unsigned n : {1u << 20, 1u << 16}) {          // 4 MiB (decoder-activation sized) and 256 KiB
    for (int grid : {300, 40}) {
      run<0, 0>(n, iters, false, grid, "single stream, plain loads");
      total += run<0, 0>(n, iters, true, grid, "two streams, plain loads");
      run<0, 1>(n, iters, true, grid, "two streams, consumer acquire fence (buffer_inv sc1)");
      run<0, 2>(n, iters, true, grid, "two streams, consumer sc1 loads (bypass L1)");
      run<1, 0>(n, iters, true, grid, "two streams, producer __threadfence at exit");
    }
  }
  unsigned lds_total = 0;
  run_lds(1u << 20, iters, 0, 16, "LDS-staged copy alone");
  for (int big : {34, 48, 64, 76, 96, 128})
    for (int small : {8, 32, 64, 80}) {
      if (big + small > 160) continue;
      lds_total += run_lds(1u << 20, iters, big, small, "LDS-staged copy beside a large-LDS kernel");
    }
  unsigned shfl_total = 0;
  run_shfl(iters, 0, "wave butterfly reduction alone");
  for (int big : {34, 40, 64, 80, 128, 160}) shfl_total += run_shfl(iters, big, "wave butterfly reduction beside an LDS-heavy kernel");
  printf("RESULT two-stream plain-load stale elements: %u; LDS-staged corrupted elements: %u; wrong wave reductions: %u\n", total, lds_total, shfl_total);
  return 0;
}
