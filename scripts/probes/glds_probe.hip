// Probe: semantics of buffer_load_dwordx4 ... lds on gfx950 (LDS destination layout, out-of-range behaviour).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(const unsigned* x, unsigned nbytes, const unsigned* offs, uint4* out) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[8192];
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) ((unsigned*)lds)[i] = 0xABABABABu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, nbytes, 0x00020000);
  unsigned off = offs[threadIdx.x];
  int wave = threadIdx.x >> 6;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + wave * 1024), 16, off, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  out[threadIdx.x] = *(uint4*)(lds + threadIdx.x * 16);
}
int main() {
  const int N = 4096;
  std::vector<unsigned> h(N);
  for (int i = 0; i < N; ++i) h[i] = i;
  unsigned *dx, *doffs; uint4* dout;
  hipMalloc(&dx, N * 4); hipMalloc(&doffs, 256 * 4); hipMalloc(&dout, 256 * 16);
  hipMemcpy(dx, h.data(), N * 4, hipMemcpyHostToDevice);
  std::vector<unsigned> offs(256);
  for (int i = 0; i < 256; ++i) offs[i] = (i % 5 == 3) ? 0xFFFFFFF0u : (unsigned)(((i * 7) % 1000) * 16);
  offs[10] = N * 4 - 8;  // straddles the end
  hipMemcpy(doffs, offs.data(), 256 * 4, hipMemcpyHostToDevice);
  k<<<1, 256>>>(dx, N * 4, doffs, dout);
  std::vector<uint4> o(256);
  hipMemcpy(o.data(), dout, 256 * 16, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; ++i) {
    unsigned e0 = (offs[i] == 0xFFFFFFF0u) ? 0 : offs[i] / 4;
    bool oob = offs[i] == 0xFFFFFFF0u;
    if (i < 12 || (oob && i < 40)) printf("lane %3d off %08x -> %08x %08x %08x %08x\n", i, offs[i], o[i].x, o[i].y, o[i].z, o[i].w);
    if (i == 10) continue;
    if (oob ? (o[i].x | o[i].y | o[i].z | o[i].w) != 0 : (o[i].x != e0 || o[i].w != e0 + 3)) ++bad;
  }
  printf("mismatches: %d (0 means: LDS dest = wave base + lane*16, OOB lanes write zeros)\n", bad);
  return 0;
}
