// Does the SHAPE of a wave's 16-byte stores limit a write-bound kernel on MI355X?  Writes a [M][N] bf16 matrix (row stride ld) three ways:
//   0  row-per-lane, the pointwise kernels' epilogue: lane (l32, half) stores 16 B at row l32, columns (piece * 2 + half) * 8 - an instruction
//      covers 32 rows x 32 contiguous bytes, a 128-byte line is completed by 4 consecutive instructions of the same wave;
//   1  line-per-8-lanes: lane l stores 16 B at row l / 8, columns (l % 8) * 8 - an instruction covers 8 rows x 128 contiguous bytes;
//   2  row-per-wave: 64 lanes x 16 B = 1 KiB contiguous of one row.
// Same bytes, same grid; prints GB/s.  usage: store_pattern_probe [M] [N]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE>
__global__ __launch_bounds__(256) void store_kernel(uint4* __restrict__ y, int M, int N, int ld) {   // ld, N in bf16 elements; N % 512 == 0
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l32 = lane & 31, half = lane >> 5;
  const uint4 v = make_uint4(lane, wave, blockIdx.x, 7);
  // workgroup = 128 rows (4 waves x 32 rows), all N columns - like a 128-pixel tile walking its n-tiles
  const int row0 = blockIdx.x * 128 + wave * 32;
  for (int n0 = 0; n0 < N; n0 += 64) {          // 64 channels = 128 bytes per row and step
    if (MODE == 0) {
      const int row = row0 + l32;
#pragma unroll
      for (int pc = 0; pc < 4; ++pc)
        if (row < M) y[((size_t)row * ld + n0 + (pc * 2 + half) * 8) / 8] = v;
    } else if (MODE == 1) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = row0 + it * 8 + (lane >> 3);
        if (row < M) y[((size_t)row * ld + n0 + (lane & 7) * 8) / 8] = v;
      }
    }
  }
  if (MODE == 2) {
    for (int r = 0; r < 32; ++r) {
      const int row = row0 + r;
      for (int n0 = lane * 8; n0 < N; n0 += 512)
        if (row < M) y[((size_t)row * ld + n0) / 8] = v;
    }
  }
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 134400, N = argc > 2 ? atoi(argv[2]) : 1536;
  const int ld = N;
  uint4* y;
  hipMalloc(&y, (size_t)M * ld * 2 * 3);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const dim3 grid((M + 127) / 128), block(256);
  for (int mode = 0; mode < 3; ++mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      uint4* dst = y + (size_t)(rep % 3) * M * ld / 8;    // rotate over three buffers
      hipEventRecord(e0);
      if (mode == 0) store_kernel<0><<<grid, block>>>(dst, M, N, ld);
      if (mode == 1) store_kernel<1><<<grid, block>>>(dst, M, N, ld);
      if (mode == 2) store_kernel<2><<<grid, block>>>(dst, M, N, ld);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best) best = ms;
    }
    printf("mode %d: %.1f us  %.0f GB/s  (M=%d N=%d, %.0f MB)\n", mode, best * 1e3, (double)M * N * 2 / best / 1e6, M, N, (double)M * N * 2 / 1e6);
  }
  return 0;
}
