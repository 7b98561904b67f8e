// Training-path support kernels around the convolution: weight (re)packing from the fp32 master
// copy, gradient unpacking, activation / pooling backward, the zero-insertion that turns a stride-2 input gradient into a
// stride-1 convolution.  All are HBM-bound elementwise kernels, 8 bf16 (16 B) per lane.
#include "common.h"

int fx_tune(const char* env_name, int default_value);  // conv_igemm.hip: integer tuning knob from the environment

// ------------------------------------------------------------------------------------------------
// Master weights (reference layout [N][C][KH][KW], fp32) -> the two bf16 images the MFMA kernels read:
//   w_fwd [Npad][KH][KW][C]            : forward / weight layout of fx_conv2d_nhwc_bf16, optional per-out-channel scale
//                                        (eval-mode BatchNorm folded: gamma / sqrt(var + eps))
//   w_dgrad [Cpad][KH][KW][N]          : w_dgrad[c][kh][kw][n] = scale[n] * w[n][c][KH-1-kh][KW-1-kw]  (the input gradient
//                                        of a stride-1 "same" conv is the conv of dZ with the flipped, transposed filter)
// Rows beyond N (C) of the padded images are zeroed by the caller once.
__global__ __launch_bounds__(256) void pack_conv_weights_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                                                bf16_t* __restrict__ w_fwd, bf16_t* __restrict__ w_dgrad, int N, int C, int KH,
                                                                int KW) {
  const int64_t total = (int64_t)N * C * KH * KW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int kw = (int)(i % KW);
    int64_t r = i / KW;
    int kh = (int)(r % KH);
    r /= KH;
    int c = (int)(r % C);
    int n = (int)(r / C);
    const float v = w[i] * (scale ? scale[n] : 1.0f);
    const bf16_t b = f32_to_bf16(v);
    if (w_fwd) w_fwd[(((int64_t)n * KH + kh) * KW + kw) * C + c] = b;
    if (w_dgrad) w_dgrad[(((int64_t)c * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)) * N + n] = b;
  }
}

// [rows][K] bf16 weight image -> MFMA fragment order [rows/32][K/16][lane][8] (fx_conv_desc.w_frag): lane l of fragment (nb, ks) holds
// w[nb*32 + l%32][ks*16 + (l/32)*8 .. +8] - a permutation of 16-byte chunks, coalesced on the store side
__global__ __launch_bounds__(256) void frag_from_rows_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ frag, int rows, int K) {
  const int K16 = K / 16;
  const int64_t total = (int64_t)rows * K / 8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int l = (int)(i & 63);
    const int64_t f = i >> 6;
    const int ks = (int)(f % K16);
    const int nb = (int)(f / K16);
    *reinterpret_cast<uint4*>(frag + i * 8) =
        *reinterpret_cast<const uint4*>(w + ((int64_t)nb * 32 + (l & 31)) * K + ks * 16 + (l >> 5) * 8);
  }
}

extern "C" int fx_pack_frag_bf16(const void* w_rows, void* frag, int rows, int K, fx_stream_t stream_) {
  FX_CHECK_ARG(w_rows && frag && rows > 0 && K > 0 && rows % 32 == 0 && K % 16 == 0);
  FX_CHECK_ARG(((uintptr_t)w_rows % 16) == 0 && ((uintptr_t)frag % 16) == 0);
  int64_t grid = ((int64_t)rows * K / 8 + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(frag_from_rows_kernel, dim3((int)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), (const bf16_t*)w_rows,
                     (bf16_t*)frag, rows, K);
  return fx_launch_status();
}

extern "C" int fx_pack_conv_weights_f32(const float* w, const float* scale, void* w_fwd, void* w_dgrad, void* w_fwd_frag, void* w_dgrad_frag,
                                        int N, int C, int KH, int KW, fx_stream_t stream_) {
  FX_CHECK_ARG(w && (w_fwd || w_dgrad) && N > 0 && C > 0 && KH > 0 && KW > 0);
  FX_CHECK_ARG(!w_fwd_frag || (w_fwd && N % 32 == 0 && (KH * KW * C) % 16 == 0 && ((uintptr_t)w_fwd_frag % 16) == 0 && ((uintptr_t)w_fwd % 16) == 0));
  FX_CHECK_ARG(!w_dgrad_frag || (w_dgrad && C % 32 == 0 && (KH * KW * N) % 16 == 0 && ((uintptr_t)w_dgrad_frag % 16) == 0 && ((uintptr_t)w_dgrad % 16) == 0));
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  int64_t total = (int64_t)N * C * KH * KW;
  int64_t grid = (total + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(pack_conv_weights_kernel, dim3((int)grid), dim3(256), 0, stream, w, scale, (bf16_t*)w_fwd, (bf16_t*)w_dgrad, N, C, KH, KW);
  grid = (total / 8 + 255) / 256;
  if (grid > 4096) grid = 4096;
  if (w_fwd_frag)
    hipLaunchKernelGGL(frag_from_rows_kernel, dim3((int)grid), dim3(256), 0, stream, (const bf16_t*)w_fwd, (bf16_t*)w_fwd_frag, N, KH * KW * C);
  if (w_dgrad_frag)
    hipLaunchKernelGGL(frag_from_rows_kernel, dim3((int)grid), dim3(256), 0, stream, (const bf16_t*)w_dgrad, (bf16_t*)w_dgrad_frag, C, KH * KW * N);
  return fx_launch_status();
}

// nn.Linear master weights fp32 [N][K] (+ bias) -> the bf16 images of the forward GEMM (w_fwd [.][Kp], row n) and of the input-gradient
// GEMM (w_t [.][Np], row k), dimensions padded to the kernels' granules (padding pre-zeroed by the caller, never touched), and the
// bias into its padded fp32 vector: one launch per layer and step instead of pad + transpose + copy launches.
__global__ __launch_bounds__(256) void pack_linear_weights_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                                                  bf16_t* __restrict__ w_fwd, bf16_t* __restrict__ w_t, float* __restrict__ bias_out,
                                                                  int N, int K, int Np, int Kp) {
  const int64_t total = (int64_t)N * K;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int k = (int)(i % K), n = (int)(i / K);
    const bf16_t b = f32_to_bf16(w[i]);
    w_fwd[(int64_t)n * Kp + k] = b;
    w_t[(int64_t)k * Np + n] = b;
    if (k == 0 && bias_out) bias_out[n] = bias[n];
  }
}

extern "C" int fx_pack_linear_weights_f32(const float* w, const float* bias, void* w_fwd, void* w_t, float* bias_out, int N, int K, int Np, int Kp,
                                          fx_stream_t stream_) {
  FX_CHECK_ARG(w && w_fwd && w_t && N > 0 && K > 0 && Np >= N && Kp >= K && (!bias_out || bias));
  int64_t grid = ((int64_t)N * K + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(pack_linear_weights_kernel, dim3((int)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), w, bias, (bf16_t*)w_fwd,
                     (bf16_t*)w_t, bias_out, N, K, Np, Kp);
  return fx_launch_status();
}

// ------------------------------------------------------------------------------------------------
// All weight images of a model in ONE launch (multi-tensor form of fx_pack_conv_weights_f32 / fx_pack_linear_weights_f32): after every
// optimizer step each of the ~190 layers of RT-DETR needs its bf16 images rebuilt from the fp32 masters - ~340 launches of a few
// microseconds each, bound by launch latency on the GPU and by the Python launch path on the host.  A table entry describes one master
// tensor; a workgroup finds its entry by binary search over the entries' first workgroup index and converts a tile of 8 output channels
// x up to PK_TILE (input channel, tap) elements: the masters are read coalesced into LDS, and all four images - forward, flipped-transposed
// and their fragment-order copies - are written in 16-byte pieces (8 consecutive input channels, or the tile's 8 output channels).
// (Element-wise scattered 2-byte stores made this launch 0.78 ms for RT-DETR's 42 M parameters; the data is 0.5 GB.)
#define PK_TILE 2304   // (input channels x taps) per tile: 256 x 9, or up to 2304 channels of a pointwise layer; 8 rows x 2 B = 36 KiB of LDS

__device__ __forceinline__ int64_t frag_offset(int row, int k, int K) {
  return ((((int64_t)(row >> 5) * (K >> 4) + (k >> 4)) * 64 + (row & 31) + 32 * ((k >> 3) & 1)) << 3) + (k & 7);
}

static inline int pk_chunk_c(int C, int T) {   // input channels per tile: a multiple of 8 (or all of C)
  int cc = (PK_TILE / T) & ~7;
  if (cc < 8) cc = 8;
  return C < cc ? C : cc;
}

extern "C" int fx_pack_entry_blocks(int N, int C, int KH, int KW) {
  if (N <= 0 || C <= 0 || KH <= 0 || KW <= 0 || KH * KW > PK_TILE / 8) return -1;
  const int cc = pk_chunk_c(C, KH * KW);
  return ((N + 7) / 8) * ((C + cc - 1) / cc);
}

__global__ __launch_bounds__(256) void pack_weights_many_kernel(const fx_pack_entry* __restrict__ tab, int n_entries) {
  __shared__ int s_e;
  __shared__ __attribute__((aligned(16))) bf16_t tile[8][PK_TILE];
  if (threadIdx.x == 0) {
    int lo = 0, hi = n_entries - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (tab[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    s_e = lo;
  }
  __syncthreads();
  const fx_pack_entry e = tab[s_e];
  const int N = e.N, C = e.C, KH = e.KH, KW = e.KW, T = KH * KW;
  int CC = (PK_TILE / T) & ~7;
  if (CC < 8) CC = 8;
  if (C < CC) CC = C;
  const int ncc = (C + CC - 1) / CC;
  const int blk = (int)blockIdx.x - e.first_block;
  const int n0 = (blk / ncc) * 8, c0 = (blk % ncc) * CC;
  const int nn = min(8, N - n0), cw = min(CC, C - c0);
  const int ng0 = e.n_offset + n0;   // first output channel of the tile within the (possibly shared) images
  // ---- masters -> LDS (bf16, folded-BN scale applied): row r = output channel n0 + r, column cl * T + tap
  for (int r = 0; r < nn; ++r) {
    const float sc = e.scale ? e.scale[n0 + r] : 1.0f;
    const float* __restrict__ src = e.w + ((int64_t)(n0 + r) * C + c0) * T;
    for (int i = threadIdx.x; i < cw * T; i += 256) tile[r][i] = f32_to_bf16(src[i] * sc);
  }
  if (e.bias_out && c0 == 0 && (int)threadIdx.x < nn) e.bias_out[ng0 + threadIdx.x] = e.bias[n0 + threadIdx.x];
  __syncthreads();
  bf16_t* __restrict__ w_fwd = (bf16_t*)e.w_fwd;
  bf16_t* __restrict__ w_dgrad = (bf16_t*)e.w_dgrad;
  bf16_t* __restrict__ f_fwd = (bf16_t*)e.w_fwd_frag;
  bf16_t* __restrict__ f_dgrad = (bf16_t*)e.w_dgrad_frag;
  const int Kf = T * C, Kd = T * e.n_total;
  // ---- row-major along the input channel: forward image and its fragment copy, 8 consecutive channels per store
  const bool c_vec = (C % 8 == 0) && (e.ld_fwd % 8 == 0);   // (c0 and cw are multiples of 8 then)
  if (c_vec) {
    const int c8n = cw / 8;
    for (int i = threadIdx.x; i < nn * T * c8n; i += 256) {
      const int c8 = i % c8n, tap = (i / c8n) % T, r = i / (c8n * T);
      bf16_t v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = tile[r][(c8 * 8 + j) * T + tap];
      const int kf = tap * C + c0 + c8 * 8;
      if (w_fwd) *reinterpret_cast<uint4*>(w_fwd + (int64_t)(ng0 + r) * e.ld_fwd + kf) = *reinterpret_cast<const uint4*>(v);
      if (f_fwd) *reinterpret_cast<uint4*>(f_fwd + frag_offset(ng0 + r, kf, Kf)) = *reinterpret_cast<const uint4*>(v);
    }
  } else {
    for (int i = threadIdx.x; i < nn * T * cw; i += 256) {
      const int cl = i % cw, tap = (i / cw) % T, r = i / (cw * T);
      const int kf = tap * C + c0 + cl;
      const bf16_t b = tile[r][cl * T + tap];
      if (w_fwd) w_fwd[(int64_t)(ng0 + r) * e.ld_fwd + kf] = b;
      if (f_fwd) f_fwd[frag_offset(ng0 + r, kf, Kf)] = b;
    }
  }
  // ---- row-major along the output channel: flipped-transposed image and its fragment copy, the tile's 8 output channels per store
  const bool n_vec = nn == 8 && (ng0 % 8 == 0) && (e.n_total % 8 == 0) && (e.ld_dgrad % 8 == 0);
  for (int i = threadIdx.x; i < cw * T; i += 256) {
    const int cl = i / T, tap = i - cl * T;      // i is also the LDS column
    const int kh = tap / KW, kw = tap - kh * KW;
    const int kd0 = ((KH - 1 - kh) * KW + (KW - 1 - kw)) * e.n_total + ng0;
    const int c = c0 + cl;
    if (n_vec) {
      bf16_t v[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = tile[r][i];
      if (w_dgrad) *reinterpret_cast<uint4*>(w_dgrad + (int64_t)c * e.ld_dgrad + kd0) = *reinterpret_cast<const uint4*>(v);
      if (f_dgrad) *reinterpret_cast<uint4*>(f_dgrad + frag_offset(c, kd0, Kd)) = *reinterpret_cast<const uint4*>(v);
    } else {
      for (int r = 0; r < nn; ++r) {
        if (w_dgrad) w_dgrad[(int64_t)c * e.ld_dgrad + kd0 + r] = tile[r][i];
        if (f_dgrad) f_dgrad[frag_offset(c, kd0 + r, Kd)] = tile[r][i];
      }
    }
  }
}

extern "C" int fx_pack_weights_many_f32(const fx_pack_entry* entries_dev, int n_entries, int total_blocks, fx_stream_t stream_) {
  FX_CHECK_ARG(entries_dev && n_entries > 0 && total_blocks > 0);
  hipLaunchKernelGGL(pack_weights_many_kernel, dim3(total_blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), entries_dev, n_entries);
  return fx_launch_status();
}

// dw_master[n][c][kh][kw] (+)= scale[n] * sum_s dw_eff[s][n][kh][kw][c]   (chain rule through the folded BatchNorm scale; the sum runs
// over the pixel-range partials of fx_conv2d_wgrad_partial_nhwc_bf16 - one slab when the gradient was accumulated with atomics)
__global__ __launch_bounds__(256) void unpack_conv_wgrad_kernel(const float* __restrict__ dw_eff, int64_t split_stride, int splits,
                                                                const float* __restrict__ scale, float* __restrict__ dw, int N, int C, int KH,
                                                                int KW, int Ceff, int accumulate) {
  const int64_t total = (int64_t)N * C * KH * KW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int kw = (int)(i % KW);
    int64_t r = i / KW;
    int kh = (int)(r % KH);
    r /= KH;
    int c = (int)(r % C);
    int n = (int)(r / C);
    const int64_t src = (((int64_t)n * KH + kh) * KW + kw) * Ceff + c;
    float v = 0.0f;
    for (int s = 0; s < splits; ++s) v += dw_eff[(int64_t)s * split_stride + src];
    v *= scale ? scale[n] : 1.0f;
    dw[i] = accumulate ? dw[i] + v : v;
  }
}

// slab 0 += slabs 1..S-1 (coalesced float4 streams; the slabs are read at the HBM rate, which is what makes per-pixel-range
// partial stores cheaper than fp32 atomics)
__global__ __launch_bounds__(256) void slab_sum_kernel(float* __restrict__ part, int64_t split_stride, int splits, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 a = reinterpret_cast<const float4*>(part)[i];
    int s = 1;
    for (; s + 3 < splits; s += 4) {
      const float4 b0 = reinterpret_cast<const float4*>(part + (int64_t)s * split_stride)[i];
      const float4 b1 = reinterpret_cast<const float4*>(part + (int64_t)(s + 1) * split_stride)[i];
      const float4 b2 = reinterpret_cast<const float4*>(part + (int64_t)(s + 2) * split_stride)[i];
      const float4 b3 = reinterpret_cast<const float4*>(part + (int64_t)(s + 3) * split_stride)[i];
      a.x += (b0.x + b1.x) + (b2.x + b3.x); a.y += (b0.y + b1.y) + (b2.y + b3.y);
      a.z += (b0.z + b1.z) + (b2.z + b3.z); a.w += (b0.w + b1.w) + (b2.w + b3.w);
    }
    for (; s < splits; ++s) {
      const float4 b = reinterpret_cast<const float4*>(part + (int64_t)s * split_stride)[i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    reinterpret_cast<float4*>(part)[i] = a;
  }
}

// The two passes above in one, for partial slabs: walks the SOURCE order [n][kh][kw][c] with float4 loads (every slab read once, coalesced),
// sums the pixel-range partials in registers and scatters the four results to dw[n][c][kh][kw] (4-byte stores KH*KW*4 bytes apart - one
// slab's worth of writes, merged in L2) - instead of a slab-sum pass that writes the sum back and a second pass that re-reads it.
__global__ __launch_bounds__(256) void unpack_sum_kernel(const float* __restrict__ part, int64_t split_stride, int splits, const float* __restrict__ scale,
                                                         float* __restrict__ dw, int N, int C, int KH, int KW, int Ceff, int accumulate) {
  const int c4n = Ceff / 4, taps = KH * KW;
  const int n4 = N * taps * c4n;   // < 2^31 (checked by the launcher): 32-bit index arithmetic (three 64-bit divisions per element cost more than the loads)
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
    const int r = i / c4n;
    const int c0 = (i - r * c4n) * 4;
    const int n = r / taps, tap = r - n * taps;
    float4 a = reinterpret_cast<const float4*>(part)[i];
    int s = 1;
    for (; s + 3 < splits; s += 4) {
      const float4 b0 = reinterpret_cast<const float4*>(part + (int64_t)s * split_stride)[i];
      const float4 b1 = reinterpret_cast<const float4*>(part + (int64_t)(s + 1) * split_stride)[i];
      const float4 b2 = reinterpret_cast<const float4*>(part + (int64_t)(s + 2) * split_stride)[i];
      const float4 b3 = reinterpret_cast<const float4*>(part + (int64_t)(s + 3) * split_stride)[i];
      a.x += (b0.x + b1.x) + (b2.x + b3.x); a.y += (b0.y + b1.y) + (b2.y + b3.y);
      a.z += (b0.z + b1.z) + (b2.z + b3.z); a.w += (b0.w + b1.w) + (b2.w + b3.w);
    }
    for (; s < splits; ++s) {
      const float4 b = reinterpret_cast<const float4*>(part + (int64_t)s * split_stride)[i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    const float sc = scale ? scale[n] : 1.0f;
    const float v[4] = {a.x * sc, a.y * sc, a.z * sc, a.w * sc};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (c0 + k < C) {
        float* dst = dw + ((int64_t)n * C + c0 + k) * taps + tap;
        *dst = accumulate ? *dst + v[k] : v[k];
      }
    }
  }
}

static int unpack_launch(const float* dw_eff, int64_t split_stride, int splits, const float* scale, float* dw_master, int N, int C, int KH, int KW,
                         int C_eff, int accumulate, fx_stream_t stream_) {
  FX_CHECK_ARG(dw_eff && dw_master && N > 0 && C > 0 && KH > 0 && KW > 0 && C_eff >= C && splits >= 1);
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const int64_t slab = (int64_t)N * KH * KW * C_eff;
  static const int fused = fx_tune("FX_UNPACK_FUSED", 1);
  if (splits > 1 && C_eff % 4 == 0 && split_stride % 4 == 0 && ((uintptr_t)dw_eff % 16) == 0 && slab < (1ll << 31)) {
    int64_t grid = (slab / 4 + 255) / 256;
    if (grid > 8192) grid = 8192;
    if (fused) {
      hipLaunchKernelGGL(unpack_sum_kernel, dim3((int)grid), dim3(256), 0, stream, dw_eff, split_stride, splits, scale, dw_master, N, C, KH, KW, C_eff,
                         accumulate);
      return fx_launch_status();
    }
    hipLaunchKernelGGL(slab_sum_kernel, dim3((int)grid), dim3(256), 0, stream, const_cast<float*>(dw_eff), split_stride, splits, slab / 4);
    splits = 1;
  }
  int64_t total = (int64_t)N * C * KH * KW;
  int64_t grid = (total + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(unpack_conv_wgrad_kernel, dim3((int)grid), dim3(256), 0, stream, dw_eff, split_stride, splits, scale, dw_master, N, C, KH, KW,
                     C_eff, accumulate);
  return fx_launch_status();
}

extern "C" int fx_unpack_conv_wgrad_f32(const float* dw_eff, const float* scale, float* dw_master, int N, int C, int KH, int KW, int C_eff,
                                        int accumulate, fx_stream_t stream_) {
  return unpack_launch(dw_eff, 0, 1, scale, dw_master, N, C, KH, KW, C_eff, accumulate, stream_);
}

extern "C" int fx_unpack_conv_wgrad_sum_f32(const float* partials, int64_t split_stride, int splits, const float* scale, float* dw_master, int N,
                                            int C, int KH, int KW, int C_eff, int accumulate, fx_stream_t stream_) {
  FX_CHECK_ARG(split_stride >= (int64_t)N * KH * KW * C_eff);
  return unpack_launch(partials, split_stride, splits, scale, dw_master, N, C, KH, KW, C_eff, accumulate, stream_);
}

// ------------------------------------------------------------------------------------------------
// Backward of the fused conv epilogue  y = act(conv + bias [+ residual]) :  dz = dy * act'(.)  (dz is both the gradient of
// the conv output and, for a pre-activation residual, of the residual branch).  ReLU needs only the saved OUTPUT
// (y > 0); an optional second upstream gradient dy2 (the tensor had two consumers) is added first.
__global__ __launch_bounds__(256) void relu_bwd_kernel(const bf16_t* __restrict__ dy, int lddy, const bf16_t* __restrict__ dy2, int lddy2,
                                                       const bf16_t* __restrict__ y, int ldy, bf16_t* __restrict__ dz, int lddz, int64_t rows,
                                                       int C8, int use_relu) {
  const int64_t total = rows * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    const int64_t r = i / C8;
    float g[8], o[8];
    unpack_bf16x8(*reinterpret_cast<const uint4*>(dy + r * lddy + c8 * 8), g);
    if (dy2) {
      float g2[8];
      unpack_bf16x8(*reinterpret_cast<const uint4*>(dy2 + r * lddy2 + c8 * 8), g2);
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] += g2[j];
    }
    if (use_relu) {
      unpack_bf16x8(*reinterpret_cast<const uint4*>(y + r * ldy + c8 * 8), o);
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] = o[j] > 0.0f ? g[j] : 0.0f;
    }
    *reinterpret_cast<uint4*>(dz + r * lddz + c8 * 8) = pack_bf16x8(g);
  }
}

extern "C" int fx_relu_bwd_bf16(const void* dy, int lddy, const void* dy2, int lddy2, const void* y, int ldy, void* dz, int lddz, int64_t rows,
                                int cols, int use_relu, fx_stream_t stream_) {
  FX_CHECK_ARG(dy && dz && rows > 0 && cols > 0 && cols % 8 == 0 && lddy >= cols && lddz >= cols && lddy % 8 == 0 && lddz % 8 == 0);
  FX_CHECK_ARG(!use_relu || (y && ldy >= cols && ldy % 8 == 0));
  FX_CHECK_ARG(!dy2 || (lddy2 >= cols && lddy2 % 8 == 0));
  int64_t total = rows * (cols / 8);
  int64_t grid = (total + 255) / 256;
  if (grid > 256 * 32) grid = 256 * 32;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3((int)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), (const bf16_t*)dy, lddy,
                     (const bf16_t*)dy2, lddy2, (const bf16_t*)y, ldy, (bf16_t*)dz, lddz, rows, cols / 8, use_relu);
  return fx_launch_status();
}

// ------------------------------------------------------------------------------------------------
// u[b, 2*ho, 2*wo, :] = dz[b, ho, wo, :], zero elsewhere (u is [B, H, W, C] with H >= 2*Ho-1): the input gradient of a
// stride-2 3x3 pad-1 conv is the stride-1 conv of u with the flipped filter.
__global__ __launch_bounds__(256) void zero_insert2_kernel(const bf16_t* __restrict__ dz, int lddz, bf16_t* __restrict__ u, int ldu, int B,
                                                           int Ho, int Wo, int H, int W, int C8) {
  const int64_t total = (int64_t)B * H * W * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    int64_t p = i / C8;
    const int x = (int)(p % W);
    p /= W;
    const int y = (int)(p % H);
    const int b = (int)(p / H);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (!(y & 1) && !(x & 1) && (y >> 1) < Ho && (x >> 1) < Wo)
      v = *reinterpret_cast<const uint4*>(dz + (((int64_t)b * Ho + (y >> 1)) * Wo + (x >> 1)) * lddz + c8 * 8);
    *reinterpret_cast<uint4*>(u + (((int64_t)b * H + y) * W + x) * ldu + c8 * 8) = v;
  }
}

extern "C" int fx_zero_insert2_nhwc_bf16(const void* dz, int lddz, void* u, int ldu, int B, int Ho, int Wo, int H, int W, int C,
                                         fx_stream_t stream_) {
  FX_CHECK_ARG(dz && u && B > 0 && Ho > 0 && Wo > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && lddz >= C && ldu >= C);
  int64_t total = (int64_t)B * H * W * (C / 8);
  int64_t grid = (total + 255) / 256;
  if (grid > 256 * 32) grid = 256 * 32;
  hipLaunchKernelGGL(zero_insert2_kernel, dim3((int)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), (const bf16_t*)dz, lddz,
                     (bf16_t*)u, ldu, B, Ho, Wo, H, W, C / 8);
  return fx_launch_status();
}

// ------------------------------------------------------------------------------------------------
// AvgPool2d(2, 2, 0, ceil_mode=True) backward: dx[2ho+dy, 2wo+dx] = dp[ho, wo] / (#in-bounds taps of that window)
__global__ __launch_bounds__(256) void avgpool2_bwd_kernel(const bf16_t* __restrict__ dp, int lddp, bf16_t* __restrict__ dx, int lddx, int B,
                                                           int H, int W, int C8, int Ho, int Wo) {
  const int64_t total = (int64_t)B * H * W * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    int64_t p = i / C8;
    const int x = (int)(p % W);
    p /= W;
    const int y = (int)(p % H);
    const int b = (int)(p / H);
    const int ho = y >> 1, wo = x >> 1;
    const int cnt = ((2 * ho + 1 < H) ? 2 : 1) * ((2 * wo + 1 < W) ? 2 : 1);
    float g[8];
    unpack_bf16x8(*reinterpret_cast<const uint4*>(dp + (((int64_t)b * Ho + ho) * Wo + wo) * lddp + c8 * 8), g);
    const float inv = 1.0f / (float)cnt;
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] *= inv;
    *reinterpret_cast<uint4*>(dx + (((int64_t)b * H + y) * W + x) * lddx + c8 * 8) = pack_bf16x8(g);
  }
}

extern "C" int fx_avgpool2x2_bwd_nhwc_bf16(const void* dp, int lddp, void* dx, int lddx, int B, int H, int W, int C, fx_stream_t stream_) {
  FX_CHECK_ARG(dp && dx && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && lddp >= C && lddx >= C && lddp % 8 == 0 && lddx % 8 == 0);
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  int64_t total = (int64_t)B * H * W * (C / 8);
  int64_t grid = (total + 255) / 256;
  if (grid > 256 * 32) grid = 256 * 32;
  hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3((int)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), (const bf16_t*)dp, lddp,
                     (bf16_t*)dx, lddx, B, H, W, C / 8, Ho, Wo);
  return fx_launch_status();
}

// ------------------------------------------------------------------------------------------------
// MaxPool2d(3, 2, 1) backward, gather form (no atomics, deterministic), two passes:
//   1. per output window: the tap (kh*3 + kw) of its maximum -> one byte per output element.  PyTorch's CPU/GPU kernels keep the FIRST
//      maximum in window scan order (kh-major) - after a ReLU ties (zeros) are common, so the scan order is reproduced: a later tap wins
//      only if strictly greater;
//   2. per input pixel: dy of every window (1, 2 or 4 of them) whose recorded tap it is.
// (A single pass that re-derives the arg-max of each of the up to 4 windows per input pixel reads 36 taps per pixel through the
// caches: 0.75 ms for the [16,320,320,64] stem activation; the arg-max bytes cut that to 9 taps per WINDOW plus <= 4 x 24 bytes per pixel.)
__global__ __launch_bounds__(256) void maxpool3x3s2_argmax_kernel(const bf16_t* __restrict__ x, int ldx, unsigned char* __restrict__ arg, int B,
                                                                  int H, int W, int C8, int Ho, int Wo) {
  const int64_t total = (int64_t)B * Ho * Wo * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    int64_t p = i / C8;
    const int wo = (int)(p % Wo);
    p /= Wo;
    const int ho = (int)(p % Ho);
    const int b = (int)(p / Ho);
    float best[8];
    int tap[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) best[j] = -INFINITY, tap[j] = -1;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int hi = ho * 2 - 1 + kh;
      if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int wi = wo * 2 - 1 + kw;
        if ((unsigned)wi >= (unsigned)W) continue;
        float f[8];
        unpack_bf16x8(*reinterpret_cast<const uint4*>(x + (((int64_t)b * H + hi) * W + wi) * ldx + c8 * 8), f);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (f[j] > best[j] || tap[j] < 0) best[j] = f[j], tap[j] = kh * 3 + kw;
      }
    }
    uint2 o;
    o.x = (unsigned)tap[0] | ((unsigned)tap[1] << 8) | ((unsigned)tap[2] << 16) | ((unsigned)tap[3] << 24);
    o.y = (unsigned)tap[4] | ((unsigned)tap[5] << 8) | ((unsigned)tap[6] << 16) | ((unsigned)tap[7] << 24);
    *reinterpret_cast<uint2*>(arg + i * 8) = o;
  }
}

__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_kernel(const unsigned char* __restrict__ arg, const bf16_t* __restrict__ dy, int lddy,
                                                               bf16_t* __restrict__ dx, int lddx, int B, int H, int W, int C8, int Ho,
                                                               int Wo) {
  const int64_t total = (int64_t)B * H * W * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    int64_t p = i / C8;
    const int xi = (int)(p % W);
    p /= W;
    const int yi = (int)(p % H);
    const int b = (int)(p / H);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // windows (ho, wo) with 2*ho - 1 <= yi <= 2*ho + 1
    const int ho_lo = max(yi / 2, 0), ho_hi = min((yi + 1) / 2, Ho - 1);
    const int wo_lo = max(xi / 2, 0), wo_hi = min((xi + 1) / 2, Wo - 1);
    for (int ho = ho_lo; ho <= ho_hi; ++ho)
      for (int wo = wo_lo; wo <= wo_hi; ++wo) {
        const unsigned me = (unsigned)((yi - (ho * 2 - 1)) * 3 + (xi - (wo * 2 - 1)));
        const int64_t w = (((int64_t)b * Ho + ho) * Wo + wo);
        const uint2 a = *reinterpret_cast<const uint2*>(arg + (w * C8 + c8) * 8);
        if (a.x == me * 0x01010101u && a.y == a.x) {   // all eight channels chose this pixel
          float g[8];
          unpack_bf16x8(*reinterpret_cast<const uint4*>(dy + w * lddy + c8 * 8), g);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += g[j];
          continue;
        }
        bool any = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) any |= ((a.x >> (8 * j)) & 0xffu) == me || ((a.y >> (8 * j)) & 0xffu) == me;
        if (!any) continue;
        float g[8];
        unpack_bf16x8(*reinterpret_cast<const uint4*>(dy + w * lddy + c8 * 8), g);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (((a.x >> (8 * j)) & 0xffu) == me) acc[j] += g[j];
          if (((a.y >> (8 * j)) & 0xffu) == me) acc[4 + j] += g[4 + j];
        }
      }
    *reinterpret_cast<uint4*>(dx + (((int64_t)b * H + yi) * W + xi) * lddx + c8 * 8) = pack_bf16x8(acc);
  }
}

extern "C" int fx_maxpool3x3s2_bwd_nhwc_bf16(const void* x, int ldx, const void* dy, int lddy, void* dx, int lddx, int B, int H, int W, int C,
                                             void* workspace, fx_stream_t stream_) {
  FX_CHECK_ARG(x && dy && dx && workspace && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0);
  FX_CHECK_ARG(ldx >= C && lddy >= C && lddx >= C && ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0 && ((uintptr_t)workspace % 8) == 0);
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  int64_t total = (int64_t)B * Ho * Wo * (C / 8);
  int64_t grid = (total + 255) / 256;
  if (grid > 256 * 32) grid = 256 * 32;
  hipLaunchKernelGGL(maxpool3x3s2_argmax_kernel, dim3((int)grid), dim3(256), 0, stream, (const bf16_t*)x, ldx, (unsigned char*)workspace, B, H, W,
                     C / 8, Ho, Wo);
  total = (int64_t)B * H * W * (C / 8);
  grid = (total + 255) / 256;
  if (grid > 256 * 32) grid = 256 * 32;
  hipLaunchKernelGGL(maxpool3x3s2_bwd_kernel, dim3((int)grid), dim3(256), 0, stream, (const unsigned char*)workspace, (const bf16_t*)dy, lddy,
                     (bf16_t*)dx, lddx, B, H, W, C / 8, Ho, Wo);
  return fx_launch_status();
}

// ------------------------------------------------------------------------------------------------
// (x - mean) / std of the uint8 / fp32 HWC image, as bf16 NHWC with the 3 channels padded to 8 (zeros): the activation
// tensor the stem conv's weight gradient reads (fx_conv2d_wgrad_nhwc_bf16 needs C % 8 == 0).
__global__ __launch_bounds__(256) void normalize_pad8_kernel(const void* __restrict__ img, int is_f32, const float* __restrict__ mean,
                                                             const float* __restrict__ inv_std, bf16_t* __restrict__ out, int64_t pixels) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < pixels; i += (int64_t)gridDim.x * 256) {
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float raw = is_f32 ? reinterpret_cast<const float*>(img)[i * 3 + c] : (float)reinterpret_cast<const unsigned char*>(img)[i * 3 + c];
      v[c] = (raw - mean[c]) * inv_std[c];
    }
    *reinterpret_cast<uint4*>(out + i * 8) = pack_bf16x8(v);
  }
}

extern "C" int fx_normalize_pad8(const void* img, int is_f32, const float* mean, const float* inv_std, void* out, int64_t pixels,
                                 fx_stream_t stream_) {
  FX_CHECK_ARG(img && mean && inv_std && out && pixels > 0);
  int64_t grid = (pixels + 255) / 256;
  if (grid > 256 * 32) grid = 256 * 32;
  hipLaunchKernelGGL(normalize_pad8_kernel, dim3((int)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), img, is_f32, mean, inv_std,
                     (bf16_t*)out, pixels);
  return fx_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Train-mode BatchNorm2d around the convolution (ConvNormLayer under model.train(), focoos/nn/layers/conv.py:78-98 with
// nn.BatchNorm2d / SyncBatchNorm: batch statistics over (B, H, W) per channel).  NHWC bf16 activations [rows][C], fp32
// statistics.  Four HBM-bound passes, each thread 8 channels x a strided set of rows:
//   forward : fx_bn_stats_bf16  -> sums[c] = sum z, sums[C + c] = sum z^2           (host: all-reduce for SyncBN, mean / var)
//             fx_bn_apply_bf16  -> y = act(z * scale[c] + shift[c] [+ residual])     scale = gamma * rstd, shift = beta - mean * scale
//   backward: da = dy * act'(a), a = z * scale + shift (recomputed, nothing but z is kept from the forward)
//             fx_bn_bwd_stats_bf16 -> sums[c] = sum da (= dbeta), sums[C + c] = sum da * xhat (= dgamma), xhat = (z - mean) * rstd
//             fx_bn_bwd_apply_bf16 -> dz = scale * (da - sums[c] / n - xhat * sums[C + c] / n)   [and da itself for a residual branch]
// The conv output z is read as bf16 or - z_f32 - as fp32: with batch statistics y depends on z - mean, and a bf16 z keeps 8 bits of z,
// not of z - mean; channels whose |mean| is several standard deviations lose most of their signal, and the error compounds over the
// 50-100 normalised layers of a backbone.  The trainable graphs therefore keep z in fp32 (conv epilogue out_f32).
template <typename ZT>
__device__ __forceinline__ void bn_ld8(const ZT* __restrict__ p, float* v);
template <>
__device__ __forceinline__ void bn_ld8<bf16_t>(const bf16_t* __restrict__ p, float* v) {
  unpack_bf16x8(*reinterpret_cast<const uint4*>(p), v);
}
template <>
__device__ __forceinline__ void bn_ld8<float>(const float* __restrict__ p, float* v) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

#define BN_ROWS(C) ((C) >= 256 ? 256 : 1024)   // rows per workgroup: narrow tensors have many more rows and 4-8x the row lanes
__device__ __forceinline__ float bn_act_grad(float a, int act) {
  switch (act) {
    case FX_ACT_RELU: return a > 0.0f ? 1.0f : 0.0f;
    case FX_ACT_SILU: {
      const float s = 1.0f / (1.0f + __expf(-a));
      return s * (1.0f + a * (1.0f - s));
    }
    case FX_ACT_GELU: return 0.5f * (1.0f + erff(a * 0.70710678118654752f)) + a * 0.3989422804014327f * __expf(-0.5f * a * a);
    default: return 1.0f;
  }
}

// shared column-reduction skeleton: 32 column groups x 8 row lanes per workgroup, BN_ROWS rows, two sums per channel
template <int MODE, typename ZT>  // 0: (z, z^2)   1: (da, da * xhat)
__global__ __launch_bounds__(256) void bn_reduce_kernel(const ZT* __restrict__ z, int ldz, const bf16_t* __restrict__ dy, int lddy,
                                                        const bf16_t* __restrict__ res, int ldr, const float* __restrict__ scale, const float* __restrict__ shift,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd, int act,
                                                        float* __restrict__ sums, int64_t rows, int C, int rpb) {
  __shared__ float part[2][2048];
  // column groups x row lanes: 32 x 8 for wide tensors; narrow ones (C = 32 / 64 / 128 - most of the early backbone) get
  // 4 x 64 / 8 x 32 / 16 x 16 so that every thread has work
  const int cb = C - blockIdx.x * 256;
  const int ncg = cb == 32 ? 4 : (cb == 64 ? 8 : (cb == 128 ? 16 : 32));
  const int nrl = 256 / ncg;
  const int cg = threadIdx.x % ncg, rl = threadIdx.x / ncg;
  const int c0 = blockIdx.x * 256 + cg * 8;
  const int64_t r0 = (int64_t)blockIdx.y * rpb;
  const int64_t r1 = r0 + rpb < rows ? r0 + rpb : rows;
  float s0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c0 < C) {
    float sc[8], sh[8], mu[8], rs[8];
    if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) sc[j] = scale[c0 + j], sh[j] = shift[c0 + j], mu[j] = mean[c0 + j], rs[j] = rstd[c0 + j];
    }
    for (int64_t r = r0 + rl; r < r1; r += nrl) {
      float v[8];
      bn_ld8<ZT>(z + r * ldz + c0, v);
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) s0[j] += v[j], s1[j] += v[j] * v[j];
      } else {
        float g[8], rr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        unpack_bf16x8(*reinterpret_cast<const uint4*>(dy + r * lddy + c0), g);
        if (res) unpack_bf16x8(*reinterpret_cast<const uint4*>(res + r * ldr + c0), rr);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float da = g[j] * bn_act_grad(v[j] * sc[j] + sh[j] + rr[j], act);
          s0[j] += da;
          s1[j] += da * (v[j] - mu[j]) * rs[j];
        }
      }
    }
  }
  const int cols = ncg * 8;
#pragma unroll
  for (int j = 0; j < 8; ++j) part[0][rl * cols + cg * 8 + j] = s0[j], part[1][rl * cols + cg * 8 + j] = s1[j];
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if ((int)threadIdx.x < cols && c < C) {
    float a = 0.0f, b = 0.0f;
    for (int i = 0; i < nrl; ++i) a += part[0][i * cols + threadIdx.x], b += part[1][i * cols + threadIdx.x];
    unsafeAtomicAdd(sums + c, a);
    unsafeAtomicAdd(sums + C + c, b);
  }
}

extern "C" int fx_bn_stats_bf16(const void* z, int ldz, int z_f32, float* sums, int64_t rows, int C, fx_stream_t stream_) {
  FX_CHECK_ARG(z && sums && rows > 0 && C > 0 && C % 8 == 0 && ldz >= C && ldz % 8 == 0);
  const dim3 grid((C + 255) / 256, (unsigned)((rows + BN_ROWS(C) - 1) / BN_ROWS(C)));
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (z_f32)
    hipLaunchKernelGGL((bn_reduce_kernel<0, float>), grid, dim3(256), 0, stream, (const float*)z, ldz, nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr,
                       nullptr, 0, sums, rows, C, BN_ROWS(C));
  else
    hipLaunchKernelGGL((bn_reduce_kernel<0, bf16_t>), grid, dim3(256), 0, stream, (const bf16_t*)z, ldz, nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr,
                       nullptr, 0, sums, rows, C, BN_ROWS(C));
  return fx_launch_status();
}

extern "C" int fx_bn_bwd_stats_bf16(const void* dy, int lddy, const void* z, int ldz, int z_f32, const void* residual, int ldr, const float* scale,
                                    const float* shift, const float* mean, const float* rstd, int act, float* sums, int64_t rows, int C,
                                    fx_stream_t stream_) {
  FX_CHECK_ARG(!residual || (ldr >= C && ldr % 8 == 0));
  FX_CHECK_ARG(dy && z && scale && shift && mean && rstd && sums && rows > 0 && C > 0 && C % 8 == 0);
  FX_CHECK_ARG(ldz >= C && lddy >= C && ldz % 8 == 0 && lddy % 8 == 0);
  const dim3 grid((C + 255) / 256, (unsigned)((rows + BN_ROWS(C) - 1) / BN_ROWS(C)));
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (z_f32)
    hipLaunchKernelGGL((bn_reduce_kernel<1, float>), grid, dim3(256), 0, stream, (const float*)z, ldz, (const bf16_t*)dy, lddy, (const bf16_t*)residual, ldr,
                       scale, shift, mean, rstd, act, sums, rows, C, BN_ROWS(C));
  else
    hipLaunchKernelGGL((bn_reduce_kernel<1, bf16_t>), grid, dim3(256), 0, stream, (const bf16_t*)z, ldz, (const bf16_t*)dy, lddy, (const bf16_t*)residual,
                       ldr, scale, shift, mean, rstd, act, sums, rows, C, BN_ROWS(C));
  return fx_launch_status();
}

template <typename ZT>
__global__ __launch_bounds__(256) void bn_apply_kernel(const ZT* __restrict__ z, int ldz, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const bf16_t* __restrict__ res, int ldr, int act,
                                                       bf16_t* __restrict__ y, int ldy, int64_t rows, int C8) {
  const int64_t total = rows * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    const int64_t r = i / C8;
    float v[8];
    bn_ld8<ZT>(z + r * ldz + c8 * 8, v);
    const float4 s0 = *reinterpret_cast<const float4*>(scale + c8 * 8), s1 = *reinterpret_cast<const float4*>(scale + c8 * 8 + 4);
    const float4 h0 = *reinterpret_cast<const float4*>(shift + c8 * 8), h1 = *reinterpret_cast<const float4*>(shift + c8 * 8 + 4);
    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = v[j] * sc[j] + sh[j];
    if (res) {
      float rr[8];
      unpack_bf16x8(*reinterpret_cast<const uint4*>(res + r * ldr + c8 * 8), rr);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += rr[j];
    }
    if (act != FX_ACT_NONE) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = fx_act(v[j], act);
    }
    *reinterpret_cast<uint4*>(y + r * ldy + c8 * 8) = pack_bf16x8(v);
  }
}

extern "C" int fx_bn_apply_bf16(const void* z, int ldz, int z_f32, const float* scale, const float* shift, const void* residual, int ldr, int act, void* y,
                                int ldy, int64_t rows, int C, fx_stream_t stream_) {
  FX_CHECK_ARG(z && scale && shift && y && rows > 0 && C > 0 && C % 8 == 0 && ldz >= C && ldy >= C && ldz % 8 == 0 && ldy % 8 == 0);
  FX_CHECK_ARG(!residual || (ldr >= C && ldr % 8 == 0));
  int64_t total = rows * (C / 8);
  int64_t grid = (total + 255) / 256;
  if (grid > 256 * 32) grid = 256 * 32;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (z_f32)
    hipLaunchKernelGGL(bn_apply_kernel<float>, dim3((int)grid), dim3(256), 0, stream, (const float*)z, ldz, scale, shift, (const bf16_t*)residual, ldr, act,
                       (bf16_t*)y, ldy, rows, C / 8);
  else
    hipLaunchKernelGGL(bn_apply_kernel<bf16_t>, dim3((int)grid), dim3(256), 0, stream, (const bf16_t*)z, ldz, scale, shift, (const bf16_t*)residual, ldr,
                       act, (bf16_t*)y, ldy, rows, C / 8);
  return fx_launch_status();
}

// With a residual the activation input is a = z * scale + shift + residual; the residual is re-read rather than keeping a.
__device__ __forceinline__ void ld8(const float* __restrict__ p, float* f) {   // 8 consecutive per-channel values (32-byte aligned)
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}

template <typename ZT>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const bf16_t* __restrict__ dy, int lddy, const ZT* __restrict__ z, int ldz,
                                                           const bf16_t* __restrict__ res, int ldr, const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd, int act,
                                                           const float* __restrict__ sums, float inv_n, bf16_t* __restrict__ da_out, int ldda,
                                                           bf16_t* __restrict__ dz, int lddz, int64_t rows, int C, float* __restrict__ dgamma_acc,
                                                           float* __restrict__ dbeta_acc) {
  // the affine gradients ARE the reduction results (dbeta = sums[c], dgamma = sums[C + c]): one workgroup adds them to the parameter
  // gradients here instead of two tiny tensor-add launches per BatchNorm layer
  if (blockIdx.x == 0 && dgamma_acc)
    for (int c = threadIdx.x; c < C; c += 256) {
      dgamma_acc[c] += sums[C + c];
      dbeta_acc[c] += sums[c];
    }
  const int C8 = C / 8;
  const int64_t total = rows * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    const int64_t r = i / C8;
    float v[8], g[8], o[8], d[8], rr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bn_ld8<ZT>(z + r * ldz + c8 * 8, v);
    unpack_bf16x8(*reinterpret_cast<const uint4*>(dy + r * lddy + c8 * 8), g);
    if (res) unpack_bf16x8(*reinterpret_cast<const uint4*>(res + r * ldr + c8 * 8), rr);
    float sc[8], sh[8], mu[8], rs[8], s0[8], s1[8];
    ld8(scale + c8 * 8, sc); ld8(shift + c8 * 8, sh); ld8(mean + c8 * 8, mu); ld8(rstd + c8 * 8, rs);
    ld8(sums + c8 * 8, s0); ld8(sums + C + c8 * 8, s1);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float da = g[j] * bn_act_grad(v[j] * sc[j] + sh[j] + rr[j], act);
      const float xh = (v[j] - mu[j]) * rs[j];
      d[j] = da;
      o[j] = sc[j] * (da - s0[j] * inv_n - xh * s1[j] * inv_n);
    }
    if (da_out) *reinterpret_cast<uint4*>(da_out + r * ldda + c8 * 8) = pack_bf16x8(d);
    *reinterpret_cast<uint4*>(dz + r * lddz + c8 * 8) = pack_bf16x8(o);
  }
}

extern "C" int fx_bn_bwd_apply_bf16(const void* dy, int lddy, const void* z, int ldz, int z_f32, const void* residual, int ldr, const float* scale,
                                    const float* shift, const float* mean, const float* rstd, int act, const float* sums, float inv_n,
                                    void* da_out, int ldda, void* dz, int lddz, int64_t rows, int C, float* dgamma_acc, float* dbeta_acc,
                                    fx_stream_t stream_) {
  FX_CHECK_ARG(!residual || (ldr >= C && ldr % 8 == 0));
  FX_CHECK_ARG(!dgamma_acc == !dbeta_acc);
  FX_CHECK_ARG(dy && z && scale && shift && mean && rstd && sums && dz && rows > 0 && C > 0 && C % 8 == 0);
  FX_CHECK_ARG(ldz >= C && lddy >= C && lddz >= C && ldz % 8 == 0 && lddy % 8 == 0 && lddz % 8 == 0 && (!da_out || (ldda >= C && ldda % 8 == 0)));
  int64_t total = rows * (C / 8);
  int64_t grid = (total + 255) / 256;
  if (grid > 256 * 32) grid = 256 * 32;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (z_f32)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3((int)grid), dim3(256), 0, stream, (const bf16_t*)dy, lddy, (const float*)z, ldz,
                       (const bf16_t*)residual, ldr, scale, shift, mean, rstd, act, sums, inv_n, (bf16_t*)da_out, ldda, (bf16_t*)dz, lddz, rows, C,
                       dgamma_acc, dbeta_acc);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3((int)grid), dim3(256), 0, stream, (const bf16_t*)dy, lddy, (const bf16_t*)z, ldz,
                       (const bf16_t*)residual, ldr, scale, shift, mean, rstd, act, sums, inv_n, (bf16_t*)da_out, ldda, (bf16_t*)dz, lddz, rows, C,
                       dgamma_acc, dbeta_acc);
  return fx_launch_status();
}

// Per-channel epilogue of the forward statistics (one launch instead of a dozen tiny tensor ops): batch mean / biased
// variance from the (all-reduced) sums, the affine that fx_bn_apply_bf16 uses, and the running-statistics update of
// nn.BatchNorm2d (momentum, unbiased variance; num_batches_tracked += 1).
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ sums, float n, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, float momentum,
                                                          float* __restrict__ running_mean, float* __restrict__ running_var,
                                                          int64_t* __restrict__ num_batches_tracked, float* __restrict__ mean,
                                                          float* __restrict__ rstd, float* __restrict__ scale, float* __restrict__ shift, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
  if (c >= C) return;
  const float m = sums[c] / n;
  const float var = fmaxf(sums[C + c] / n - m * m, 0.0f);
  const float r = rsqrtf(var + eps);
  const float sc = gamma[c] * r;
  mean[c] = m;
  rstd[c] = r;
  scale[c] = sc;
  shift[c] = beta[c] - m * sc;
  if (running_mean) {
    running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * m;
    running_var[c] = (1.0f - momentum) * running_var[c] + momentum * var * (n > 1.0f ? n / (n - 1.0f) : 1.0f);
  }
}

extern "C" int fx_bn_finalize_f32(const float* sums, float n, const float* gamma, const float* beta, float eps, float momentum,
                                  float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean, float* rstd, float* scale,
                                  float* shift, int C, fx_stream_t stream_) {
  FX_CHECK_ARG(sums && gamma && beta && mean && rstd && scale && shift && C > 0 && n > 0.0f && (!running_mean == !running_var));
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), sums, n, gamma, beta, eps,
                     momentum, running_mean, running_var, num_batches_tracked, mean, rstd, scale, shift, C);
  return fx_launch_status();
}
