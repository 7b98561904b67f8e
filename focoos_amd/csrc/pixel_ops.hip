// Bandwidth-bound pixel kernels of the RT-DETR path (gfx950): stem conv with fused normalisation,
// bilinear resizes, 3x3/s2 max-pool.  All are coalesced-HBM wavefront kernels: 8 bf16 channels
// (16 bytes) per lane, NHWC, no LDS staging (no reuse beyond what L2 already gives).
#include <stdlib.h>

#include "common.h"

static int fx_env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

// ------------------------------------------------------------------------------------------------
// Stem: (x-mean)*inv_std  ->  3x3/s2/p1 conv (3 -> 32)  ->  +bias (BN folded)  ->  ReLU  -> bf16 NHWC.
// Four lanes per output pixel, 8 output channels each: the 27 normalised taps + 8 fp32 accumulators live in
// registers, the [27][32] fp32 weight table sits in LDS and is read as float4 (4 distinct addresses per
// wave instruction -> conflict-free broadcast), and the wave stores 16 pixels x 64 B = 1 KiB contiguous.
// Zero padding applies to the NORMALISED image (the reference pads after normalising): padded taps add 0.
template <typename TIn, bool RELU>
__global__ __launch_bounds__(256) void stem_conv_kernel(const TIn* __restrict__ x, const float* __restrict__ wt /*[27][32]*/,
                                                         const float* __restrict__ bias, const float* __restrict__ mean,
                                                         const float* __restrict__ inv_std, bf16_t* __restrict__ y, int B, int H,
                                                         int W, int Ho, int Wo) {
  __shared__ __attribute__((aligned(16))) float ws[27 * 32];
  for (int i = threadIdx.x; i < 27 * 32; i += 256) ws[i] = wt[i];
  __syncthreads();
  const int total = B * Ho * Wo;
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int o = gid >> 2, cg = gid & 3;
  if (o >= total) return;
  const int b = o / (Ho * Wo), rem = o - b * Ho * Wo;
  const int ho = rem / Wo, wo = rem - ho * Wo;
  const float m0 = mean[0], m1 = mean[1], m2 = mean[2], s0 = inv_std[0], s1 = inv_std[1], s2 = inv_std[2];
  float acc[8];
  {
    float4 b0 = *reinterpret_cast<const float4*>(bias + cg * 8), b1 = *reinterpret_cast<const float4*>(bias + cg * 8 + 4);
    acc[0] = b0.x; acc[1] = b0.y; acc[2] = b0.z; acc[3] = b0.w; acc[4] = b1.x; acc[5] = b1.y; acc[6] = b1.z; acc[7] = b1.w;
  }
#pragma unroll 1
  for (int kh = 0; kh < 3; ++kh) {
    const int hi = ho * 2 - 1 + kh;
    const int hc = min(max(hi, 0), H - 1);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int wi = wo * 2 - 1 + kw;
      const int wc = min(max(wi, 0), W - 1);
      const bool ok = (hi == hc) && (wi == wc);
      const TIn* px = x + ((int64_t)(b * H + hc) * W + wc) * 3;
      float v[3] = {ok ? ((float)px[0] - m0) * s0 : 0.0f, ok ? ((float)px[1] - m1) * s1 : 0.0f, ok ? ((float)px[2] - m2) * s2 : 0.0f};
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float* wr = ws + ((kh * 3 + kw) * 3 + c) * 32 + cg * 8;
        float4 w0 = *reinterpret_cast<const float4*>(wr), w1 = *reinterpret_cast<const float4*>(wr + 4);
        acc[0] = fmaf(v[c], w0.x, acc[0]); acc[1] = fmaf(v[c], w0.y, acc[1]); acc[2] = fmaf(v[c], w0.z, acc[2]); acc[3] = fmaf(v[c], w0.w, acc[3]);
        acc[4] = fmaf(v[c], w1.x, acc[4]); acc[5] = fmaf(v[c], w1.y, acc[5]); acc[6] = fmaf(v[c], w1.z, acc[6]); acc[7] = fmaf(v[c], w1.w, acc[7]);
      }
    }
  }
  if (RELU) {
#pragma unroll
    for (int n = 0; n < 8; ++n) acc[n] = fmaxf(acc[n], 0.0f);
  }
  *reinterpret_cast<uint4*>(y + (int64_t)o * 32 + cg * 8) = pack_bf16x8(acc);
}

// ------------------------------------------------------------------------------------------------
// The same convolution for uint8 images on the matrix cores.  The VALU kernel above runs at ~20 fp32 TFLOP/s (27 byte loads and
// 216 FMAs per lane; 0.9 TB/s of HBM traffic for a layer that reads 1.2 MB and writes 6.5 MB per 640^2 image); as a GEMM it is
// [32 channels x K = 32] x [K x pixels] with K = the 27 taps padded to 32, i.e. two v_mfma_f32_32x32x16_bf16 per 32 pixels, and
// the kernel becomes a stream of byte loads, conversions and 8-byte stores.
//   * A operand = the (BN-folded) weights, built once per wave from the fp32 table; B operand = normalised taps of the lane's
//     pixel.  The K order is free as long as A and B agree: a pixel's taps are 3 image rows of 9 consecutive bytes
//     (3 pixels x RGB), so K is laid out as  [row0 bytes 0-7 | row2 bytes 0-7]  (first MFMA: lane halves 0 / 1) and
//     [row1 bytes 0-7 | byte 8 of rows 0,1,2 + 5 zeros]  (second MFMA) - each lane fetches one or two 8-byte runs.
//   * runs start at arbitrary byte offsets: three aligned dword buffer loads + v_alignbyte; the buffer descriptor returns zeros
//     past the end, and whatever is read left of the image or above it is masked after normalisation (zero padding applies to the
//     NORMALISED image).
//   * with weights as A, a lane's accumulators are 4 consecutive channels of one pixel -> bias, ReLU, 8-byte bf16 stores.
// Inputs and weights are rounded to bf16 (like every later layer).  The VALU kernel above remains as the FX_STEM_MFMA=0 fallback.
__device__ __forceinline__ void stem_run8(__amdgpu_buffer_rsrc_t r, int a, unsigned& lo, unsigned& hi, unsigned& b8) {
  const unsigned a4 = (unsigned)a & ~3u;
  const unsigned sh = (unsigned)a & 3u;
  const unsigned d0 = __builtin_amdgcn_raw_buffer_load_b32(r, a4, 0, 0);
  const unsigned d1 = __builtin_amdgcn_raw_buffer_load_b32(r, a4 + 4u, 0, 0);
  const unsigned d2 = __builtin_amdgcn_raw_buffer_load_b32(r, a4 + 8u, 0, 0);
  lo = __builtin_amdgcn_alignbyte(d1, d0, sh);
  hi = __builtin_amdgcn_alignbyte(d2, d1, sh);
  b8 = (sh == 0 ? d2 : __builtin_amdgcn_alignbyte(0u, d2, sh)) & 0xffu;   // byte a+8 (sh <= 3: it lies in d2)
}

// fp32 images (the reference's float NCHW contract, values on the 0..255 scale): the same 9 taps as floats; identical arithmetic
// after the load, so a float image holding integer values gives bit-identical results to its uint8 form
__device__ __forceinline__ void stem_run9_f32(__amdgpu_buffer_rsrc_t r, int elem, float* f, float& f8) {
  const unsigned off = (unsigned)elem * 4u;
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, off + 4u * j, 0, 0));
  f8 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, off + 32u, 0, 0));
}

template <typename TIn, bool RELU>
__global__ __launch_bounds__(256) void stem_mfma_kernel(const TIn* __restrict__ x, unsigned x_bytes, const float* __restrict__ wt /*[27][32]*/,
                                                         const float* __restrict__ bias, const float* __restrict__ mean,
                                                         const float* __restrict__ inv_std, bf16_t* __restrict__ y, int B, int H, int W, int Ho,
                                                         int Wo, int tiles_per_wave) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, hh = lane >> 5;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, x_bytes, 0x00020000);
  // ---- A fragments: channel = col; slot j of the lane's 8 K-values -> (image row rr, byte bb) per the layout above
  bf16x8 a1, a2;
  {
    float f1[8], f2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int rr1 = hh ? 2 : 0;                        // first MFMA: row 0 / row 2, byte j
      f1[j] = wt[((rr1 * 3 + j / 3) * 3 + j % 3) * 32 + col];
      if (hh == 0) f2[j] = wt[((1 * 3 + j / 3) * 3 + j % 3) * 32 + col];           // second MFMA, low half: row 1, byte j
      else f2[j] = j < 3 ? wt[((j * 3 + 2) * 3 + 2) * 32 + col] : 0.0f;            // high half: byte 8 (kw = 2, c = 2) of rows 0, 1, 2
    }
    a1 = __builtin_bit_cast(bf16x8, pack_bf16x8(f1));
    a2 = __builtin_bit_cast(bf16x8, pack_bf16x8(f2));
  }
  const float m0 = mean[0], m1 = mean[1], m2 = mean[2], s0 = inv_std[0], s1 = inv_std[1], s2 = inv_std[2];
  // byte j of a run is channel j % 3
  const float mj[8] = {m0, m1, m2, m0, m1, m2, m0, m1}, sj[8] = {s0, s1, s2, s0, s1, s2, s0, s1};
  // epilogue constants: accumulator r holds channel (r & 3) + 8 * (r >> 2) + 4 * hh of pixel `col`
  float bs[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bs[r] = bias[(r & 3) + 8 * (r >> 2) + 4 * hh];
  const int total = B * Ho * Wo;
  const int tile0 = (blockIdx.x * 4 + wave) * tiles_per_wave;
#pragma unroll 1
  for (int ti = 0; ti < tiles_per_wave; ++ti) {
    const int o = (tile0 + ti) * 32 + col;
    if ((tile0 + ti) * 32 >= total) break;
    const bool act = o < total;
    const int oo = act ? o : total - 1;
    const int b = oo / (Ho * Wo), rem = oo - b * Ho * Wo;
    const int ho = rem / Wo, wo = rem - ho * Wo;
    const int wi0 = 2 * wo - 1;
    // column validity of the 3 taps (pixels wi0, wi0+1, wi0+2) -> per byte j: pixel j / 3
    const bool c0 = wi0 >= 0, c2 = wi0 + 2 < W;   // the middle one always exists
    auto run = [&](int rr, float* f, float& f8) {
      const int hi_ = 2 * ho - 1 + rr;
      const bool rv = act && hi_ >= 0 && hi_ < H;
      // leftmost pixels: the run would start 3 elements before the row (before the buffer for the very first pixel, where a wrapped
      // offset is not reliably range-checked) - fetch from the row start and shift the taps up instead; taps 0-2 are masked anyway
      const bool neg = wi0 < 0;
      const int elem = rv ? ((b * H + hi_) * W + (neg ? 0 : wi0)) * 3 : 0;
      float raw[8], raw8;
      if (sizeof(TIn) == 1) {
        unsigned lo, hi, b8;
        stem_run8(xr, elem, lo, hi, b8);
        if (neg) {
          b8 = (hi >> 8) & 0xffu;
          hi = __builtin_amdgcn_alignbyte(hi, lo, 1);
          lo = __builtin_amdgcn_alignbyte(lo, 0u, 1);
        }
        const unsigned w[2] = {lo, hi};
#pragma unroll
        for (int j = 0; j < 8; ++j) raw[j] = (float)((w[j >> 2] >> (8 * (j & 3))) & 0xffu);
        raw8 = (float)b8;
      } else {
        float t[8], t8;
        stem_run9_f32(xr, elem, t, t8);
#pragma unroll
        for (int j = 0; j < 8; ++j) raw[j] = neg ? (j >= 3 ? t[j - 3] : 0.0f) : t[j];
        raw8 = neg ? t[5] : t8;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool ok = rv && (j < 3 ? c0 : (j < 6 ? true : c2));
        f[j] = ok ? (raw[j] - mj[j]) * sj[j] : 0.0f;
      }
      f8 = (rv && c2) ? (raw8 - m2) * s2 : 0.0f;
    };
    // both lane halves run the same code: rows (0 | 2) and 1; the upper half gets byte 8 of row 0 from its partner lane
    float f1[8], fb[8], ta, tb;
    run(hh ? 2 : 0, f1, ta);
    run(1, fb, tb);
    const float ta_row0 = __shfl_xor(ta, 32, 64);
    float f2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f2[j] = hh ? 0.0f : fb[j];
    if (hh) f2[0] = ta_row0, f2[1] = tb, f2[2] = ta;
    const bf16x8 b1 = __builtin_bit_cast(bf16x8, pack_bf16x8(f1)), b2 = __builtin_bit_cast(bf16x8, pack_bf16x8(f2));
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bs[r];
    acc = FX_MFMA_32x32x16(a1, b1, acc);
    acc = FX_MFMA_32x32x16(a2, b2, acc);
    if (act) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v0 = acc[4 * g], v1 = acc[4 * g + 1], v2 = acc[4 * g + 2], v3 = acc[4 * g + 3];
        if (RELU) v0 = fmaxf(v0, 0.0f), v1 = fmaxf(v1, 0.0f), v2 = fmaxf(v2, 0.0f), v3 = fmaxf(v3, 0.0f);
        uint2 pk;
        pk.x = pack_bf16x2(v0, v1);
        pk.y = pack_bf16x2(v2, v3);
        *reinterpret_cast<uint2*>(y + (int64_t)o * 32 + 8 * g + 4 * hh) = pk;
      }
    }
  }
}

static int stem_launch(const void* x, int in_f32, const float* w, const float* bias, const float* mean, const float* inv_std, void* y, int B,
                       int H, int W, int Cout, int relu, fx_stream_t stream_) {
  FX_CHECK_ARG(x && w && bias && mean && inv_std && y && B > 0 && H > 0 && W > 0);
  if (Cout != 32) return FX_ERR_UNSUPPORTED;
  int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  int64_t total = (int64_t)B * Ho * Wo;
  if (total * 4 >= (1ll << 31)) return FX_ERR_UNSUPPORTED;
  int grid = (int)((total * 4 + 255) / 256);
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
#define STEM_GO(T, R) \
  hipLaunchKernelGGL((stem_conv_kernel<T, R>), dim3(grid), dim3(256), 0, stream, (const T*)x, w, bias, mean, inv_std, (bf16_t*)y, B, H, W, Ho, Wo)
  const int64_t x_bytes = (int64_t)B * H * W * 3 * (in_f32 ? 4 : 1);
  static const int use_mfma = fx_env_int("FX_STEM_MFMA", 1);
  if (use_mfma && x_bytes < 0xFFFFFFF0ll) {
    // dword loads are range-checked as a whole: round the record count up so that the last (partial) dword of a uint8 image whose
    // byte count is not a multiple of 4 is still returned (the bytes past the end belong to masked taps; allocations are 512-byte padded)
    const unsigned x_records = (unsigned)((x_bytes + 3) & ~3ll);
    const int tiles = (int)((total + 31) / 32);
    const int tpw = 8;                                  // 256 pixels per wave: amortises the A-fragment build
    const int blocks = (tiles + 4 * tpw - 1) / (4 * tpw);
#define STEM_MFMA(T, R)                                                                                                                       \
  hipLaunchKernelGGL((stem_mfma_kernel<T, R>), dim3(blocks), dim3(256), 0, stream, (const T*)x, x_records, w, bias, mean, inv_std, (bf16_t*)y, B, H, \
                     W, Ho, Wo, tpw)
    if (in_f32) {
      if (relu) STEM_MFMA(float, true); else STEM_MFMA(float, false);
    } else {
      if (relu) STEM_MFMA(uint8_t, true); else STEM_MFMA(uint8_t, false);
    }
#undef STEM_MFMA
    return fx_launch_status();
  }
  if (in_f32) {
    if (relu) STEM_GO(float, true); else STEM_GO(float, false);
  } else {
    if (relu) STEM_GO(uint8_t, true); else STEM_GO(uint8_t, false);
  }
#undef STEM_GO
  return fx_launch_status();
}

extern "C" int fx_stem_conv3x3s2(const void* x, int in_f32, const float* w, const float* bias, const float* mean, const float* inv_std,
                                 void* y, int B, int H, int W, int Cout, fx_stream_t stream_) {
  return stem_launch(x, in_f32, w, bias, mean, inv_std, y, B, H, W, Cout, 1, stream_);
}

// The same convolution without the ReLU: the pre-BatchNorm tensor of the training path with batch statistics.
extern "C" int fx_stem_conv3x3s2_linear(const void* x, int in_f32, const float* w, const float* bias, const float* mean, const float* inv_std,
                                        void* y, int B, int H, int W, int Cout, fx_stream_t stream_) {
  return stem_launch(x, in_f32, w, bias, mean, inv_std, y, B, H, W, Cout, 0, stream_);
}

// ------------------------------------------------------------------------------------------------
// Bilinear source index exactly as ATen's upsample_bilinear2d (align_corners=False):
//   src = scale * (dst + 0.5) - 0.5, clamped at 0; scale = in / out.
__device__ __forceinline__ void bilinear_src(int dst, float scale, int in_size, int& i0, int& i1, float& l0, float& l1) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  src = src < 0.0f ? 0.0f : src;
  i0 = (int)src;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.0f - l1;
}

__global__ __launch_bounds__(256) void resize_u8_kernel(const uint8_t* __restrict__ x, int H, int W, float* __restrict__ y, int Ho,
                                                         int Wo, float sh, float sw) {
  int total = Ho * Wo * 3;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    int c = i % 3, p = i / 3;
    int wo = p % Wo, ho = p / Wo;
    int h0, h1, w0, w1;
    float lh0, lh1, lw0, lw1;
    bilinear_src(ho, sh, H, h0, h1, lh0, lh1);
    bilinear_src(wo, sw, W, w0, w1, lw0, lw1);
    float p00 = x[((int64_t)h0 * W + w0) * 3 + c], p01 = x[((int64_t)h0 * W + w1) * 3 + c];
    float p10 = x[((int64_t)h1 * W + w0) * 3 + c], p11 = x[((int64_t)h1 * W + w1) * 3 + c];
    y[i] = lh0 * (lw0 * p00 + lw1 * p01) + lh1 * (lw0 * p10 + lw1 * p11);
  }
}

extern "C" int fx_resize_bilinear_u8(const uint8_t* x, int H, int W, float* y, int Ho, int Wo, fx_stream_t stream_) {
  FX_CHECK_ARG(x && y && H > 0 && W > 0 && Ho > 0 && Wo > 0);
  int total = Ho * Wo * 3;
  int grid = (total + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(resize_u8_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), x, H, W, y, Ho, Wo,
                     (float)H / (float)Ho, (float)W / (float)Wo);
  return fx_launch_status();
}

// One workgroup = 256 consecutive (column, 8-channel vector) slots of ONE output row: the row (image, ho) comes from the block index with two
// 32-bit divisions per WORKGROUP, a thread's column and vector from one shift (C8 a power of two) or one 32-bit division.  (Until round 5 every
// thread decomposed a flat 64-bit index with three 64-bit divisions - ~300 VALU instructions in front of four loads and a store: the 40 -> 80
// up-sampling of the hybrid encoder took 28.8 us for 65 MB.)
__global__ __launch_bounds__(256) void resize_nhwc_kernel(const bf16_t* __restrict__ x, int ldx, bf16_t* __restrict__ y, int ldy, int B,
                                                           int H, int W, int C8, int c8_shift, int bpr, int Ho, int Wo, float sh, float sw) {
  // XCD-aware block order (round 5): consecutive blocks are dealt round-robin to the 8 XCDs, each with its own L2, and vertically adjacent
  // output rows share an input row - in launch order every XCD fetched its own copy of it (PMC: 318 MB read for a 210 MB tensor).  With the
  // remap an XCD owns a contiguous band of output rows and the shared rows are re-fetched only at the eight band seams.
  const unsigned bid = (unsigned)fx_xcd_remap(blockIdx.x, gridDim.x);
  const unsigned row = bid / (unsigned)bpr;                       // (image, output row): uniform
  const unsigned t = (bid - row * (unsigned)bpr) * 256u + threadIdx.x;
  if (t >= (unsigned)(Wo * C8)) return;
  const unsigned b = row / (unsigned)Ho, ho = row - b * (unsigned)Ho;
  const unsigned wo = c8_shift >= 0 ? t >> c8_shift : t / (unsigned)C8;
  const unsigned c8 = t - wo * (unsigned)C8;
  int h0, h1, w0, w1;
  float lh0, lh1, lw0, lw1;
  bilinear_src((int)ho, sh, H, h0, h1, lh0, lh1);
  bilinear_src((int)wo, sw, W, w0, w1, lw0, lw1);
  const bf16_t* base = x + (int64_t)b * H * W * ldx + c8 * 8;
  float f00[8], f01[8], f10[8], f11[8], o[8];
  unpack_bf16x8(*reinterpret_cast<const uint4*>(base + ((int64_t)h0 * W + w0) * ldx), f00);
  unpack_bf16x8(*reinterpret_cast<const uint4*>(base + ((int64_t)h0 * W + w1) * ldx), f01);
  unpack_bf16x8(*reinterpret_cast<const uint4*>(base + ((int64_t)h1 * W + w0) * ldx), f10);
  unpack_bf16x8(*reinterpret_cast<const uint4*>(base + ((int64_t)h1 * W + w1) * ldx), f11);
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = lh0 * (lw0 * f00[j] + lw1 * f01[j]) + lh1 * (lw0 * f10[j] + lw1 * f11[j]);
  *reinterpret_cast<uint4*>(y + ((int64_t)row * Wo + wo) * ldy + c8 * 8) = pack_bf16x8(o);
}

extern "C" int fx_resize_bilinear_nhwc_bf16(const void* x, int ldx, void* y, int ldy, int B, int H, int W, int C, int Ho, int Wo,
                                            fx_stream_t stream_) {
  FX_CHECK_ARG(x && y && B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 8 == 0);
  FX_CHECK_ARG(ldx >= C && ldy >= C && ldx % 8 == 0 && ldy % 8 == 0);
  const int C8 = C / 8;
  const int64_t per_row = (int64_t)Wo * C8;
  const int64_t bpr = (per_row + 255) / 256;
  const int64_t grid = (int64_t)B * Ho * bpr;
  if (per_row >= (1ll << 31) || grid >= (1ll << 31)) return FX_ERR_UNSUPPORTED;
  const int shift = (C8 & (C8 - 1)) == 0 ? __builtin_ctz((unsigned)C8) : -1;
  hipLaunchKernelGGL(resize_nhwc_kernel, dim3((unsigned)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), (const bf16_t*)x, ldx,
                     (bf16_t*)y, ldy, B, H, W, C8, shift, (int)bpr, Ho, Wo, (float)H / (float)Ho, (float)W / (float)Wo);
  return fx_launch_status();
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_kernel(const bf16_t* __restrict__ x, int ldx, bf16_t* __restrict__ y, int ldy, int B, int H,
                                                       int W, int C8, int Ho, int Wo) {
  int64_t total = (int64_t)B * Ho * Wo * C8;
  // XCD-aware block order (round 5): consecutive blocks are dealt round-robin to the 8 XCDs, each with its own L2, and vertically adjacent
  // output rows share an input row - in launch order every XCD fetched its own copy of it (PMC: 318 MB read for a 210 MB tensor).  With the
  // remap an XCD owns a contiguous band of output rows and the shared rows are re-fetched only at the eight band seams.
  const int bid = fx_xcd_remap(blockIdx.x, gridDim.x);
  for (int64_t i = (int64_t)bid * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int c8 = (int)(i % C8);
    int64_t p = i / C8;
    int wo = (int)(p % Wo);
    int64_t q = p / Wo;
    int ho = (int)(q % Ho), b = (int)(q / Ho);
    float m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      int hi = ho * 2 - 1 + kh;
      if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        int wi = wo * 2 - 1 + kw;
        if ((unsigned)wi >= (unsigned)W) continue;
        float f[8];
        unpack_bf16x8(*reinterpret_cast<const uint4*>(x + ((int64_t)(b * H + hi) * W + wi) * ldx + c8 * 8), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], f[j]);
      }
    }
    *reinterpret_cast<uint4*>(y + p * ldy + c8 * 8) = pack_bf16x8(m);
  }
}

extern "C" int fx_maxpool3x3s2_nhwc_bf16(const void* x, int ldx, void* y, int ldy, int B, int H, int W, int C, fx_stream_t stream_) {
  FX_CHECK_ARG(x && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && ldx >= C && ldy >= C && ldx % 8 == 0 && ldy % 8 == 0);
  int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  int64_t total = (int64_t)B * Ho * Wo * (C / 8);
  int64_t grid = (total + 255) / 256;
  if (grid > 256 * 256) grid = 256 * 256;   // one pass for every registry shape (the band order of the remap needs it)
  hipLaunchKernelGGL(maxpool_kernel, dim3((int)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), (const bf16_t*)x, ldx,
                     (bf16_t*)y, ldy, B, H, W, C / 8, Ho, Wo);
  return fx_launch_status();
}

// ------------------------------------------------------------------------------------------------
// nn.AvgPool2d(2, 2, 0, ceil_mode=True) of the ResNet-vd "d" shortcut (resnet.py:89-100): average over the in-bounds
// taps (ceil mode with pad 0 divides by the number of valid elements).  Done ONCE here instead of inside the 1x1
// shortcut conv's A-load, where it was recomputed for every N tile (up to 16x).
__global__ __launch_bounds__(256) void avgpool2_kernel(const bf16_t* __restrict__ x, int ldx, bf16_t* __restrict__ y, int ldy, int B, int H,
                                                        int W, int C8, int Ho, int Wo) {
  int64_t total = (int64_t)B * Ho * Wo * C8;
  // XCD-aware block order (round 5): consecutive blocks are dealt round-robin to the 8 XCDs, each with its own L2, and vertically adjacent
  // output rows share an input row - in launch order every XCD fetched its own copy of it (PMC: 318 MB read for a 210 MB tensor).  With the
  // remap an XCD owns a contiguous band of output rows and the shared rows are re-fetched only at the eight band seams.
  const int bid = fx_xcd_remap(blockIdx.x, gridDim.x);
  for (int64_t i = (int64_t)bid * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int c8 = (int)(i % C8);
    int64_t p = i / C8;
    int wo = (int)(p % Wo);
    int64_t q = p / Wo;
    int ho = (int)(q % Ho), b = (int)(q / Ho);
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int cnt = 0;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        int hi = 2 * ho + dy, wi = 2 * wo + dx;
        if (hi < H && wi < W) {
          float f[8];
          unpack_bf16x8(*reinterpret_cast<const uint4*>(x + ((int64_t)(b * H + hi) * W + wi) * ldx + c8 * 8), f);
#pragma unroll
          for (int j = 0; j < 8; ++j) s[j] += f[j];
          ++cnt;
        }
      }
    float inv = 1.0f / (float)cnt;
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] *= inv;
    *reinterpret_cast<uint4*>(y + p * ldy + c8 * 8) = pack_bf16x8(s);
  }
}

extern "C" int fx_avgpool2x2_nhwc_bf16(const void* x, int ldx, void* y, int ldy, int B, int H, int W, int C, fx_stream_t stream_) {
  FX_CHECK_ARG(x && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && ldx >= C && ldy >= C && ldx % 8 == 0 && ldy % 8 == 0);
  int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  int64_t total = (int64_t)B * Ho * Wo * (C / 8);
  int64_t grid = (total + 255) / 256;
  if (grid > 256 * 32) grid = 256 * 32;
  hipLaunchKernelGGL(avgpool2_kernel, dim3((int)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), (const bf16_t*)x, ldx,
                     (bf16_t*)y, ldy, B, H, W, C / 8, Ho, Wo);
  return fx_launch_status();
}
