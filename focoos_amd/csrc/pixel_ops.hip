// Bandwidth-bound pixel kernels of the RT-DETR path (gfx950): stem conv with fused normalisation,
// bilinear resizes, 3x3/s2 max-pool.  All are coalesced-HBM wavefront kernels: 8 bf16 channels
// (16 bytes) per lane, NHWC, no LDS staging (no reuse beyond what L2 already gives).
#include "common.h"

// ------------------------------------------------------------------------------------------------
// Stem: (x-mean)*inv_std  ->  3x3/s2/p1 conv (3 -> 32)  ->  +bias (BN folded)  ->  ReLU  -> bf16 NHWC.
// One lane = one output pixel x 32 channels; the 27x32 fp32 weights live in LDS and are read as
// broadcast float4 (all lanes same address).  Zero padding applies to the NORMALISED image
// (reference pads after normalising), i.e. padded taps contribute nothing.
template <typename TIn>
__global__ __launch_bounds__(256) void stem_conv_kernel(const TIn* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, const float* __restrict__ mean,
                                                         const float* __restrict__ inv_std, bf16_t* __restrict__ y, int B, int H,
                                                         int W, int Ho, int Wo) {
  __shared__ __attribute__((aligned(16))) float ws[27 * 32];  // [tap(kh,kw,c)][n]
  __shared__ float bs[32];
  for (int i = threadIdx.x; i < 27 * 32; i += 256) {
    int n = i & 31, tap = i >> 5;  // w is [n][kh][kw][c] = [n][tap]
    ws[i] = w[n * 27 + tap];
  }
  if (threadIdx.x < 32) bs[threadIdx.x] = bias[threadIdx.x];
  __syncthreads();
  const int total = B * Ho * Wo;
  float mu[3] = {mean[0], mean[1], mean[2]}, is[3] = {inv_std[0], inv_std[1], inv_std[2]};
  for (int o = blockIdx.x * 256 + threadIdx.x; o < total; o += gridDim.x * 256) {
    int b = o / (Ho * Wo), rem = o - b * Ho * Wo;
    int ho = rem / Wo, wo = rem - ho * Wo;
    float acc[32];
#pragma unroll
    for (int n = 0; n < 32; ++n) acc[n] = bs[n];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      int hi = ho * 2 - 1 + kh;
      if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        int wi = wo * 2 - 1 + kw;
        if ((unsigned)wi >= (unsigned)W) continue;
        const TIn* px = x + ((int64_t)(b * H + hi) * W + wi) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float v = ((float)px[c] - mu[c]) * is[c];
          const float4* wr = reinterpret_cast<const float4*>(ws + ((kh * 3 + kw) * 3 + c) * 32);
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            float4 wv = wr[q];
            acc[4 * q + 0] += v * wv.x;
            acc[4 * q + 1] += v * wv.y;
            acc[4 * q + 2] += v * wv.z;
            acc[4 * q + 3] += v * wv.w;
          }
        }
      }
    }
#pragma unroll
    for (int n = 0; n < 32; ++n) acc[n] = fmaxf(acc[n], 0.0f);
    uint4* dst = reinterpret_cast<uint4*>(y + (int64_t)o * 32);
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = pack_bf16x8(acc + 8 * q);
  }
}

extern "C" int fx_stem_conv3x3s2(const void* x, int in_f32, const float* w, const float* bias, const float* mean, const float* inv_std,
                                 void* y, int B, int H, int W, int Cout, fx_stream_t stream_) {
  FX_CHECK_ARG(x && w && bias && mean && inv_std && y && B > 0 && H > 0 && W > 0);
  if (Cout != 32) return FX_ERR_UNSUPPORTED;
  int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  int64_t total = (int64_t)B * Ho * Wo;
  if (total >= (1ll << 31)) return FX_ERR_UNSUPPORTED;
  int grid = (int)((total + 255) / 256);
  if (grid > 256 * 16) grid = 256 * 16;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (in_f32)
    hipLaunchKernelGGL(stem_conv_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)x, w, bias, mean, inv_std, (bf16_t*)y, B,
                       H, W, Ho, Wo);
  else
    hipLaunchKernelGGL(stem_conv_kernel<uint8_t>, dim3(grid), dim3(256), 0, stream, (const uint8_t*)x, w, bias, mean, inv_std,
                       (bf16_t*)y, B, H, W, Ho, Wo);
  return fx_launch_status();
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

__global__ __launch_bounds__(256) void resize_nhwc_kernel(const bf16_t* __restrict__ x, int ldx, bf16_t* __restrict__ y, int ldy, int B,
                                                           int H, int W, int C8, int Ho, int Wo, float sh, float sw) {
  int64_t total = (int64_t)B * Ho * Wo * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int c8 = (int)(i % C8);
    int64_t p = i / C8;
    int wo = (int)(p % Wo);
    int64_t q = p / Wo;
    int ho = (int)(q % Ho), b = (int)(q / Ho);
    int h0, h1, w0, w1;
    float lh0, lh1, lw0, lw1;
    bilinear_src(ho, sh, H, h0, h1, lh0, lh1);
    bilinear_src(wo, sw, W, w0, w1, lw0, lw1);
    const bf16_t* base = x + (int64_t)b * H * W * ldx + c8 * 8;
    float f00[8], f01[8], f10[8], f11[8], o[8];
    unpack_bf16x8(*reinterpret_cast<const uint4*>(base + ((int64_t)h0 * W + w0) * ldx), f00);
    unpack_bf16x8(*reinterpret_cast<const uint4*>(base + ((int64_t)h0 * W + w1) * ldx), f01);
    unpack_bf16x8(*reinterpret_cast<const uint4*>(base + ((int64_t)h1 * W + w0) * ldx), f10);
    unpack_bf16x8(*reinterpret_cast<const uint4*>(base + ((int64_t)h1 * W + w1) * ldx), f11);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = lh0 * (lw0 * f00[j] + lw1 * f01[j]) + lh1 * (lw0 * f10[j] + lw1 * f11[j]);
    *reinterpret_cast<uint4*>(y + p * ldy + c8 * 8) = pack_bf16x8(o);
  }
}

extern "C" int fx_resize_bilinear_nhwc_bf16(const void* x, int ldx, void* y, int ldy, int B, int H, int W, int C, int Ho, int Wo,
                                            fx_stream_t stream_) {
  FX_CHECK_ARG(x && y && B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 8 == 0);
  FX_CHECK_ARG(ldx >= C && ldy >= C && ldx % 8 == 0 && ldy % 8 == 0);
  int64_t total = (int64_t)B * Ho * Wo * (C / 8);
  int64_t grid = (total + 255) / 256;
  if (grid > 256 * 32) grid = 256 * 32;
  hipLaunchKernelGGL(resize_nhwc_kernel, dim3((int)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), (const bf16_t*)x, ldx,
                     (bf16_t*)y, ldy, B, H, W, C / 8, Ho, Wo, (float)H / (float)Ho, (float)W / (float)Wo);
  return fx_launch_status();
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_kernel(const bf16_t* __restrict__ x, int ldx, bf16_t* __restrict__ y, int ldy, int B, int H,
                                                       int W, int C8, int Ho, int Wo) {
  int64_t total = (int64_t)B * Ho * Wo * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
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
  if (grid > 256 * 32) grid = 256 * 32;
  hipLaunchKernelGGL(maxpool_kernel, dim3((int)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), (const bf16_t*)x, ldx,
                     (bf16_t*)y, ldy, B, H, W, C / 8, Ho, Wo);
  return fx_launch_status();
}
