// Bandwidth-bound pixel kernels of the RT-DETR path (gfx950): stem conv with fused normalisation,
// bilinear resizes, 3x3/s2 max-pool.  All are coalesced-HBM wavefront kernels: 8 bf16 channels
// (16 bytes) per lane, NHWC, no LDS staging (no reuse beyond what L2 already gives).
#include "common.h"

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

// ------------------------------------------------------------------------------------------------
// nn.AvgPool2d(2, 2, 0, ceil_mode=True) of the ResNet-vd "d" shortcut (resnet.py:89-100): average over the in-bounds
// taps (ceil mode with pad 0 divides by the number of valid elements).  Done ONCE here instead of inside the 1x1
// shortcut conv's A-load, where it was recomputed for every N tile (up to 16x).
__global__ __launch_bounds__(256) void avgpool2_kernel(const bf16_t* __restrict__ x, int ldx, bf16_t* __restrict__ y, int ldy, int B, int H,
                                                        int W, int C8, int Ho, int Wo) {
  int64_t total = (int64_t)B * Ho * Wo * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
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
