// Backward kernels of the BiSeNetFormer-specific layers (training direction):
//   CatBottleneck stride-2 branch: depthwise 3x3 s2 conv + AvgPool2d(3,2,1) skip      focoos/nn/backbone/stdc.py:120-166
//   AttentionRefinementModule / FeatureFusionModule gates: feat.mean((2,3)), feat * atten   bisenetformer/modelling.py:159-167, 226-237
//   mask einsum "bqc,bchw->bqhw" gradient layout change                                bisenetformer/modelling.py:84
// All bandwidth kernels: 8 channels (16 B) per lane, fp32 accumulation, fixed-order LDS trees before the final atomics.
#include "common.h"

// ------------------------------------------------------------------------------------------------
// dx[b,hi,wi,c] = sum over the taps (kh,kw) whose output position (ho,wo) = ((hi+1-kh)/2, (wi+1-kw)/2) is integral and in range
//               of w[kh*3+kw][c] * dy[b,ho,wo,c]       (transpose of fx_dwconv3x3s2_nhwc_bf16; AvgPool2d(3,2,1): w = 1/9)
__global__ __launch_bounds__(256) void dwconv3x3s2_dgrad_kernel(const bf16_t* __restrict__ dy, int lddy, const float* __restrict__ w,
                                                                bf16_t* __restrict__ dx, int lddx, int B, int H, int W, int Ho, int Wo, int C8) {
  const int64_t total = (int64_t)B * H * W * C8;
  const int C = C8 * 8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    int64_t p = i / C8;
    const int wi = (int)(p % W);
    p /= W;
    const int hi = (int)(p % H);
    const int b = (int)(p / H);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int th = hi + 1 - kh;
      if (th < 0 || (th & 1) || (th >> 1) >= Ho) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int tw = wi + 1 - kw;
        if (tw < 0 || (tw & 1) || (tw >> 1) >= Wo) continue;
        float v[8];
        unpack_bf16x8(*reinterpret_cast<const uint4*>(dy + (((int64_t)b * Ho + (th >> 1)) * Wo + (tw >> 1)) * lddy + c8 * 8), v);
        const float* wr = w + (kh * 3 + kw) * C + c8 * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaf(v[j], wr[j], acc[j]);
      }
    }
    *reinterpret_cast<uint4*>(dx + (((int64_t)b * H + hi) * W + wi) * lddx + c8 * 8) = pack_bf16x8(acc);
  }
}

// dw[kh*3+kw][c] += sum_{b,ho,wo} dy[b,ho,wo,c] * x[b,2ho-1+kh,2wo-1+kw,c].  Workgroup = 64 channels x a range of output pixels
// (8 channel lanes across, 32 pixel lanes down), LDS tree per tap, one fp32 atomic per (tap, channel, workgroup).
__global__ __launch_bounds__(256) void dwconv3x3s2_wgrad_kernel(const bf16_t* __restrict__ dy, int lddy, const bf16_t* __restrict__ x, int ldx,
                                                                float* __restrict__ dw, int B, int H, int W, int Ho, int Wo, int C) {
  __shared__ float part[32][65];
  const int cg = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int c0 = blockIdx.x * 64 + cg * 8;
  const int64_t P = (int64_t)B * Ho * Wo;
  const int64_t per = (P + gridDim.y - 1) / gridDim.y;
  const int64_t p_lo = (int64_t)blockIdx.y * per, p_hi = p_lo + per < P ? p_lo + per : P;
  float acc[9][8];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[t][j] = 0.0f;
  if (c0 < C) {
    for (int64_t p = p_lo + pl; p < p_hi; p += 32) {
      const int wo = (int)(p % Wo);
      const int64_t q = p / Wo;
      const int ho = (int)(q % Ho), b = (int)(q / Ho);
      float g[8];
      unpack_bf16x8(*reinterpret_cast<const uint4*>(dy + p * lddy + c0), g);
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int hi = 2 * ho - 1 + kh;
        if (hi < 0 || hi >= H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int wi = 2 * wo - 1 + kw;
          if (wi < 0 || wi >= W) continue;
          float v[8];
          unpack_bf16x8(*reinterpret_cast<const uint4*>(x + (((int64_t)b * H + hi) * W + wi) * ldx + c0), v);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[kh * 3 + kw][j] = fmaf(g[j], v[j], acc[kh * 3 + kw][j]);
        }
      }
    }
  }
#pragma unroll 1
  for (int t = 0; t < 9; ++t) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) part[pl][cg * 8 + j] = acc[t][j];
    __syncthreads();
    if (threadIdx.x < 64) {
      float s = 0.0f;
#pragma unroll
      for (int i = 0; i < 32; ++i) s += part[i][threadIdx.x];
      const int c = blockIdx.x * 64 + threadIdx.x;
      if (c < C) unsafeAtomicAdd(dw + (int64_t)t * C + c, s);
    }
  }
}

extern "C" int fx_dwconv3x3s2_bwd_nhwc_bf16(const void* dy, int lddy, const void* x, int ldx, const float* w, void* dx, int lddx, float* dw, int B,
                                            int H, int W, int C, fx_stream_t stream_) {
  FX_CHECK_ARG(dy && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && lddy >= C && lddy % 8 == 0 && ((uintptr_t)dy % 16) == 0);
  FX_CHECK_ARG((dx == nullptr || (w && lddx >= C && lddx % 8 == 0 && ((uintptr_t)dx % 16) == 0)));
  FX_CHECK_ARG((dw == nullptr || (x && ldx >= C && ldx % 8 == 0 && ((uintptr_t)x % 16) == 0)));
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  if (dx) {
    const int64_t total = (int64_t)B * H * W * (C / 8);
    int64_t grid = (total + 255) / 256;
    if (grid > 256 * 32) grid = 256 * 32;
    hipLaunchKernelGGL(dwconv3x3s2_dgrad_kernel, dim3((int)grid), dim3(256), 0, stream, (const bf16_t*)dy, lddy, w, (bf16_t*)dx, lddx, B, H, W, Ho, Wo,
                       C / 8);
  }
  if (dw) {
    const int64_t P = (int64_t)B * Ho * Wo;
    int splits = (int)((P + 511) / 512);   // >= 16 pixels per pixel lane and workgroup; the reduction tail is 9 atomics per channel and workgroup
    const int cgs = (C + 63) / 64;
    if (splits * cgs > 1024) splits = 1024 / cgs;
    if (splits < 1) splits = 1;
    hipLaunchKernelGGL(dwconv3x3s2_wgrad_kernel, dim3(cgs, splits), dim3(256), 0, stream, (const bf16_t*)dy, lddy, (const bf16_t*)x, ldx, dw, B, H, W, Ho,
                       Wo, C);
  }
  return fx_launch_status();
}

// ------------------------------------------------------------------------------------------------
// out[b][c] (+)= scale * sum_p a[b,p,c] * (b_ ? b_[b,p,c] : 1): the gate gradient sum_p dy * feat (d (feat * atten) / d atten) and the
// broadcast-add gradient sum_p dy.  f32 output; pixel ranges split over blockIdx.z with one atomic per (channel, workgroup) when
// there is more than one range (out zero-initialised by the caller in that case).
__global__ __launch_bounds__(256) void rowdot_kernel(const bf16_t* __restrict__ a, int lda, const bf16_t* __restrict__ b_, int ldb, float scale,
                                                     float* __restrict__ out, int ldo, int P, int C) {
  __shared__ float part[32][65];
  const int b = blockIdx.y, cg = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int c0 = blockIdx.x * 64 + cg * 8;
  const int per = (P + gridDim.z - 1) / gridDim.z;
  const int p_lo = blockIdx.z * per, p_hi = p_lo + per < P ? p_lo + per : P;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c0 < C) {
    const bf16_t* ab = a + (int64_t)b * P * lda + c0;
    const bf16_t* bb = b_ ? b_ + (int64_t)b * P * ldb + c0 : nullptr;
    for (int p = p_lo + pl; p < p_hi; p += 32) {
      float v[8];
      unpack_bf16x8(*reinterpret_cast<const uint4*>(ab + (int64_t)p * lda), v);
      if (bb) {
        float u[8];
        unpack_bf16x8(*reinterpret_cast<const uint4*>(bb + (int64_t)p * ldb), u);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaf(v[j], u[j], acc[j]);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += v[j];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) part[pl][cg * 8 + j] = acc[j];
  __syncthreads();
  if (threadIdx.x < 64) {
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += part[i][threadIdx.x];
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c < C) {
      if (gridDim.z == 1) out[(int64_t)b * ldo + c] = s * scale;
      else unsafeAtomicAdd(out + (int64_t)b * ldo + c, s * scale);
    }
  }
}

extern "C" int fx_rowdot_nhwc_bf16(const void* a, int lda, const void* b, int ldb, float scale, float* out, int ldo, int B, int P, int C, int splits,
                                   fx_stream_t stream_) {
  FX_CHECK_ARG(a && out && B > 0 && P > 0 && C > 0 && C % 8 == 0 && lda >= C && lda % 8 == 0 && ldo >= C && ((uintptr_t)a % 16) == 0);
  FX_CHECK_ARG(b == nullptr || (ldb >= C && ldb % 8 == 0 && ((uintptr_t)b % 16) == 0));
  FX_CHECK_ARG(splits >= 1 && splits <= 1024);
  hipLaunchKernelGGL(rowdot_kernel, dim3((C + 63) / 64, B, splits), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), (const bf16_t*)a, lda,
                     (const bf16_t*)b, ldb, scale, out, ldo, P, C);
  return fx_launch_status();
}

// y[b,p,c] = scale * vec[b][c]  (gradient of feat.mean((2,3)) towards feat: scale = 1/P)
__global__ __launch_bounds__(256) void bcast_vec_kernel(const float* __restrict__ vec, int ldv, float scale, bf16_t* __restrict__ y, int ldy, int P,
                                                        int C8, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    const int64_t row = i / C8;
    const int b = (int)(row / P);
    const float* vp = vec + (int64_t)b * ldv + c8 * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = vp[j] * scale;
    *reinterpret_cast<uint4*>(y + row * ldy + c8 * 8) = pack_bf16x8(v);
  }
}

extern "C" int fx_bcast_vec_nhwc_bf16(const float* vec, int ldv, float scale, void* y, int ldy, int B, int P, int C, fx_stream_t stream_) {
  FX_CHECK_ARG(vec && y && B > 0 && P > 0 && C > 0 && C % 8 == 0 && ldv >= C && ldy >= C && ldy % 8 == 0 && ((uintptr_t)y % 16) == 0);
  const int64_t total = (int64_t)B * P * (C / 8);
  int64_t grid = (total + 255) / 256;
  if (grid > 256 * 32) grid = 256 * 32;
  hipLaunchKernelGGL(bcast_vec_kernel, dim3((int)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), vec, ldv, scale, (bf16_t*)y, ldy, P, C / 8,
                     total);
  return fx_launch_status();
}

// ------------------------------------------------------------------------------------------------
// rows[b][p][q] = bf16(planes[b][q][p]) for q < Q, 0 for Q <= q < Qp: the [B,Q,h,w] fp32 mask-logit gradient (what the point-sampled
// criterion scatters into) as pixel-major bf16 rows, the operand layout of the two GEMMs of the einsum backward
// (d embed = dM x F: weight-gradient kernel; d F = dM^T x embed: 1x1 conv kernel).  64-pixel x Qp tiles through LDS.
__global__ __launch_bounds__(256) void planes_to_rows_kernel(const float* __restrict__ planes, int Q, int P, bf16_t* __restrict__ rows, int ldr, int Qp) {
  extern __shared__ float tile[];   // [Qp][65]
  const int b = blockIdx.y, p0 = blockIdx.x * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* src = planes + (int64_t)b * Q * P;
  for (int q = wave; q < Qp; q += 4) tile[q * 65 + lane] = (q < Q && p0 + lane < P) ? src[(int64_t)q * P + p0 + lane] : 0.0f;
  __syncthreads();
  const int Q8 = Qp / 8;
  for (int i = threadIdx.x; i < 64 * Q8; i += 256) {
    const int p = i / Q8, q8 = i % Q8;
    if (p0 + p >= P) continue;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = tile[(q8 * 8 + j) * 65 + p];
    *reinterpret_cast<uint4*>(rows + ((int64_t)b * P + p0 + p) * ldr + q8 * 8) = pack_bf16x8(v);
  }
}

extern "C" int fx_planes_to_rows_bf16(const float* planes, int Q, int P, void* rows, int ld_rows, int Qp, int B, fx_stream_t stream_) {
  FX_CHECK_ARG(planes && rows && B > 0 && Q > 0 && P > 0 && Qp >= Q && Qp % 8 == 0 && Qp <= 240 && ld_rows >= Qp && ld_rows % 8 == 0);
  FX_CHECK_ARG(((uintptr_t)rows % 16) == 0);
  hipLaunchKernelGGL(planes_to_rows_kernel, dim3((P + 63) / 64, B), dim3(256), (size_t)Qp * 65 * sizeof(float), reinterpret_cast<hipStream_t>(stream_),
                     planes, Q, P, (bf16_t*)rows, ld_rows, Qp);
  return fx_launch_status();
}
