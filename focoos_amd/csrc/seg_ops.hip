// BiSeNetFormer-specific kernels of the segmentation path (bisenetformer-l-ade) for gfx950:
//   * depthwise 3x3 stride-2 convolution (+ folded BatchNorm) / AvgPool2d(3, 2, 1) of the STDC CatBottleneck
//     (focoos/nn/backbone/stdc.py:114-128, 146-166) - one HBM-bound kernel, 8 channels (16 B) per lane;
//   * global average pooling, the pooled 1x1 convolutions and the channel gate of AttentionRefinementModule /
//     ContextPath / FeatureFusionModule (focoos/models/bisenetformer/modelling.py:149-237);
//   * the `predict_all_pixels` branch of BisenetFormerProcessor.postprocess (bisenetformer/processor.py:215-229): every image
//     pixel is assigned to the query maximising class score x upsampled mask probability - fused with the x8 bilinear
//     upsample, so neither the [B,Q,H,W] probability tensor nor the [B,Q,H,W] boolean tensor is materialised.
#include "common.h"
int fx_tune(const char* env_name, int default_value);   // runtime.hip

// ------------------------------------------------------------------------------------------------
// y[b,ho,wo,c] = bias[c] + sum_{kh,kw} w[kh*3+kw][c] * x[b, 2ho-1+kh, 2wo-1+kw, c]   (zero padding; AvgPool2d's default
// count_include_pad=True is the same sum with w = 1/9).  x bf16 NHWC (row stride ldx), w f32 [9][C], bias f32 [C] or NULL.
template <typename OT>
__global__ __launch_bounds__(256) void dwconv3x3s2_kernel(const bf16_t* __restrict__ x, int ldx, const float* __restrict__ w,
                                                          const float* __restrict__ bias, OT* __restrict__ y, int ldy, int B, int H, int W,
                                                          int Ho, int Wo, int C8) {
  const int64_t total = (int64_t)B * Ho * Wo * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    int64_t p = i / C8;
    const int wo = (int)(p % Wo);
    p /= Wo;
    const int ho = (int)(p % Ho);
    const int b = (int)(p / Ho);
    const int C = C8 * 8;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = bias ? bias[c8 * 8 + j] : 0.0f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int hi = 2 * ho - 1 + kh;
      if (hi < 0 || hi >= H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int wi = 2 * wo - 1 + kw;
        if (wi < 0 || wi >= W) continue;
        float v[8];
        unpack_bf16x8(*reinterpret_cast<const uint4*>(x + (((int64_t)b * H + hi) * W + wi) * ldx + c8 * 8), v);
        const float* wr = w + (kh * 3 + kw) * C + c8 * 8;
        const float4 w0 = *reinterpret_cast<const float4*>(wr), w1 = *reinterpret_cast<const float4*>(wr + 4);
        acc[0] = fmaf(v[0], w0.x, acc[0]); acc[1] = fmaf(v[1], w0.y, acc[1]); acc[2] = fmaf(v[2], w0.z, acc[2]); acc[3] = fmaf(v[3], w0.w, acc[3]);
        acc[4] = fmaf(v[4], w1.x, acc[4]); acc[5] = fmaf(v[5], w1.y, acc[5]); acc[6] = fmaf(v[6], w1.z, acc[6]); acc[7] = fmaf(v[7], w1.w, acc[7]);
      }
    }
    OT* yp = y + (((int64_t)b * Ho + ho) * Wo + wo) * ldy + c8 * 8;
    if constexpr (sizeof(OT) == 4) {
      *reinterpret_cast<float4*>(yp) = make_float4(acc[0], acc[1], acc[2], acc[3]);
      *reinterpret_cast<float4*>(yp + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    } else {
      *reinterpret_cast<uint4*>(yp) = pack_bf16x8(acc);
    }
  }
}

extern "C" int fx_dwconv3x3s2_nhwc_bf16(const void* x, int ldx, const float* w, const float* bias, void* y, int ldy, int B, int H, int W, int C,
                                        fx_stream_t stream_) {
  FX_CHECK_ARG(x && w && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && ldx >= C && ldy >= C && ldx % 8 == 0 && ldy % 8 == 0);
  FX_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)w % 16) == 0);
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int64_t total = (int64_t)B * Ho * Wo * (C / 8);
  int64_t grid = (total + 255) / 256;
  if (grid > 256 * 32) grid = 256 * 32;
  hipLaunchKernelGGL(dwconv3x3s2_kernel<bf16_t>, dim3((int)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), (const bf16_t*)x, ldx, w, bias,
                     (bf16_t*)y, ldy, B, H, W, Ho, Wo, C / 8);
  return fx_launch_status();
}

extern "C" int fx_dwconv3x3s2_nhwc_f32out(const void* x, int ldx, const float* w, const float* bias, float* y, int ldy, int B, int H, int W, int C,
                                          fx_stream_t stream_) {
  FX_CHECK_ARG(x && w && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && ldx >= C && ldy >= C && ldx % 8 == 0 && ldy % 8 == 0);
  FX_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)w % 16) == 0);
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int64_t total = (int64_t)B * Ho * Wo * (C / 8);
  int64_t grid = (total + 255) / 256;
  if (grid > 256 * 32) grid = 256 * 32;
  hipLaunchKernelGGL(dwconv3x3s2_kernel<float>, dim3((int)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), (const bf16_t*)x, ldx, w, bias, y,
                     ldy, B, H, W, Ho, Wo, C / 8);
  return fx_launch_status();
}

// ------------------------------------------------------------------------------------------------
// mean[b][c] = (1/P) sum_p x[b,p,c]   (feat.mean(dim=(2,3)) / adaptive_avg_pool2d(feat, 1)); f32 output.  One workgroup per
// (image, 64-channel group): 8 lanes x 8 channels across, 128 lanes down the pixels, fixed-order LDS tree -> deterministic.
// 1024 threads per (image, 64-channel group) since round 5: 128 pixel lanes x 8 channel vectors, four independent loads in flight per
// lane - the 256-thread form walked 6 400 pixels with 32 lanes and one load at a time (56 us for the 105 MB of the feature-fusion mean:
// 64 workgroups cannot draw more than 0.9 TB/s that way).  Fixed summation order: lane partials (pixels p, p + 128, ... in order, four
// interleaved accumulators folded ((a0 + a1) + (a2 + a3))), then the 128 lanes in order.
#define GM_THREADS 1024
#define GM_LANES (GM_THREADS / 8)
__global__ __launch_bounds__(GM_THREADS) void global_mean_kernel(const bf16_t* __restrict__ x, int ldx, float* __restrict__ out, int ldo, int P, int C) {
  __shared__ float part[GM_LANES][64];
  const int b = blockIdx.y, cg = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int c0 = blockIdx.x * 64 + cg * 8;
  float acc[4][8];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[u][j] = 0.0f;
  if (c0 < C) {
    const bf16_t* xb = x + (int64_t)b * P * ldx + c0;
    int p = pl;
    for (; p + 3 * GM_LANES < P; p += 4 * GM_LANES) {
      uint4 r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) r[u] = *reinterpret_cast<const uint4*>(xb + (int64_t)(p + u * GM_LANES) * ldx);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float v[8];
        unpack_bf16x8(r[u], v);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[u][j] += v[j];
      }
    }
    for (int u = 0; p < P; p += GM_LANES, ++u) {   // at most three left: they continue the interleaving
      float v[8];
      unpack_bf16x8(*reinterpret_cast<const uint4*>(xb + (int64_t)p * ldx), v);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (u == 0) acc[0][j] += v[j];
        else if (u == 1) acc[1][j] += v[j];
        else acc[2][j] += v[j];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) part[pl][cg * 8 + j] = (acc[0][j] + acc[1][j]) + (acc[2][j] + acc[3][j]);
  __syncthreads();
  if (threadIdx.x < 64) {
    float s = 0.0f;
#pragma unroll 8
    for (int i = 0; i < GM_LANES; ++i) s += part[i][threadIdx.x];
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c < C) out[(int64_t)b * ldo + c] = s / (float)P;
  }
}

extern "C" int fx_global_mean_nhwc_bf16(const void* x, int ldx, float* out, int ldo, int B, int P, int C, fx_stream_t stream_) {
  FX_CHECK_ARG(x && out && B > 0 && P > 0 && C > 0 && C % 8 == 0 && ldx >= C && ldx % 8 == 0 && ldo >= C && ((uintptr_t)x % 16) == 0);
  hipLaunchKernelGGL(global_mean_kernel, dim3((C + 63) / 64, B), dim3(GM_THREADS), 0, reinterpret_cast<hipStream_t>(stream_), (const bf16_t*)x, ldx, out,
                     ldo, P, C);
  return fx_launch_status();
}

// out[b][n] = act(bias[n] + sum_c W[n][c] * in[b][c])  on pooled vectors (the 1x1 convs applied to [B,C,1,1] tensors:
// conv_avg, conv_atten + bn_atten, FFM conv1 / conv2).  f32 throughout (a few MFLOP); one wave per output row n.
// act: FX_ACT_NONE / FX_ACT_RELU / 4 = sigmoid.
#define FX_ACT_SIGMOID_LOCAL 4
__global__ __launch_bounds__(256) void pooled_linear_kernel(const float* __restrict__ in, int ldi, const float* __restrict__ W,
                                                            const float* __restrict__ bias, int act, float* __restrict__ out, int ldo, int C,
                                                            int N) {
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* ib = in + (int64_t)b * ldi;
  const int n_end = min(N, (int)(blockIdx.y + 1) * 16);   // 16 output rows per workgroup, 4 per wave
  for (int n = blockIdx.y * 16 + wave; n < n_end; n += 4) {
    const float* wr = W + (int64_t)n * C;
    float s = 0.0f;
    for (int c = lane; c < C; c += 64) s = fmaf(wr[c], ib[c], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) {
      s += bias ? bias[n] : 0.0f;
      if (act == FX_ACT_RELU) s = fmaxf(s, 0.0f);
      else if (act == FX_ACT_SIGMOID_LOCAL) s = 1.0f / (1.0f + __expf(-s));
      out[(int64_t)b * ldo + n] = s;
    }
  }
}

extern "C" int fx_pooled_linear_f32(const float* in, int ldi, const float* W, const float* bias, int act, float* out, int ldo, int B, int C,
                                    int N, fx_stream_t stream_) {
  FX_CHECK_ARG(in && W && out && B > 0 && C > 0 && N > 0 && ldi >= C && ldo >= N);
  FX_CHECK_ARG(act == FX_ACT_NONE || act == FX_ACT_RELU || act == FX_ACT_SIGMOID_LOCAL);
  hipLaunchKernelGGL(pooled_linear_kernel, dim3(B, (N + 15) / 16), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), in, ldi, W, bias, act, out, ldo, C, N);
  return fx_launch_status();
}

// y = x * gate[b][c] (+ x if self_add) (+ add_vec[b][c]) (+ add_map[b,p,c]):
//   ARM:  feat * atten            (+ the pooled context vector for arm32, + the upsampled coarser level for arm16)
//   FFM:  feat * atten + feat
__global__ __launch_bounds__(256) void channel_gate_kernel(const bf16_t* __restrict__ x, int ldx, const float* __restrict__ gate, int ldg,
                                                           int self_add, const float* __restrict__ add_vec, int ldv,
                                                           const bf16_t* __restrict__ add_map, int ldm, bf16_t* __restrict__ y, int ldy,
                                                           int64_t P, int C8, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    const int64_t r = i / C8;
    const int b = (int)(r / P);
    float v[8], o[8];
    unpack_bf16x8(*reinterpret_cast<const uint4*>(x + r * ldx + c8 * 8), v);
    const float* g = gate + (int64_t)b * ldg + c8 * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = v[j] * g[j] + (self_add ? v[j] : 0.0f);
    if (add_vec) {
      const float* a = add_vec + (int64_t)b * ldv + c8 * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += a[j];
    }
    if (add_map) {
      float m[8];
      unpack_bf16x8(*reinterpret_cast<const uint4*>(add_map + r * ldm + c8 * 8), m);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += m[j];
    }
    *reinterpret_cast<uint4*>(y + r * ldy + c8 * 8) = pack_bf16x8(o);
  }
}

extern "C" int fx_channel_gate_nhwc_bf16(const void* x, int ldx, const float* gate, int ldg, int self_add, const float* add_vec, int ldv,
                                         const void* add_map, int ldm, void* y, int ldy, int B, int P, int C, fx_stream_t stream_) {
  FX_CHECK_ARG(x && gate && y && B > 0 && P > 0 && C > 0 && C % 8 == 0 && ldx >= C && ldy >= C && ldx % 8 == 0 && ldy % 8 == 0 && ldg >= C);
  FX_CHECK_ARG((!add_vec || ldv >= C) && (!add_map || (ldm >= C && ldm % 8 == 0)));
  const int64_t total = (int64_t)B * P * (C / 8);
  int64_t grid = (total + 255) / 256;
  if (grid > 256 * 32) grid = 256 * 32;
  hipLaunchKernelGGL(channel_gate_kernel, dim3((int)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), (const bf16_t*)x, ldx, gate, ldg,
                     self_add, add_vec, ldv, (const bf16_t*)add_map, ldm, (bf16_t*)y, ldy, (int64_t)P, C / 8, total);
  return fx_launch_status();
}

// ------------------------------------------------------------------------------------------------
// predict_all_pixels post-process.  winner[b,y,x] = argmax_q score[b,q] * up(prob[b,q])[y,x] (first maximum, like
// torch.argmax), up = F.interpolate(bilinear, align_corners=False) of the low-resolution sigmoid probabilities (the tail of
// BisenetFormer.forward, modelling.py:606-607).  Per (image, query): pixel count, bounding box (masks_to_xyxy) and, for
// use_mask_score, the sum of the winning probabilities - accumulated in LDS per workgroup, written as per-workgroup partials
// and reduced in a fixed order by the shared selection kernel (integers exact; the float sum uses LDS float atomics).
struct SegPartial {   // same layout as MfPartial (mask_ops.hip) - the selection kernel below mirrors mf_select_kernel
  int32_t cnt;
  float sum;
  int32_t x0, x1, y0, y1;
};

struct SegAxis {
  int i0, i1;
  float w0, w1;
};

__device__ __forceinline__ SegAxis seg_lerp_axis(int dst, float scale, int in) {   // = lerp_axis of mask_ops.hip (ATen's source index rule)
  float src = ((float)dst + 0.5f) * scale - 0.5f;
  src = src < 0.0f ? 0.0f : src;
  SegAxis a;
  a.i0 = (int)src;
  if (a.i0 > in - 1) a.i0 = in - 1;
  a.i1 = a.i0 + (a.i0 < in - 1 ? 1 : 0);
  a.w1 = src - (float)a.i0;
  a.w0 = 1.0f - a.w1;
  return a;
}

__device__ __forceinline__ float seg_lerp_row(float p0, float p1, float w0, float w1) {
#pragma clang fp contract(off)
  return w0 * p0 + w1 * p1;
}

#define SEG_QMAX 128

__device__ __forceinline__ void seg_stats_init(int32_t* cnt, int32_t* bx, float* sum, int Q) {
  for (int q = threadIdx.x; q < Q; q += 256) {
    cnt[q] = 0;
    sum[q] = 0.0f;
    bx[q * 4 + 0] = 0x7fffffff; bx[q * 4 + 1] = -1; bx[q * 4 + 2] = 0x7fffffff; bx[q * 4 + 3] = -1;
  }
}

__device__ __forceinline__ void seg_stats_add(int32_t* cnt, int32_t* bx, float* sum, int q, int n, float s, int xa, int xb, int ya, int yb,
                                              bool use_sum) {
  atomicAdd(&cnt[q], n);
  if (use_sum) atomicAdd(&sum[q], s);
  atomicMin(&bx[q * 4 + 0], xa); atomicMax(&bx[q * 4 + 1], xb);
  atomicMin(&bx[q * 4 + 2], ya); atomicMax(&bx[q * 4 + 3], yb);
}

// S x S output pixels per low-resolution cell (H == S*h, W == S*w; S = 8 for BiSeNetFormer's stride-8 masks): one thread per
// cell keeps the S*S running maxima in registers and reads each query's 3x3 neighbourhood once (9 loads per S*S pixels).
// Offset r inside a cell: src = cell + (r + 0.5)/S - 0.5 -> taps (cell-1, cell) with lambda 1 + (r+.5)/S - .5 for r < S/2 and
// (cell, cell+1) with lambda (r+.5)/S - .5 otherwise; at the first cell the source clamps to 0 (weights 1, 0) and at the
// last the right tap clamps onto the cell - exactly seg_lerp_axis, all values exact in fp32, so the result is bit-identical
// to the generic kernel.
template <int S, bool USE_SUM, bool SKIP>
__global__ __launch_bounds__(256) void seg_winner_cell_kernel(const float* __restrict__ lo, int h, int w, const float* __restrict__ score, int Q,
                                                              uint8_t* __restrict__ winner, SegPartial* __restrict__ part, int nblk) {
  __shared__ int32_t s_cnt[SEG_QMAX];
  __shared__ int32_t s_bx[SEG_QMAX * 4];
  __shared__ float s_sum[SEG_QMAX];
  __shared__ float s_score[SEG_QMAX];
  const int b = blockIdx.y;
  seg_stats_init(s_cnt, s_bx, s_sum, Q);
  for (int q = threadIdx.x; q < Q; q += 256) s_score[q] = score[b * Q + q];
  __syncthreads();
  const int cell = blockIdx.x * 256 + threadIdx.x;
  const int H = S * h, W = S * w;
  if (cell < h * w) {
    const int i = cell / w, j = cell - i * w;
    const int im = max(i - 1, 0), ip = min(i + 1, h - 1), jm = max(j - 1, 0), jp = min(j + 1, w - 1);
    // weights of the second tap for the S offsets (first half: taps (cell-1, cell); second half: (cell, cell+1)); at the
    // first cell of an axis the clamped source is index 0 with weights (1, 0) on taps (cell, cell+1)
    const bool y_first = i == 0, x_first = j == 0;
    float wy1[S], wx1[S];
#pragma unroll
    for (int r = 0; r < S; ++r) {
      const float f = ((float)r + 0.5f) / (float)S - 0.5f;   // exact: S is a power of two
      wy1[r] = r < S / 2 ? (y_first ? 0.0f : 1.0f + f) : f;
      wx1[r] = r < S / 2 ? (x_first ? 0.0f : 1.0f + f) : f;
    }
    float best[S * S], bprob[USE_SUM ? S * S : 1];
    uint8_t bidx[S * S];
#pragma unroll
    for (int k = 0; k < S * S; ++k) best[k] = -INFINITY, bidx[k] = 0;
    const float* pb = lo + (int64_t)b * Q * h * w;
    // window rows (im, i, ip) x columns (jm, j, jp), clamped at the borders; the next query's window is fetched while the current
    // one is evaluated (one wave per SIMD at this register count: nothing else hides the L2 latency)
    const int o00 = im * w + jm, o01 = im * w + j, o02 = im * w + jp, o10 = i * w + jm, o11 = i * w + j, o12 = i * w + jp,
              o20 = ip * w + jm, o21 = ip * w + j, o22 = ip * w + jp;
    float nx[3][3];
    nx[0][0] = pb[o00]; nx[0][1] = pb[o01]; nx[0][2] = pb[o02];
    nx[1][0] = pb[o10]; nx[1][1] = pb[o11]; nx[1][2] = pb[o12];
    nx[2][0] = pb[o20]; nx[2][1] = pb[o21]; nx[2][2] = pb[o22];
    // A query whose score x the largest of its nine window values cannot exceed the smallest running maximum of the cell is skipped: every
    // output pixel is a convex combination of window values (weights w0 + w1 = 1 exactly, all dyadic), so its product with the score is at
    // most sc * max9 up to two roundings - the 1e-6 margin covers them - and an update needs a STRICTLY larger value.  Same winner map, bit
    // for bit; with trained weights most of the Q queries (low class score) leave after nine loads and ten VALU operations.
    // bmin is refreshed only every eighth query (63 fmin per refresh): the running maxima only grow, so a stale bmin is a valid - merely
    // weaker - bound.  SKIP is a template parameter, not a kernel argument: with the test compiled in, the loop costs 0.40 instead of
    // 0.26 ms per part on inputs where nothing can be skipped (random weights - every query competitive), whatever the runtime flag says
    // (register allocation / loop shape; measured with the pre-test kernel linked in, profiles/r04_wgrad_table_after.txt).
    float bmin = -INFINITY;
    bool touched = false;
#pragma unroll 1
    for (int q = 0; q < Q; ++q) {
      const float sc = s_score[q];
      float v[3][3];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) v[a][c] = nx[a][c];
      if (q + 1 < Q) {
        const float* p = pb + (int64_t)(q + 1) * h * w;
        nx[0][0] = p[o00]; nx[0][1] = p[o01]; nx[0][2] = p[o02];
        nx[1][0] = p[o10]; nx[1][1] = p[o11]; nx[1][2] = p[o12];
        nx[2][0] = p[o20]; nx[2][1] = p[o21]; nx[2][2] = p[o22];
      }
      if constexpr (SKIP) {
        const float m9 = fmaxf(fmaxf(fmaxf(v[0][0], v[0][1]), fmaxf(v[0][2], v[1][0])), fmaxf(fmaxf(fmaxf(v[1][1], v[1][2]), fmaxf(v[2][0], v[2][1])), v[2][2]));
        if (sc * m9 * 1.000001f <= bmin) continue;
      }
      float rowi[3][S];   // x-interpolated window rows (ATen interpolates along x first; same operation order as lerp_taps)
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float l0 = x_first ? v[a][1] : v[a][0], l1 = x_first ? v[a][2] : v[a][1];
#pragma unroll
        for (int c = 0; c < S; ++c)
          rowi[a][c] = c < S / 2 ? seg_lerp_row(l0, l1, 1.0f - wx1[c], wx1[c]) : seg_lerp_row(v[a][1], v[a][2], 1.0f - wx1[c], wx1[c]);
      }
#pragma unroll
      for (int c = 0; c < S; ++c) {
        const float t0 = y_first ? rowi[1][c] : rowi[0][c], t1 = y_first ? rowi[2][c] : rowi[1][c];
#pragma unroll
        for (int r = 0; r < S; ++r) {
          const float pr = r < S / 2 ? seg_lerp_row(t0, t1, 1.0f - wy1[r], wy1[r]) : seg_lerp_row(rowi[1][c], rowi[2][c], 1.0f - wy1[r], wy1[r]);
          const float val = sc * pr;
          if (val > best[r * S + c]) {
            best[r * S + c] = val;
            bidx[r * S + c] = (uint8_t)q;
            if (USE_SUM) bprob[r * S + c] = pr;
            if constexpr (SKIP) touched = true;
          }
        }
      }
      if constexpr (SKIP) if (touched && (q & 7) == 7) {
        float m = best[0];
#pragma unroll
        for (int k = 1; k < S * S; ++k) m = fminf(m, best[k]);
        bmin = m;
        touched = false;
      }
    }
    // winner map + statistics
    uint8_t* wo = winner + ((int64_t)b * H + (int64_t)i * S) * W + j * S;
    bool uniform = true;
#pragma unroll
    for (int k = 1; k < S * S; ++k) uniform = uniform && bidx[k] == bidx[0];
#pragma unroll
    for (int r = 0; r < S; ++r) {
      uint32_t pk[S / 4];
#pragma unroll
      for (int c4 = 0; c4 < S / 4; ++c4)
        pk[c4] = (uint32_t)bidx[r * S + c4 * 4] | ((uint32_t)bidx[r * S + c4 * 4 + 1] << 8) | ((uint32_t)bidx[r * S + c4 * 4 + 2] << 16) |
                 ((uint32_t)bidx[r * S + c4 * 4 + 3] << 24);
#pragma unroll
      for (int c4 = 0; c4 < S / 4; ++c4) *reinterpret_cast<uint32_t*>(wo + (int64_t)r * W + c4 * 4) = pk[c4];
    }
    if (uniform) {
      float s = 0.0f;
      if (USE_SUM) {
#pragma unroll
        for (int k = 0; k < S * S; ++k) s += bprob[k];
      }
      seg_stats_add(s_cnt, s_bx, s_sum, bidx[0], S * S, s, j * S, j * S + S - 1, i * S, i * S + S - 1, USE_SUM);
    } else {
#pragma unroll
      for (int r = 0; r < S; ++r)
#pragma unroll
        for (int c = 0; c < S; ++c)
          seg_stats_add(s_cnt, s_bx, s_sum, bidx[r * S + c], 1, USE_SUM ? bprob[r * S + c] : 0.0f, j * S + c, j * S + c, i * S + r, i * S + r,
                        USE_SUM);
    }
  }
  __syncthreads();
  for (int q = threadIdx.x; q < Q; q += 256)
    part[((int64_t)b * Q + q) * nblk + blockIdx.x] = SegPartial{s_cnt[q], s_sum[q], s_bx[q * 4], s_bx[q * 4 + 1], s_bx[q * 4 + 2], s_bx[q * 4 + 3]};
}

// Any size: one thread per output pixel, four taps per (pixel, query).
template <bool USE_SUM>
__global__ __launch_bounds__(256) void seg_winner_generic_kernel(const float* __restrict__ lo, int h, int w, int H, int W, float sy, float sx,
                                                                 const float* __restrict__ score, int Q, uint8_t* __restrict__ winner,
                                                                 SegPartial* __restrict__ part, int nblk) {
  __shared__ int32_t s_cnt[SEG_QMAX];
  __shared__ int32_t s_bx[SEG_QMAX * 4];
  __shared__ float s_sum[SEG_QMAX];
  __shared__ float s_score[SEG_QMAX];
  const int b = blockIdx.y;
  seg_stats_init(s_cnt, s_bx, s_sum, Q);
  for (int q = threadIdx.x; q < Q; q += 256) s_score[q] = score[b * Q + q];
  __syncthreads();
  const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (pix < (int64_t)H * W) {
    const int y = (int)(pix / W), x = (int)(pix - (int64_t)y * W);
    const SegAxis ay = seg_lerp_axis(y, sy, h), ax = seg_lerp_axis(x, sx, w);
    float best = -INFINITY, bp = 0.0f;
    int bi = 0;
    for (int q = 0; q < Q; ++q) {
      const float* p = lo + ((int64_t)b * Q + q) * h * w;
      const float top = seg_lerp_row(p[ay.i0 * w + ax.i0], p[ay.i0 * w + ax.i1], ax.w0, ax.w1);
      const float bot = seg_lerp_row(p[ay.i1 * w + ax.i0], p[ay.i1 * w + ax.i1], ax.w0, ax.w1);
      const float pr = seg_lerp_row(top, bot, ay.w0, ay.w1);
      const float val = s_score[q] * pr;
      if (val > best) best = val, bi = q, bp = pr;
    }
    winner[(int64_t)b * H * W + pix] = (uint8_t)bi;
    seg_stats_add(s_cnt, s_bx, s_sum, bi, 1, bp, x, x, y, y, USE_SUM);
  }
  __syncthreads();
  for (int q = threadIdx.x; q < Q; q += 256)
    part[((int64_t)b * Q + q) * nblk + blockIdx.x] = SegPartial{s_cnt[q], s_sum[q], s_bx[q * 4], s_bx[q * 4 + 1], s_bx[q * 4 + 2], s_bx[q * 4 + 3]};
}

// Selection (bisenetformer/processor.py:232-262, same rules as the MaskFormer processor): masks with more than one pixel;
// score = class score [x mean winning probability, with the reference's 1e-3 / 1e-5 constants]; score > threshold.
__global__ __launch_bounds__(128) void seg_select_kernel(const SegPartial* __restrict__ part, int nblk, const float* __restrict__ score,
                                                         const int32_t* __restrict__ label, int Q, float thr, int use_mask_score,
                                                         int32_t* __restrict__ det_count, int32_t* __restrict__ det_query,
                                                         float* __restrict__ det_score, int32_t* __restrict__ det_label,
                                                         int32_t* __restrict__ det_box, int32_t* __restrict__ mask_area) {
  const int b = blockIdx.x, q = threadIdx.x;
  SegPartial r{0, 0.0f, 0x7fffffff, -1, 0x7fffffff, -1};
  if (q < Q)
    for (int i = 0; i < nblk; ++i) {
      const SegPartial t = part[((int64_t)b * Q + q) * nblk + i];
      r.cnt += t.cnt; r.sum += t.sum;
      r.x0 = min(r.x0, t.x0); r.x1 = max(r.x1, t.x1);
      r.y0 = min(r.y0, t.y0); r.y1 = max(r.y1, t.y1);
    }
  float s = q < Q ? score[b * Q + q] : 0.0f;
  bool keep = q < Q && r.cnt > 1;
  if (keep && use_mask_score) s *= (1e-3f * r.sum) / (1e-3f * (float)r.cnt + 1e-5f);
  if (thr > 0.0f) keep = keep && s > thr;
  __shared__ int wcount[2];
  const unsigned long long bal = __ballot(keep);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wcount[wave] = __popcll(bal);
  __syncthreads();
  const int pos = (wave ? wcount[0] : 0) + __popcll(bal & ((1ull << lane) - 1ull));
  if (keep) {
    const int o = b * Q + pos;
    det_query[o] = q;
    det_score[o] = s;
    det_label[o] = label[b * Q + q];
    det_box[o * 4 + 0] = r.x0; det_box[o * 4 + 1] = r.y0; det_box[o * 4 + 2] = r.x1; det_box[o * 4 + 3] = r.y1;
    mask_area[o] = r.cnt;
  }
  if (threadIdx.x == 0) det_count[b] = wcount[0] + wcount[1];
}

// Bit-packed binary masks of the kept detections from the winner map: bit x&31 of words[((b*Q + slot)*H + y)*ceil(W/32) + x/32].
__global__ __launch_bounds__(256) void seg_pack_masks_kernel(const uint8_t* __restrict__ winner, int H, int W, const int32_t* __restrict__ det_count,
                                                             const int32_t* __restrict__ det_query, int Q, uint32_t* __restrict__ words) {
  const int slot = blockIdx.y, b = blockIdx.z;
  if (slot >= det_count[b]) return;
  const int q = det_query[b * Q + slot];
  const int W32 = (W + 31) >> 5;
  const uint8_t* wb = winner + (int64_t)b * H * W;
  uint32_t* wo = words + ((int64_t)b * Q + slot) * H * W32;
  const int64_t nwords = (int64_t)H * W32;
  if (W & 31) {   // rows are not whole words: byte-wise, bits >= W stay 0 (odd image widths; the aligned form below is the hot one)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (int64_t)gridDim.x * 256) {
      const int y = (int)(i / W32), x0 = (int)(i % W32) * 32;
      const uint8_t* row = wb + (int64_t)y * W + x0;
      const int n = min(32, W - x0);
      uint32_t bits = 0;
      for (int t = 0; t < n; ++t) bits |= (uint32_t)(row[t] == (uint8_t)q) << t;
      wo[i] = bits;
    }
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (int64_t)gridDim.x * 256) {
    const uint4 a = *reinterpret_cast<const uint4*>(wb + i * 32), c = *reinterpret_cast<const uint4*>(wb + i * 32 + 16);
    const uint32_t v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
    uint32_t bits = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int t = 0; t < 4; ++t) bits |= (uint32_t)(((v[k] >> (8 * t)) & 0xffu) == (uint32_t)q) << (k * 4 + t);
    wo[i] = bits;
  }
}

static int seg_blocks(int h, int w, int H, int W) {
  if ((H == 8 * h && W == 8 * w) || (H == 4 * h && W == 4 * w)) return (h * w + 255) / 256;
  return (int)(((int64_t)H * W + 255) / 256);
}

extern "C" size_t fx_seg_postprocess_workspace_bytes(int B, int Q, int h, int w, int H, int W) {
  if (B <= 0 || Q <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return 0;
  const size_t win = ((size_t)B * H * W + 255) / 256 * 256;
  return win + (size_t)B * Q * seg_blocks(h, w, H, W) * sizeof(SegPartial);
}

extern "C" int fx_seg_postprocess(const float* mask_probs_lowres, int h, int w, int H, int W, const float* score, const int32_t* label, int B,
                                  int Q, float threshold, int use_mask_score, void* workspace, size_t workspace_bytes, int32_t* det_count,
                                  int32_t* det_query, float* det_score, int32_t* det_label, int32_t* det_box, int32_t* det_area,
                                  uint32_t* mask_words, uint8_t* winner_out, fx_stream_t stream_) {
  FX_CHECK_ARG(mask_probs_lowres && score && label && workspace && det_count && det_query && det_score && det_label && det_box && det_area);
  FX_CHECK_ARG(B > 0 && Q > 0 && h > 0 && w > 0 && H > 0 && W > 0);
  if (Q > SEG_QMAX) return FX_ERR_UNSUPPORTED;
  FX_CHECK_ARG(workspace_bytes >= fx_seg_postprocess_workspace_bytes(B, Q, h, w, H, W) && ((uintptr_t)workspace % 16) == 0);
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const size_t win_bytes = ((size_t)B * H * W + 255) / 256 * 256;
  uint8_t* winner = winner_out ? winner_out : reinterpret_cast<uint8_t*>(workspace);
  SegPartial* part = reinterpret_cast<SegPartial*>(reinterpret_cast<uint8_t*>(workspace) + win_bytes);
  const int nblk = seg_blocks(h, w, H, W);
  // the hopeless-query test of seg_winner_cell_kernel: pays when most queries carry a small class score (trained weights), costs 0.14 ms
  // per 16-image part when none does (the random-weight bench) - default from the measurement, see scripts/dev/seg_skip_bench.py
  const int skip = fx_tune("FX_SEG_SKIP", 0);   // read per call (capture time under a graph): tests flip it in-process
#define SEG_LAUNCH(S, SUM, SK)                                                                                                             \
  hipLaunchKernelGGL((seg_winner_cell_kernel<S, SUM, SK>), dim3(nblk, B), dim3(256), 0, stream, mask_probs_lowres, h, w, score, Q, winner, \
                     part, nblk)
#define SEG_CELL(S)                                                                                                                        \
  do {                                                                                                                                     \
    if (use_mask_score) { if (skip) SEG_LAUNCH(S, true, true); else SEG_LAUNCH(S, true, false); }                                          \
    else { if (skip) SEG_LAUNCH(S, false, true); else SEG_LAUNCH(S, false, false); }                                                       \
  } while (0)
  if (H == 8 * h && W == 8 * w && W % 4 == 0) SEG_CELL(8);
  else if (H == 4 * h && W == 4 * w) SEG_CELL(4);
  else if (use_mask_score)
    hipLaunchKernelGGL(seg_winner_generic_kernel<true>, dim3(nblk, B), dim3(256), 0, stream, mask_probs_lowres, h, w, H, W, (float)h / (float)H,
                       (float)w / (float)W, score, Q, winner, part, nblk);
  else
    hipLaunchKernelGGL(seg_winner_generic_kernel<false>, dim3(nblk, B), dim3(256), 0, stream, mask_probs_lowres, h, w, H, W, (float)h / (float)H,
                       (float)w / (float)W, score, Q, winner, part, nblk);
#undef SEG_CELL
#undef SEG_LAUNCH
  hipLaunchKernelGGL(seg_select_kernel, dim3(B), dim3(128), 0, stream, part, nblk, score, label, Q, threshold, use_mask_score, det_count,
                     det_query, det_score, det_label, det_box, det_area);
  if (mask_words) {
    const int64_t nwords = (int64_t)H * ((W + 31) / 32);
    int gx = (int)((nwords + 255) / 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(seg_pack_masks_kernel, dim3(gx, Q, B), dim3(256), 0, stream, winner, H, W, det_count, det_query, Q, mask_words);
  }
  return fx_launch_status();
}
