// Training-path kernels of the token-space layers: activation forward/backward on a saved
// pre-activation, bias gradient (column sum), LayerNorm backward, bilinear-resize backward, attention backward, row
// scatter (backward of the query gather).  Correctness-first versions: fp32 math, wavefront reductions, fp32 atomics for
// the cross-row parameter gradients; the decoder / AIFI tensors they touch are small (<= B*8400 rows of 256).
#include "common.h"

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ------------------------------------------------------------------------------------------------ activations
__device__ __forceinline__ float act_grad(float z, int act) {
  switch (act) {
    case FX_ACT_RELU: return z > 0.0f ? 1.0f : 0.0f;
    case FX_ACT_SILU: {
      const float s = 1.0f / (1.0f + __expf(-z));
      return s * (1.0f + z * (1.0f - s));
    }
    case FX_ACT_GELU: {  // exact (erf) GELU, like F.gelu
      const float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752f));
      return cdf + z * 0.3989422804014327f * __expf(-0.5f * z * z);
    }
    default: return 1.0f;
  }
}

__global__ __launch_bounds__(256) void act_fwd_kernel(const bf16_t* __restrict__ z, int ldz, bf16_t* __restrict__ y, int ldy, int64_t rows, int C8,
                                                      int act) {
  const int64_t total = rows * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    const int64_t r = i / C8;
    float v[8];
    unpack_bf16x8(*reinterpret_cast<const uint4*>(z + r * ldz + c8 * 8), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = fx_act(v[j], act);
    *reinterpret_cast<uint4*>(y + r * ldy + c8 * 8) = pack_bf16x8(v);
  }
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const bf16_t* __restrict__ dy, int lddy, const bf16_t* __restrict__ z, int ldz,
                                                      bf16_t* __restrict__ dz, int lddz, int64_t rows, int C8, int act) {
  const int64_t total = rows * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    const int64_t r = i / C8;
    float g[8], v[8];
    unpack_bf16x8(*reinterpret_cast<const uint4*>(dy + r * lddy + c8 * 8), g);
    unpack_bf16x8(*reinterpret_cast<const uint4*>(z + r * ldz + c8 * 8), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] *= act_grad(v[j], act);
    *reinterpret_cast<uint4*>(dz + r * lddz + c8 * 8) = pack_bf16x8(g);
  }
}

static inline int ew_grid(int64_t total) {
  int64_t grid = (total + 255) / 256;
  return (int)(grid > 256 * 32 ? 256 * 32 : grid);
}

extern "C" int fx_act_fwd_bf16(const void* z, int ldz, void* y, int ldy, int64_t rows, int cols, int act, fx_stream_t stream_) {
  FX_CHECK_ARG(z && y && rows > 0 && cols > 0 && cols % 8 == 0 && ldz >= cols && ldy >= cols && ldz % 8 == 0 && ldy % 8 == 0);
  hipLaunchKernelGGL(act_fwd_kernel, dim3(ew_grid(rows * (cols / 8))), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), (const bf16_t*)z, ldz,
                     (bf16_t*)y, ldy, rows, cols / 8, act);
  return fx_launch_status();
}

extern "C" int fx_act_bwd_bf16(const void* dy, int lddy, const void* z, int ldz, void* dz, int lddz, int64_t rows, int cols, int act,
                               fx_stream_t stream_) {
  FX_CHECK_ARG(dy && z && dz && rows > 0 && cols > 0 && cols % 8 == 0 && lddy >= cols && ldz >= cols && lddz >= cols);
  FX_CHECK_ARG(lddy % 8 == 0 && ldz % 8 == 0 && lddz % 8 == 0);
  hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_grid(rows * (cols / 8))), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), (const bf16_t*)dy,
                     lddy, (const bf16_t*)z, ldz, (bf16_t*)dz, lddz, rows, cols / 8, act);
  return fx_launch_status();
}

// ------------------------------------------------------------------------------------------------ bias gradient
// out[c] += sum_r x[r][c].  Thread = 8 consecutive columns (one 16-byte load) x a strided set of rows; a workgroup covers
// 32 column groups (256 columns) x 8 row lanes and walks COLSUM_ROWS rows, then reduces the row lanes through LDS and
// issues one atomic per column.
#define COLSUM_ROWS 256
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ x, int ldx, float* __restrict__ out, int64_t rows, int cols) {
  __shared__ float part[8][256];
  const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c0 = blockIdx.x * 256 + cg * 8;
  const int64_t r0 = (int64_t)blockIdx.y * COLSUM_ROWS;
  const int64_t r1 = r0 + COLSUM_ROWS < rows ? r0 + COLSUM_ROWS : rows;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c0 < cols)
    for (int64_t r = r0 + rl; r < r1; r += 8) {
      float v[8];
      unpack_bf16x8(*reinterpret_cast<const uint4*>(x + r * ldx + c0), v);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += v[j];
    }
#pragma unroll
  for (int j = 0; j < 8; ++j) part[rl][cg * 8 + j] = s[j];
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < cols) {
    float t = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += part[i][threadIdx.x];
    unsafeAtomicAdd(out + c, t);
  }
}

extern "C" int fx_colsum_bf16(const void* x, int ldx, float* out, int64_t rows, int cols, fx_stream_t stream_) {
  FX_CHECK_ARG(x && out && rows > 0 && cols > 0 && cols % 8 == 0 && ldx >= cols && ldx % 8 == 0);
  hipLaunchKernelGGL(colsum_kernel, dim3((cols + 255) / 256, (unsigned)((rows + COLSUM_ROWS - 1) / COLSUM_ROWS)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream_), (const bf16_t*)x, ldx, out, rows, cols);
  return fx_launch_status();
}

// ------------------------------------------------------------------------------------------------ LayerNorm backward
// y = (x - mu) * rstd * gamma + beta over 256 (or 128) channels, one wave per row (4 or 2 channels per lane):
//   g = dy * gamma;  dx = rstd * (g - mean(g) - xhat * mean(g * xhat));  dgamma += dy * xhat;  dbeta += dy.
template <int CPL>   // channels per lane: 4 (256 columns) or 2 (128 columns: the narrow pixel-decoder encoders of fai-mf-{m,s}-coco-ins)
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const bf16_t* __restrict__ dy, int lddy, const bf16_t* __restrict__ x, int ldx,
                                                            const float* __restrict__ gamma, bf16_t* __restrict__ dx, int lddx,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int rows) {
  constexpr int COLS = 64 * CPL;
  constexpr float INV = 1.0f / (float)COLS;
  __shared__ float pg[4][COLS], pb[4][COLS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float ag[CPL], ab[CPL], gmv[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) ag[j] = 0.0f, ab[j] = 0.0f, gmv[j] = gamma[lane * CPL + j];
  auto load = [&](const bf16_t* p, float* v) {   // CPL consecutive bf16 of this lane
    if constexpr (CPL == 4) {
      const uint2 w = *reinterpret_cast<const uint2*>(p);
      v[0] = bf16lo_to_f32(w.x), v[1] = bf16hi_to_f32(w.x), v[2] = bf16lo_to_f32(w.y), v[3] = bf16hi_to_f32(w.y);
    } else {
      const unsigned w = *reinterpret_cast<const unsigned*>(p);
      v[0] = bf16lo_to_f32(w), v[1] = bf16hi_to_f32(w);
    }
  };
  // a wave walks ~16 rows: the next row's loads are issued before this row's four wave reductions (the loop was one memory round
  // trip per row: 24 us for the decoder's 4800-row LayerNorms, 21 of them on the critical path of a training step)
  const int step = gridDim.x * 4;
  int row = blockIdx.x * 4 + wave;
  float xn[CPL], dn[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) xn[j] = dn[j] = 0.0f;
  if (row < rows) {
    load(x + (int64_t)row * ldx + lane * CPL, xn);
    load(dy + (int64_t)row * lddy + lane * CPL, dn);
  }
  for (; row < rows; row += step) {
    float v[CPL], d[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) v[j] = xn[j], d[j] = dn[j];
    if (row + step < rows) {
      load(x + (int64_t)(row + step) * ldx + lane * CPL, xn);
      load(dy + (int64_t)(row + step) * lddy + lane * CPL, dn);
    }
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) sum += v[j];
    const float mean = wsum(sum) * INV;
    float c[CPL], var = 0.0f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) c[j] = v[j] - mean, var += c[j] * c[j];
    const float rstd = rsqrtf(wsum(var) * INV + 1e-5f);
    float xh[CPL], g[CPL], sg = 0.0f, sgx = 0.0f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      xh[j] = c[j] * rstd;
      g[j] = d[j] * gmv[j];
      sg += g[j];
      sgx += g[j] * xh[j];
      ag[j] += d[j] * xh[j];
      ab[j] += d[j];
    }
    sg = wsum(sg) * INV;
    sgx = wsum(sgx) * INV;
    if constexpr (CPL == 4) {
      uint2 o;
      o.x = pack_bf16x2(rstd * (g[0] - sg - xh[0] * sgx), rstd * (g[1] - sg - xh[1] * sgx));
      o.y = pack_bf16x2(rstd * (g[2] - sg - xh[2] * sgx), rstd * (g[3] - sg - xh[3] * sgx));
      *reinterpret_cast<uint2*>(dx + (int64_t)row * lddx + lane * 4) = o;
    } else {
      *reinterpret_cast<unsigned*>(dx + (int64_t)row * lddx + lane * 2) = pack_bf16x2(rstd * (g[0] - sg - xh[0] * sgx), rstd * (g[1] - sg - xh[1] * sgx));
    }
  }
#pragma unroll
  for (int j = 0; j < CPL; ++j) pg[wave][lane * CPL + j] = ag[j], pb[wave][lane * CPL + j] = ab[j];
  __syncthreads();
  const int c = threadIdx.x;
  if (c < COLS) {
    if (dgamma) unsafeAtomicAdd(dgamma + c, pg[0][c] + pg[1][c] + pg[2][c] + pg[3][c]);
    if (dbeta) unsafeAtomicAdd(dbeta + c, pb[0][c] + pb[1][c] + pb[2][c] + pb[3][c]);
  }
}

extern "C" int fx_layernorm_bwd_bf16(const void* dy, int lddy, const void* x, int ldx, const float* gamma, void* dx, int lddx, float* dgamma,
                                     float* dbeta, int rows, int cols, fx_stream_t stream_) {
  FX_CHECK_ARG(dy && x && gamma && dx && rows > 0);
  if (cols != 256 && cols != 128) return FX_ERR_UNSUPPORTED;
  FX_CHECK_ARG(lddy >= cols && ldx >= cols && lddx >= cols && lddy % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0);
  // every workgroup ends with 512 fp32 atomics onto the same dgamma / dbeta addresses: ~64 rows per workgroup keep that tail short
  // (one workgroup per 4 rows made the 4800-row decoder LayerNorms 36 us each - 1024-deep contention per address)
  int grid = (rows + 63) / 64;
  if (grid < 64) grid = (rows + 3) / 4 < 64 ? (rows + 3) / 4 : 64;
  if (grid > 1024) grid = 1024;
  if (cols == 128)
    hipLaunchKernelGGL(layernorm_bwd_kernel<2>, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), (const bf16_t*)dy, lddy,
                       (const bf16_t*)x, ldx, gamma, (bf16_t*)dx, lddx, dgamma, dbeta, rows);
  else
    hipLaunchKernelGGL(layernorm_bwd_kernel<4>, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), (const bf16_t*)dy, lddy,
                       (const bf16_t*)x, ldx, gamma, (bf16_t*)dx, lddx, dgamma, dbeta, rows);
  return fx_launch_status();
}

// ------------------------------------------------------------------------------------------------ bilinear resize backward
// Gather form of the adjoint of fx_resize_bilinear_nhwc_bf16 (align_corners=False): an input pixel sums, over the output
// pixels whose two taps per axis include it, dy times the tap weight.  Deterministic, no atomics, writes bf16 directly.
__device__ __forceinline__ void bil_src(int dst, float scale, int in, int& i0, int& i1, float& w0, float& w1) {
  float src = ((float)dst + 0.5f) * scale - 0.5f;
  src = src < 0.0f ? 0.0f : src;
  i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  w1 = src - (float)i0;
  w0 = 1.0f - w1;
}

// candidate output range whose source coordinate can fall in [i - 1, i + 1]
__device__ __forceinline__ void bil_range(int i, float scale, int out, int& lo, int& hi) {
  const float inv = 1.0f / scale;
  lo = (int)floorf(((float)i - 1.0f + 0.5f) * inv - 0.5f) - 1;
  hi = (int)ceilf(((float)i + 1.0f + 0.5f) * inv - 0.5f) + 1;
  lo = lo < 0 ? 0 : lo;
  hi = hi > out - 1 ? out - 1 : hi;
}

__global__ __launch_bounds__(256) void resize_bwd_kernel(const bf16_t* __restrict__ dy, int lddy, bf16_t* __restrict__ dx, int lddx, int B, int H,
                                                         int W, int C8, int Ho, int Wo, float sh, float sw) {
  const int64_t total = (int64_t)B * H * W * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    int64_t p = i / C8;
    const int w = (int)(p % W);
    p /= W;
    const int h = (int)(p % H);
    const int b = (int)(p / H);
    int holo, hohi, wolo, wohi;
    bil_range(h, sh, Ho, holo, hohi);
    bil_range(w, sw, Wo, wolo, wohi);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int ho = holo; ho <= hohi; ++ho) {
      int h0, h1;
      float lh0, lh1;
      bil_src(ho, sh, H, h0, h1, lh0, lh1);
      const float wh = (h0 == h ? lh0 : 0.0f) + (h1 == h ? lh1 : 0.0f);
      if (wh == 0.0f) continue;
      for (int wo = wolo; wo <= wohi; ++wo) {
        int w0, w1;
        float lw0, lw1;
        bil_src(wo, sw, W, w0, w1, lw0, lw1);
        const float ww = (w0 == w ? lw0 : 0.0f) + (w1 == w ? lw1 : 0.0f);
        if (ww == 0.0f) continue;
        float g[8];
        unpack_bf16x8(*reinterpret_cast<const uint4*>(dy + (((int64_t)b * Ho + ho) * Wo + wo) * lddy + c8 * 8), g);
        const float k = wh * ww;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += k * g[j];
      }
    }
    *reinterpret_cast<uint4*>(dx + (((int64_t)b * H + h) * W + w) * lddx + c8 * 8) = pack_bf16x8(acc);
  }
}

extern "C" int fx_resize_bilinear_bwd_nhwc_bf16(const void* dy, int lddy, void* dx, int lddx, int B, int H, int W, int C, int Ho, int Wo,
                                                fx_stream_t stream_) {
  FX_CHECK_ARG(dy && dx && B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 8 == 0 && lddy >= C && lddy % 8 == 0 && lddx >= C && lddx % 8 == 0);
  hipLaunchKernelGGL(resize_bwd_kernel, dim3(ew_grid((int64_t)B * H * W * (C / 8))), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_),
                     (const bf16_t*)dy, lddy, (bf16_t*)dx, lddx, B, H, W, C / 8, Ho, Wo, (float)H / (float)Ho, (float)W / (float)Wo);
  return fx_launch_status();
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const float4 a = *reinterpret_cast<const float4*>(x + i * 8), b = *reinterpret_cast<const float4*>(x + i * 8 + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    *reinterpret_cast<uint4*>(y + i * 8) = pack_bf16x8(v);
  }
}

extern "C" int fx_cast_f32_bf16(const float* x, void* y, int64_t n, fx_stream_t stream_) {
  FX_CHECK_ARG(x && y && n > 0 && n % 8 == 0);
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), x, (bf16_t*)y, n / 8);
  return fx_launch_status();
}

// (attention backward: attn_bwd.hip)

// ------------------------------------------------------------------------------------------------ gather backward
// dsrc[b, idx[b,j], :] = dout[b, j, :]  (indices are a top-k result: unique per image; dsrc zeroed by the caller)
__global__ __launch_bounds__(256) void scatter_rows_kernel(const bf16_t* __restrict__ dout, int ldo, const int32_t* __restrict__ idx, int k,
                                                           bf16_t* __restrict__ dsrc, int lds, int rpb, int B, int C8) {
  const int64_t total = (int64_t)B * k * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    const int64_t r = i / C8;
    const int b = (int)(r / k);
    *reinterpret_cast<uint4*>(dsrc + ((int64_t)b * rpb + idx[r]) * lds + c8 * 8) = *reinterpret_cast<const uint4*>(dout + r * ldo + c8 * 8);
  }
}

extern "C" int fx_scatter_rows_bf16(const void* dout, int ldo, const int32_t* idx, int k, void* dsrc, int lds, int rows_per_batch, int B, int cols,
                                    fx_stream_t stream_) {
  FX_CHECK_ARG(dout && idx && dsrc && k > 0 && B > 0 && cols > 0 && cols % 8 == 0 && ldo >= cols && lds >= cols && ldo % 8 == 0 && lds % 8 == 0);
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(ew_grid((int64_t)B * k * (cols / 8))), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_),
                     (const bf16_t*)dout, ldo, idx, k, (bf16_t*)dsrc, lds, rows_per_batch, B, cols / 8);
  return fx_launch_status();
}

// ------------------------------------------------------------------------------------------------ VFL loss forward + gradient
// SetCriterion.loss_labels_vfl (fai_detr/modelling.py:464-497) fused with its gradient:
//   t[r][c] = score[r] if c == cls[r] else 0;  w = alpha * sigmoid(x)^gamma * (1 - onehot) + t   (sigmoid detached)
//   loss = sum_{r,c} w * BCEwithLogits(x, t) * scale,   dx = w * (sigmoid(x) - t) * scale        (scale = weight / num_boxes)
// logits bf16 [rows][ld]; cls i32 [rows] (K = no object); loss_out fp32 scalar accumulated with an atomic per block.
__global__ __launch_bounds__(256) void vfl_loss_kernel(const bf16_t* __restrict__ logits, int ld, const int32_t* __restrict__ cls,
                                                       const float* __restrict__ score, float alpha, float gamma, float scale,
                                                       float* __restrict__ loss_out, bf16_t* __restrict__ dlogits, int lddl, int64_t rows, int K) {
  __shared__ float red[4];
  const int64_t total = rows * K;
  float acc = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % K);
    const int64_t r = i / K;
    const float x = bf16_to_f32(logits[r * ld + c]);
    const bool pos = cls[r] == c;
    const float t = pos ? score[r] : 0.0f;
    const float p = 1.0f / (1.0f + __expf(-x));
    const float w = pos ? t : alpha * __powf(p, gamma);
    const float bce = fmaxf(x, 0.0f) - x * t + __logf(1.0f + __expf(-fabsf(x)));
    acc += w * bce;
    if (dlogits) dlogits[r * lddl + c] = f32_to_bf16(w * (p - t) * scale);
  }
  acc = wsum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(loss_out, (red[0] + red[1] + red[2] + red[3]) * scale);
}

extern "C" int fx_vfl_loss_bf16(const void* logits, int ld, const int32_t* cls, const float* score, float alpha, float gamma, float scale,
                                float* loss_out, void* dlogits, int lddl, int64_t rows, int K, fx_stream_t stream_) {
  FX_CHECK_ARG(logits && cls && score && loss_out && rows > 0 && K > 0 && ld >= K && (!dlogits || lddl >= K));
  hipLaunchKernelGGL(vfl_loss_kernel, dim3(ew_grid(rows * K / 4 + 1)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), (const bf16_t*)logits, ld,
                     cls, score, alpha, gamma, scale, loss_out, (bf16_t*)dlogits, lddl, rows, K);
  return fx_launch_status();
}


// ------------------------------------------------------------------------------------------------ iterative box refinement
// TransformerDecoder (fai_detr/modelling.py:996-1013): box = sigmoid(delta + inverse_sigmoid(ref)), inverse_sigmoid(x) =
// log(clamp(clamp(x, 0, 1), eps) / clamp(1 - clamp(x, 0, 1), eps)), eps = 1e-5 (focoos/nn/layers/functional.py).  delta bf16 (the bbox head's
// output), ref / box fp32, n elements.  Backward: ds = g * box * (1 - box); d delta = ds; d ref = ds * d inverse_sigmoid / d ref with
// autograd's clamp conventions (gradient passes where the input is inside the closed clamp range).
__global__ __launch_bounds__(256) void box_refine_kernel(const bf16_t* __restrict__ delta, const float* __restrict__ ref, float* __restrict__ box,
                                                          int64_t n, float eps) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float x = fminf(fmaxf(ref[i], 0.0f), 1.0f);
    const float u = logf(fmaxf(x, eps) / fmaxf(1.0f - x, eps)) + bf16_to_f32(delta[i]);
    box[i] = 1.0f / (1.0f + expf(-u));
  }
}

__global__ __launch_bounds__(256) void box_refine_bwd_kernel(const float* __restrict__ g, const float* __restrict__ box, const float* __restrict__ ref,
                                                              bf16_t* __restrict__ d_delta, float* __restrict__ d_ref, int64_t n, float eps) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float s = box[i];
    const float ds = g[i] * s * (1.0f - s);
    d_delta[i] = f32_to_bf16(ds);
    if (d_ref) {
      const float r = ref[i];
      const float x = fminf(fmaxf(r, 0.0f), 1.0f);
      const float inside = (r >= 0.0f && r <= 1.0f) ? 1.0f : 0.0f;
      const float dinv = (x >= eps ? 1.0f / x : 0.0f) + (1.0f - x >= eps ? 1.0f / (1.0f - x) : 0.0f);
      d_ref[i] = ds * dinv * inside;
    }
  }
}

extern "C" int fx_box_refine_f32(const void* delta, const float* ref, float* box, int64_t n, float eps, fx_stream_t stream_) {
  FX_CHECK_ARG(delta && ref && box && n > 0 && eps > 0.0f);
  hipLaunchKernelGGL(box_refine_kernel, dim3(ew_grid(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), (const bf16_t*)delta, ref, box, n, eps);
  return fx_launch_status();
}

extern "C" int fx_box_refine_bwd_f32(const float* grad_box, const float* box, const float* ref, void* d_delta, float* d_ref, int64_t n, float eps,
                                     fx_stream_t stream_) {
  FX_CHECK_ARG(grad_box && box && ref && d_delta && n > 0 && eps > 0.0f);
  hipLaunchKernelGGL(box_refine_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), grad_box, box, ref,
                     (bf16_t*)d_delta, d_ref, n, eps);
  return fx_launch_status();
}
