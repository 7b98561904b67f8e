// Mask-classification training criterion of the MaskFormer / BiSeNetFormer families, values and gradients:
//   point_sample                                   focoos/nn/layers/point_rend.py:29-52 (F.grid_sample bilinear / zeros / align_corners=False)
//   MaskHungarianMatcher.memory_efficient_forward  focoos/models/fai_mf/loss.py:661-723 (cost blocks; the assignment itself is fx_lsa_f32)
//   SetCriterion.loss_labels (ce_loss) / loss_masks :411-431, :463-523 with get_uncertain_point_coords_with_randomness point_rend.py:73-128
// The reference draws its sample points with torch.rand inside these functions; here the uniform draws are INPUTS (the host
// wrapper generates them on the device), so results are a pure function of the arguments.  fp32 arithmetic like the reference
// (autocast is disabled there, loss.py:702), float64 for the final reductions, fixed summation order -> deterministic.
#include "common.h"
int fx_tune(const char* env_name, int default_value);   // runtime.hip

// ATen's grid_sampler_compute_source_index (align_corners=False) applied to the wrapper's `2 * c - 1`: ((g + 1) * size - 1) / 2.
__device__ __forceinline__ float ps_unnormalize(float c, int size) {
#pragma clang fp contract(off)
  const float g = 2.0f * c - 1.0f;
  return ((g + 1.0f) * (float)size - 1.0f) / 2.0f;
}

template <typename T>
__device__ __forceinline__ float ps_load(const T* p, int H, int W, int y, int x) {
  return (y >= 0 && y < H && x >= 0 && x < W) ? (float)p[(int64_t)y * W + x] : 0.0f;
}

// bilinear sample with zero padding; weights as in ATen's CPU kernel: w = ix - floor(ix), e = 1 - w, n = iy - floor(iy), s = 1 - n;
// out = nw_val * (e*s) + ne_val * (w*s) + sw_val * (e*n) + se_val * (w*n)
template <typename T>
__device__ __forceinline__ float ps_sample(const T* p, int H, int W, float cx, float cy) {
#pragma clang fp contract(off)
  const float ix = ps_unnormalize(cx, W), iy = ps_unnormalize(cy, H);
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy;
  const float w = ix - fx, e = 1.0f - w, n = iy - fy, s = 1.0f - n;
  float acc = ps_load(p, H, W, y0, x0) * (e * s);
  acc += ps_load(p, H, W, y0, x0 + 1) * (w * s);
  acc += ps_load(p, H, W, y0 + 1, x0) * (e * n);
  acc += ps_load(p, H, W, y0 + 1, x0 + 1) * (w * n);
  return acc;
}

template <typename T>
__global__ __launch_bounds__(256) void point_sample_kernel(const T* __restrict__ src, int H, int W, const int32_t* __restrict__ src_index,
                                                           const float* __restrict__ coords, const int32_t* __restrict__ coord_index,
                                                           float* __restrict__ out, int P) {
  const int r = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const T* m = src + (int64_t)(src_index ? src_index[r] : r) * H * W;
  const float* c = coords + ((int64_t)(coord_index ? coord_index[r] : r) * P + p) * 2;
  out[(int64_t)r * P + p] = ps_sample(m, H, W, c[0], c[1]);
}

extern "C" int fx_point_sample_f32(const void* src, int src_is_u8, int H, int W, const int32_t* src_index, const float* coords,
                                   const int32_t* coord_index, float* out, int R, int P, fx_stream_t stream_) {
  FX_CHECK_ARG(src && coords && out && H > 0 && W > 0 && R > 0 && P > 0);
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  dim3 grid((P + 255) / 256, R);
  if (src_is_u8)
    hipLaunchKernelGGL(point_sample_kernel<uint8_t>, grid, dim3(256), 0, stream, (const uint8_t*)src, H, W, src_index, coords, coord_index, out, P);
  else
    hipLaunchKernelGGL(point_sample_kernel<float>, grid, dim3(256), 0, stream, (const float*)src, H, W, src_index, coords, coord_index, out, P);
  return fx_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Matching cost of one prediction set.  One workgroup per (image, query): the query's P sampled logits sit in LDS; wave w
// reduces targets t = w, w+4, ... over the points.  With pos = softplus(-x), neg = softplus(x) (binary_cross_entropy_with_logits
// against ones / zeros):  sum_p pos*tgt + neg*(1-tgt) = sum_p neg - sum_p x*tgt.
__device__ __forceinline__ float softplus_f(float x) { return fmaxf(x, 0.0f) + log1pf(__expf(-fabsf(x))); }

__global__ __launch_bounds__(256) void mask_match_cost_kernel(const float* __restrict__ logits, int ldl, const float* __restrict__ pred_pts,
                                                              const float* __restrict__ tgt_pts, const int32_t* __restrict__ tgt_labels,
                                                              const int32_t* __restrict__ tgt_offsets, int Q, int K, int P, int Tmax,
                                                              float w_class, float w_mask, float w_dice, int cls_sigmoid,
                                                              float* __restrict__ cost) {
  extern __shared__ float xs[];   // [P]
  __shared__ float red[2][4];
  __shared__ float s_norm[2];     // max logit, 1 / sum exp
  const int b = blockIdx.y, q = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t0 = tgt_offsets[b], T = tgt_offsets[b + 1] - t0;
  float* crow = cost + ((int64_t)b * Q + q) * Tmax;
  for (int t = T + threadIdx.x; t < Tmax; t += 256) crow[t] = 0.0f;
  if (T == 0) return;
  const float* xp = pred_pts + ((int64_t)b * Q + q) * P;
  float sp = 0.0f, sg = 0.0f;
  for (int p = threadIdx.x; p < P; p += 256) {
    const float x = xp[p];
    xs[p] = x;
    sp += softplus_f(x);
    sg += fx_sigmoid(x);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sp += __shfl_xor(sp, o, 64), sg += __shfl_xor(sg, o, 64);
  if (lane == 0) red[0][wave] = sp, red[1][wave] = sg;
  // class probabilities: softmax over the K+1 logits of this query (sigmoid when cls_sigmoid)
  const float* lp = logits + ((int64_t)b * Q + q) * ldl;
  if (wave == 0 && !cls_sigmoid) {
    float mx = -INFINITY;
    for (int c = lane; c <= K; c += 64) mx = fmaxf(mx, lp[c]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float se = 0.0f;
    for (int c = lane; c <= K; c += 64) se += __expf(lp[c] - mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o, 64);
    if (lane == 0) s_norm[0] = mx, s_norm[1] = 1.0f / se;
  }
  __syncthreads();
  const float SP = red[0][0] + red[0][1] + red[0][2] + red[0][3], SG = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  for (int t = wave; t < T; t += 4) {
    const float* tp = tgt_pts + (int64_t)(t0 + t) * P;
    float a = 0.0f, bs = 0.0f, st = 0.0f;
    for (int p = lane; p < P; p += 64) {
      const float tv = tp[p], x = xs[p];
      a = fmaf(x, tv, a);
      bs = fmaf(fx_sigmoid(x), tv, bs);
      st += tv;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64), bs += __shfl_xor(bs, o, 64), st += __shfl_xor(st, o, 64);
    if (lane == 0) {
      const int lab = tgt_labels[t0 + t];
      const float prob = cls_sigmoid ? fx_sigmoid(lp[lab]) : __expf(lp[lab] - s_norm[0]) * s_norm[1];
      const float c_mask = (SP - a) / (float)P;
      const float c_dice = 1.0f - (2.0f * bs + 1.0f) / (SG + st + 1.0f);
      crow[t] = w_mask * c_mask + w_class * (-prob) + w_dice * c_dice;
    }
  }
}

extern "C" int fx_mask_match_cost_f32(const float* logits, int ldl, const float* pred_pts, const float* tgt_pts, const int32_t* tgt_labels,
                                      const int32_t* tgt_offsets, int B, int Q, int K, int P, int Tmax, float w_class, float w_mask, float w_dice,
                                      int cls_sigmoid, float* cost, fx_stream_t stream_) {
  FX_CHECK_ARG(logits && pred_pts && tgt_pts && tgt_labels && tgt_offsets && cost && B > 0 && Q > 0 && K > 0 && P > 0 && Tmax > 0 && ldl >= K + 1);
  if ((size_t)P * sizeof(float) > 150 * 1024) return FX_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(mask_match_cost_kernel, dim3(Q, B), dim3(256), (size_t)P * sizeof(float), reinterpret_cast<hipStream_t>(stream_), logits, ldl,
                     pred_pts, tgt_pts, tgt_labels, tgt_offsets, Q, K, P, Tmax, w_class, w_mask, w_dice, cls_sigmoid, cost);
  return fx_launch_status();
}

// The same cost blocks as two small fp32 GEMMs (round 4): per image  A[q][t] = sum_p x[q][p] tgt[t][p],  S[q][t] = sum_p sigmoid(x[q][p]) tgt[t][p]
// over the shared points.  The kernel above is one workgroup per (image, query) that streams every target's P points for ITS query - 800
// workgroups x T x 50 KB = 600 MB of L2 reads per prediction set, a sigmoid per (point, target), 245 us.  Here a workgroup owns 16 queries x
// 16 targets x one slice of the points: 64-point tiles of x (+ sigmoid, softplus: computed once per point) and of the targets go through
// LDS, thread (q, t) accumulates its three sums; slices are reduced in a fixed order by the finishing kernel (deterministic).
#define MMC_QB 16
#define MMC_TB 16
#define MMC_PC 64
#define MMC_NTB 4          // target blocks held in registers: Tmax <= 64 (above: the kernel above)
__global__ __launch_bounds__(256) void mask_match_partial_kernel(const float* __restrict__ pred_pts, const float* __restrict__ tgt_pts,
                                                                 const int32_t* __restrict__ tgt_offsets, int Q, int P, int Tmax, int nsplit,
                                                                 float* __restrict__ part /*[B][nsplit][Q][3*Tmax + 2]*/) {
  __shared__ float xsT[MMC_QB][MMC_PC + 1], sgT[MMC_QB][MMC_PC + 1], spT[MMC_QB][MMC_PC + 1], tgT[MMC_TB][MMC_PC + 1];
  const int b = blockIdx.y, qb = blockIdx.x, sp_i = blockIdx.z;
  const int t0 = tgt_offsets[b], T = tgt_offsets[b + 1] - t0;
  const int tid = threadIdx.x, qi = tid >> 4, ti = tid & 15;
  const int q = qb * MMC_QB + qi;
  const int row = 3 * Tmax + 2;
  float* out = part + (((int64_t)b * nsplit + sp_i) * Q + q) * row;
  if (T == 0) return;
  const int per = ((P + nsplit - 1) / nsplit + MMC_PC - 1) / MMC_PC * MMC_PC;
  const int p_lo = sp_i * per, p_hi = min(P, p_lo + per);
  float a[MMC_NTB], bs[MMC_NTB], st[MMC_NTB], SP = 0.0f, SG = 0.0f;
#pragma unroll
  for (int k = 0; k < MMC_NTB; ++k) a[k] = bs[k] = st[k] = 0.0f;
  const int ntb = (T + MMC_TB - 1) / MMC_TB;
  for (int p0 = p_lo; p0 < p_hi; p0 += MMC_PC) {
    __syncthreads();
    for (int e = tid; e < MMC_QB * MMC_PC; e += 256) {          // the 16 queries' points of this tile: logit, sigmoid, softplus
      const int r = e / MMC_PC, c = e - r * MMC_PC;
      const int qq = qb * MMC_QB + r, p = p0 + c;
      float x = 0.0f, sx = 0.0f, px = 0.0f;
      if (qq < Q && p < p_hi) {
        x = pred_pts[((int64_t)b * Q + qq) * P + p];
        sx = fx_sigmoid(x);
        px = softplus_f(x);
      }
      xsT[r][c] = x; sgT[r][c] = sx; spT[r][c] = px;
    }
#pragma unroll
    for (int k = 0; k < MMC_NTB; ++k) {
      if (k >= ntb) break;
      if (k > 0) __syncthreads();
      for (int e = tid; e < MMC_TB * MMC_PC; e += 256) {
        const int r = e / MMC_PC, c = e - r * MMC_PC;
        const int t = k * MMC_TB + r, p = p0 + c;
        tgT[r][c] = (t < T && p < p_hi) ? tgt_pts[(int64_t)(t0 + t) * P + p] : 0.0f;
      }
      __syncthreads();
      float aa = a[k], bb = bs[k], ss = st[k];
#pragma unroll 8
      for (int c = 0; c < MMC_PC; ++c) {
        const float tv = tgT[ti][c];
        aa = fmaf(xsT[qi][c], tv, aa);
        bb = fmaf(sgT[qi][c], tv, bb);
        ss += tv;
      }
      a[k] = aa; bs[k] = bb; st[k] = ss;
      if (k == 0 && ti == 0) {
#pragma unroll 8
        for (int c = 0; c < MMC_PC; ++c) SP += spT[qi][c], SG += sgT[qi][c];
      }
    }
  }
  if (q < Q) {
#pragma unroll
    for (int k = 0; k < MMC_NTB; ++k) {
      const int t = k * MMC_TB + ti;
      if (t < T) out[3 * t] = a[k], out[3 * t + 1] = bs[k], out[3 * t + 2] = st[k];
    }
    if (ti == 0) out[3 * Tmax] = SP, out[3 * Tmax + 1] = SG;
  }
}

__global__ __launch_bounds__(256) void mask_match_finish_kernel(const float* __restrict__ logits, int ldl, const float* __restrict__ part, int nsplit,
                                                                const int32_t* __restrict__ tgt_labels, const int32_t* __restrict__ tgt_offsets, int Q,
                                                                int K, int P, int Tmax, float w_class, float w_mask, float w_dice, int cls_sigmoid,
                                                                float* __restrict__ cost) {
  // one wave per (image, query): class probabilities, then lane t finishes target t
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y, q = blockIdx.x * 4 + wave;
  if (q >= Q) return;
  const int t0 = tgt_offsets[b], T = tgt_offsets[b + 1] - t0;
  float* crow = cost + ((int64_t)b * Q + q) * Tmax;
  for (int t = T + lane; t < Tmax; t += 64) crow[t] = 0.0f;
  if (T == 0) return;
  const float* lp = logits + ((int64_t)b * Q + q) * ldl;
  float mx = 0.0f, inv = 1.0f;
  if (!cls_sigmoid) {
    mx = -INFINITY;
    for (int c = lane; c <= K; c += 64) mx = fmaxf(mx, lp[c]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float se = 0.0f;
    for (int c = lane; c <= K; c += 64) se += __expf(lp[c] - mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o, 64);
    inv = 1.0f / se;
  }
  const int row = 3 * Tmax + 2;
  float SP = 0.0f, SG = 0.0f;
  for (int s = 0; s < nsplit; ++s) {
    const float* pr = part + (((int64_t)b * nsplit + s) * Q + q) * row;
    SP += pr[3 * Tmax];
    SG += pr[3 * Tmax + 1];
  }
  for (int t = lane; t < T; t += 64) {
    float a = 0.0f, bs = 0.0f, st = 0.0f;
    for (int s = 0; s < nsplit; ++s) {
      const float* pr = part + (((int64_t)b * nsplit + s) * Q + q) * row;
      a += pr[3 * t]; bs += pr[3 * t + 1]; st += pr[3 * t + 2];
    }
    const int lab = tgt_labels[t0 + t];
    const float prob = cls_sigmoid ? fx_sigmoid(lp[lab]) : __expf(lp[lab] - mx) * inv;
    const float c_mask = (SP - a) / (float)P;
    const float c_dice = 1.0f - (2.0f * bs + 1.0f) / (SG + st + 1.0f);
    crow[t] = w_mask * c_mask + w_class * (-prob) + w_dice * c_dice;
  }
}

static int mmc_splits(int B, int Q) {
  const int wgs = B * ((Q + MMC_QB - 1) / MMC_QB);
  int s = (512 + wgs - 1) / wgs;     // ~2 workgroups per CU
  return s < 1 ? 1 : (s > 16 ? 16 : s);
}

extern "C" size_t fx_mask_match_cost_workspace_bytes(int B, int Q, int Tmax) {
  if (B <= 0 || Q <= 0 || Tmax <= 0 || Tmax > MMC_NTB * MMC_TB) return 0;
  return (size_t)B * mmc_splits(B, Q) * Q * (3 * (size_t)Tmax + 2) * sizeof(float);
}

extern "C" int fx_mask_match_cost_f32(const float* logits, int ldl, const float* pred_pts, const float* tgt_pts, const int32_t* tgt_labels,
                                      const int32_t* tgt_offsets, int B, int Q, int K, int P, int Tmax, float w_class, float w_mask, float w_dice,
                                      int cls_sigmoid, float* cost, fx_stream_t stream_);

extern "C" int fx_mask_match_cost_ws_f32(const float* logits, int ldl, const float* pred_pts, const float* tgt_pts, const int32_t* tgt_labels,
                                         const int32_t* tgt_offsets, int B, int Q, int K, int P, int Tmax, float w_class, float w_mask, float w_dice,
                                         int cls_sigmoid, float* cost, void* workspace, size_t workspace_bytes, fx_stream_t stream_) {
  const size_t need = fx_mask_match_cost_workspace_bytes(B, Q, Tmax);
  static const int tiled = fx_tune("FX_MASK_COST_TILED", 1);
  if (!tiled || !workspace || need == 0 || workspace_bytes < need || ((uintptr_t)workspace % 16) != 0)
    return fx_mask_match_cost_f32(logits, ldl, pred_pts, tgt_pts, tgt_labels, tgt_offsets, B, Q, K, P, Tmax, w_class, w_mask, w_dice, cls_sigmoid, cost, stream_);
  FX_CHECK_ARG(logits && pred_pts && tgt_pts && tgt_labels && tgt_offsets && cost && B > 0 && Q > 0 && K > 0 && P > 0 && Tmax > 0 && ldl >= K + 1);
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const int ns = mmc_splits(B, Q);
  hipLaunchKernelGGL(mask_match_partial_kernel, dim3((Q + MMC_QB - 1) / MMC_QB, B, ns), dim3(256), 0, stream, pred_pts, tgt_pts, tgt_offsets, Q, P, Tmax, ns,
                     reinterpret_cast<float*>(workspace));
  hipLaunchKernelGGL(mask_match_finish_kernel, dim3((Q + 3) / 4, B), dim3(256), 0, stream, logits, ldl, reinterpret_cast<const float*>(workspace), ns, tgt_labels,
                     tgt_offsets, Q, K, P, Tmax, w_class, w_mask, w_dice, cls_sigmoid, cost);
  return fx_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Losses of one prediction set.
// (1) labels: target class of (b, q) = label of its matched target, else K ("no object"); loss_ce = sum w[y] * nll / sum w[y]
//     (F.cross_entropy with class weights, weight eos_coef on K).  One wave per query row, partial sums in float64.
__global__ __launch_bounds__(256) void mask_label_ce_kernel(const float* __restrict__ logits, int ldl, const int32_t* __restrict__ tgt_labels,
                                                            const int32_t* __restrict__ tgt_offsets, const int32_t* __restrict__ pred_idx,
                                                            const int32_t* __restrict__ tgt_idx, int n_rows, int Q, int K, float eos_coef,
                                                            double* __restrict__ partial /*[B*Q][2]*/) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int b = row / Q, q = row - b * Q;
  const int t0 = tgt_offsets[b], T = tgt_offsets[b + 1] - t0;
  int y = K;
  for (int i = lane; i < T; i += 64)
    if (pred_idx[t0 + i] == q) y = tgt_labels[t0 + tgt_idx[t0 + i]];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) y = min(y, __shfl_xor(y, o, 64));   // at most one lane found a match; K is the maximum
  const float* lp = logits + (int64_t)row * ldl;
  float mx = -INFINITY;
  for (int c = lane; c <= K; c += 64) mx = fmaxf(mx, lp[c]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float se = 0.0f;
  for (int c = lane; c <= K; c += 64) se += expf(lp[c] - mx);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o, 64);
  if (lane == 0) {
    const float nll = (mx + logf(se)) - lp[y];
    const float w = y == K ? eos_coef : 1.0f;
    partial[(int64_t)row * 2 + 0] = (double)(w * nll);
    partial[(int64_t)row * 2 + 1] = (double)w;
  }
}

// (2) masks: one workgroup per matched (prediction, target) pair.  Importance sampling: of the n_over uniformly drawn points keep
// the k = num_points - n_extra with the smallest |logit| (torch.topk of -|logit|; ties -> lowest index), found by a 4-pass
// radix select on the bit pattern of |x| with the samples recomputed per pass (4 taps each) instead of stored; then the
// BCE / dice sums over those points plus the n_extra uniformly drawn ones, target values sampled bilinearly at the same points.
// Transpose of ps_sample: the gradient v of a sampled value goes to its (in-bounds) taps with the same weights.
__device__ __forceinline__ void ps_scatter(float* g, int H, int W, float cx, float cy, float v) {
#pragma clang fp contract(off)
  const float ix = ps_unnormalize(cx, W), iy = ps_unnormalize(cy, H);
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy;
  const float w = ix - fx, e = 1.0f - w, n = iy - fy, s = 1.0f - n;
  const bool xa = x0 >= 0 && x0 < W, xb = x0 + 1 >= 0 && x0 + 1 < W, ya = y0 >= 0 && y0 < H, yb = y0 + 1 >= 0 && y0 + 1 < H;
  if (ya && xa) unsafeAtomicAdd(g + (int64_t)y0 * W + x0, v * (e * s));
  if (ya && xb) unsafeAtomicAdd(g + (int64_t)y0 * W + x0 + 1, v * (w * s));
  if (yb && xa) unsafeAtomicAdd(g + (int64_t)(y0 + 1) * W + x0, v * (e * n));
  if (yb && xb) unsafeAtomicAdd(g + (int64_t)(y0 + 1) * W + x0 + 1, v * (w * n));
}

// Forward: one workgroup per matched pair.  The n_over candidate logits are sampled ONCE into the workspace (xs; 150 KB per pair at the
// registry size: L2-resident), the four radix passes and the selection read them back with coalesced loads, and the selection is
// recorded (sel[i] = 1) so that the backward neither re-samples the candidates nor repeats the select.
#define MPL_THREADS 1024
template <typename TT>
__global__ __launch_bounds__(MPL_THREADS) void mask_point_loss_kernel(const float* __restrict__ pred_masks, int h, int w, const TT* __restrict__ tgt_masks,
                                                                      int H, int W, const int32_t* __restrict__ tgt_offsets, int B, int Q,
                                                                      const int32_t* __restrict__ pred_idx, const int32_t* __restrict__ tgt_idx,
                                                                      const float* __restrict__ rand_over, int n_over,
                                                                      const float* __restrict__ rand_extra, int n_extra, int k_imp,
                                                                      double* __restrict__ rows /*[N][4]: bce sum, sum s*t, sum s, sum t*/,
                                                                      float* __restrict__ xs_all /*[N][n_over]*/, uint8_t* __restrict__ sel_all /*[N][n_over]*/) {
  __shared__ unsigned hist[256];
  __shared__ unsigned s_prefix, s_need;
  __shared__ unsigned s_scan[MPL_THREADS];
  __shared__ double red[4][MPL_THREADS / 64];
  const int n = blockIdx.x, tid = threadIdx.x;
  int b = 0;
  while (b + 1 < B && tgt_offsets[b + 1] <= n) ++b;
  const float* pm = pred_masks + ((int64_t)b * Q + pred_idx[n]) * h * w;
  const TT* tm = tgt_masks + (int64_t)(tgt_offsets[b] + tgt_idx[n]) * H * W;
  const float* ro = rand_over + (int64_t)n * n_over * 2;
  float* xs = xs_all + (int64_t)n * n_over;
  uint8_t* sel = sel_all + (int64_t)n * n_over;
  if (k_imp > 0) {
    for (int i = tid; i < n_over; i += MPL_THREADS) xs[i] = ps_sample(pm, h, w, ro[2 * i], ro[2 * i + 1]);
    __syncthreads();   // the workgroup reads back its own global writes (same CU: L1 write-through + barrier)
  }
  // ---- radix select: key of the k_imp-th smallest |x| (prefix), and how many keys equal to it are needed
  unsigned prefix = 0, need = (unsigned)k_imp;
  if (k_imp > 0) {
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      for (int i = tid; i < 256; i += MPL_THREADS) hist[i] = 0;
      __syncthreads();
      const unsigned hi_mask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
      for (int i = tid; i < n_over; i += MPL_THREADS) {
        const unsigned key = __float_as_uint(fabsf(xs[i]));
        if ((key & hi_mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (tid == 0) {
        unsigned acc = 0, d = 0;
        for (; d < 256; ++d) {
          if (acc + hist[d] >= need) break;
          acc += hist[d];
        }
        s_prefix = prefix | (d << shift);
        s_need = need - acc;
      }
      __syncthreads();
      prefix = s_prefix;
      need = s_need;
      __syncthreads();
    }
  }
  // ---- selected points: key < prefix, plus the first `need` (index order) with key == prefix
  double bce = 0.0, st = 0.0, ss = 0.0, tt = 0.0;
  const int chunk = (n_over + MPL_THREADS - 1) / MPL_THREADS;   // contiguous index ranges so tie ranks follow the index order
  const int i0 = tid * chunk, i1 = min(n_over, i0 + chunk);
  unsigned ties = 0;
  if (k_imp > 0)
    for (int i = i0; i < i1; ++i) ties += __float_as_uint(fabsf(xs[i])) == prefix;
  s_scan[tid] = ties;
  __syncthreads();
  for (int o = 1; o < MPL_THREADS; o <<= 1) {   // inclusive Hillis-Steele scan
    const unsigned v = tid >= o ? s_scan[tid - o] : 0u;
    __syncthreads();
    s_scan[tid] += v;
    __syncthreads();
  }
  unsigned tie_rank = s_scan[tid] - ties;   // ties before this thread's range
  auto add_point = [&](float x, float t) {
    const float s = 1.0f / (1.0f + expf(-x));
    bce += (double)(fmaxf(x, 0.0f) - x * t + log1pf(expf(-fabsf(x))));
    st += (double)(s * t);
    ss += (double)s;
    tt += (double)t;
  };
  if (k_imp > 0)
    for (int i = i0; i < i1; ++i) {
      const float x = xs[i];
      const unsigned key = __float_as_uint(fabsf(x));
      bool take = key < prefix;
      if (key == prefix) take = tie_rank++ < need;
      sel[i] = take ? 1 : 0;
      if (take) add_point(x, ps_sample(tm, H, W, ro[2 * i], ro[2 * i + 1]));
    }
  const float* re = rand_extra + (int64_t)n * n_extra * 2;
  for (int i = tid; i < n_extra; i += MPL_THREADS) {
    const float cx = re[2 * i], cy = re[2 * i + 1];
    add_point(ps_sample(pm, h, w, cx, cy), ps_sample(tm, H, W, cx, cy));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    bce += __shfl_xor(bce, o, 64); st += __shfl_xor(st, o, 64); ss += __shfl_xor(ss, o, 64); tt += __shfl_xor(tt, o, 64);
  }
  const int lane = tid & 63, wave = tid >> 6;
  if (lane == 0) red[0][wave] = bce, red[1][wave] = st, red[2][wave] = ss, red[3][wave] = tt;
  __syncthreads();
  if (tid < 4) {
    double a = 0.0;
    for (int i = 0; i < MPL_THREADS / 64; ++i) a += red[tid][i];
    rows[(int64_t)n * 4 + tid] = a;
  }
}

// Backward: d loss / d mask logit of every selected point, g_bce * (sigmoid(x) - t) + g_dice * s (1 - s) * d dice / d s with
// dice = 1 - (2 st + 1) / (ss + tt + 1) from the forward's rows[n], scattered to the four taps of the point in the pair's [h,w] plane of
// dmasks (zero-initialised by the caller; the sample coordinates carry no gradient: the reference draws / selects them under no_grad,
// loss.py:487-497).  blockIdx.y splits a pair's points over several workgroups (the forward left xs / sel in the workspace).
__global__ __launch_bounds__(256) void mask_point_loss_bwd_kernel(const float* __restrict__ pred_masks, int h, int w, const void* __restrict__ tgt_masks,
                                                                  int tgt_is_u8, int H, int W, const int32_t* __restrict__ tgt_offsets, int B, int Q,
                                                                  const int32_t* __restrict__ pred_idx, const int32_t* __restrict__ tgt_idx,
                                                                  const float* __restrict__ rand_over, int n_over,
                                                                  const float* __restrict__ rand_extra, int n_extra, int k_imp,
                                                                  const double* __restrict__ rows, const float* __restrict__ xs_all,
                                                                  const uint8_t* __restrict__ sel_all, float c_bce, float c_dice,
                                                                  const float* __restrict__ g3, float* __restrict__ dmasks) {
  const int n = blockIdx.x;
  int b = 0;
  while (b + 1 < B && tgt_offsets[b + 1] <= n) ++b;
  const int q = pred_idx[n];
  const float* pm = pred_masks + ((int64_t)b * Q + q) * h * w;
  const int64_t toff = (int64_t)(tgt_offsets[b] + tgt_idx[n]) * H * W;
  const uint8_t* tm8 = reinterpret_cast<const uint8_t*>(tgt_masks) + toff;
  const float* tmf = reinterpret_cast<const float*>(tgt_masks) + toff;
  float* gplane = dmasks + ((int64_t)b * Q + q) * h * w;
  const double ST = rows[(int64_t)n * 4 + 1], D = rows[(int64_t)n * 4 + 2] + rows[(int64_t)n * 4 + 3] + 1.0;
  const float dice_a = (float)(-2.0 / D), dice_b = (float)((2.0 * ST + 1.0) / (D * D));   // d dice / d s_p = dice_a * t_p + dice_b
  const float g_bce = g3[1] * c_bce, g_dice = g3[2] * c_dice;
  auto point = [&](float x, float cx, float cy) {
    const float t = tgt_is_u8 ? ps_sample(tm8, H, W, cx, cy) : ps_sample(tmf, H, W, cx, cy);
    const float s = 1.0f / (1.0f + expf(-x));
    ps_scatter(gplane, h, w, cx, cy, g_bce * (s - t) + g_dice * (s * (1.0f - s)) * (dice_a * t + dice_b));
  };
  const int stride = gridDim.y * 256, t0 = blockIdx.y * 256 + threadIdx.x;
  if (k_imp > 0) {
    const float* ro = rand_over + (int64_t)n * n_over * 2;
    const float* xs = xs_all + (int64_t)n * n_over;
    const uint8_t* sel = sel_all + (int64_t)n * n_over;
    for (int i = t0; i < n_over; i += stride)
      if (sel[i]) point(xs[i], ro[2 * i], ro[2 * i + 1]);
  }
  const float* re = rand_extra + (int64_t)n * n_extra * 2;
  for (int i = t0; i < n_extra; i += stride) {
    const float cx = re[2 * i], cy = re[2 * i + 1];
    point(ps_sample(pm, h, w, cx, cy), cx, cy);
  }
}

// The same gradient with the plane accumulated in LDS (round 4): the kernel above spends its time in four scattered fp32 atomics per point on
// L2 (BiSeNetFormer 8 x ~10 pairs x 12 544 points: 200 us; caching the target values changed nothing).  Here a workgroup owns a band of rows
// of ONE pair's [h, w] plane (<= 64 KiB of fp32: the whole 128 x 128 plane of BiSeNetFormer at 1024^2, two or three bands of MaskFormer's
// 200 x 200), walks all the pair's points, adds the taps that fall into its band with LDS atomics and stores the band once (a prediction
// is matched to at most one target, so its plane has ONE writer: plain stores over the caller's zeros).
#define MPB_THREADS 1024
#define MPB_LDS_FLOATS 16384
__global__ __launch_bounds__(MPB_THREADS) void mask_point_loss_bwd_lds_kernel(const float* __restrict__ pred_masks, int h, int w, const void* __restrict__ tgt_masks,
                                                                              int tgt_is_u8, int H, int W, const int32_t* __restrict__ tgt_offsets, int B, int Q,
                                                                              const int32_t* __restrict__ pred_idx, const int32_t* __restrict__ tgt_idx,
                                                                              const float* __restrict__ rand_over, int n_over, const float* __restrict__ rand_extra,
                                                                              int n_extra, int k_imp, const double* __restrict__ rows,
                                                                              const float* __restrict__ xs_all, const uint8_t* __restrict__ sel_all, float c_bce,
                                                                              float c_dice, const float* __restrict__ g3, float* __restrict__ dmasks, int band_rows) {
  extern __shared__ float plane[];   // [band_rows][w]
  const int n = blockIdx.x, r0 = blockIdx.y * band_rows, r1 = min(h, r0 + band_rows), tid = threadIdx.x;
  int b = 0;
  while (b + 1 < B && tgt_offsets[b + 1] <= n) ++b;
  const int q = pred_idx[n];
  const float* pm = pred_masks + ((int64_t)b * Q + q) * h * w;
  const int64_t toff = (int64_t)(tgt_offsets[b] + tgt_idx[n]) * H * W;
  const uint8_t* tm8 = reinterpret_cast<const uint8_t*>(tgt_masks) + toff;
  const float* tmf = reinterpret_cast<const float*>(tgt_masks) + toff;
  const int nband = (r1 - r0) * w;
  for (int i = tid; i < nband; i += MPB_THREADS) plane[i] = 0.0f;
  __syncthreads();
  const double ST = rows[(int64_t)n * 4 + 1], D = rows[(int64_t)n * 4 + 2] + rows[(int64_t)n * 4 + 3] + 1.0;
  const float dice_a = (float)(-2.0 / D), dice_b = (float)((2.0 * ST + 1.0) / (D * D));
  const float g_bce = g3[1] * c_bce, g_dice = g3[2] * c_dice;
  auto point = [&](bool have_x, float x_in, float cx, float cy) {
    // taps of the point in the prediction plane (ps_scatter's arithmetic); nothing to do if none lies in this band
    const float ix = ps_unnormalize(cx, w), iy = ps_unnormalize(cy, h);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const bool ya = y0 >= r0 && y0 < r1, yb = y0 + 1 >= r0 && y0 + 1 < r1;
    if (!ya && !yb) return;
    const float x = have_x ? x_in : ps_sample(pm, h, w, cx, cy);
    const float t = tgt_is_u8 ? ps_sample(tm8, H, W, cx, cy) : ps_sample(tmf, H, W, cx, cy);
    const float s = 1.0f / (1.0f + expf(-x));
    const float v = g_bce * (s - t) + g_dice * (s * (1.0f - s)) * (dice_a * t + dice_b);
    const float wx = ix - fx, ex = 1.0f - wx, ny = iy - fy, sy = 1.0f - ny;
    const bool xa = x0 >= 0 && x0 < w, xb = x0 + 1 >= 0 && x0 + 1 < w;
    if (ya && xa) atomicAdd(&plane[(y0 - r0) * w + x0], v * (ex * sy));
    if (ya && xb) atomicAdd(&plane[(y0 - r0) * w + x0 + 1], v * (wx * sy));
    if (yb && xa) atomicAdd(&plane[(y0 + 1 - r0) * w + x0], v * (ex * ny));
    if (yb && xb) atomicAdd(&plane[(y0 + 1 - r0) * w + x0 + 1], v * (wx * ny));
  };
  if (k_imp > 0) {
    const float* ro = rand_over + (int64_t)n * n_over * 2;
    const float* xs = xs_all + (int64_t)n * n_over;
    const uint8_t* sel = sel_all + (int64_t)n * n_over;
    for (int i = tid; i < n_over; i += MPB_THREADS)
      if (sel[i]) point(true, xs[i], ro[2 * i], ro[2 * i + 1]);
  }
  const float* re = rand_extra + (int64_t)n * n_extra * 2;
  for (int i = tid; i < n_extra; i += MPB_THREADS) point(false, 0.0f, re[2 * i], re[2 * i + 1]);
  __syncthreads();
  float* gband = dmasks + ((int64_t)b * Q + q) * h * w + (int64_t)r0 * w;
  for (int i = tid; i < nband; i += MPB_THREADS) gband[i] = plane[i];
}

__global__ __launch_bounds__(256) void mask_loss_final_kernel(const double* __restrict__ ce_partial, int n_rows, const double* __restrict__ rows, int N,
                                                              int num_points, float num_masks, float w_ce, float w_mask, float w_dice,
                                                              float* __restrict__ out3, double* __restrict__ wsum_out) {
  __shared__ double red[4][256];
  double a = 0.0, wsum = 0.0, lm = 0.0, ld = 0.0;
  for (int i = threadIdx.x; i < n_rows; i += 256) a += ce_partial[2 * i], wsum += ce_partial[2 * i + 1];
  for (int i = threadIdx.x; i < N; i += 256) {
    lm += rows[4 * i] / (double)num_points;
    ld += 1.0 - (2.0 * rows[4 * i + 1] + 1.0) / (rows[4 * i + 2] + rows[4 * i + 3] + 1.0);
  }
  red[0][threadIdx.x] = a; red[1][threadIdx.x] = wsum; red[2][threadIdx.x] = lm; red[3][threadIdx.x] = ld;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o)
      for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (wsum_out) *wsum_out = red[1][0];
    out3[0] = w_ce * (float)(red[0][0] / red[1][0]);
    out3[1] = w_mask * (float)(red[2][0] / (double)num_masks);
    out3[2] = w_dice * (float)(red[3][0] / (double)num_masks);
  }
}

// workspace: CE partials f64 [B*Q][2] | pair sums f64 [sum_T][4] | sum of the CE class weights f64 | candidate logits f32 [sum_T][n_over] |
// selection flags u8 [sum_T][n_over]
extern "C" size_t fx_mask_set_loss_workspace_bytes(int B, int Q, int sum_T, int n_over) {
  if (B <= 0 || Q <= 0 || sum_T < 0 || n_over < 0) return 0;
  return ((size_t)B * Q * 2 + (size_t)sum_T * 4 + 1) * sizeof(double) + (size_t)sum_T * n_over * 5 + 16;
}

extern "C" int fx_mask_set_loss_f32(const float* logits, int ldl, const float* pred_masks, int h, int w, const void* tgt_masks, int tgt_is_u8, int H,
                                    int W, const int32_t* tgt_labels, const int32_t* tgt_offsets, int sum_T, const int32_t* pred_idx,
                                    const int32_t* tgt_idx, const float* rand_over, int n_over, const float* rand_extra, int n_extra, int num_points,
                                    int B, int Q, int K, float eos_coef, float num_masks, float w_ce, float w_mask, float w_dice, void* workspace,
                                    size_t workspace_bytes, float* out3, fx_stream_t stream_) {
  FX_CHECK_ARG(logits && pred_masks && tgt_offsets && workspace && out3 && B > 0 && Q > 0 && K > 0 && ldl >= K + 1 && h > 0 && w > 0 && H > 0 && W > 0);
  FX_CHECK_ARG(sum_T >= 0 && num_points > 0 && n_extra >= 0 && n_extra <= num_points && n_over >= num_points - n_extra && num_masks > 0.0f);
  FX_CHECK_ARG(sum_T == 0 || (tgt_masks && tgt_labels && pred_idx && tgt_idx && (n_extra == 0 || rand_extra) && (num_points == n_extra || rand_over)));
  FX_CHECK_ARG(workspace_bytes >= fx_mask_set_loss_workspace_bytes(B, Q, sum_T, n_over) && ((uintptr_t)workspace % 8) == 0);
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  double* ce_partial = reinterpret_cast<double*>(workspace);
  double* rows = ce_partial + (size_t)B * Q * 2;
  hipLaunchKernelGGL(mask_label_ce_kernel, dim3((B * Q + 3) / 4), dim3(256), 0, stream, logits, ldl, tgt_labels, tgt_offsets, pred_idx, tgt_idx, B * Q, Q, K, eos_coef,
                     ce_partial);
  float* xs = reinterpret_cast<float*>(rows + (size_t)sum_T * 4 + 1);
  uint8_t* sel = reinterpret_cast<uint8_t*>(xs + (size_t)sum_T * n_over);
  if (sum_T > 0) {
    if (tgt_is_u8)
      hipLaunchKernelGGL(mask_point_loss_kernel<uint8_t>, dim3(sum_T), dim3(MPL_THREADS), 0, stream, pred_masks, h, w, (const uint8_t*)tgt_masks, H, W,
                         tgt_offsets, B, Q, pred_idx, tgt_idx, rand_over, n_over, rand_extra, n_extra, num_points - n_extra, rows, xs, sel);
    else
      hipLaunchKernelGGL(mask_point_loss_kernel<float>, dim3(sum_T), dim3(MPL_THREADS), 0, stream, pred_masks, h, w, (const float*)tgt_masks, H, W,
                         tgt_offsets, B, Q, pred_idx, tgt_idx, rand_over, n_over, rand_extra, n_extra, num_points - n_extra, rows, xs, sel);
  }
  hipLaunchKernelGGL(mask_loss_final_kernel, dim3(1), dim3(256), 0, stream, ce_partial, B * Q, rows, sum_T, num_points, num_masks, w_ce, w_mask, w_dice,
                     out3, rows + (size_t)sum_T * 4);
  return fx_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Gradients of the three losses of one prediction set (the backward of fx_mask_set_loss_f32 under upstream gradients g[0..2]):
//   dlogits[b,q,c] = g0 * w_ce * w[y] / sum_rows w[y] * (softmax_c - [c == y])        (F.cross_entropy with class weights, mean)
//   dmasks: see mask_point_loss_kernel<.., true>; planes of unmatched queries stay zero.
__global__ __launch_bounds__(256) void mask_label_ce_bwd_kernel(const float* __restrict__ logits, int ldl, const int32_t* __restrict__ tgt_labels,
                                                                const int32_t* __restrict__ tgt_offsets, const int32_t* __restrict__ pred_idx,
                                                                const int32_t* __restrict__ tgt_idx, int n_rows, int Q, int K, float eos_coef,
                                                                const double* __restrict__ wsum, const float* __restrict__ g, float w_ce,
                                                                float* __restrict__ dlogits, int lddl) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int b = row / Q, q = row - b * Q;
  const int t0 = tgt_offsets[b], T = tgt_offsets[b + 1] - t0;
  int y = K;
  for (int i = lane; i < T; i += 64)
    if (pred_idx[t0 + i] == q) y = tgt_labels[t0 + tgt_idx[t0 + i]];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) y = min(y, __shfl_xor(y, o, 64));
  const float* lp = logits + (int64_t)row * ldl;
  float mx = -INFINITY;
  for (int c = lane; c <= K; c += 64) mx = fmaxf(mx, lp[c]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float se = 0.0f;
  for (int c = lane; c <= K; c += 64) se += expf(lp[c] - mx);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o, 64);
  const float coef = g[0] * w_ce * (y == K ? eos_coef : 1.0f) / (float)wsum[0];
  float* dp = dlogits + (int64_t)row * lddl;
  for (int c = lane; c <= K; c += 64) dp[c] = coef * (expf(lp[c] - mx) / se - (c == y ? 1.0f : 0.0f));
}

extern "C" int fx_mask_set_loss_bwd_f32(const float* logits, int ldl, const float* pred_masks, int h, int w, const void* tgt_masks, int tgt_is_u8, int H,
                                        int W, const int32_t* tgt_labels, const int32_t* tgt_offsets, int sum_T, const int32_t* pred_idx,
                                        const int32_t* tgt_idx, const float* rand_over, int n_over, const float* rand_extra, int n_extra,
                                        int num_points, int B, int Q, int K, float eos_coef, float num_masks, float w_ce, float w_mask, float w_dice,
                                        const void* workspace, size_t workspace_bytes, const float* grad3, float* dlogits, int lddl, float* dmasks,
                                        fx_stream_t stream_) {
  FX_CHECK_ARG(logits && pred_masks && tgt_offsets && workspace && grad3 && dlogits && dmasks && B > 0 && Q > 0 && K > 0);
  FX_CHECK_ARG(ldl >= K + 1 && lddl >= K + 1 && h > 0 && w > 0 && H > 0 && W > 0 && sum_T >= 0 && num_points > 0 && n_extra >= 0 && n_extra <= num_points);
  FX_CHECK_ARG(n_over >= num_points - n_extra && num_masks > 0.0f);
  FX_CHECK_ARG(sum_T == 0 || (tgt_masks && tgt_labels && pred_idx && tgt_idx && (n_extra == 0 || rand_extra) && (num_points == n_extra || rand_over)));
  FX_CHECK_ARG(workspace_bytes >= fx_mask_set_loss_workspace_bytes(B, Q, sum_T, n_over) && ((uintptr_t)workspace % 8) == 0);
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const double* ce_partial = reinterpret_cast<const double*>(workspace);
  double* rows = const_cast<double*>(ce_partial) + (size_t)B * Q * 2;
  hipLaunchKernelGGL(mask_label_ce_bwd_kernel, dim3((B * Q + 3) / 4), dim3(256), 0, stream, logits, ldl, tgt_labels, tgt_offsets, pred_idx, tgt_idx, B * Q, Q,
                     K, eos_coef, rows + (size_t)sum_T * 4, grad3, w_ce, dlogits, lddl);
  if (sum_T > 0) {
    const float c_bce = w_mask / (num_masks * (float)num_points), c_dice = w_dice / num_masks;
    const float* xs = reinterpret_cast<const float*>(rows + (size_t)sum_T * 4 + 1);
    const uint8_t* sel = reinterpret_cast<const uint8_t*>(xs + (size_t)sum_T * n_over);
    static const int lds_on = fx_tune("FX_MPL_BWD_LDS", 1);
    if (lds_on && w <= MPB_LDS_FLOATS) {
      const int band_rows = min(h, MPB_LDS_FLOATS / w);
      const int nbands = (h + band_rows - 1) / band_rows;
      const size_t smem = (size_t)band_rows * w * sizeof(float);
      static size_t attr_smem = 0;
      if (smem > attr_smem) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(mask_point_loss_bwd_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
          return FX_ERR_RUNTIME;
        attr_smem = smem;
      }
      hipLaunchKernelGGL(mask_point_loss_bwd_lds_kernel, dim3(sum_T, nbands), dim3(MPB_THREADS), smem, stream, pred_masks, h, w, tgt_masks, tgt_is_u8, H, W,
                         tgt_offsets, B, Q, pred_idx, tgt_idx, rand_over, n_over, rand_extra, n_extra, num_points - n_extra, rows, xs, sel, c_bce, c_dice, grad3,
                         dmasks, band_rows);
      return fx_launch_status();
    }
    const int split = sum_T >= 256 ? 2 : (sum_T >= 64 ? 8 : 16);   // >= 512 workgroups
    hipLaunchKernelGGL(mask_point_loss_bwd_kernel, dim3(sum_T, split), dim3(256), 0, stream, pred_masks, h, w, tgt_masks, tgt_is_u8, H, W, tgt_offsets, B, Q,
                       pred_idx, tgt_idx, rand_over, n_over, rand_extra, n_extra, num_points - n_extra, rows, xs, sel, c_bce, c_dice, grad3, dmasks);
  }
  return fx_launch_status();
}
