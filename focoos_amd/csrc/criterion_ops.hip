// RT-DETR set-criterion kernels (gfx950): the matching cost, the Hungarian / linear-sum-assignment solve and the
// VFL + L1 + GIoU losses of one prediction set - SURVEY §8a rows A14/A15, "next" row N2 (GPU-resident Hungarian).
// The reference computes the cost on the device, copies it to the host and calls SciPy per image
// (fai_detr/modelling.py:746-750, "FIXME ... Can we use GPU?"); here everything stays in HBM and the index results
// are bit-identical to SciPy's (same algorithm, same float64 arithmetic, same tie rule).
#include "common.h"

// ------------------------------------------------------------------------------------------------
// Cost blocks C[b][q][t] = w_bbox * L1(box_q, tbox_t) + w_class * focal_cost(p[q][label_t]) - w_giou * GIoU(q, t)
// (modelling.py:717-745, use_focal_loss branch; box math utils/box.py:14-64).  One lane per (q, t) of an image.
__device__ __forceinline__ void cxcywh_to_xyxy(const float* b, float& x0, float& y0, float& x1, float& y1) {
  x0 = b[0] - 0.5f * b[2];
  y0 = b[1] - 0.5f * b[3];
  x1 = b[0] + 0.5f * b[2];
  y1 = b[1] + 0.5f * b[3];
}

__device__ __forceinline__ void iou_giou(float ax0, float ay0, float ax1, float ay1, float bx0, float by0, float bx1, float by1, float& iou,
                                         float& giou) {
  const float a1 = (ax1 - ax0) * (ay1 - ay0), a2 = (bx1 - bx0) * (by1 - by0);
  const float iw = fmaxf(fminf(ax1, bx1) - fmaxf(ax0, bx0), 0.0f), ih = fmaxf(fminf(ay1, by1) - fmaxf(ay0, by0), 0.0f);
  const float inter = iw * ih;
  const float uni = a1 + a2 - inter;
  iou = inter / uni;
  const float cw = fmaxf(fmaxf(ax1, bx1) - fminf(ax0, bx0), 0.0f), ch = fmaxf(fmaxf(ay1, by1) - fminf(ay0, by0), 0.0f);
  const float area = cw * ch;
  giou = iou - (area - uni) / (area + 1e-5f);
}

__device__ __forceinline__ float pow_gamma(float x, float gamma) { return gamma == 2.0f ? x * x : powf(x, gamma); }

__global__ __launch_bounds__(256) void match_cost_kernel(const float* __restrict__ logits, int ldl, const float* __restrict__ boxes,
                                                          const int32_t* __restrict__ tlabels, const float* __restrict__ tboxes,
                                                          const int32_t* __restrict__ toff, int Q, int Tmax, float wc, float wb, float wg, float alpha,
                                                          float gamma, float* __restrict__ cost) {
  const int b = blockIdx.y;
  const int t0 = toff[b], T = toff[b + 1] - t0;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Q * Tmax) return;
  const int q = i / Tmax, t = i - q * Tmax;
  float out = 0.0f;
  if (t < T) {
    const float x = logits[((int64_t)b * Q + q) * ldl + tlabels[t0 + t]];
    const float p = 1.0f / (1.0f + expf(-x));
    const float neg = (1.0f - alpha) * pow_gamma(p, gamma) * (-logf(1.0f - p + 1e-8f));
    const float pos = alpha * pow_gamma(1.0f - p, gamma) * (-logf(p + 1e-8f));
    const float* pb = boxes + ((int64_t)b * Q + q) * 4;
    const float* tb = tboxes + (int64_t)(t0 + t) * 4;
    const float l1 = fabsf(pb[0] - tb[0]) + fabsf(pb[1] - tb[1]) + fabsf(pb[2] - tb[2]) + fabsf(pb[3] - tb[3]);
    float ax0, ay0, ax1, ay1, bx0, by0, bx1, by1, iou, giou;
    cxcywh_to_xyxy(pb, ax0, ay0, ax1, ay1);
    cxcywh_to_xyxy(tb, bx0, by0, bx1, by1);
    iou_giou(ax0, ay0, ax1, ay1, bx0, by0, bx1, by1, iou, giou);
    out = wb * l1 + wc * (pos - neg) + wg * (-giou);
  }
  cost[((int64_t)b * Q + q) * Tmax + t] = out;
}

extern "C" int fx_detr_match_cost_f32(const float* logits, int ldl, const float* boxes, const int32_t* tgt_labels, const float* tgt_boxes,
                                      const int32_t* tgt_offsets, int B, int Q, int K, int Tmax, float w_class, float w_bbox, float w_giou, float alpha,
                                      float gamma, float* cost, fx_stream_t stream_) {
  FX_CHECK_ARG(logits && boxes && tgt_offsets && cost && B > 0 && Q > 0 && K > 0 && Tmax >= 0 && ldl >= K);
  if (Tmax == 0) return FX_OK;
  FX_CHECK_ARG(tgt_labels && tgt_boxes);
  dim3 grid((Q * Tmax + 255) / 256, B);
  hipLaunchKernelGGL(match_cost_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), logits, ldl, boxes, tgt_labels, tgt_boxes,
                     tgt_offsets, Q, Tmax, w_class, w_bbox, w_giou, alpha, gamma, cost);
  return fx_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Linear sum assignment of each image's [Q, T] cost block, T <= Q: the modified Jonker-Volgenant shortest-augmenting-
// path algorithm of scipy.optimize.linear_sum_assignment (Crouse 2016).  SciPy transposes when rows > columns, so
// rows = targets, columns = queries.  One wave per image; the column scan of every path-extension step runs 64 wide,
// everything is float64 like SciPy.  SciPy's choice among equal-cost columns ("an equal column replaces the current
// one only if it is unassigned", scanning its swap-removed `remaining` list in order) is reproduced exactly:
// among minimum-cost entries pick the LAST unassigned one in list order, else the FIRST.
#define LSA_MAXQ 1024
#define LSA_MAXT 1024

struct LsaBest {
  double val;
  int un_it;   // largest list position among minimum-cost unassigned columns (-1: none)
  int as_it;   // smallest list position among minimum-cost columns
};

__device__ __forceinline__ LsaBest lsa_merge(const LsaBest& a, const LsaBest& b) {
  if (a.val < b.val) return a;
  if (b.val < a.val) return b;
  LsaBest r;
  r.val = a.val;
  r.un_it = max(a.un_it, b.un_it);
  r.as_it = min(a.as_it, b.as_it);
  return r;
}

__global__ __launch_bounds__(64) void lsa_kernel(const float* __restrict__ cost, int Q, int Tmax, const int32_t* __restrict__ toff,
                                                  int32_t* __restrict__ pred_idx, int32_t* __restrict__ tgt_idx, int32_t* __restrict__ status) {
  __shared__ double u[LSA_MAXT], v[LSA_MAXQ], spc[LSA_MAXQ];
  __shared__ int path[LSA_MAXQ], row4col[LSA_MAXQ], remaining[LSA_MAXQ], col4row[LSA_MAXT];
  __shared__ unsigned char SR[LSA_MAXT], SC[LSA_MAXQ];
  __shared__ int s_i, s_sink, s_num;
  __shared__ double s_min;
  const int b = blockIdx.x, lane = threadIdx.x;
  const int t0 = toff[b], T = toff[b + 1] - t0;
  if (T <= 0) return;
  const float* C = cost + (int64_t)b * Q * Tmax;  // C[q*Tmax + t]; transposed problem: row = t, col = q
  const double INF = __longlong_as_double(0x7ff0000000000000ll);
  {  // SciPy validates first: NaN or -inf anywhere -> ValueError("matrix contains invalid numeric entries") (scipy/optimize/_lsap.c)
    bool invalid = false;
    for (int idx = lane; idx < Q * T; idx += 64) {
      const float c = C[(int64_t)(idx / T) * Tmax + idx % T];
      invalid |= (c != c) || (c == -__builtin_inff());
    }
    if (__ballot(invalid) != 0ull) {
      if (lane == 0 && status) atomicOr(status, 2);
      return;
    }
  }
  for (int j = lane; j < Q; j += 64) { v[j] = 0.0; row4col[j] = -1; path[j] = -1; }
  for (int i = lane; i < T; i += 64) { u[i] = 0.0; col4row[i] = -1; }
  __syncthreads();
  for (int cur = 0; cur < T; ++cur) {
    for (int j = lane; j < Q; j += 64) { spc[j] = INF; SC[j] = 0; remaining[j] = Q - j - 1; }
    for (int i = lane; i < T; i += 64) SR[i] = 0;
    if (lane == 0) { s_i = cur; s_sink = -1; s_num = Q; s_min = 0.0; }
    __syncthreads();
    while (true) {
      const int i = s_i, num = s_num;
      const double minv = s_min, ui = u[i];
      LsaBest best;
      best.val = INF; best.un_it = -1; best.as_it = 0x7fffffff;
      for (int it = lane; it < num; it += 64) {
        const int j = remaining[it];
        const double r = minv + (double)C[(int64_t)j * Tmax + i] - ui - v[j];
        double s = spc[j];
        if (r < s) { path[j] = i; spc[j] = r; s = r; }
        LsaBest c;
        c.val = s;
        const bool un = row4col[j] == -1;
        c.un_it = un ? it : -1;
        c.as_it = it;
        best = lsa_merge(best, c);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        LsaBest other;
        other.val = __shfl_xor(best.val, o, 64);
        other.un_it = __shfl_xor(best.un_it, o, 64);
        other.as_it = __shfl_xor(best.as_it, o, 64);
        best = lsa_merge(best, other);
      }
      __syncthreads();  // all spc/path updates visible; everyone has read s_i/s_num/s_min
      if (lane == 0) {
        s_min = best.val;
        if (!(best.val < INF)) {
          // SciPy: "if (lowest == INFINITY) return -1" BEFORE the column is taken (rectangular_lsap.cpp) - a row whose reachable
          // costs are all inf / NaN.  (Until round 3 the sink was set first: with a NaN row the duals became inf and the search
          // could run past its candidate list.)
          s_sink = -2;
        } else {
          SR[i] = 1;
          const int index = best.un_it >= 0 ? best.un_it : best.as_it;
          const int j = remaining[index];
          if (row4col[j] == -1) s_sink = j; else s_i = row4col[j];
          SC[j] = 1;
          remaining[index] = remaining[num - 1];
          s_num = num - 1;
        }
      }
      __syncthreads();
      if (s_sink != -1) break;
    }
    const double minv = s_min;
    const int sink = s_sink;
    if (sink < 0) {  // infeasible (inf / nan costs): outputs untouched; SciPy raises "cost matrix is infeasible" here - the flag lets the host do so
      if (lane == 0 && status) atomicOr(status, 1);
      return;
    }
    // dual updates (SciPy order of operations: u[cur] += minVal; u[i] += minVal - spc[col4row[i]]; v[j] -= minVal - spc[j])
    for (int i = lane; i < T; i += 64)
      if (i == cur) u[i] += minv; else if (SR[i]) u[i] += minv - spc[col4row[i]];
    for (int j = lane; j < Q; j += 64)
      if (SC[j]) v[j] -= minv - spc[j];
    __syncthreads();
    if (lane == 0) {  // augment along the path
      int j = sink;
      while (true) {
        const int i = path[j];
        row4col[j] = i;
        const int tmp = col4row[i];
        col4row[i] = j;
        j = tmp;
        if (i == cur) break;
      }
    }
    __syncthreads();
  }
  // output sorted by query index (SciPy returns row_ind ascending after undoing the transpose)
  for (int t = lane; t < T; t += 64) {
    const int q = col4row[t];
    int rank = 0;
    for (int k = 0; k < T; ++k) rank += (col4row[k] < q) ? 1 : 0;
    pred_idx[t0 + rank] = q;
    tgt_idx[t0 + rank] = t;
  }
}

extern "C" int fx_lsa_status_f32(const float* cost, int B, int Q, int Tmax, const int32_t* tgt_offsets, int32_t* pred_idx, int32_t* tgt_idx,
                                 int32_t* status, fx_stream_t stream_) {
  FX_CHECK_ARG(tgt_offsets && B > 0 && Q > 0 && Tmax >= 0);
  if (Tmax == 0) return FX_OK;
  FX_CHECK_ARG(cost && pred_idx && tgt_idx);
  if (Q > LSA_MAXQ || Tmax > LSA_MAXT || Tmax > Q) return FX_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(lsa_kernel, dim3(B), dim3(64), 0, reinterpret_cast<hipStream_t>(stream_), cost, Q, Tmax, tgt_offsets, pred_idx, tgt_idx, status);
  return fx_launch_status();
}

extern "C" int fx_lsa_f32(const float* cost, int B, int Q, int Tmax, const int32_t* tgt_offsets, int32_t* pred_idx, int32_t* tgt_idx,
                          fx_stream_t stream_) {
  return fx_lsa_status_f32(cost, B, Q, Tmax, tgt_offsets, pred_idx, tgt_idx, nullptr, stream_);
}

// ------------------------------------------------------------------------------------------------
// Losses of one prediction set (modelling.py:464-497, 513-530).  Pass 1 (one lane per matched pair): IoU/GIoU/L1 of the
// pair, scatter (label, IoU) to the matched query.  Pass 2: varifocal BCE over all B*Q*K logits, per-block partial sums.
// Pass 3: fixed-order float64 reduction -> deterministic results.
__global__ __launch_bounds__(256) void loss_pairs_kernel(const float* __restrict__ boxes, const int32_t* __restrict__ tlabels,
                                                          const float* __restrict__ tboxes, const int32_t* __restrict__ toff,
                                                          const int32_t* __restrict__ pred_idx, const int32_t* __restrict__ tgt_idx, int B, int Q,
                                                          int32_t* __restrict__ q_label, float* __restrict__ q_iou, float* __restrict__ pair_l1,
                                                          float* __restrict__ pair_giou) {
  const int n = toff[B];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    int b = 0;
    while (i >= toff[b + 1]) ++b;
    const int q = pred_idx[i], t = toff[b] + tgt_idx[i];
    const float* pb = boxes + ((int64_t)b * Q + q) * 4;
    const float* tb = tboxes + (int64_t)t * 4;
    float ax0, ay0, ax1, ay1, bx0, by0, bx1, by1, iou, giou;
    cxcywh_to_xyxy(pb, ax0, ay0, ax1, ay1);
    cxcywh_to_xyxy(tb, bx0, by0, bx1, by1);
    iou_giou(ax0, ay0, ax1, ay1, bx0, by0, bx1, by1, iou, giou);
    q_label[b * Q + q] = tlabels[t];
    q_iou[b * Q + q] = iou;
    pair_l1[i] = fabsf(pb[0] - tb[0]) + fabsf(pb[1] - tb[1]) + fabsf(pb[2] - tb[2]) + fabsf(pb[3] - tb[3]);
    pair_giou[i] = 1.0f - giou;
  }
}

__global__ __launch_bounds__(256) void loss_vfl_kernel(const float* __restrict__ logits, int ldl, const int32_t* __restrict__ q_label,
                                                        const float* __restrict__ q_iou, int rows, int K, float alpha, float gamma,
                                                        double* __restrict__ partial) {
  __shared__ double red[256];
  double acc = 0.0;
  const int64_t total = (int64_t)rows * K;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / K), k = (int)(i - (int64_t)r * K);
    const float x = logits[(int64_t)r * ldl + k];
    const bool pos = q_label[r] == k;
    const float z = pos ? q_iou[r] : 0.0f;             // target_score
    const float p = 1.0f / (1.0f + expf(-x));
    const float w = alpha * pow_gamma(p, gamma) * (pos ? 0.0f : 1.0f) + z;
    const float bce = fmaxf(x, 0.0f) - x * z + log1pf(expf(-fabsf(x)));
    acc += (double)(w * bce);
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void loss_final_kernel(const double* __restrict__ partial, int nblocks, const float* __restrict__ pair_l1,
                                                          const float* __restrict__ pair_giou, const int32_t* __restrict__ toff, int B, float num_boxes,
                                                          float w_vfl, float w_bbox, float w_giou, float* __restrict__ out3) {
  __shared__ double red[3][256];
  double a = 0.0, l1 = 0.0, gi = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) a += partial[i];
  const int n = toff[B];
  for (int i = threadIdx.x; i < n; i += 256) {
    l1 += (double)pair_l1[i];
    gi += (double)pair_giou[i];
  }
  red[0][threadIdx.x] = a; red[1][threadIdx.x] = l1; red[2][threadIdx.x] = gi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s)
      for (int c = 0; c < 3; ++c) red[c][threadIdx.x] += red[c][threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out3[0] = w_vfl * (float)(red[0][0] / (double)num_boxes);
    out3[1] = w_bbox * (float)(red[1][0] / (double)num_boxes);
    out3[2] = w_giou * (float)(red[2][0] / (double)num_boxes);
  }
}

#define LOSS_BLOCKS 1024

extern "C" int fx_detr_set_loss_workspace_bytes(int B, int Q, int sum_T) {
  return B * Q * 8 + (sum_T > 0 ? sum_T : 1) * 8 + LOSS_BLOCKS * 8 + 64;
}

extern "C" int fx_detr_set_loss_f32(const float* logits, int ldl, const float* boxes, const int32_t* tgt_labels, const float* tgt_boxes,
                                    const int32_t* tgt_offsets, const int32_t* pred_idx, const int32_t* tgt_idx, int B, int Q, int K, int sum_T,
                                    float num_boxes, float focal_alpha, float focal_gamma, float w_vfl, float w_bbox, float w_giou, void* workspace,
                                    float* out3, fx_stream_t stream_) {
  FX_CHECK_ARG(logits && boxes && tgt_offsets && workspace && out3 && B > 0 && Q > 0 && K > 0 && sum_T >= 0 && ldl >= K && num_boxes > 0.0f);
  FX_CHECK_ARG(sum_T == 0 || (tgt_labels && tgt_boxes && pred_idx && tgt_idx));
  FX_CHECK_ARG(((uintptr_t)workspace % 8) == 0);
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  // workspace layout: partial[LOSS_BLOCKS] f64 | q_label[B*Q] i32 | q_iou[B*Q] f32 | pair_l1[sum_T] f32 | pair_giou[sum_T] f32
  double* partial = reinterpret_cast<double*>(workspace);
  int32_t* q_label = reinterpret_cast<int32_t*>(partial + LOSS_BLOCKS);
  float* q_iou = reinterpret_cast<float*>(q_label + (int64_t)B * Q);
  float* pair_l1 = q_iou + (int64_t)B * Q;
  float* pair_giou = pair_l1 + (sum_T > 0 ? sum_T : 1);
  if (hipMemsetAsync(q_label, 0xff, (size_t)B * Q * 4, stream) != hipSuccess) return FX_ERR_RUNTIME;  // -1 = unmatched
  if (sum_T > 0)
    hipLaunchKernelGGL(loss_pairs_kernel, dim3((sum_T + 255) / 256), dim3(256), 0, stream, boxes, tgt_labels, tgt_boxes, tgt_offsets, pred_idx, tgt_idx,
                       B, Q, q_label, q_iou, pair_l1, pair_giou);
  hipLaunchKernelGGL(loss_vfl_kernel, dim3(LOSS_BLOCKS), dim3(256), 0, stream, logits, ldl, q_label, q_iou, B * Q, K, focal_alpha, focal_gamma, partial);
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, stream, partial, LOSS_BLOCKS, pair_l1, pair_giou, tgt_offsets, B, num_boxes, w_vfl,
                     w_bbox, w_giou, out3);
  return fx_launch_status();
}


// ------------------------------------------------------------------------------------------------
// Box losses of one prediction set WITH their gradient (training path): SetCriterion.loss_boxes (modelling.py:513-530: L1 + GIoU of the
// matched pairs, / num_boxes) and the (label, IoU) targets loss_labels_vfl scatters to the matched queries (:464-480), one launch.
// The PyTorch formulation of the same lines costs ~65 elementwise launches forward and ~90 backward per prediction set (7 sets per
// step); here one single-workgroup kernel walks the (at most a few hundred) pairs: fixed summation order, float64 partials.
// Gradient conventions are autograd's: max / min split a tie half-half, clamp(min=0) passes the gradient at 0, sgn(0) = 0.
__device__ __forceinline__ float tie_gt(float a, float b) { return a > b ? 1.0f : (a == b ? 0.5f : 0.0f); }

__global__ __launch_bounds__(256) void box_loss_kernel(const float* __restrict__ boxes, const int32_t* __restrict__ tlabels,
                                                        const float* __restrict__ tboxes, const int32_t* __restrict__ toff,
                                                        const int32_t* __restrict__ pred_idx, const int32_t* __restrict__ tgt_idx, int B, int Q, int K,
                                                        float scale_bbox, float scale_giou, int32_t* __restrict__ q_cls, float* __restrict__ q_score,
                                                        float* __restrict__ loss2, float* __restrict__ pair_grad) {
  __shared__ double red[2][256];
  for (int i = threadIdx.x; i < B * Q; i += 256) {
    q_cls[i] = K;
    q_score[i] = 0.0f;
  }
  __syncthreads();
  const int n = toff[B];
  double l1s = 0.0, gis = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    int b = 0;
    while (i >= toff[b + 1]) ++b;
    const int q = pred_idx[i], t = toff[b] + tgt_idx[i];
    const float* pb = boxes + ((int64_t)b * Q + q) * 4;
    const float* tb = tboxes + (int64_t)t * 4;
    float ax0, ay0, ax1, ay1, bx0, by0, bx1, by1;
    cxcywh_to_xyxy(pb, ax0, ay0, ax1, ay1);
    cxcywh_to_xyxy(tb, bx0, by0, bx1, by1);
    const float wa = ax1 - ax0, ha = ay1 - ay0;
    const float area_a = wa * ha, area_b = (bx1 - bx0) * (by1 - by0);
    const float dx = fminf(ax1, bx1) - fmaxf(ax0, bx0), dy = fminf(ay1, by1) - fmaxf(ay0, by0);
    const float iw = fmaxf(dx, 0.0f), ih = fmaxf(dy, 0.0f);
    const float I = iw * ih, U = area_a + area_b - I, iou = I / U;
    const float cdx = fmaxf(ax1, bx1) - fminf(ax0, bx0), cdy = fmaxf(ay1, by1) - fminf(ay0, by0);
    const float cw = fmaxf(cdx, 0.0f), ch = fmaxf(cdy, 0.0f);
    const float A = cw * ch, E = A + 1e-5f;
    const float giou = iou - (A - U) / E;
    q_cls[b * Q + q] = tlabels[t];
    q_score[b * Q + q] = iou;
    float l1 = 0.0f;
    for (int c = 0; c < 4; ++c) {
      const float d = pb[c] - tb[c];
      l1 += fabsf(d);
      pair_grad[(int64_t)i * 8 + c] = scale_bbox * (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f));
    }
    l1s += (double)l1;
    gis += (double)(1.0f - giou);
    // d giou / d (ax0, ay0, ax1, ay1)
    const float px = dx >= 0.0f ? 1.0f : 0.0f, py = dy >= 0.0f ? 1.0f : 0.0f, pcx = cdx >= 0.0f ? 1.0f : 0.0f, pcy = cdy >= 0.0f ? 1.0f : 0.0f;
    const float dI[4] = {-px * tie_gt(ax0, bx0) * ih, -py * tie_gt(ay0, by0) * iw, px * tie_gt(bx1, ax1) * ih, py * tie_gt(by1, ay1) * iw};
    const float dA[4] = {-pcx * tie_gt(bx0, ax0) * ch, -pcy * tie_gt(by0, ay0) * cw, pcx * tie_gt(ax1, bx1) * ch, pcy * tie_gt(ay1, by1) * cw};
    const float dAa[4] = {-ha, -wa, ha, wa};
    float g[4];
    for (int c = 0; c < 4; ++c) {
      const float dU = dAa[c] - dI[c];
      g[c] = dI[c] / U - I / (U * U) * dU - dA[c] * (U + 1e-5f) / (E * E) + dU / E;
    }
    // loss = 1 - giou;  cx -> x0 + x1, w -> (x1 - x0) / 2
    pair_grad[(int64_t)i * 8 + 4] = -scale_giou * (g[0] + g[2]);
    pair_grad[(int64_t)i * 8 + 5] = -scale_giou * (g[1] + g[3]);
    pair_grad[(int64_t)i * 8 + 6] = -scale_giou * 0.5f * (g[2] - g[0]);
    pair_grad[(int64_t)i * 8 + 7] = -scale_giou * 0.5f * (g[3] - g[1]);
  }
  red[0][threadIdx.x] = l1s;
  red[1][threadIdx.x] = gis;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      red[0][threadIdx.x] += red[0][threadIdx.x + s];
      red[1][threadIdx.x] += red[1][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    loss2[0] = (float)((double)scale_bbox * red[0][0]);
    loss2[1] = (float)((double)scale_giou * red[1][0]);
  }
}

__global__ __launch_bounds__(256) void box_loss_bwd_kernel(const float* __restrict__ pair_grad, const int32_t* __restrict__ toff,
                                                            const int32_t* __restrict__ pred_idx, int B, int Q, const float* __restrict__ g_bbox,
                                                            const float* __restrict__ g_giou, float* __restrict__ dboxes) {
  const int n = toff[B];
  const float g1 = g_bbox ? *g_bbox : 0.0f, g2 = g_giou ? *g_giou : 0.0f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    int b = 0;
    while (i >= toff[b + 1]) ++b;
    float* d = dboxes + ((int64_t)b * Q + pred_idx[i]) * 4;
    for (int c = 0; c < 4; ++c) d[c] = g1 * pair_grad[(int64_t)i * 8 + c] + g2 * pair_grad[(int64_t)i * 8 + 4 + c];
  }
}

extern "C" int fx_detr_box_loss_f32(const float* boxes, const int32_t* tgt_labels, const float* tgt_boxes, const int32_t* tgt_offsets,
                                    const int32_t* pred_idx, const int32_t* tgt_idx, int B, int Q, int K, int sum_T, float scale_bbox,
                                    float scale_giou, int32_t* q_cls, float* q_score, float* loss2, float* pair_grad, fx_stream_t stream_) {
  FX_CHECK_ARG(boxes && tgt_offsets && q_cls && q_score && loss2 && B > 0 && Q > 0 && K > 0 && sum_T >= 0);
  FX_CHECK_ARG(sum_T == 0 || (tgt_labels && tgt_boxes && pred_idx && tgt_idx && pair_grad));
  hipLaunchKernelGGL(box_loss_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), boxes, tgt_labels, tgt_boxes, tgt_offsets, pred_idx,
                     tgt_idx, B, Q, K, scale_bbox, scale_giou, q_cls, q_score, loss2, pair_grad);
  return fx_launch_status();
}

extern "C" int fx_detr_box_loss_bwd_f32(const float* pair_grad, const int32_t* tgt_offsets, const int32_t* pred_idx, int B, int Q, int sum_T,
                                        const float* g_bbox, const float* g_giou, float* dboxes, fx_stream_t stream_) {
  FX_CHECK_ARG(tgt_offsets && dboxes && B > 0 && Q > 0 && sum_T >= 0 && (sum_T == 0 || (pair_grad && pred_idx)));
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (hipMemsetAsync(dboxes, 0, (size_t)B * Q * 16, stream) != hipSuccess) return FX_ERR_RUNTIME;
  if (sum_T > 0)
    hipLaunchKernelGGL(box_loss_bwd_kernel, dim3((sum_T + 255) / 256), dim3(256), 0, stream, pair_grad, tgt_offsets, pred_idx, B, Q, g_bbox, g_giou, dboxes);
  return fx_launch_status();
}
