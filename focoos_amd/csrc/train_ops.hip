// Training-side kernels (gfx950) that do not depend on the conv backward: the multi-scale deformable attention core
// in fp32 with its backward (the autograd half of seam B4), and a fused multi-tensor AdamW step with global-norm gradient
// clipping (the reference runs ~500 single-tensor param groups and clips twice).
#include <math.h>

#include "common.h"

int fx_tune(const char* env_name, int default_value);  // conv_igemm.hip: integer tuning knob from the environment

// ------------------------------------------------------------------------------------------------
// ms_deform_attn_core (focoos/nn/layers/deformable.py:10-35), fp32.  One wave per (batch, query); lane = (head = lane>>3,
// 4 channels = (lane&7)*4), M*D = 256.  Forward writes out[b,q,h*32+c]; backward scatters grad_value with fp32 atomics
// (different queries hit the same value pixels) and reduces grad_loc / grad_attn over the 8 lanes of a head.
struct Tap4 {
  float w[4];
  int idx[4];  // y*W+x or -1 (out of the map: zeros padding)
  float tx, ty;
};

__device__ __forceinline__ Tap4 make_taps(float lx, float ly, int Hl, int Wl) {
  Tap4 t;
  const float gx = 2.0f * lx - 1.0f, gy = 2.0f * ly - 1.0f;
  const float ix = ((gx + 1.0f) * (float)Wl - 1.0f) * 0.5f, iy = ((gy + 1.0f) * (float)Hl - 1.0f) * 0.5f;
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  const int x0 = (int)fx0, y0 = (int)fy0;
  t.tx = ix - fx0;
  t.ty = iy - fy0;
  t.w[0] = (1.f - t.tx) * (1.f - t.ty);
  t.w[1] = t.tx * (1.f - t.ty);
  t.w[2] = (1.f - t.tx) * t.ty;
  t.w[3] = t.tx * t.ty;
  const bool xin0 = (unsigned)x0 < (unsigned)Wl, xin1 = (unsigned)(x0 + 1) < (unsigned)Wl;
  const bool yin0 = (unsigned)y0 < (unsigned)Hl, yin1 = (unsigned)(y0 + 1) < (unsigned)Hl;
  t.idx[0] = (yin0 && xin0) ? y0 * Wl + x0 : -1;
  t.idx[1] = (yin0 && xin1) ? y0 * Wl + x0 + 1 : -1;
  t.idx[2] = (yin1 && xin0) ? (y0 + 1) * Wl + x0 : -1;
  t.idx[3] = (yin1 && xin1) ? (y0 + 1) * Wl + x0 + 1 : -1;
  return t;
}

// value element type: fp32 (the reference-shaped entry points) or bf16 (the training graph reads the value projection's bf16 output in
// place - a column slice of the six layers' shared [B,S,6*256] buffer, row stride ldv - instead of an fp32 copy of it)
__device__ __forceinline__ void msda_ld4(const float* p, float* v) {
  const float4 x = *reinterpret_cast<const float4*>(p);
  v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
}
__device__ __forceinline__ void msda_ld4(const bf16_t* p, float* v) {
  const uint2 x = *reinterpret_cast<const uint2*>(p);
  v[0] = bf16lo_to_f32(x.x); v[1] = bf16hi_to_f32(x.x);
  v[2] = bf16lo_to_f32(x.y); v[3] = bf16hi_to_f32(x.y);
}
__device__ __forceinline__ float msda_ld1(const float* p) { return *p; }
__device__ __forceinline__ float msda_ld1(const bf16_t* p) { return bf16_to_f32(*p); }

template <bool BWD, typename VT>
__global__ __launch_bounds__(256) void msda_f32_kernel(const VT* __restrict__ value, int ldv, const int32_t* __restrict__ shapes,
                                                        const int32_t* __restrict__ lstart, int L, int P, const float* __restrict__ loc,
                                                        const float* __restrict__ attn, const float* __restrict__ grad_out, float* __restrict__ out,
                                                        float* __restrict__ grad_value, float* __restrict__ grad_loc, float* __restrict__ grad_attn,
                                                        int B, int S, int Q, int M) {
  const int lane = threadIdx.x & 63;
  const int bq = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bq >= B * Q) return;
  const int b = bq / Q;
  const int h = lane >> 3, cg = lane & 7;
  const int LP = L * P, CH = M * 32;
  const int64_t vbase = (int64_t)b * S * ldv + h * 32 + cg * 4;
  const float* locp = loc + ((int64_t)bq * M + h) * LP * 2;
  const float* attp = attn + ((int64_t)bq * M + h) * LP;
  float go[4] = {0.f, 0.f, 0.f, 0.f};
  if (BWD) {
    const float4 g = *reinterpret_cast<const float4*>(grad_out + (int64_t)bq * CH + h * 32 + cg * 4);
    go[0] = g.x; go[1] = g.y; go[2] = g.z; go[3] = g.w;
  }
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int l = 0; l < L; ++l) {
    const int Hl = shapes[2 * l], Wl = shapes[2 * l + 1];
    const int64_t lbase = vbase + (int64_t)lstart[l] * ldv;
    for (int pt = 0; pt < P; ++pt) {
      const int i = l * P + pt;
      const float aw = attp[i];
      const Tap4 t = make_taps(locp[2 * i], locp[2 * i + 1], Hl, Wl);
      float v[4][4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (t.idx[k] >= 0) {
          msda_ld4(value + lbase + (int64_t)t.idx[k] * ldv, v[k]);
        } else {
          v[k][0] = v[k][1] = v[k][2] = v[k][3] = 0.f;
        }
      }
      if (!BWD) {
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] += aw * (t.w[0] * v[0][c] + t.w[1] * v[1][c] + t.w[2] * v[2][c] + t.w[3] * v[3][c]);
      } else {
        float g_a = 0.f, g_x = 0.f, g_y = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float s = t.w[0] * v[0][c] + t.w[1] * v[1][c] + t.w[2] * v[2][c] + t.w[3] * v[3][c];
          g_a += go[c] * s;
          g_x += go[c] * ((v[1][c] - v[0][c]) * (1.f - t.ty) + (v[3][c] - v[2][c]) * t.ty);
          g_y += go[c] * ((v[2][c] - v[0][c]) * (1.f - t.tx) + (v[3][c] - v[1][c]) * t.tx);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (t.idx[k] >= 0) {
            float* gv = grad_value + lbase + (int64_t)t.idx[k] * ldv;
            const float wk = aw * t.w[k];
#pragma unroll
            for (int c = 0; c < 4; ++c) unsafeAtomicAdd(gv + c, wk * go[c]);  // hardware fp32 atomic (plain atomicAdd lowers to a CAS loop)
          }
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
          g_a += __shfl_xor(g_a, o, 64);
          g_x += __shfl_xor(g_x, o, 64);
          g_y += __shfl_xor(g_y, o, 64);
        }
        if (cg == 0) {
          grad_attn[((int64_t)bq * M + h) * LP + i] = g_a;
          // d(ix)/d(loc_x) = W, d(iy)/d(loc_y) = H  (ix = loc_x * W - 0.5)
          grad_loc[(((int64_t)bq * M + h) * LP + i) * 2 + 0] = aw * g_x * (float)Wl;
          grad_loc[(((int64_t)bq * M + h) * LP + i) * 2 + 1] = aw * g_y * (float)Hl;
        }
      }
    }
  }
  if (!BWD) *reinterpret_cast<float4*>(out + (int64_t)bq * CH + h * 32 + cg * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

// Backward: one wave per (batch, query); lane = (head parity, channel): every atomic instruction covers the 32 consecutive
// channels (one 128-byte line) of two heads, so a tap costs 8 line-wide L2 atomic requests instead of the 32 quarter-filled
// ones of the forward's lane layout (4 channels per lane) - the L2 atomic units, not the gathers, bound this kernel.
// (A wave per sampling point - 12x the waves - was measured slower: 1.25 vs 1.06 ms per layer; the request count is what matters.)
template <typename VT>
__global__ __launch_bounds__(256) void msda_f32_bwd_kernel(const VT* __restrict__ value, int ldv, const int32_t* __restrict__ shapes,
                                                            const int32_t* __restrict__ lstart, int L, int P, const float* __restrict__ loc,
                                                            const float* __restrict__ attn, const float* __restrict__ grad_out,
                                                            float* __restrict__ grad_value, int ldg, float* __restrict__ grad_loc,
                                                            float* __restrict__ grad_attn, int B, int S, int Q, int M, int lsplit, int psplit) {
  const int lane = threadIdx.x & 63;
  // lsplit = L: one wave per (batch, query, LEVEL) - B*Q waves alone (4800 for 16 x 300) are ~5 per SIMD, too few to hide the
  // loc -> taps -> value-load dependency chain of every sampling point; the atomic request count is unchanged
  const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int per_q = lsplit * psplit;
  const int bq = unit / per_q;
  if (bq >= B * Q) return;
  const int sub = unit - bq * per_q;
  const int l_lo = lsplit > 1 ? sub / psplit : 0, l_hi = lsplit > 1 ? l_lo + 1 : L;
  const int pt_lo = (sub % psplit) * (P / psplit), pt_hi = pt_lo + P / psplit;
  const int b = bq / Q;
  const int hh = lane >> 5, c = lane & 31;
  const int LP = L * P, CH = M * 32;
  float go[4];
#pragma unroll
  for (int hp = 0; hp < 4; ++hp) go[hp] = grad_out[(int64_t)bq * CH + (hp * 2 + hh) * 32 + c];
  for (int l = l_lo; l < l_hi; ++l) {
    const int Hl = shapes[2 * l], Wl = shapes[2 * l + 1];
    const int64_t lbase0 = ((int64_t)b * S + lstart[l]) * ldv + c;
    const int64_t gbase0 = ((int64_t)b * S + lstart[l]) * ldg + c;
    for (int pt = pt_lo; pt < pt_hi; ++pt) {
      const int i = l * P + pt;
#pragma unroll
      for (int hp = 0; hp < 4; ++hp) {
        const int h = hp * 2 + hh;
        const int64_t pidx = ((int64_t)bq * M + h) * LP + i;
        const float aw = attn[pidx];
        const Tap4 t = make_taps(loc[2 * pidx], loc[2 * pidx + 1], Hl, Wl);
        const int64_t lbase = lbase0 + h * 32, gbase = gbase0 + h * 32;
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = t.idx[k] >= 0 ? msda_ld1(value + lbase + (int64_t)t.idx[k] * ldv) : 0.f;
        const float g = go[hp];
        float g_a = g * (t.w[0] * v[0] + t.w[1] * v[1] + t.w[2] * v[2] + t.w[3] * v[3]);
        float g_x = g * ((v[1] - v[0]) * (1.f - t.ty) + (v[3] - v[2]) * t.ty);
        float g_y = g * ((v[2] - v[0]) * (1.f - t.tx) + (v[3] - v[1]) * t.tx);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (t.idx[k] >= 0) unsafeAtomicAdd(grad_value + gbase + (int64_t)t.idx[k] * ldg, aw * t.w[k] * g);
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
          g_a += __shfl_xor(g_a, o, 64);
          g_x += __shfl_xor(g_x, o, 64);
          g_y += __shfl_xor(g_y, o, 64);
        }
        if (c == 0) {
          grad_attn[pidx] = g_a;
          // d(ix)/d(loc_x) = W, d(iy)/d(loc_y) = H  (ix = loc_x * W - 0.5)
          grad_loc[pidx * 2 + 0] = aw * g_x * (float)Wl;
          grad_loc[pidx * 2 + 1] = aw * g_y * (float)Hl;
        }
      }
    }
  }
}

// ---- the backward without L2 atomics (fx_msda_train_bwd_slab): two kernels.
// (1) msda_point_grad_kernel: grad_loc / grad_attn in the FORWARD's lane layout (lane = head x 4 channels, 8-byte value loads) - the
//     atomic form above needs lane = channel so that an atomic covers whole lines, and pays for it with 2-byte gathers (148 us of its
//     304 us per layer are gathers); without the atomics the gathers cost what the forward's do.
// (2) msda_bwd_value_kernel: the value gradient by BINNING.  A workgroup owns (batch, head, slab of whole rows of one level, at most
//     MS_PIX pixels).  It computes the taps of all Q*P sampling points of its (batch, head, level) twice: once to count the taps per
//     pixel (LDS integer atomics), then - after an exclusive scan of the counts - to file every tap (query, weight) under its pixel.
//     Each half-wave then sums the entries of one pixel at a time (lane = channel; grad_out rows staged in LDS) and stores the pixel
//     once, as bf16 - the dtype the value projection's backward reads - so the shared gradient buffer needs no zero-fill, no fp32
//     image and no cast, and there is no floating-point atomic anywhere.  (First attempt, kept in profiles/r03_msda_bwd.txt: the same
//     slabs with ds_add_f32 accumulation - 580 us per layer, 377 us of it the LDS float atomics at ~3 cycles per lane.)
constexpr int MS_PIX = 3200;   // pixels per slab: level 80 x 80 = two slabs -> 4 slabs x 128 (batch, head) = 512 workgroups = 2 per CU
constexpr int MS_MAXL = 8;
constexpr int MS_HOT = 96;     // entries from which a pixel is summed by the whole workgroup
// ints reserved for the set-aside pixel list, rounded so that (MS_PIX + 8 + 2 + 512 + this) is a multiple of 4 ints = 16 bytes
__host__ __device__ constexpr int msda_hot_ints(int Q, int P) {
  return ((MS_PIX + 8 + 2 + 512 + (Q * P * 4 / MS_HOT + 2)) + 3) / 4 * 4 - (MS_PIX + 8 + 2 + 512);
}

struct MsdaBinArgs {
  int L, P, B, S, Q, M, ldg;
  int H[MS_MAXL], W[MS_MAXL], start[MS_MAXL], rows[MS_MAXL], slab0[MS_MAXL + 1];
};

__device__ __forceinline__ float msda_go1(const float* p) { return *p; }
__device__ __forceinline__ float msda_go1(const bf16_t* p) { return bf16_to_f32(*p); }

template <typename VT, typename GT>
__global__ __launch_bounds__(256) void msda_point_grad_kernel(const VT* __restrict__ value, int ldv, const int32_t* __restrict__ shapes,
                                                               const int32_t* __restrict__ lstart, int L, int P, const float* __restrict__ loc,
                                                               const float* __restrict__ attn, const GT* __restrict__ grad_out,
                                                               float* __restrict__ grad_loc, float* __restrict__ grad_attn, int B, int S, int Q, int M) {
  const int lane = threadIdx.x & 63;
  const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);   // one wave per (batch, query, level)
  const int bq = unit / L;
  if (bq >= B * Q) return;
  const int l = unit - bq * L;
  const int b = bq / Q;
  const int h = lane >> 3, cg = lane & 7;
  const int LP = L * P, CH = M * 32;
  float go[4];
  msda_ld4(grad_out + (int64_t)bq * CH + h * 32 + cg * 4, go);
  const int Hl = shapes[2 * l], Wl = shapes[2 * l + 1];
  const int64_t lbase = ((int64_t)b * S + lstart[l]) * ldv + h * 32 + cg * 4;
  const int64_t pbase = ((int64_t)bq * M + h) * LP + l * P;
  for (int pt = 0; pt < P; ++pt) {
    const float aw = attn[pbase + pt];
    const Tap4 t = make_taps(loc[2 * (pbase + pt)], loc[2 * (pbase + pt) + 1], Hl, Wl);
    float v[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (t.idx[k] >= 0) {
        msda_ld4(value + lbase + (int64_t)t.idx[k] * ldv, v[k]);
      } else {
        v[k][0] = v[k][1] = v[k][2] = v[k][3] = 0.f;
      }
    }
    float g_a = 0.f, g_x = 0.f, g_y = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      g_a += go[c] * (t.w[0] * v[0][c] + t.w[1] * v[1][c] + t.w[2] * v[2][c] + t.w[3] * v[3][c]);
      g_x += go[c] * ((v[1][c] - v[0][c]) * (1.f - t.ty) + (v[3][c] - v[2][c]) * t.ty);
      g_y += go[c] * ((v[2][c] - v[0][c]) * (1.f - t.tx) + (v[3][c] - v[1][c]) * t.tx);
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
      g_a += __shfl_xor(g_a, o, 64);
      g_x += __shfl_xor(g_x, o, 64);
      g_y += __shfl_xor(g_y, o, 64);
    }
    if (cg == 0) {
      grad_attn[pbase + pt] = g_a;
      grad_loc[2 * (pbase + pt) + 0] = aw * g_x * (float)Wl;   // d(ix)/d(loc_x) = W, d(iy)/d(loc_y) = H
      grad_loc[2 * (pbase + pt) + 1] = aw * g_y * (float)Hl;
    }
  }
}

template <typename GT>
__global__ __launch_bounds__(512) void msda_bwd_value_kernel(MsdaBinArgs a, const float* __restrict__ loc, const float* __restrict__ attn,
                                                             const GT* __restrict__ grad_out, bf16_t* __restrict__ gv) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ms_smem[];
  int* cnt = reinterpret_cast<int*>(ms_smem);                              // [MS_PIX]: tap count -> start offset -> end offset per pixel
  int* wsum = cnt + MS_PIX;                                                // [8] wave totals of the scan
  int* hotn = wsum + 8;                                                    // [1] (+1 pad) number of set-aside pixels
  float* part = reinterpret_cast<float*>(hotn + 2);                        // [16][32] partial sums of a set-aside pixel
  int* hot = reinterpret_cast<int*>(part + 512);                           // [Q*P*4 / MS_HOT + 2] set-aside pixels
  // `ent` starts on a 16-byte boundary (MS_HOT_INTS pads the set-aside list) and holds Q*P*4 8-byte entries (a multiple of 32 bytes), so goL -
  // written with 16-byte stores in the fp32 instantiation - is 16-byte aligned too (it was 8-byte aligned for Q*P = 1200)
  uint2* ent = reinterpret_cast<uint2*>(hot + msda_hot_ints(a.Q, a.P));   // [Q*P*4]: (query, weight bits), filed by pixel
  GT* goL = reinterpret_cast<GT*>(ent + (size_t)a.Q * a.P * 4);            // [Q][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int slab = blockIdx.x, b = blockIdx.y / a.M, h = blockIdx.y - b * a.M;
  int l = 0;
  while (l + 1 < a.L && slab >= a.slab0[l + 1]) ++l;
  const int Hl = a.H[l], Wl = a.W[l];
  const int r0 = (slab - a.slab0[l]) * a.rows[l];
  const int r1 = r0 + a.rows[l] < Hl ? r0 + a.rows[l] : Hl;
  const int p0 = r0 * Wl, np = (r1 - r0) * Wl;
  const int P = a.P, LP = a.L * P, CH = a.M * 32, NPT = a.Q * P;
  // grad_out rows of this (batch, head): 4 channels per thread and load
  for (int i = tid; i < a.Q * 8; i += 512) {
    const GT* src = grad_out + ((int64_t)b * a.Q + (i >> 3)) * CH + h * 32 + (i & 7) * 4;
    if (sizeof(GT) == 2) reinterpret_cast<uint2*>(goL)[i] = *reinterpret_cast<const uint2*>(src);
    else reinterpret_cast<float4*>(goL)[i] = *reinterpret_cast<const float4*>(src);
  }
  for (int i = tid; i < np; i += 512) cnt[i] = 0;
  if (tid == 0) *hotn = 0;
  __syncthreads();
  // pass 1: taps per pixel
  for (int i = tid; i < NPT; i += 512) {
    const int q = i / P, pt = i - q * P;
    const int64_t pidx = (((int64_t)b * a.Q + q) * a.M + h) * LP + l * P + pt;
    const float2 xy = *reinterpret_cast<const float2*>(loc + 2 * pidx);
    const Tap4 t = make_taps(xy.x, xy.y, Hl, Wl);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = t.idx[k] - p0;
      if (t.idx[k] >= 0 && (unsigned)r < (unsigned)np) atomicAdd(cnt + r, 1);
    }
  }
  __syncthreads();
  // exclusive scan of cnt[0, np): a run of consecutive pixels per thread, wave scan of the run totals, wave totals through LDS
  {
    const int run = (np + 511) / 512;
    const int i0 = tid * run, i1 = i0 + run < np ? i0 + run : np;
    int tot = 0;
    for (int i = i0; i < i1; ++i) tot += cnt[i];
    int inc = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(inc, o, 64);
      if (lane >= o) inc += up;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int base = inc - tot;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    for (int i = i0; i < i1; ++i) {
      const int n = cnt[i];
      cnt[i] = base;
      base += n;
    }
  }
  __syncthreads();
  // pass 2: file every tap under its pixel (the returning atomic advances the pixel's cursor: afterwards cnt[p] = END of pixel p)
  for (int i = tid; i < NPT; i += 512) {
    const int q = i / P, pt = i - q * P;
    const int64_t pidx = (((int64_t)b * a.Q + q) * a.M + h) * LP + l * P + pt;
    const float2 xy = *reinterpret_cast<const float2*>(loc + 2 * pidx);
    const float aw = attn[pidx];
    const Tap4 t = make_taps(xy.x, xy.y, Hl, Wl);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = t.idx[k] - p0;
      if (t.idx[k] >= 0 && (unsigned)r < (unsigned)np) {
        const int slot = atomicAdd(cnt + r, 1);
        ent[slot] = make_uint2((unsigned)q, __float_as_uint(aw * t.w[k]));
      }
    }
  }
  __syncthreads();
  // sum per pixel: half-wave = pixel, lane = channel; four entries in flight.  Pixels with MS_HOT or more entries (queries crowd on a few
  // pixels of the coarse level) are set aside and summed afterwards by all 16 half-waves together.
  const int hw = tid >> 5, c = tid & 31;
  const int64_t orow = ((int64_t)b * a.S + a.start[l] + p0) * a.ldg + h * 32 + c;
  for (int p = hw; p < np; p += 16) {
    const int s = p ? cnt[p - 1] : 0, e = cnt[p];
    if (e - s >= MS_HOT) {
      if (c == 0) hot[atomicAdd(hotn, 1)] = p;
      continue;
    }
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int j = s;
    for (; j + 4 <= e; j += 4) {
      const uint2 e0 = ent[j], e1 = ent[j + 1], e2 = ent[j + 2], e3 = ent[j + 3];
      a0 += __uint_as_float(e0.y) * msda_go1(goL + e0.x * 32 + c);
      a1 += __uint_as_float(e1.y) * msda_go1(goL + e1.x * 32 + c);
      a2 += __uint_as_float(e2.y) * msda_go1(goL + e2.x * 32 + c);
      a3 += __uint_as_float(e3.y) * msda_go1(goL + e3.x * 32 + c);
    }
    for (; j < e; ++j) {
      const uint2 en = ent[j];
      a0 += __uint_as_float(en.y) * msda_go1(goL + en.x * 32 + c);
    }
    gv[orow + (int64_t)p * a.ldg] = f32_to_bf16((a0 + a1) + (a2 + a3));
  }
  __syncthreads();
  const int nhot = *hotn;
  for (int k = 0; k < nhot; ++k) {
    const int p = hot[k];
    const int s = p ? cnt[p - 1] : 0, e = cnt[p];
    float a0 = 0.f, a1 = 0.f;
    int j = s + hw;
    for (; j + 16 < e; j += 32) {
      const uint2 e0 = ent[j], e1 = ent[j + 16];
      a0 += __uint_as_float(e0.y) * msda_go1(goL + e0.x * 32 + c);
      a1 += __uint_as_float(e1.y) * msda_go1(goL + e1.x * 32 + c);
    }
    if (j < e) {
      const uint2 en = ent[j];
      a0 += __uint_as_float(en.y) * msda_go1(goL + en.x * 32 + c);
    }
    part[hw * 32 + c] = a0 + a1;
    __syncthreads();
    if (hw == 0) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) t += part[i * 32 + c];
      gv[orow + (int64_t)p * a.ldg] = f32_to_bf16(t);
    }
    __syncthreads();
  }
}

// value: fp32 or bf16 (value_bf16), row stride ldv elements (>= M*32; a column slice of a wider buffer is fine), 16-byte (fp32) /
// 8-byte (bf16) aligned rows.  grad_value: fp32, row stride ldg; zeroed here unless zero_grad_value == 0 (several launches accumulating
// into column slices of one buffer zero it once, themselves).
extern "C" int fx_msda_train_fwd(const void* value, int value_bf16, int ldv, const int32_t* spatial_shapes, const int32_t* level_start, int L, int P,
                                 const float* loc, const float* attn, float* out, int B, int S, int Q, int M, fx_stream_t stream_) {
  FX_CHECK_ARG(value && spatial_shapes && level_start && loc && attn && out && B > 0 && S > 0 && Q > 0 && L > 0 && P > 0);
  if (M != 8) return FX_ERR_UNSUPPORTED;
  FX_CHECK_ARG(ldv >= M * 32 && ldv % 4 == 0 && ((uintptr_t)value % (value_bf16 ? 8 : 16)) == 0);
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (value_bf16)
    hipLaunchKernelGGL((msda_f32_kernel<false, bf16_t>), dim3((B * Q + 3) / 4), dim3(256), 0, stream, (const bf16_t*)value, ldv, spatial_shapes,
                       level_start, L, P, loc, attn, nullptr, out, nullptr, nullptr, nullptr, B, S, Q, M);
  else
    hipLaunchKernelGGL((msda_f32_kernel<false, float>), dim3((B * Q + 3) / 4), dim3(256), 0, stream, (const float*)value, ldv, spatial_shapes,
                       level_start, L, P, loc, attn, nullptr, out, nullptr, nullptr, nullptr, B, S, Q, M);
  return fx_launch_status();
}

extern "C" int fx_msda_train_bwd(const void* value, int value_bf16, int ldv, const int32_t* spatial_shapes, const int32_t* level_start, int L, int P,
                                 const float* loc, const float* attn, const float* grad_out, float* grad_value, int ldg, int zero_grad_value,
                                 float* grad_loc, float* grad_attn, int B, int S, int Q, int M, fx_stream_t stream_) {
  FX_CHECK_ARG(value && spatial_shapes && level_start && loc && attn && grad_out && grad_value && grad_loc && grad_attn);
  FX_CHECK_ARG(B > 0 && S > 0 && Q > 0 && L > 0 && P > 0);
  if (M != 8) return FX_ERR_UNSUPPORTED;
  FX_CHECK_ARG(ldv >= M * 32 && ldg >= M * 32 && (!zero_grad_value || ldg == M * 32));
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (zero_grad_value && hipMemsetAsync(grad_value, 0, (size_t)B * S * M * 32 * sizeof(float), stream) != hipSuccess) return FX_ERR_RUNTIME;
  static const int split_on = fx_tune("FX_MSDA_BWD_LEVEL_SPLIT", 1);
  static const int psplit_env = fx_tune("FX_MSDA_BWD_POINT_SPLIT", 1);
  const int lsplit = split_on ? L : 1;
  const int psplit = (split_on && psplit_env > 0 && P % psplit_env == 0) ? psplit_env : 1;
  const dim3 grid((B * Q * lsplit * psplit + 3) / 4);
  if (value_bf16)
    hipLaunchKernelGGL(msda_f32_bwd_kernel<bf16_t>, grid, dim3(256), 0, stream, (const bf16_t*)value, ldv, spatial_shapes, level_start, L, P, loc, attn,
                       grad_out, grad_value, ldg, grad_loc, grad_attn, B, S, Q, M, lsplit, psplit);
  else
    hipLaunchKernelGGL(msda_f32_bwd_kernel<float>, grid, dim3(256), 0, stream, (const float*)value, ldv, spatial_shapes, level_start, L, P, loc, attn,
                       grad_out, grad_value, ldg, grad_loc, grad_attn, B, S, Q, M, lsplit, psplit);
  return fx_launch_status();
}

// The same backward without floating-point atomics (kernels above): grad_value_bf16 [B,S,ldg] bf16 - the M*32 columns of this layer are
// fully overwritten (no zero-fill needed, nothing accumulated across calls).  grad_out fp32 or bf16 (grad_out_bf16).  shapes_host: the
// L (H, W) pairs on the host (the slab grid is sized from them).  FX_ERR_UNSUPPORTED when a level is wider than MS_PIX pixels or the
// taps of Q*P points do not fit the LDS (fx_msda_bwd_slab_supported tells beforehand).
static int msda_slab_smem(int P, int Q, int go_bf16) {
  return (MS_PIX + 8 + 2 + 512 + msda_hot_ints(Q, P)) * 4 + Q * P * 4 * 8 + Q * 32 * (go_bf16 ? 2 : 4);
}

extern "C" int fx_msda_bwd_slab_supported(const int32_t* shapes_host, int L, int P, int Q, int M, int grad_out_bf16) {
  if (!shapes_host || M != 8 || L < 1 || L > MS_MAXL || P < 1 || Q < 1 || Q >= (1 << 24)) return 0;
  for (int l = 0; l < L; ++l)
    if (shapes_host[2 * l] < 1 || shapes_host[2 * l + 1] < 1 || shapes_host[2 * l + 1] > MS_PIX) return 0;
  return msda_slab_smem(P, Q, grad_out_bf16) <= 160 * 1024 ? 1 : 0;
}

extern "C" int fx_msda_train_bwd_slab(const void* value, int value_bf16, int ldv, const int32_t* spatial_shapes, const int32_t* level_start,
                                      const int32_t* shapes_host, int L, int P, const float* loc, const float* attn, const void* grad_out,
                                      int grad_out_bf16, void* grad_value_bf16, int ldg, float* grad_loc, float* grad_attn, int B, int S, int Q,
                                      int M, fx_stream_t stream_) {
  FX_CHECK_ARG(value && spatial_shapes && level_start && shapes_host && loc && attn && grad_out && grad_value_bf16 && grad_loc && grad_attn);
  FX_CHECK_ARG(B > 0 && S > 0 && Q > 0 && L > 0 && P > 0);
  if (!fx_msda_bwd_slab_supported(shapes_host, L, P, Q, M, grad_out_bf16)) return FX_ERR_UNSUPPORTED;
  FX_CHECK_ARG(ldv >= M * 32 && ldv % 4 == 0 && ((uintptr_t)value % (value_bf16 ? 8 : 16)) == 0);
  FX_CHECK_ARG(ldg >= M * 32 && ((uintptr_t)grad_value_bf16 % 2) == 0 && ((uintptr_t)grad_out % (grad_out_bf16 ? 8 : 16)) == 0);
  MsdaBinArgs a;
  a.L = L; a.P = P; a.B = B; a.S = S; a.Q = Q; a.M = M; a.ldg = ldg;
  int start = 0, slabs = 0;
  for (int l = 0; l < MS_MAXL; ++l) {
    if (l < L) {
      const int H = shapes_host[2 * l], W = shapes_host[2 * l + 1];
      a.H[l] = H; a.W[l] = W; a.start[l] = start; a.rows[l] = MS_PIX / W < H ? MS_PIX / W : H; a.slab0[l] = slabs;
      start += H * W;
      slabs += (H + a.rows[l] - 1) / a.rows[l];
    } else {
      a.H[l] = a.W[l] = a.start[l] = a.rows[l] = 0; a.slab0[l] = slabs;
    }
  }
  a.slab0[MS_MAXL] = slabs;
  FX_CHECK_ARG(start == S);
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const dim3 grid((B * Q * L + 3) / 4);
#define FX_MSDA_PG(VT_, GT_)                                                                                                                   \
  hipLaunchKernelGGL((msda_point_grad_kernel<VT_, GT_>), grid, dim3(256), 0, stream, (const VT_*)value, ldv, spatial_shapes, level_start, L, P, \
                     loc, attn, (const GT_*)grad_out, grad_loc, grad_attn, B, S, Q, M)
  if (value_bf16 && grad_out_bf16) FX_MSDA_PG(bf16_t, bf16_t);
  else if (value_bf16) FX_MSDA_PG(bf16_t, float);
  else if (grad_out_bf16) FX_MSDA_PG(float, bf16_t);
  else FX_MSDA_PG(float, float);
#undef FX_MSDA_PG
  const int smem = msda_slab_smem(P, Q, grad_out_bf16);
  static int attr_smem[2] = {0, 0};
  const void* kern = grad_out_bf16 ? reinterpret_cast<const void*>(msda_bwd_value_kernel<bf16_t>) : reinterpret_cast<const void*>(msda_bwd_value_kernel<float>);
  if (smem > attr_smem[grad_out_bf16 ? 1 : 0]) {
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) return FX_ERR_RUNTIME;
    attr_smem[grad_out_bf16 ? 1 : 0] = smem;
  }
  bf16_t* gv = reinterpret_cast<bf16_t*>(grad_value_bf16);
  if (grad_out_bf16) hipLaunchKernelGGL(msda_bwd_value_kernel<bf16_t>, dim3(slabs, B * M), dim3(512), smem, stream, a, loc, attn, (const bf16_t*)grad_out, gv);
  else hipLaunchKernelGGL(msda_bwd_value_kernel<float>, dim3(slabs, B * M), dim3(512), smem, stream, a, loc, attn, (const float*)grad_out, gv);
  return fx_launch_status();
}

// ---- sampling locations / attention weights of one deformable layer from its raw projections, and the way back
// (MSDeformableAttention.forward, fai_detr/modelling.py:866-879, 4-d reference branch):
//   aw  = softmax over the L*P logits of a (query, head);   loc = ref_xy + off / P * ref_wh * 0.5
// One launch each way instead of ~7 elementwise launches forward (casts, softmax, views, mul / div / add) and ~8 backward - the decoder
// phases of a training step are bound by their launch count, not by their arithmetic.  One thread per (query, head).
__global__ __launch_bounds__(256) void msda_prep_kernel(const bf16_t* __restrict__ off, int ld_off, const bf16_t* __restrict__ logit, int ld_logit,
                                                        const float* __restrict__ ref, float* __restrict__ loc, float* __restrict__ aw, int BQ, int M,
                                                        int LP, int P) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= BQ * M) return;
  const int bq = i / M, h = i - bq * M;
  const float4 r = *reinterpret_cast<const float4*>(ref + (int64_t)bq * 4);
  const bf16_t* op = off + (int64_t)bq * ld_off + h * LP * 2;
  const bf16_t* lp = logit + (int64_t)bq * ld_logit + h * LP;
  float* locp = loc + ((int64_t)bq * M + h) * LP * 2;
  float* awp = aw + ((int64_t)bq * M + h) * LP;
  float mx = -INFINITY;
  for (int k = 0; k < LP; ++k) mx = fmaxf(mx, bf16_to_f32(lp[k]));
  float sum = 0.f;
  for (int k = 0; k < LP; ++k) sum += expf(bf16_to_f32(lp[k]) - mx);
  for (int k = 0; k < LP; ++k) {
    awp[k] = expf(bf16_to_f32(lp[k]) - mx) / sum;
    locp[2 * k] = r.x + bf16_to_f32(op[2 * k]) / (float)P * r.z * 0.5f;
    locp[2 * k + 1] = r.y + bf16_to_f32(op[2 * k + 1]) / (float)P * r.w * 0.5f;
  }
}

__global__ __launch_bounds__(256) void msda_prep_bwd_kernel(const float* __restrict__ grad_loc, const float* __restrict__ grad_attn,
                                                            const float* __restrict__ aw, const float* __restrict__ ref, bf16_t* __restrict__ grad_off,
                                                            int ld_off, bf16_t* __restrict__ grad_logit, int ld_logit, int BQ, int M, int LP, int P) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= BQ * M) return;
  const int bq = i / M, h = i - bq * M;
  const float4 r = *reinterpret_cast<const float4*>(ref + (int64_t)bq * 4);
  const float sx = r.z * 0.5f / (float)P, sy = r.w * 0.5f / (float)P;
  const float* gl = grad_loc + ((int64_t)bq * M + h) * LP * 2;
  const float* ga = grad_attn + ((int64_t)bq * M + h) * LP;
  const float* awp = aw + ((int64_t)bq * M + h) * LP;
  bf16_t* go = grad_off + (int64_t)bq * ld_off + h * LP * 2;
  bf16_t* gg = grad_logit + (int64_t)bq * ld_logit + h * LP;
  float dot = 0.f;
  for (int k = 0; k < LP; ++k) dot += awp[k] * ga[k];
  for (int k = 0; k < LP; ++k) {
    gg[k] = f32_to_bf16(awp[k] * (ga[k] - dot));   // softmax backward
    go[2 * k] = f32_to_bf16(gl[2 * k] * sx);
    go[2 * k + 1] = f32_to_bf16(gl[2 * k + 1] * sy);
  }
}

extern "C" int fx_msda_prep_bf16(const void* off, int ld_off, const void* logit, int ld_logit, const float* ref, float* loc, float* aw, int BQ, int M,
                                 int L, int P, fx_stream_t stream_) {
  FX_CHECK_ARG(off && logit && ref && loc && aw && BQ > 0 && M > 0 && L > 0 && P > 0 && ld_off >= M * L * P * 2 && ld_logit >= M * L * P);
  FX_CHECK_ARG(((uintptr_t)ref % 16) == 0);
  hipLaunchKernelGGL(msda_prep_kernel, dim3((BQ * M + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), (const bf16_t*)off, ld_off,
                     (const bf16_t*)logit, ld_logit, ref, loc, aw, BQ, M, L * P, P);
  return fx_launch_status();
}

extern "C" int fx_msda_prep_bwd_bf16(const float* grad_loc, const float* grad_attn, const float* aw, const float* ref, void* grad_off, int ld_off,
                                     void* grad_logit, int ld_logit, int BQ, int M, int L, int P, fx_stream_t stream_) {
  FX_CHECK_ARG(grad_loc && grad_attn && aw && ref && grad_off && grad_logit && BQ > 0 && M > 0 && L > 0 && P > 0);
  FX_CHECK_ARG(ld_off >= M * L * P * 2 && ld_logit >= M * L * P && ((uintptr_t)ref % 16) == 0);
  hipLaunchKernelGGL(msda_prep_bwd_kernel, dim3((BQ * M + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), grad_loc, grad_attn, aw,
                     ref, (bf16_t*)grad_off, ld_off, (bf16_t*)grad_logit, ld_logit, BQ, M, L * P, P);
  return fx_launch_status();
}

extern "C" int fx_msda_f32_fwd(const float* value, const int32_t* spatial_shapes, const int32_t* level_start, int L, int P, const float* loc,
                               const float* attn, float* out, int B, int S, int Q, int M, fx_stream_t stream_) {
  return fx_msda_train_fwd(value, 0, M * 32, spatial_shapes, level_start, L, P, loc, attn, out, B, S, Q, M, stream_);
}

extern "C" int fx_msda_f32_bwd(const float* value, const int32_t* spatial_shapes, const int32_t* level_start, int L, int P, const float* loc,
                               const float* attn, const float* grad_out, float* grad_value, float* grad_loc, float* grad_attn, int B, int S, int Q,
                               int M, fx_stream_t stream_) {
  return fx_msda_train_bwd(value, 0, M * 32, spatial_shapes, level_start, L, P, loc, attn, grad_out, grad_value, M * 32, 1, grad_loc, grad_attn, B, S, Q,
                           M, stream_);
}

// ------------------------------------------------------------------------------------------------
// Fused multi-tensor AdamW with global-norm clipping over ONE flat fp32 parameter buffer.
// The reference builds one param group per tensor (trainer/solver/build.py:39-138: backbone lr x0.1, no weight decay on
// norms/biases) and clips the full-model gradient norm (build.py:29-36, trainer.py:758-760).  Here every tensor is a
// segment of a flat buffer with its own (lr, weight_decay); a chunk table maps workgroups to segments.
//   pass 1: sum of squared gradients -> per-block partials (fixed order, float64)
//   pass 2: total norm -> clip coefficient (device side, no host sync) + AdamW update, torch.optim.AdamW arithmetic
__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ partial) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc += (double)g[i] * (double)g[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                     const int64_t* __restrict__ chunk_start, const int32_t* __restrict__ chunk_len,
                                                     const float* __restrict__ chunk_lr, const float* __restrict__ chunk_wd,
                                                     const double* __restrict__ partial, int npartial, float max_norm, float beta1, float beta2,
                                                     float eps, float bias_c1, float bias_c2_sqrt, float* __restrict__ total_norm_out) {
  __shared__ float s_clip;
  __shared__ int s_skip;
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int i = 0; i < npartial; ++i) tot += partial[i];
    // inf / NaN anywhere in the gradients (a non-finite loss, non-finite matching costs): the update is SKIPPED - parameters and moments stay
    // as they were, total_norm_out reports inf - instead of writing NaN into every parameter (TrainStep polls the Hungarian status
    // word only every check_every steps; the reference's default training runs under GradScaler, which skips such steps too, trainer.py:645)
    const bool finite = tot == tot && tot < 1.7976931348623157e308;
    const float norm = finite ? (float)sqrt(tot) : __builtin_inff();
    float c = 1.0f;
    if (max_norm > 0.0f) c = fminf(max_norm / (norm + 1e-6f), 1.0f);  // torch.nn.utils.clip_grad_norm_
    s_clip = c;
    s_skip = finite ? 0 : 1;
    if (blockIdx.x == 0 && total_norm_out) *total_norm_out = norm;
  }
  __syncthreads();
  if (s_skip) return;
  const float clip = s_clip;
  const int64_t start = chunk_start[blockIdx.x];
  const int len = chunk_len[blockIdx.x];
  const float lr = chunk_lr[blockIdx.x], wd = chunk_wd[blockIdx.x];
  const float step = lr / bias_c1;
  for (int i = threadIdx.x; i < len; i += 256) {
    const int64_t j = start + i;
    const float gg = g[j] * clip;
    float pp = p[j] * (1.0f - lr * wd);
    const float mm = beta1 * m[j] + (1.0f - beta1) * gg;
    const float vv = beta2 * v[j] + (1.0f - beta2) * gg * gg;
    const float denom = sqrtf(vv) / bias_c2_sqrt + eps;
    pp -= step * (mm / denom);
    p[j] = pp;
    m[j] = mm;
    v[j] = vv;
  }
}

#define NORM_BLOCKS 1024

extern "C" int fx_adamw_workspace_bytes(void) { return NORM_BLOCKS * 8; }

extern "C" int fx_adamw_step_f32(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t numel, const int64_t* chunk_start,
                                 const int32_t* chunk_len, const float* chunk_lr, const float* chunk_wd, int nchunks, int step, float beta1, float beta2,
                                 float eps, float max_grad_norm, void* workspace, float* total_norm_out, fx_stream_t stream_) {
  FX_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && numel > 0 && chunk_start && chunk_len && chunk_lr && chunk_wd && nchunks > 0 && step >= 1);
  FX_CHECK_ARG(workspace && ((uintptr_t)workspace % 8) == 0);
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  double* partial = reinterpret_cast<double*>(workspace);
  hipLaunchKernelGGL(sqnorm_kernel, dim3(NORM_BLOCKS), dim3(256), 0, stream, grads, numel, partial);
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  const float bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  hipLaunchKernelGGL(adamw_kernel, dim3(nchunks), dim3(256), 0, stream, params, grads, exp_avg, exp_avg_sq, chunk_start, chunk_len, chunk_lr, chunk_wd,
                     partial, NORM_BLOCKS, max_grad_norm, beta1, beta2, eps, bc1, bc2s, total_norm_out);
  return fx_launch_status();
}

// ---- the same step under a dynamic loss scale (fp16 build; torch.amp.GradScaler semantics, trainer/trainer.py:645,735-773) ----------------
// The loss was multiplied by state->scale before backward, so `grads` hold scale * g.  Everything GradScaler does around optimizer.step()
// happens on the device, in the launches of the step itself, with no host synchronisation:
//   unscale_()   : g * (1 / scale) folded into the update; the clip norm is ||g|| / scale
//   step()       : SKIPPED (parameters, moments and the step count untouched) when the gradients hold an inf / NaN - the fixed-order fp64
//                  sum of squares the clip needs anyway is non-finite exactly then
//   update()     : scale *= backoff on a skipped step, scale *= growth after `growth_interval` consecutive good ones
// Adam's bias correction uses the number of steps actually TAKEN (state->good_steps), as torch.optim.AdamW's own step counter would.
__global__ __launch_bounds__(256) void adamw_scaled_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                            const int64_t* __restrict__ chunk_start, const int32_t* __restrict__ chunk_len,
                                                            const float* __restrict__ chunk_lr, const float* __restrict__ chunk_wd,
                                                            const double* __restrict__ partial, int npartial, float max_norm, float beta1, float beta2,
                                                            float eps, const fx_loss_scale_state* __restrict__ state, float* __restrict__ total_norm_out) {
  __shared__ float s_mul, s_bc1, s_bc2s;
  __shared__ int s_skip;
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int i = 0; i < npartial; ++i) tot += partial[i];
    const float inv = 1.0f / state->scale;
    const bool finite = tot == tot && tot < 1.7976931348623157e308;   // not NaN, not +inf
    const float norm = finite ? (float)sqrt(tot) * inv : __builtin_inff();
    float c = 1.0f;
    if (max_norm > 0.0f) c = fminf(max_norm / (norm + 1e-6f), 1.0f);
    s_mul = c * inv;
    s_skip = finite ? 0 : 1;
    const double t = (double)(state->good_steps + 1);
    s_bc1 = (float)(1.0 - pow((double)beta1, t));
    s_bc2s = (float)sqrt(1.0 - pow((double)beta2, t));
    if (blockIdx.x == 0 && total_norm_out) *total_norm_out = norm;
  }
  __syncthreads();
  if (s_skip) return;
  const float mul = s_mul, bias_c1 = s_bc1, bias_c2_sqrt = s_bc2s;
  const int64_t start = chunk_start[blockIdx.x];
  const int len = chunk_len[blockIdx.x];
  const float lr = chunk_lr[blockIdx.x], wd = chunk_wd[blockIdx.x];
  const float step = lr / bias_c1;
  for (int i = threadIdx.x; i < len; i += 256) {
    const int64_t j = start + i;
    const float gg = g[j] * mul;
    float pp = p[j] * (1.0f - lr * wd);
    const float mm = beta1 * m[j] + (1.0f - beta1) * gg;
    const float vv = beta2 * v[j] + (1.0f - beta2) * gg * gg;
    const float denom = sqrtf(vv) / bias_c2_sqrt + eps;
    pp -= step * (mm / denom);
    p[j] = pp;
    m[j] = mm;
    v[j] = vv;
  }
}

// GradScaler.update(): one thread, launched behind the step (every block of the step has read the OLD scale by then)
__global__ void loss_scale_update_kernel(const double* __restrict__ partial, int npartial, fx_loss_scale_state* __restrict__ state, float growth,
                                         float backoff, int growth_interval) {
  double tot = 0.0;
  for (int i = 0; i < npartial; ++i) tot += partial[i];
  const bool finite = tot == tot && tot < 1.7976931348623157e308;
  if (!finite) {
    state->scale *= backoff;
    state->growth_tracker = 0;
    state->skipped_steps += 1;
  } else {
    state->good_steps += 1;
    if (++state->growth_tracker >= growth_interval) {
      state->scale *= growth;
      state->growth_tracker = 0;
    }
  }
}

extern "C" int fx_adamw_step_scaled_f32(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t numel, const int64_t* chunk_start,
                                        const int32_t* chunk_len, const float* chunk_lr, const float* chunk_wd, int nchunks, float beta1, float beta2,
                                        float eps, float max_grad_norm, void* workspace, float* total_norm_out, fx_loss_scale_state* state,
                                        float growth_factor, float backoff_factor, int growth_interval, fx_stream_t stream_) {
  FX_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && numel > 0 && chunk_start && chunk_len && chunk_lr && chunk_wd && nchunks > 0 && state);
  FX_CHECK_ARG(workspace && ((uintptr_t)workspace % 8) == 0 && growth_factor >= 1.0f && backoff_factor > 0.0f && backoff_factor <= 1.0f && growth_interval > 0);
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  double* partial = reinterpret_cast<double*>(workspace);
  hipLaunchKernelGGL(sqnorm_kernel, dim3(NORM_BLOCKS), dim3(256), 0, stream, grads, numel, partial);
  hipLaunchKernelGGL(adamw_scaled_kernel, dim3(nchunks), dim3(256), 0, stream, params, grads, exp_avg, exp_avg_sq, chunk_start, chunk_len, chunk_lr, chunk_wd,
                     partial, NORM_BLOCKS, max_grad_norm, beta1, beta2, eps, state, total_norm_out);
  hipLaunchKernelGGL(loss_scale_update_kernel, dim3(1), dim3(1), 0, stream, partial, NORM_BLOCKS, state, growth_factor, backoff_factor, growth_interval);
  return fx_launch_status();
}
