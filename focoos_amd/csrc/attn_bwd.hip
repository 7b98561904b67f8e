// Attention backward on the matrix cores (head_dim 32, any Lq / Lk, optional boolean attention mask) - the training direction of
// nn.MultiheadAttention's core as the reference runs it under autograd:
//   RT-DETR  AIFI / decoder self-attention          focoos/nn/layers/transformer.py:583-601, fai_detr/modelling.py:924-958
//   masked cross-attention of the mask decoders     focoos/models/bisenetformer/modelling.py:420-436 (fai_mf/modelling.py:509-523),
//                                                   incl. the "a query whose mask forbids every key attends everywhere" rule
// With S = Q K^T / sqrt(32), P = softmax(S [masked]), O = P V:  dV = P^T dO,  dP = dO V^T,  D_i = sum_j P_ij dP_ij,
// dS = P o (dP - D),  dQ = dS K / sqrt(32),  dK = dS^T Q / sqrt(32).
// Two kernels, both in the "transposed" register layout of the forward kernel (token_ops.hip: lanes = rows of the operand that stays
// in registers, 32x32x16 bf16 MFMAs, probabilities fed to the next MFMA from registers):
//   (1) per 32-query wave, two sweeps over the keys (K, V, K^T tiles through LDS): sweep 1 = online log-sum-exp and D (fp32, from the
//       recomputed probabilities - not from the bf16-rounded forward output), written to the workspace; sweep 2 = dS tiles -> dQ.
//   (2) per 32-key wave, one sweep over the queries (Q, dO, Q^T, dO^T tiles through LDS): P and dS tiles from the saved statistics
//       -> dV, dK.
#include "common.h"

#define AB_CT 8   // key tiles per LDS chunk in kernel (1): 8 * (2 KB K + 2 KB V + 2 KB K^T) = 48 KB
#define AB_QT 4   // query tiles per LDS chunk in kernel (2): 4 * (2 KB Q + 2 KB dO + 2 KB Q^T + 2 KB dO^T) = 32 KB

__device__ __forceinline__ int ab_kofs(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// rows [nrows_valid] x 32 channels of head `hd` -> row-major tile storage (64 B per row, 16-B chunks XOR-swizzled by (row>>2)&3)
// and, optionally, the transposed storage (per 32-row tile: [d][32 rows] bf16, 8-B groups XOR-swizzled by (d>>2)&7).
__device__ __forceinline__ void ab_stage(const bf16_t* __restrict__ src, int ld, int row0, int nrows_total, int ntiles, unsigned char* rm,
                                         unsigned char* tr, int tid) {
  for (int i = tid; i < ntiles * 32 * 4; i += 256) {
    const int lrow = i >> 2, c = i & 3;
    const int row = row0 + lrow;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < nrows_total) v = *reinterpret_cast<const uint4*>(src + (int64_t)row * ld + c * 8);
    *reinterpret_cast<uint4*>(rm + lrow * 64 + ((c ^ ((lrow >> 2) & 3)) << 4)) = v;
    if (tr) {
      const int tile = lrow >> 5, kk = lrow & 31;
      const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int d = c * 8 + e;
        const bf16_t val = (bf16_t)((e & 1) ? (w4[e >> 1] >> 16) : (w4[e >> 1] & 0xffffu));
        *reinterpret_cast<bf16_t*>(tr + tile * 2048 + d * 64 + ((((kk >> 2) ^ ((d >> 2) & 7))) << 3) + (kk & 3) * 2) = val;
      }
    }
  }
}

// A-operand fragments (row = lane & 31 of the tile, reduction index = channels 8h.. and 16+8h..) of a row-major tile
__device__ __forceinline__ void ab_frag_rows(const unsigned char* rm, int trow, int h, bf16x8& f0, bf16x8& f1) {
  const unsigned char* r = rm + trow * 64;
  const int sw = (trow >> 2) & 3;
  f0 = *reinterpret_cast<const bf16x8*>(r + ((h ^ sw) << 4));
  f1 = *reinterpret_cast<const bf16x8*>(r + (((2 + h) ^ sw) << 4));
}

// A-operand fragments of a transposed tile (row = channel d = lane & 31, reduction index = the tile's rows in the ab_kofs order that
// matches a 16-value accumulator column packed as two bf16x8)
__device__ __forceinline__ void ab_frag_tr(const unsigned char* tr_tile, int d, int h, bf16x8& f0, bf16x8& f1) {
  const unsigned char* r = tr_tile + d * 64;
  const int sw = (d >> 2) & 7;
  const uint2 a = *reinterpret_cast<const uint2*>(r + ((h ^ sw) << 3));
  const uint2 b = *reinterpret_cast<const uint2*>(r + (((h + 2) ^ sw) << 3));
  const uint2 c = *reinterpret_cast<const uint2*>(r + (((h + 4) ^ sw) << 3));
  const uint2 e = *reinterpret_cast<const uint2*>(r + (((h + 6) ^ sw) << 3));
  f0 = __builtin_bit_cast(bf16x8, make_uint4(a.x, a.y, b.x, b.y));
  f1 = __builtin_bit_cast(bf16x8, make_uint4(c.x, c.y, e.x, e.y));
}

__device__ __forceinline__ void ab_load_row_frags(const bf16_t* base, int64_t row, int ld, bool ok, int h, bf16x8& f0, bf16x8& f1) {
  uint4 a = make_uint4(0, 0, 0, 0), c = make_uint4(0, 0, 0, 0);
  if (ok) {
    const bf16_t* p = base + row * ld;
    a = *reinterpret_cast<const uint4*>(p + 8 * h);
    c = *reinterpret_cast<const uint4*>(p + 16 + 8 * h);
  }
  f0 = __builtin_bit_cast(bf16x8, a);
  f1 = __builtin_bit_cast(bf16x8, c);
}

#define AB_SCALE 0.17677669529663687f                        // 1 / sqrt(32)
#define AB_SCALE2 (0.17677669529663687f * 1.4426950408889634f)  // ... * log2(e): softmax in the exp2 domain

// ------------------------------------------------------------------------------------------------ (1) statistics + dQ
template <bool MASKED>
__global__ __launch_bounds__(256) void mha32_bwd_dq_kernel(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ k, int ldk,
                                                           const bf16_t* __restrict__ v, int ldv, const bf16_t* __restrict__ dout, int lddo,
                                                           bf16_t* __restrict__ dq, int lddq, float* __restrict__ stats, int Lq, int Lk, int heads,
                                                           const uint32_t* __restrict__ mask, int ldm) {
  __shared__ __attribute__((aligned(16))) unsigned char ks[AB_CT * 2048];
  __shared__ __attribute__((aligned(16))) unsigned char vs[AB_CT * 2048];
  __shared__ __attribute__((aligned(16))) unsigned char kt[AB_CT * 2048];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.y, b = bh / heads, hd = bh % heads;
  const int T = (Lk + 31) >> 5;
  const bf16_t* kb = k + (int64_t)b * Lk * ldk + hd * 32;
  const bf16_t* vb = v + (int64_t)b * Lk * ldv + hd * 32;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const bool active = q0 < Lq;
  const int j = lane & 31, h = lane >> 5;
  const int qi = q0 + j;
  bf16x8 qf0, qf1, df0, df1;
  ab_load_row_frags(q + (int64_t)b * Lq * ldq + hd * 32, qi, ldq, qi < Lq, h, qf0, qf1);
  ab_load_row_frags(dout + (int64_t)b * Lq * lddo + hd * 32, qi, lddo, qi < Lq, h, df0, df1);
  const uint32_t* mrow = MASKED ? mask + ((int64_t)b * Lq + (qi < Lq ? qi : 0)) * ldm : nullptr;
  float m = -INFINITY, l = 0.0f, Dn = 0.0f, m2 = -INFINITY, l2 = 0.0f, Dn2 = 0.0f;
  float lse = 0.0f, Dv = 0.0f;
  bool use_mask = false;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll 1
  for (int sweep = 0; sweep < 2; ++sweep) {
#pragma unroll 1
    for (int c0 = 0; c0 < T; c0 += AB_CT) {
      const int nt = (T - c0) < AB_CT ? (T - c0) : AB_CT;
      __syncthreads();
      ab_stage(kb, ldk, c0 * 32, Lk, nt, ks, sweep ? kt : nullptr, tid);
      ab_stage(vb, ldv, c0 * 32, Lk, nt, vs, nullptr, tid);
      __syncthreads();
      if (!active) continue;
      for (int t = 0; t < nt; ++t) {
        bf16x8 kf0, kf1, vf0, vf1;
        ab_frag_rows(ks, t * 32 + j, h, kf0, kf1);
        ab_frag_rows(vs, t * 32 + j, h, vf0, vf1);
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.0f, dp[r] = 0.0f;
        s = FX_MFMA_32x32x16(kf0, qf0, s);
        s = FX_MFMA_32x32x16(kf1, qf1, s);
        dp = FX_MFMA_32x32x16(vf0, df0, dp);
        dp = FX_MFMA_32x32x16(vf1, df1, dp);
        uint32_t mw = 0;
        if (MASKED) mw = mrow[c0 + t];
        if (sweep == 0) {
          float mx = -INFINITY, mx2 = -INFINITY;
          float s2[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int kofs = ab_kofs(r, h);
            s[r] = ((c0 + t) * 32 + kofs) < Lk ? s[r] * AB_SCALE2 : -INFINITY;
            mx = fmaxf(mx, s[r]);
            if (MASKED) {
              s2[r] = ((mw >> kofs) & 1u) ? -INFINITY : s[r];
              mx2 = fmaxf(mx2, s2[r]);
            }
          }
          {
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mn = fmaxf(m, mx);   // finite: every tile holds at least one real key
            const float alpha = __builtin_amdgcn_exp2f(m - mn);
            m = mn;
            float ps = 0.0f, pd = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float p = __builtin_amdgcn_exp2f(s[r] - mn);
              ps += p;
              pd = fmaf(p, dp[r], pd);
            }
            l = l * alpha + ps;
            Dn = Dn * alpha + pd;
          }
          if (MASKED) {
            mx2 = fmaxf(mx2, __shfl_xor(mx2, 32, 64));
            const float mn = fmaxf(m2, mx2);
            const float ms = mn == -INFINITY ? 0.0f : mn;   // nothing allowed so far: keep l2 = 0 without NaNs
            const float alpha = __builtin_amdgcn_exp2f(m2 - ms);
            m2 = mn;
            float ps = 0.0f, pd = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float p = __builtin_amdgcn_exp2f(s2[r] - ms);
              ps += p;
              pd = fmaf(p, dp[r], pd);
            }
            l2 = l2 * alpha + ps;
            Dn2 = Dn2 * alpha + pd;
          }
        } else {
          float ds[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int kofs = ab_kofs(r, h);
            const bool dead = ((c0 + t) * 32 + kofs) >= Lk || (MASKED && use_mask && ((mw >> kofs) & 1u));
            const float p = dead ? 0.0f : __builtin_amdgcn_exp2f(s[r] * AB_SCALE2 - lse);
            ds[r] = p * (dp[r] - Dv);
          }
          bf16x8 kt0, kt1;
          ab_frag_tr(kt + t * 2048, j, h, kt0, kt1);
          const uint4 p0 = pack_bf16x8(ds), p1 = pack_bf16x8(ds + 8);
          acc = FX_MFMA_32x32x16(kt0, __builtin_bit_cast(bf16x8, p0), acc);
          acc = FX_MFMA_32x32x16(kt1, __builtin_bit_cast(bf16x8, p1), acc);
        }
      }
    }
    if (sweep == 0 && active) {
      l += __shfl_xor(l, 32, 64);
      Dn += __shfl_xor(Dn, 32, 64);
      lse = m + __builtin_amdgcn_logf(l);   // v_log_f32 = log2
      Dv = Dn / l;
      if (MASKED) {
        l2 += __shfl_xor(l2, 32, 64);
        Dn2 += __shfl_xor(Dn2, 32, 64);
        use_mask = l2 > 0.0f;   // some key allowed; otherwise the row attends everywhere (reference: the mask row is cleared)
        if (use_mask) {
          lse = m2 + __builtin_amdgcn_logf(l2);
          Dv = Dn2 / l2;
        }
      }
      if (h == 0 && qi < Lq) {
        float* sp = stats + ((int64_t)bh * Lq + qi) * 3;
        sp[0] = lse;
        sp[1] = Dv;
        sp[2] = use_mask ? 1.0f : 0.0f;
      }
    }
  }
  if (!active || qi >= Lq) return;
  bf16_t* dp_ = dq + ((int64_t)b * Lq + qi) * lddq + hd * 32;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    uint2 o;
    o.x = pack_bf16x2(acc[4 * g] * AB_SCALE, acc[4 * g + 1] * AB_SCALE);
    o.y = pack_bf16x2(acc[4 * g + 2] * AB_SCALE, acc[4 * g + 3] * AB_SCALE);
    *reinterpret_cast<uint2*>(dp_ + 8 * g + 4 * h) = o;
  }
}

// ------------------------------------------------------------------------------------------------ (2) dK, dV
template <bool MASKED>
__global__ __launch_bounds__(256) void mha32_bwd_dkv_kernel(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ k, int ldk,
                                                            const bf16_t* __restrict__ v, int ldv, const bf16_t* __restrict__ dout, int lddo,
                                                            bf16_t* __restrict__ dk, int lddk, bf16_t* __restrict__ dv, int lddv,
                                                            const float* __restrict__ stats, int Lq, int Lk, int heads,
                                                            const uint32_t* __restrict__ mask, int ldm) {
  __shared__ __attribute__((aligned(16))) unsigned char qs[AB_QT * 2048];
  __shared__ __attribute__((aligned(16))) unsigned char dos[AB_QT * 2048];
  __shared__ __attribute__((aligned(16))) unsigned char qt[AB_QT * 2048];
  __shared__ __attribute__((aligned(16))) unsigned char dot[AB_QT * 2048];
  __shared__ float st_s[AB_QT * 32][3];
  __shared__ uint32_t mw_s[4][AB_QT * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.y, b = bh / heads, hd = bh % heads;
  const int TQ = (Lq + 31) >> 5;
  const bf16_t* qb = q + (int64_t)b * Lq * ldq + hd * 32;
  const bf16_t* db = dout + (int64_t)b * Lq * lddo + hd * 32;
  const int k0 = blockIdx.x * 128 + wave * 32;
  const bool active = k0 < Lk;
  const int j = lane & 31, h = lane >> 5;
  const int kj = k0 + j;
  bf16x8 kf0, kf1, vf0, vf1;
  ab_load_row_frags(k + (int64_t)b * Lk * ldk + hd * 32, kj, ldk, kj < Lk, h, kf0, kf1);
  ab_load_row_frags(v + (int64_t)b * Lk * ldv + hd * 32, kj, ldv, kj < Lk, h, vf0, vf1);
  f32x16 ak, av;
#pragma unroll
  for (int r = 0; r < 16; ++r) ak[r] = 0.0f, av[r] = 0.0f;
#pragma unroll 1
  for (int c0 = 0; c0 < TQ; c0 += AB_QT) {
    const int nt = (TQ - c0) < AB_QT ? (TQ - c0) : AB_QT;
    __syncthreads();
    ab_stage(qb, ldq, c0 * 32, Lq, nt, qs, qt, tid);
    ab_stage(db, lddo, c0 * 32, Lq, nt, dos, dot, tid);
    for (int i = tid; i < nt * 32; i += 256) {
      const int qi = c0 * 32 + i;
      const bool ok = qi < Lq;
      const float* sp = stats + ((int64_t)bh * Lq + (ok ? qi : 0)) * 3;
      st_s[i][0] = ok ? sp[0] : 0.0f;
      st_s[i][1] = ok ? sp[1] : 0.0f;
      st_s[i][2] = ok ? sp[2] : 0.0f;
      if (MASKED) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const int kw = (blockIdx.x * 128 + w * 32) >> 5;
          mw_s[w][i] = (ok && kw < ldm) ? mask[((int64_t)b * Lq + qi) * ldm + kw] : 0u;
        }
      }
    }
    __syncthreads();
    if (!active) continue;
    for (int t = 0; t < nt; ++t) {
      bf16x8 qf0, qf1, df0, df1;
      ab_frag_rows(qs, t * 32 + j, h, qf0, qf1);
      ab_frag_rows(dos, t * 32 + j, h, df0, df1);
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.0f, dp[r] = 0.0f;
      s = FX_MFMA_32x32x16(qf0, kf0, s);
      s = FX_MFMA_32x32x16(qf1, kf1, s);
      dp = FX_MFMA_32x32x16(df0, vf0, dp);
      dp = FX_MFMA_32x32x16(df1, vf1, dp);
      float pf[16], ds[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int li = t * 32 + ab_kofs(r, h);
        const int qi = c0 * 32 + li;
        bool dead = qi >= Lq || kj >= Lk;
        if (MASKED) dead = dead || (st_s[li][2] != 0.0f && ((mw_s[wave][li] >> j) & 1u));
        const float p = dead ? 0.0f : __builtin_amdgcn_exp2f(s[r] * AB_SCALE2 - st_s[li][0]);
        pf[r] = p;
        ds[r] = p * (dp[r] - st_s[li][1]);
      }
      bf16x8 t0, t1;
      ab_frag_tr(dot + t * 2048, j, h, t0, t1);
      uint4 p0 = pack_bf16x8(pf), p1 = pack_bf16x8(pf + 8);
      av = FX_MFMA_32x32x16(t0, __builtin_bit_cast(bf16x8, p0), av);
      av = FX_MFMA_32x32x16(t1, __builtin_bit_cast(bf16x8, p1), av);
      ab_frag_tr(qt + t * 2048, j, h, t0, t1);
      p0 = pack_bf16x8(ds), p1 = pack_bf16x8(ds + 8);
      ak = FX_MFMA_32x32x16(t0, __builtin_bit_cast(bf16x8, p0), ak);
      ak = FX_MFMA_32x32x16(t1, __builtin_bit_cast(bf16x8, p1), ak);
    }
  }
  if (!active || kj >= Lk) return;
  bf16_t* kp = dk + ((int64_t)b * Lk + kj) * lddk + hd * 32;
  bf16_t* vp = dv + ((int64_t)b * Lk + kj) * lddv + hd * 32;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    uint2 o;
    o.x = pack_bf16x2(ak[4 * g] * AB_SCALE, ak[4 * g + 1] * AB_SCALE);
    o.y = pack_bf16x2(ak[4 * g + 2] * AB_SCALE, ak[4 * g + 3] * AB_SCALE);
    *reinterpret_cast<uint2*>(kp + 8 * g + 4 * h) = o;
    o.x = pack_bf16x2(av[4 * g], av[4 * g + 1]);
    o.y = pack_bf16x2(av[4 * g + 2], av[4 * g + 3]);
    *reinterpret_cast<uint2*>(vp + 8 * g + 4 * h) = o;
  }
}

extern "C" size_t fx_mha_bwd_workspace_bytes(int B, int Lq, int Lk, int heads) {
  if (B <= 0 || Lq <= 0 || Lk <= 0 || heads <= 0) return 0;
  return (size_t)3 * B * heads * Lq * sizeof(float);   // per (image, head, query): log-sum-exp (log2 domain), D, "mask in effect" flag
}

extern "C" int fx_mha_masked_bwd_bf16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* dout, int lddo, void* dq,
                                      int lddq, void* dk, int lddk, void* dv, int lddv, int B, int Lq, int Lk, int heads, const uint32_t* mask_bits,
                                      int ld_mask_words, void* workspace, size_t workspace_bytes, fx_stream_t stream_) {
  FX_CHECK_ARG(q && k && v && dout && dq && dk && dv && workspace && B > 0 && Lq > 0 && Lk > 0 && heads > 0);
  FX_CHECK_ARG(workspace_bytes >= fx_mha_bwd_workspace_bytes(B, Lq, Lk, heads) && ((uintptr_t)workspace % 4) == 0);
  FX_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && lddo % 8 == 0 && lddq % 8 == 0 && lddk % 8 == 0 && lddv % 8 == 0);
  FX_CHECK_ARG(ldq >= heads * 32 && ldk >= heads * 32 && ldv >= heads * 32 && lddo >= heads * 32);
  FX_CHECK_ARG(mask_bits == nullptr || ld_mask_words >= (Lk + 31) / 32);
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  float* stats = reinterpret_cast<float*>(workspace);
  const dim3 gq((Lq + 127) / 128, B * heads), gk((Lk + 127) / 128, B * heads), block(256);
  if (mask_bits) {
    hipLaunchKernelGGL(mha32_bwd_dq_kernel<true>, gq, block, 0, stream, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv,
                       (const bf16_t*)dout, lddo, (bf16_t*)dq, lddq, stats, Lq, Lk, heads, mask_bits, ld_mask_words);
    hipLaunchKernelGGL(mha32_bwd_dkv_kernel<true>, gk, block, 0, stream, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv,
                       (const bf16_t*)dout, lddo, (bf16_t*)dk, lddk, (bf16_t*)dv, lddv, stats, Lq, Lk, heads, mask_bits, ld_mask_words);
  } else {
    hipLaunchKernelGGL(mha32_bwd_dq_kernel<false>, gq, block, 0, stream, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv,
                       (const bf16_t*)dout, lddo, (bf16_t*)dq, lddq, stats, Lq, Lk, heads, (const uint32_t*)nullptr, 0);
    hipLaunchKernelGGL(mha32_bwd_dkv_kernel<false>, gk, block, 0, stream, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv,
                       (const bf16_t*)dout, lddo, (bf16_t*)dk, lddk, (bf16_t*)dv, lddv, stats, Lq, Lk, heads, (const uint32_t*)nullptr, 0);
  }
  return fx_launch_status();
}

// The unmasked entry point keeps its round-1 signature (`o` is no longer read: D comes from the recomputed probabilities).
extern "C" int fx_mha_bwd_bf16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* o, int ldo, const void* dout,
                               int lddo, void* dq, int lddq, void* dk, int lddk, void* dv, int lddv, int B, int Lq, int Lk, int heads,
                               void* workspace, size_t workspace_bytes, fx_stream_t stream_) {
  (void)o;
  (void)ldo;
  return fx_mha_masked_bwd_bf16(q, ldq, k, ldk, v, ldv, dout, lddo, dq, lddq, dk, lddk, dv, lddv, B, Lq, Lk, heads, nullptr, 0, workspace,
                                workspace_bytes, stream_);
}
