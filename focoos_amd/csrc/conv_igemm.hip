// Implicit-GEMM convolution / linear layer for gfx950 (MI355X): bf16 operands on the MFMA matrix
// cores (v_mfma_f32_32x32x16_bf16), fp32 accumulate, fused bias + residual + activation epilogue.
//
// GEMM view:  D[n][m] = sum_k Wt[n][k] * X[m][k]   with  m = output pixel (b,ho,wo), n = output
// channel, k = (kh,kw,c).  Both operands are K-contiguous (NHWC activations, [N][KH][KW][C] weights),
// so every MFMA fragment is one 16-byte ds_read_b128.  The weights are the MFMA "A" operand and the
// pixels the "B" operand: the accumulator layout then gives each lane 4 CONSECUTIVE output channels of
// one pixel, which the epilogue stages through LDS and writes back as full 16-byte channel vectors.
//
// Staging: global -> registers -> LDS (XOR-swizzled, conflict-free for ds_read_b128), double-buffered,
// next tile's global loads issued before the MFMA block of the current tile (guide T14), one barrier
// per K-step.  im2col addressing is done on the fly: a K-step never straddles a filter tap because
// BK divides C.  Zero padding / M tails are predicated loads.
#include <stdio.h>

#include "conv_common.h"

template <int BM, int BN, int BK, int WM, int WN, bool POOL, int NSTAGE = 2>
__global__ __launch_bounds__(256, (BK == 32 ? 4 : 2)) void conv_igemm_kernel(const ConvArgs p) {
  constexpr int CPR = BK / 8, RPP = 256 / CPR;
  constexpr int A_PASS = BM / RPP, B_PASS = (BN + RPP - 1) / RPP;
  constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int EPI_LD = BN + 4;
  static_assert(WM * WN == 4, "4 waves");
  static_assert(BM % RPP == 0, "A passes");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int bid = fx_xcd_remap(blockIdx.x, gridDim.x);
  const int mt = bid / p.nNt, nt = bid % p.nNt;
  const int m0 = mt * BM, n0 = nt * BN;

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

  const int kc = tid % CPR, r_in = tid / CPR;
  int a_hi0[A_PASS], a_wi0[A_PASS], a_off[A_PASS];  // a_off: element offset of (b, hi0, wi0, kc*8); may be "before" the row, fixed by delta
  bool a_ok[A_PASS];
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int i = 0; i < A_PASS; ++i) {
    int m = m0 + r_in + i * RPP;
    a_ok[i] = m < p.M;
    int mm = a_ok[i] ? m : 0;
    int b = mm / HoWo, rem = mm - b * HoWo;
    int ho = rem / p.Wo, wo = rem - ho * p.Wo;
    if constexpr (POOL) {
      a_hi0[i] = ho * 2;
      a_wi0[i] = wo * 2;
    } else {
      a_hi0[i] = ho * p.stride - p.pad;
      a_wi0[i] = wo * p.stride - p.pad;
    }
    a_off[i] = ((b * p.H + a_hi0[i]) * p.W + a_wi0[i]) * p.ldx + kc * 8;
  }
  unsigned w_off[B_PASS];
#pragma unroll
  for (int i = 0; i < B_PASS; ++i) w_off[i] = (unsigned)(((n0 + r_in + i * RPP) * p.Ktot + kc * 8) * 2);

  uint4 ra[A_PASS], rb[B_PASS];

#define FX_LOAD_TILES(KH_, KW_, C0_, KBASE_)                                                                         \
  {                                                                                                                  \
    const int delta_ = ((KH_)*p.W + (KW_)) * p.ldx + (C0_);                                                          \
    _Pragma("unroll") for (int i = 0; i < A_PASS; ++i) {                                                             \
      if constexpr (!POOL) {                                                                                         \
        int hi = a_hi0[i] + (KH_), wi = a_wi0[i] + (KW_);                                                            \
        bool ok = a_ok[i] && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;                           \
        ra[i] = buf_load16(xr, ok ? (unsigned)(a_off[i] + delta_) * 2u : FX_OOB);                                     \
      } else {                                                                                                       \
        float s_[8] = {0, 0, 0, 0, 0, 0, 0, 0};                                                                      \
        int cnt_ = 0;                                                                                                \
        _Pragma("unroll") for (int dy = 0; dy < 2; ++dy) _Pragma("unroll") for (int dx = 0; dx < 2; ++dx) {          \
          bool ok = a_ok[i] && (a_hi0[i] + dy) < p.H && (a_wi0[i] + dx) < p.W;                                       \
          uint4 v_ = buf_load16(xr, ok ? (unsigned)(a_off[i] + delta_ + (dy * p.W + dx) * p.ldx) * 2u : FX_OOB);      \
          float f_[8];                                                                                               \
          unpack_bf16x8(v_, f_);                                                                                     \
          _Pragma("unroll") for (int j = 0; j < 8; ++j) s_[j] += f_[j];                                              \
          cnt_ += ok ? 1 : 0;                                                                                        \
        }                                                                                                            \
        float inv_ = cnt_ > 0 ? 1.0f / (float)cnt_ : 0.0f;                                                           \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) s_[j] *= inv_;                                                 \
        ra[i] = pack_bf16x8(s_);                                                                                     \
      }                                                                                                              \
    }                                                                                                                \
    _Pragma("unroll") for (int i = 0; i < B_PASS; ++i) {                                                             \
      if (BN % RPP == 0 || r_in + i * RPP < BN) rb[i] = buf_load16(wr, w_off[i] + (unsigned)(KBASE_)*2u);            \
    }                                                                                                                \
  }
#define FX_STORE_TILES(S_)                                                                                           \
  {                                                                                                                  \
    unsigned char* A_ = smem + (S_)*STAGE;                                                                           \
    unsigned char* B_ = A_ + A_BYTES;                                                                                \
    _Pragma("unroll") for (int i = 0; i < A_PASS; ++i) *reinterpret_cast<uint4*>(A_ + lds_off<BK>(r_in + i * RPP, kc)) = ra[i]; \
    _Pragma("unroll") for (int i = 0; i < B_PASS; ++i) {                                                             \
      if (BN % RPP == 0 || r_in + i * RPP < BN) *reinterpret_cast<uint4*>(B_ + lds_off<BK>(r_in + i * RPP, kc)) = rb[i]; \
    }                                                                                                                \
  }
#define FX_COMPUTE(S_)                                                                                               \
  {                                                                                                                  \
    const unsigned char* A_ = smem + (S_)*STAGE;                                                                     \
    const unsigned char* B_ = A_ + A_BYTES;                                                                          \
    _Pragma("unroll") for (int kk = 0; kk < BK / 16; ++kk) {                                                         \
      const int chunk = kk * 2 + lhalf;                                                                              \
      bf16x8 xa[TM], wb[TN];                                                                                         \
      _Pragma("unroll") for (int i = 0; i < TM; ++i) xa[i] =                                                         \
          *reinterpret_cast<const bf16x8*>(A_ + lds_off<BK>(wm * WTM + i * 32 + lrow, chunk));                       \
      _Pragma("unroll") for (int i = 0; i < TN; ++i) wb[i] =                                                         \
          *reinterpret_cast<const bf16x8*>(B_ + lds_off<BK>(wn * WTN + i * 32 + lrow, chunk));                       \
      _Pragma("unroll") for (int a = 0; a < TN; ++a) _Pragma("unroll") for (int b = 0; b < TM; ++b) acc[a][b] =      \
          FX_MFMA_32x32x16(wb[a], xa[b], acc[a][b]);                                 \
    }                                                                                                                \
  }

  f32x16 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  const int T = p.KH * p.KW * (p.C / BK);
  const int lrow = lane & 31, lhalf = lane >> 5;
  int kh = 0, kw = 0, c0 = 0, kbase = 0;
  FX_LOAD_TILES(kh, kw, c0, kbase);
  FX_STORE_TILES(0);
  __syncthreads();
  for (int t = 0; t < T - 1; ++t) {
    const int cur = (NSTAGE == 2) ? (t & 1) : 0;
    c0 += BK;
    kbase += BK;
    if (c0 == p.C) {
      c0 = 0;
      if (++kw == p.KW) {
        kw = 0;
        ++kh;
      }
    }
    FX_LOAD_TILES(kh, kw, c0, kbase);  // next tile's global loads fly under this tile's MFMAs
    FX_COMPUTE(cur);
    if constexpr (NSTAGE == 2) {
      FX_STORE_TILES(cur ^ 1);
    } else {
      __syncthreads();  // single LDS stage (whole-K-in-one-shot tiles): everyone done reading before it is overwritten
      FX_STORE_TILES(0);
    }
    __syncthreads();
  }
  // The residual tile is independent of the GEMM: fetch it now (into the registers the tile prefetch no longer
  // needs) so its latency hides under the last MFMA block and the LDS staging instead of serialising the epilogue.
  constexpr int TPR = BN / 8, RPP2 = 256 / TPR, HALF = 64, PER_HALF = HALF / RPP2, NRES = BM / RPP2;
  const int col8 = (tid % TPR) * 8;
  const int n = n0 + col8;
  const bool n_ok = n < p.Nstore;
  uint4 rres[NRES];
  if (p.res) {
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, p.r_bytes, 0x00020000);
#pragma unroll
    for (int q = 0; q < NRES; ++q) {
      const int m = m0 + (q / PER_HALF) * HALF + tid / TPR + (q % PER_HALF) * RPP2;
      rres[q] = buf_load16(rr, (m < p.M && n_ok) ? (unsigned)(m * p.ldr + n) * 2u : FX_OOB);
    }
  }
  FX_COMPUTE((NSTAGE == 2) ? ((T - 1) & 1) : 0);
  __syncthreads();
#undef FX_LOAD_TILES
#undef FX_STORE_TILES
#undef FX_COMPUTE

  // ---- epilogue: accumulators -> LDS (fp32, 64 rows at a time) -> coalesced 16-byte channel vectors
  float* stg = reinterpret_cast<float*>(smem);
  float bs[8];
  if (p.bias && n_ok) {
    float4 b0 = *reinterpret_cast<const float4*>(p.bias + n), b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
    bs[0] = b0.x; bs[1] = b0.y; bs[2] = b0.z; bs[3] = b0.w; bs[4] = b1.x; bs[5] = b1.y; bs[6] = b1.z; bs[7] = b1.w;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) bs[j] = 0.0f;
  }
#pragma unroll
  for (int hh = 0; hh < BM / HALF; ++hh) {
    if ((wm * WTM) / HALF == hh) {
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            int ml = wm * WTM + b * 32 + lrow - hh * HALF;
            int nl = wn * WTN + a * 32 + 8 * g + 4 * lhalf;
            float4 v = make_float4(acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]);
            *reinterpret_cast<float4*>(stg + ml * EPI_LD + nl) = v;
          }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER_HALF; ++j) {
      const int r = tid / TPR + j * RPP2;
      const int m = m0 + hh * HALF + r;
      if (m < p.M && n_ok) {
        float4 v0 = *reinterpret_cast<const float4*>(stg + r * EPI_LD + col8);
        float4 v1 = *reinterpret_cast<const float4*>(stg + r * EPI_LD + col8 + 4);
        float v[8] = {v0.x + bs[0], v0.y + bs[1], v0.z + bs[2], v0.w + bs[3], v1.x + bs[4], v1.y + bs[5], v1.z + bs[6], v1.w + bs[7]};
        float rf[8];
        if (p.res) {
          unpack_bf16x8(rres[hh * PER_HALF + j], rf);
          if (!p.res_after) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += rf[i];
          }
        }
        if (p.act != FX_ACT_NONE) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = fx_act(v[i], p.act);
        }
        if (p.res && p.res_after == 1) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] += rf[i];
        } else if (p.res && p.res_after == 2) {  // ReLU mask of a saved activation (training: relu backward fused into the dgrad conv)
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = rf[i] > 0.0f ? v[i] : 0.0f;
        }
        if (p.mask) {   // ReLU backward of the layer that consumes this gradient (see fx_conv_desc.mask)
          float mf[8];
          unpack_bf16x8(*reinterpret_cast<const uint4*>(p.mask + (size_t)m * p.ldm + n), mf);
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = mf[i] > 0.0f ? v[i] : 0.0f;
        }
        int64_t yoff;
        if (p.y_bstride) {
          int bb = m / HoWo;
          yoff = (int64_t)bb * p.y_bstride + (int64_t)(m - bb * HoWo) * p.ldy + n;
        } else {
          yoff = (int64_t)m * p.ldy + n;
        }
        if (p.out_f32) {
          float* dst = reinterpret_cast<float*>(p.y) + yoff;
          *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
          bf16_t* dst = reinterpret_cast<bf16_t*>(p.y) + yoff;
          *reinterpret_cast<uint4*>(dst) = pack_bf16x8(v);
        }
      }
    }
    if (hh + 1 < BM / HALF) __syncthreads();
  }
}

template <int BM, int BN, int BK, int WM, int WN, bool POOL, int NSTAGE = 2>
static int launch_conv(ConvArgs& a, hipStream_t stream) {
  constexpr int STAGE = (BM + BN) * BK * 2;
  constexpr int EPI = 64 * (BN + 4) * 4;  // the epilogue stages 64 rows at a time
  constexpr int SMEM = (NSTAGE * STAGE > EPI) ? NSTAGE * STAGE : EPI;
  static bool attr_set = false;
  auto kern = conv_igemm_kernel<BM, BN, BK, WM, WN, POOL, NSTAGE>;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess)
      return FX_ERR_RUNTIME;
    attr_set = true;
  }
  a.nNt = (a.N + BN - 1) / BN;
  int nMt = (a.M + BM - 1) / BM;
  dim3 grid(nMt * a.nNt), block(256);
  hipLaunchKernelGGL(kern, grid, block, SMEM, stream, a);
  return fx_launch_status();
}

// Validates a descriptor and fills the kernel arguments (shared by the launch and by fx_conv2d_variant).
static int conv_prepare(const fx_conv_desc* d, ConvArgs& a) {
  FX_CHECK_ARG(d && d->x && d->w && d->y);
  FX_CHECK_ARG(d->B > 0 && d->H > 0 && d->W > 0 && d->Ho > 0 && d->Wo > 0 && d->N > 0);
  FX_CHECK_ARG(d->C > 0 && d->C % 32 == 0 && d->ldx >= d->C && d->ldx % 8 == 0);
  FX_CHECK_ARG(d->KH >= 1 && d->KW >= 1 && d->stride >= 1 && d->pad >= 0);
  const int Nstore = (d->N + 7) / 8 * 8;
  FX_CHECK_ARG(d->ldy >= Nstore && d->ldy % 8 == 0);
  FX_CHECK_ARG(!d->residual || (d->ldr >= Nstore && d->ldr % 8 == 0));
  FX_CHECK_ARG(((uintptr_t)d->x % 16) == 0 && ((uintptr_t)d->w % 16) == 0 && ((uintptr_t)d->y % 16) == 0);
  FX_CHECK_ARG(!d->residual || ((uintptr_t)d->residual % 16) == 0);
  FX_CHECK_ARG(!d->bias || ((uintptr_t)d->bias % 16) == 0);
  if (d->pool2) {
    FX_CHECK_ARG(d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0);
    FX_CHECK_ARG(d->Ho == (d->H + 1) / 2 && d->Wo == (d->W + 1) / 2);
  } else {
    FX_CHECK_ARG(d->Ho == (d->H + 2 * d->pad - d->KH) / d->stride + 1);
    FX_CHECK_ARG(d->Wo == (d->W + 2 * d->pad - d->KW) / d->stride + 1);
  }
  if ((int64_t)d->B * d->H * d->W >= (1ll << 31) || (int64_t)d->B * d->Ho * d->Wo >= (1ll << 31)) return FX_ERR_UNSUPPORTED;
  // 32-bit byte offsets in the buffer loads: activation / weight views must stay below 4 GiB
  const int64_t x_bytes = ((int64_t)d->B * d->H * d->W - 1) * d->ldx * 2 + (int64_t)d->C * 2;
  const int64_t Npad = (int64_t)(d->N + 127) / 128 * 128;
  const int64_t w_bytes = Npad * d->KH * d->KW * d->C * 2;
  const int64_t r_bytes = d->residual ? (((int64_t)d->B * d->Ho * d->Wo - 1) * d->ldr + Nstore) * 2 : 0;
  if (x_bytes >= 0xFFFFFFF0ll || w_bytes >= 0xFFFFFFF0ll || r_bytes >= 0xFFFFFFF0ll) return FX_ERR_UNSUPPORTED;
  a.x = reinterpret_cast<const bf16_t*>(d->x);
  a.w = reinterpret_cast<const bf16_t*>(d->w);
  a.bias = d->bias;
  a.res = reinterpret_cast<const bf16_t*>(d->residual);
  a.y = d->y;
  a.B = d->B; a.H = d->H; a.W = d->W; a.C = d->C; a.ldx = d->ldx;
  a.Ho = d->Ho; a.Wo = d->Wo; a.N = d->N; a.ldy = d->ldy; a.ldr = d->ldr;
  a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
  a.act = d->act; a.out_f32 = d->out_f32; a.res_after = d->residual_after_act;
  a.y_bstride = d->y_batch_stride;
  FX_CHECK_ARG(d->y_batch_stride >= 0 && d->y_batch_stride % 8 == 0);
  a.M = d->B * d->Ho * d->Wo;
  a.Ktot = d->KH * d->KW * d->C;
  a.Nstore = Nstore;
  a.nNt = 0;
  a.x_bytes = (unsigned)x_bytes;
  a.w_bytes = (unsigned)w_bytes;
  a.r_bytes = (unsigned)r_bytes;
  a.mask = reinterpret_cast<const bf16_t*>(d->mask);
  a.ldm = d->ldm;
  a.m_bytes = 0;
  if (d->mask) {
    FX_CHECK_ARG(d->ldm >= Nstore && d->ldm % 8 == 0 && ((uintptr_t)d->mask % 16) == 0 && !d->out_f32 && !d->pool2);
    const int64_t m_bytes = (((int64_t)d->B * d->Ho * d->Wo - 1) * d->ldm + Nstore) * 2;
    if (m_bytes >= 0xFFFFFFF0ll) return FX_ERR_UNSUPPORTED;
    a.m_bytes = (unsigned)m_bytes;
  }
  return FX_OK;
}

// Which kernel runs a layer - ONE routing function for the launch and for the label bench.py reports (fx_conv2d_variant).
enum ConvRoute { R_C3_FLAT, R_C3_S2, R_PW_FLAT, R_SMALL_M, R_DMA, R_POOL, R_K64_N128, R_K64_N64, R_K64_N32, R_K32_N128, R_K32_N64, R_K32_N32, R_UNSUPPORTED };

static ConvRoute conv_route(const fx_conv_desc* d, const ConvArgs& a) {
  // Layers with a fragment-ordered weight copy: 3x3 / stride 1 -> halo kernel (pixels fetched once for all nine taps);
  // 1x1 with C, N multiples of 256 -> the same machinery with 256-channel LDS rows (conv3x3_flat.hip)
  // (the two size thresholds are re-read per call - a getenv each, nothing under graph replay - so that kernel tests can route
  // small shapes here while the end-to-end tests keep the production routing)
  static const int c3_on = fx_tune("FX_CONV3_FLAT", 1), pw_on = fx_tune("FX_PW_FLAT", 1);
  // 5 000 pixels (round 3; was 20 000): with two concurrent half-batch parts a layer's own latency matters less than the CU time it
  // occupies - the 20x20-level layers (M = 6 400 per part) fill only 50-100 workgroups of the flat kernels, but those run at 2-3x the
  // per-CU rate of the implicit-GEMM tiles and the other part's launches take the idle CUs: RT-DETR 3883 -> 3990 img/s
  // (profiles/r03_threshold_sweep.txt); one part alone (FX_STREAMS=1) 3334 -> 3420
  const int c3_min_m = fx_tune("FX_CONV3_MIN_M", 5000), pw_min_m = fx_tune("FX_PW_MIN_M", 5000);
  if (!d->mask && d->w_frag && ((uintptr_t)d->w_frag % 16) == 0 && d->stride == 1 && !d->pool2 && !d->out_f32) {
    const int mode = fx_c3_epilogue_mode(d->act, d->residual != nullptr, d->residual_after_act);
    static const int c32_on = fx_tune("FX_C3_C32", 1);
    if (c3_on && d->KH == 3 && d->KW == 3 && d->pad == 1 && !d->y_batch_stride && ((mode >= 0 && mode <= 3) || mode == 5) && a.M >= c3_min_m &&
        fx_conv3x3_flat_supported(d->C, d->N, d->W) && (d->C != 32 || (c32_on && fx_conv3x3_c32_supported(d->C, d->N, d->W, mode))))
      return R_C3_FLAT;
    // pointwise (FX_PW_SMALL_TILES = t > 0: below 20 000 pixels only layers with >= t (pixel tile x 256-channel tile) pairs - the narrow
    // 20x20-level layers are faster on the 64x64-tile kernels in isolation (serial kernel sum 10.34 -> 10.25 ms at t = 200), but the
    // two-part step is not: 4190 vs 4170 img/s, profiles/r03_threshold_sweep.txt - so the rule is off)
    if (pw_on && d->KH == 1 && d->KW == 1 && d->pad == 0 && d->C % 256 == 0 && d->N % 256 == 0 && a.M >= pw_min_m &&
        (a.M >= 20000 || (int64_t)((a.M + 127) / 128) * (d->N / 256) >= fx_tune("FX_PW_SMALL_TILES", 0)) && (mode == 0 || mode == 1 || (mode >= 3 && mode <= 6)))
      return R_PW_FLAT;
  }
  // 3x3 / stride 2 (branch2b of the first block of res3 / res4 / res5): k-plane kernel over the parity planes of the input
  // (conv3x3s2_kplane.hip; FX_C3S2_KPLANE=0: the implicit-GEMM tiles of rounds 1-3)
  static const int s2_on = fx_tune("FX_C3S2_KPLANE", 1);
  if (s2_on && !d->mask && d->w_frag && ((uintptr_t)d->w_frag % 16) == 0 && d->stride == 2 && d->KH == 3 && d->KW == 3 && d->pad == 1 && !d->pool2 &&
      !d->out_f32 && !d->residual && d->H == 2 * d->Ho && d->W == 2 * d->Wo && a.M >= c3_min_m) {
    const int mode = fx_c3_epilogue_mode(d->act, false, 0);
    if ((mode == 0 || mode == 1 || mode == 3) && fx_conv3x3s2_kplane_supported(d->C, d->N, d->Wo, a.M)) return R_C3_S2;
  }
  // BK=64 (2 workgroups/CU, 64 KiB LDS) for deep-K compute-bound layers; BK=32 (4 workgroups/CU, 34 KiB LDS: more
  // tiles and bytes in flight per CU) for the short-K layers, which are HBM/latency-bound.
  static const int k64_min = fx_tune("FX_K64_MIN_KTOT", FX_K64_MIN_KTOT);
  static const int small_m = fx_tune("FX_SMALL_M", 16384);
  // Small-M GEMMs (decoder / 20x20 level: a few hundred tiles, latency-bound): 64x64 tiles with a 256-deep K slab per
  // step - for K = 256 the whole reduction is ONE load phase (all 16 loads per lane in flight at once), no K loop.
  // (Ktot <= 2048 since round 4: MaskFormer-L's FFN linear2 - 800 rows, K = 2048 - took 45 us on 14 workgroups of the 128 x 128 tiles)
  static const int small_k = fx_tune("FX_SMALL_M_KTOT", 2048);
  if (!d->pool2 && a.M <= small_m && d->C % 256 == 0 && a.Ktot <= small_k) return R_SMALL_M;
  const bool k64 = (d->C % 64 == 0) && (a.Ktot >= k64_min);
  if (!d->pool2 && !d->mask && fx_conv_dma_eligible(a)) return R_DMA;
  if (d->pool2) return d->C % 64 != 0 ? R_UNSUPPORTED : R_POOL;
  if (k64) return d->N > 64 ? R_K64_N128 : (d->N > 32 ? R_K64_N64 : R_K64_N32);
  return d->N > 64 ? R_K32_N128 : (d->N > 32 ? R_K32_N64 : R_K32_N32);
}

extern "C" int fx_conv2d_nhwc_bf16(const fx_conv_desc* d, fx_stream_t stream_) {
  ConvArgs a;
  const int rc = conv_prepare(d, a);
  if (rc != FX_OK) return rc;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  switch (conv_route(d, a)) {
    case R_C3_FLAT: return fx_launch_conv3x3_flat(a, reinterpret_cast<const bf16_t*>(d->w_frag), stream);
    case R_C3_S2: return fx_launch_conv3x3s2_kplane(a, reinterpret_cast<const bf16_t*>(d->w_frag), stream);
    case R_PW_FLAT: return fx_launch_pw_flat(a, reinterpret_cast<const bf16_t*>(d->w_frag), stream);
    case R_SMALL_M: return launch_conv<64, 64, 256, 2, 2, false, 1>(a, stream);
    case R_DMA: return fx_launch_conv_dma(a, stream);
    case R_POOL: return launch_conv<128, 128, 64, 2, 2, true>(a, stream);
    case R_K64_N128: return launch_conv<128, 128, 64, 2, 2, false>(a, stream);
    case R_K64_N64: return launch_conv<128, 64, 64, 2, 2, false>(a, stream);
    case R_K64_N32: return launch_conv<128, 32, 64, 4, 1, false>(a, stream);
    case R_K32_N128: return launch_conv<128, 128, 32, 2, 2, false>(a, stream);
    case R_K32_N64: return launch_conv<128, 64, 32, 2, 2, false>(a, stream);
    case R_K32_N32: return launch_conv<128, 32, 32, 4, 1, false>(a, stream);
    default: return FX_ERR_UNSUPPORTED;
  }
}

// The label of the kernel fx_conv2d_nhwc_bf16 would run for this descriptor (pointers only checked for presence / alignment): what
// bench.py groups its per-kernel roofline by.  Writes a NUL-terminated string, e.g. "conv3x3_flat<256>", "conv_igemm<128,128,64>".
extern "C" int fx_conv2d_variant(const fx_conv_desc* d, char* out, int cap) {
  FX_CHECK_ARG(out && cap >= 48);
  ConvArgs a;
  const int rc = conv_prepare(d, a);
  if (rc != FX_OK) return rc;
  switch (conv_route(d, a)) {
    case R_C3_FLAT: {   // the same decision fx_launch_conv3x3_flat takes
      static const int kplane_on = fx_tune("FX_C3_KPLANE", 1);
      const bool c64 = fx_conv3x3_c64_supported(d->C, d->N, fx_c3_epilogue_mode(d->act, d->residual != nullptr, d->residual_after_act)) && !d->y_batch_stride;
      const char* fmt = d->C == 32 ? "conv3x3_c32<%d>" : (c64 ? "conv3x3_c64<%d>" : ((kplane_on && fx_conv3x3_kplane_supported(d->C, d->N, d->W)) ? "conv3x3_kplane<%d>" : "conv3x3_flat<%d>"));
      snprintf(out, cap, fmt, d->N);
      break;
    }
    case R_PW_FLAT: {
      static const int pwk_on = fx_tune("FX_PW_KPLANE", 1);
      const int mode = fx_c3_epilogue_mode(d->act, d->residual != nullptr, d->residual_after_act);
      const bool res_tile = d->residual != nullptr;
      const bool kp = pwk_on && fx_pw_kplane_supported(d->C, d->N, mode) && d->N * 4 + 128 * d->C * 2 <= 160 * 1024;
      snprintf(out, cap, kp ? "pw_kplane<K%d>" : "pw_flat<K%d>", d->C);
      break;
    }
    case R_C3_S2: snprintf(out, cap, "conv3x3s2_kplane<%d>", d->N); break;
    case R_SMALL_M: snprintf(out, cap, "conv_igemm<64,64,256,1stage>"); break;
    case R_DMA: snprintf(out, cap, "conv_igemm_dma<256,%d>", d->N % 256 == 0 ? 256 : 128); break;
    case R_POOL: snprintf(out, cap, "conv_igemm<128,128,64,pool>"); break;
    case R_K64_N128: snprintf(out, cap, "conv_igemm<128,128,64>"); break;
    case R_K64_N64: snprintf(out, cap, "conv_igemm<128,64,64>"); break;
    case R_K64_N32: snprintf(out, cap, "conv_igemm<128,32,64>"); break;
    case R_K32_N128: snprintf(out, cap, "conv_igemm<128,128,32>"); break;
    case R_K32_N64: snprintf(out, cap, "conv_igemm<128,64,32>"); break;
    case R_K32_N32: snprintf(out, cap, "conv_igemm<128,32,32>"); break;
    default: return FX_ERR_UNSUPPORTED;
  }
  return FX_OK;
}
