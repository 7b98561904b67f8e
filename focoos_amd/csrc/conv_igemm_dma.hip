// Implicit-GEMM convolution, large-tile variant for the deep-K / large-M layers (gfx950).
//
// Why a second kernel: with 128x128 tiles the operand stream needs ~64 B/clk/CU at full MFMA rate, which is
// about what L2 can deliver per CU (34.5 TB/s / 256 CUs / 2.4 GHz = 56 B/clk): the 128^2 kernel tops out near
// 30 % of the MFMA peak however well it is scheduled.  This kernel uses 256-row tiles (256x256 or 256x128 per
// workgroup of 8 waves, one workgroup per CU), which halves the bytes per flop, and moves the operands
// global -> LDS with the DMA path (buffer_load_dwordx4 ... lds): no staging VGPRs, no ds_write pass.
//   * LDS image: rows of 128 B (BK = 64 bf16), 16-byte chunk c of row r stored at chunk c ^ ((r>>1)&7)
//     (conflict-free ds_read_b128).  The DMA writes wave-base + lane*16 linearly, so the swizzle is applied to
//     the per-lane SOURCE address: lane L of a wave-instruction fills (row = 8*i + L/8, physical chunk L%8) and
//     therefore fetches logical chunk (L%8) ^ ((row>>1)&7) of that row (guide rule 21).
//   * zero padding / M tail: out-of-range buffer offsets make the DMA write zeros (probe: scripts/probes/glds_probe.hip).
//   * pipeline: 2 LDS stages; tile t+1's DMA is issued right after the barrier that publishes tile t and flies
//     under tile t's MFMAs; one s_waitcnt vmcnt(0) + barrier per K-step.
#include "conv_common.h"

typedef __attribute__((address_space(3))) void lds_void_t;

// One DMA wave-instruction: 64 lanes x 16 B from the buffer (per-lane byte offset, out-of-range -> zeros) to
// LDS at lds_base + lane*16.  Kept in a non-template __device__ helper: the host pass of hipcc silently drops a
// __global__ template whose body it cannot type-check (the address-space cast), leaving its launch stub undefined.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, unsigned char* lds_base, unsigned voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t*)lds_base, 16, voff, soff, 0, 0);
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64, 1) void conv_igemm_dma_kernel(const ConvArgs p) {
  constexpr int BK = 64, NW = WM * WN, NT = NW * 64;
  constexpr int A_PW = BM / 8 / NW, B_PW = BN / 8 / NW;  // DMA wave-instructions (8 rows = 1 KiB each) per wave per stage
  constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int EPI_LD = BN + 4;
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "DMA split");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int bid = fx_xcd_remap(blockIdx.x, gridDim.x);
  const int mt = bid / p.nNt, nt = bid % p.nNt;
  const int m0 = mt * BM, n0 = nt * BN;

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

  const int lrow = lane >> 3, pc = lane & 7;
  // Per DMA slot: byte offset of (b, hi0, wi0, swizzled chunk) and a bit mask of the filter taps that fall inside the
  // image (bit kh*KW+kw).  The K loop then needs one add + one bit test + one select per DMA instruction instead of
  // re-deriving the bounds checks (the main loop is issue-bound: every VALU op there competes with the MFMAs).
  unsigned a_off2[A_PW], a_mask[A_PW];
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int i = 0; i < A_PW; ++i) {
    const int row = (wave * A_PW + i) * 8 + lrow;
    const int m = m0 + row;
    const bool ok = m < p.M;
    const int mm = ok ? m : 0;
    const int b = mm / HoWo, rem = mm - b * HoWo;
    const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
    const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
    a_off2[i] = (unsigned)((((b * p.H + hi0) * p.W + wi0) * p.ldx + ((pc ^ ((row >> 1) & 7)) << 3)) * 2);
    unsigned mask = 0;
    for (int th = 0; th < p.KH; ++th)
      for (int tw = 0; tw < p.KW; ++tw)
        if (ok && (unsigned)(hi0 + th) < (unsigned)p.H && (unsigned)(wi0 + tw) < (unsigned)p.W) mask |= 1u << (th * p.KW + tw);
    a_mask[i] = mask;
  }
  unsigned w_off[B_PW];
#pragma unroll
  for (int i = 0; i < B_PW; ++i) {
    const int row = (wave * B_PW + i) * 8 + lrow;
    w_off[i] = (unsigned)(((n0 + row) * p.Ktot + ((pc ^ ((row >> 1) & 7)) << 3)) * 2);
  }

#define FX_DMA(S_, TAP_, DELTA2_, KBASE_)                                                                             \
  {                                                                                                                    \
    unsigned char* sa_ = smem + (S_)*STAGE + wave * (A_PW * 1024);                                                     \
    unsigned char* sb_ = smem + (S_)*STAGE + A_BYTES + wave * (B_PW * 1024);                                           \
    _Pragma("unroll") for (int i = 0; i < A_PW; ++i)                                                                   \
        dma16(xr, sa_ + i * 1024, ((a_mask[i] >> (TAP_)) & 1u) ? a_off2[i] + (unsigned)(DELTA2_) : FX_OOB, 0);          \
    _Pragma("unroll") for (int i = 0; i < B_PW; ++i) dma16(wr, sb_ + i * 1024, w_off[i], (KBASE_)*2);                  \
  }

  f32x16 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  const int T = p.KH * p.KW * (p.C / BK);
  const int l32 = lane & 31, lhalf = lane >> 5;
  // LDS fragment addresses, hoisted: row r of this lane's fragments has swizzle (r>>1)&7 = (l32>>1)&7 for every 32-row
  // sub-tile (sub-tile bases are multiples of 32 rows), so one address register per k-slice serves all sub-tiles
  // through the immediate offset of ds_read_b128.
  unsigned a_rd[BK / 16], b_rd[BK / 16];
#pragma unroll
  for (int kk = 0; kk < BK / 16; ++kk) {
    const unsigned x = (unsigned)(((kk * 2 + lhalf) ^ ((l32 >> 1) & 7)) << 4);
    a_rd[kk] = (unsigned)((wm * WTM + l32) * 128) + x;
    b_rd[kk] = (unsigned)(A_BYTES + (wn * WTN + l32) * 128) + x;
  }
  int kh = 0, kw = 0, c0 = 0, kbase = 0, tap = 0;
  FX_DMA(0, 0, 0, 0);
  for (int t = 0; t < T; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tile t has landed for every wave; everyone is done reading the other stage
    if (t + 1 < T) {
      c0 += BK;
      kbase += BK;
      if (c0 == p.C) {
        c0 = 0;
        ++tap;
        if (++kw == p.KW) {
          kw = 0;
          ++kh;
        }
      }
      FX_DMA((t + 1) & 1, tap, (((kh * p.W + kw) * p.ldx + c0) * 2), kbase);
    }
    const unsigned char* S_ = smem + (t & 1) * STAGE;
    // Fragment double-buffering: the k-slice kk+1 fragments are fetched into a second register set BEFORE the MFMAs of
    // slice kk issue, so the LDS latency hides under 8 MFMAs instead of stalling the matrix pipe once per slice.
    bf16x8 xa[2][TM], wb[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) xa[0][i] = *reinterpret_cast<const bf16x8*>(S_ + a_rd[0] + i * (32 * 128));
#pragma unroll
    for (int i = 0; i < TN; ++i) wb[0][i] = *reinterpret_cast<const bf16x8*>(S_ + b_rd[0] + i * (32 * 128));
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      if (kk + 1 < BK / 16) {
#pragma unroll
        for (int i = 0; i < TM; ++i) xa[(kk + 1) & 1][i] = *reinterpret_cast<const bf16x8*>(S_ + a_rd[kk + 1] + i * (32 * 128));
#pragma unroll
        for (int i = 0; i < TN; ++i) wb[(kk + 1) & 1][i] = *reinterpret_cast<const bf16x8*>(S_ + b_rd[kk + 1] + i * (32 * 128));
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ABOVE this slice's MFMAs (hipcc otherwise sinks it to save VGPRs)
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
          acc[a][b] = FX_MFMA_32x32x16(wb[kk & 1][a], xa[kk & 1][b], acc[a][b]);
    }
  }
  __syncthreads();
#undef FX_DMA

  // ---- epilogue: 64 rows at a time through LDS (fp32), residual of the next slab prefetched while this one is written
  constexpr int TPR = BN / 8, RPP2 = NT / TPR, HALF = 64, PER_HALF = HALF / RPP2, NHALF = BM / HALF;
  const int col8 = (tid % TPR) * 8;
  const int n = n0 + col8;
  const bool n_ok = n < p.Nstore;
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res ? p.res : p.x), 0, p.res ? p.r_bytes : 0u, 0x00020000);
  uint4 rres[2][PER_HALF];
#define FX_RES_LOAD(SET_, HH_)                                                                         \
  if (p.res) {                                                                                         \
    _Pragma("unroll") for (int j = 0; j < PER_HALF; ++j) {                                             \
      const int m_ = m0 + (HH_)*HALF + tid / TPR + j * RPP2;                                           \
      rres[SET_][j] = buf_load16(rr, (m_ < p.M && n_ok) ? (unsigned)(m_ * p.ldr + n) * 2u : FX_OOB);   \
    }                                                                                                  \
  }
  FX_RES_LOAD(0, 0);
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
  for (int hh = 0; hh < NHALF; ++hh) {
    if (hh + 1 < NHALF) {
      if ((hh & 1) == 0) { FX_RES_LOAD(1, hh + 1); } else { FX_RES_LOAD(0, hh + 1); }
    }
#pragma unroll
    for (int b = 0; b < TM; ++b) {
      if ((wm * WTM + b * 32) / HALF == hh) {
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int ml = wm * WTM + b * 32 + l32 - hh * HALF;
            const int nl = wn * WTN + a * 32 + 8 * g + 4 * lhalf;
            *reinterpret_cast<float4*>(stg + ml * EPI_LD + nl) =
                make_float4(acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]);
          }
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
          unpack_bf16x8(rres[hh & 1][j], rf);
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
        int64_t yoff;
        if (p.y_bstride) {
          const int bb = m / HoWo;
          yoff = (int64_t)bb * p.y_bstride + (int64_t)(m - bb * HoWo) * p.ldy + n;
        } else {
          yoff = (int64_t)m * p.ldy + n;
        }
        if (p.out_f32) {
          float* dst = reinterpret_cast<float*>(p.y) + yoff;
          *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
          *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.y) + yoff) = pack_bf16x8(v);
        }
      }
    }
    if (hh + 1 < NHALF) __syncthreads();
  }
#undef FX_RES_LOAD
}

template <int BM, int BN, int WM, int WN>
static int launch_dma(ConvArgs& a, hipStream_t stream) {
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int EPI = 64 * (BN + 4) * 4;
  constexpr int SMEM = (2 * STAGE > EPI) ? 2 * STAGE : EPI;
  static bool attr_set = false;
  auto kern = conv_igemm_dma_kernel<BM, BN, WM, WN>;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess)
      return FX_ERR_RUNTIME;
    attr_set = true;
  }
  a.nNt = (a.N + BN - 1) / BN;
  const int nMt = (a.M + BM - 1) / BM;
  hipLaunchKernelGGL(kern, dim3(nMt * a.nNt), dim3(WM * WN * 64), SMEM, stream, a);
  return fx_launch_status();
}

#ifndef FX_DMA_MIN_M
#define FX_DMA_MIN_M 40000
#endif
#ifndef FX_DMA_MIN_KTOT
#define FX_DMA_MIN_KTOT 1024
#endif

// Deep-K layers with enough rows to fill the chip with 256-row tiles.  Weights are padded to a multiple of 128
// rows, so BN = 256 needs N % 256 == 0 (or N <= 128 -> BN = 128).
bool fx_conv_dma_eligible(const ConvArgs& a) {
  static const int min_k = fx_tune("FX_DMA_MIN_KTOT", FX_DMA_MIN_KTOT), min_m = fx_tune("FX_DMA_MIN_M", FX_DMA_MIN_M);
  if (a.C % 64 != 0 || a.Ktot < min_k || a.M < min_m) return false;
  return (a.N % 256 == 0) || (a.N % 128 == 0);
}

int fx_launch_conv_dma(ConvArgs& a, hipStream_t stream) {
  static const int force_bn = fx_tune("FX_DMA_FORCE_BN", 0);
  if (a.N % 256 == 0 && force_bn != 128) return launch_dma<256, 256, 2, 4>(a, stream);
  return launch_dma<256, 128, 4, 2>(a, stream);
}
