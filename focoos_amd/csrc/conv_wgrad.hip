// Weight gradient of the NHWC bf16 convolution (training path) on the gfx950 matrix cores:
//   dW[n][kh][kw][c] += sum over output pixels m of dZ[m][n] * X[pixel(m, kh, kw)][c]
// i.e. the GEMM  D[n][kc] = sum_m  dZ^T[n][m] * Xcol[m][kc]  whose REDUCTION index is the pixel.  Both operands live in
// memory pixel-major (NHWC), the opposite of what an MFMA fragment wants (8 consecutive reduction indices per lane), so:
//   * tiles are staged in LDS exactly as they sit in memory ([64 pixels][128 channels], rows padded to 288 B), filled
//     with coalesced 16-byte buffer loads (im2col addressing per row; out-of-image taps read zeros);
//   * fragments are fetched with the gfx950 transpose read ds_read_b64_tr_b16: 16 lanes hand in the addresses of a
//     [4 pixels][16 channels] block (4 contiguous bf16 each) and get it back column-major, so two reads give a lane the 8
//     consecutive pixels of its channel.  The 32-byte row padding spreads the 4 rows of a block over distinct banks.
//   * the pixel range is split across workgroups (grid.y) - the output tile count alone (N/128 x Ktot/128) cannot fill
//     256 CUs - and partial sums are added into the fp32 gradient with hardware float atomics.
#include <stdlib.h>

#include "pw_common.h"

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

struct WgradArgs {
  const bf16_t* x;
  const bf16_t* dz;
  float* dw;
  float* dbias;  // optional: dbias[n] += sum_m dz[m][n] (the bias gradient), taken from the dZ tiles already staged here
  int B, H, W, C, ldx;
  int Ho, Wo, N, lddz;
  int KH, KW, stride, pad;
  int M, Ktot, nKt, mchunk, tiles;
  long long split_stride;  // 0: fp32 atomics into one dW; else pixel range `by` stores its partial tile sums at dw + by * split_stride
  int n_store, k_store, ld_dw;   // rows / columns of dW actually stored and its row stride (N, Ktot, Ktot unless the operands are zero-padded views of a narrower layer)
  unsigned x_bytes, dz_bytes;
};

#define WG_BP 64          // pixels per K-step (32 measured slower: half the MFMA work between barriers)
#define WG_LD (WG_BP / 16) // staging loads per thread and tile
#define WG_ROW 288        // LDS row stride in bytes: 128 channels * 2 B + 32 B padding
#define WG_TILE (WG_BP * WG_ROW)

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* base) {
  // two transpose reads: pixels +0..3 and +4..7 of the lane's channel
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + 4 * WG_ROW));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// PW: 1x1 / stride 1 / pad 0 - the pixel IS the input row: no im2col arithmetic at all (round 3: the generic form spent ~25 VALU instructions
// per staged row on two divisions and the tap bounds - more issue time than the step's MFMAs; SQ counters: MFMA busy 12 %).
template <bool PW>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * WG_TILE];  // [stage][dZ tile | X tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware order: workgroup ids are dealt round-robin to the 8 XCDs (each with its own L2); remapped so that one XCD runs
  // ALL output tiles of a pixel range back to back - they share the same X / dZ rows, which then come from that XCD's L2 once
  // instead of from every XCD's (measured ceiling before: ~320 TFLOP/s = the tile's 128 flop/byte x ~2.5 TB/s of L2 misses)
  const int bid = fx_xcd_remap(blockIdx.x, gridDim.x);
  const int bx = bid % p.tiles, by = bid / p.tiles;
  const int nt = bx / p.nKt, kt = bx % p.nKt;
  const int n0 = nt * 128, kc0 = kt * 128;
  const int m_lo = by * p.mchunk;
  const int m_hi = min(p.M, m_lo + p.mchunk);
  if (m_lo >= m_hi) return;

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc((void*)p.dz, 0, p.dz_bytes, 0x00020000);

  // staging: thread -> (row r0 + 16*i, 16-byte chunk cc) of both tiles, i = 0..WG_LD-1
  const int cc = tid & 15, r0 = tid >> 4;
  const int HoWo = p.Ho * p.Wo;
  // this thread's X columns: kc = kc0 + 8*cc .. +7 -> one filter tap and channel offset (C % 8 == 0, a chunk never straddles taps)
  const int kc = kc0 + cc * 8;
  const bool kc_ok = kc < p.Ktot;
  const int tap = kc_ok ? kc / p.C : 0;
  const int cch = kc - tap * p.C;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  const int nch = n0 + cc * 8;
  const bool n_ok = nch < p.N;  // N % 8 == 0

  // Pipeline: two register stages + two LDS stages.  The reduction index (pixel) advances every step, so every step needs fresh
  // global loads; with a single register stage each step waited a full memory round trip (the kernel sat at a ~45 us floor for
  // layers whose MFMA time is 5 us).  Now the loads of tile st+2 are issued before the MFMAs of tile st.
  uint4 rzA[WG_LD], rxA[WG_LD], rzB[WG_LD], rxB[WG_LD];
  float bsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const bool want_bias = p.dbias != nullptr && kt == 0;
  const float inv_howo = 1.0f / (float)HoWo, inv_wo = 1.0f / (float)p.Wo;
  auto fdiv = [](int m, int d, float inv) {   // m / d for 0 <= m < 2^24-ish: float estimate + exact correction (no integer divide)
    int q = (int)((float)m * inv);
    q -= (q * d > m);
    q += ((q + 1) * d <= m);
    return q;
  };
  // Generic form: (image, output row, output column) of this thread's WG_LD staged rows, carried from step to step (every row advances
  // by WG_BP pixels per step: add the quotient / remainder of WG_BP by Wo, one carry into the row, one into the image) instead of two
  // divisions per row and step.
  int pb[WG_LD], pho[WG_LD], pwo[WG_LD];
  const int q64 = WG_BP / p.Wo, r64 = WG_BP - q64 * p.Wo;
  const bool carried = WG_BP < HoWo;   // images smaller than a step (tiny test shapes): the divisions stay
  if constexpr (!PW) {
#pragma unroll
    for (int i = 0; i < WG_LD; ++i) {
      const int m = m_lo + r0 + 16 * i;
      pb[i] = fdiv(m, HoWo, inv_howo);
      const int rem = m - pb[i] * HoWo;
      pho[i] = fdiv(rem, p.Wo, inv_wo);
      pwo[i] = rem - pho[i] * p.Wo;
    }
  }
  auto load = [&](uint4* rz, uint4* rx, int mbase) {   // called with mbase = m_lo, m_lo + WG_BP, ... in order: the carried state is that of mbase
#pragma unroll
    for (int i = 0; i < WG_LD; ++i) {
      const int m = mbase + r0 + 16 * i;
      const bool ok = m < m_hi;
      rz[i] = buf_load16(zr, (ok && n_ok) ? ((unsigned)m * (unsigned)p.lddz + nch) * 2u : FX_OOB);
      if constexpr (PW) {
        rx[i] = buf_load16(xr, (ok && kc_ok) ? ((unsigned)m * (unsigned)p.ldx + kc) * 2u : FX_OOB);
      } else {
        if (!carried) {
          const int mm = ok ? m : 0;
          pb[i] = fdiv(mm, HoWo, inv_howo);
          const int rem = mm - pb[i] * HoWo;
          pho[i] = fdiv(rem, p.Wo, inv_wo);
          pwo[i] = rem - pho[i] * p.Wo;
        }
        const int hi = pho[i] * p.stride - p.pad + kh, wi = pwo[i] * p.stride - p.pad + kw;
        const bool in = ok && kc_ok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
        rx[i] = buf_load16(xr, in ? ((unsigned)((pb[i] * p.H + hi) * p.W + wi) * (unsigned)p.ldx + cch) * 2u : FX_OOB);
        // advance to the same row of the next step
        int wo = pwo[i] + r64, ho = pho[i] + q64;
        const bool cw = wo >= p.Wo;
        wo -= cw ? p.Wo : 0;
        ho += cw ? 1 : 0;
        const bool ch = ho >= p.Ho;      // carried: WG_BP < Ho * Wo, at most one carry into the image index
        ho -= ch ? p.Ho : 0;
        pb[i] += ch ? 1 : 0;
        pwo[i] = wo;
        pho[i] = ho;
      }
    }
  };
  auto store = [&](const uint4* rz, const uint4* rx, int s) {
    unsigned char* Z = smem + s * 2 * WG_TILE;
    unsigned char* X = Z + WG_TILE;
    if (want_bias) {
#pragma unroll
      for (int i = 0; i < WG_LD; ++i) {
        float f[8];
        unpack_bf16x8(rz[i], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum[j] += f[j];
      }
    }
#pragma unroll
    for (int i = 0; i < WG_LD; ++i) {
      *reinterpret_cast<uint4*>(Z + (r0 + 16 * i) * WG_ROW + cc * 16) = rz[i];
      *reinterpret_cast<uint4*>(X + (r0 + 16 * i) * WG_ROW + cc * 16) = rx[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  // fragment addressing (see header): lane l -> group g = l>>4 (channels 16*(g&1).., pixel half g>>1), t = l&15 supplies
  // row (t>>2), 8-byte piece (t&3) of the [4][16] block
  const int g = lane >> 4, t = lane & 15;
  const int frag_off = ((g >> 1) * 8 + (t >> 2)) * WG_ROW + ((g & 1) * 16 + (t & 3) * 4) * 2;
  auto compute = [&](int cur) {
    const unsigned char* Z = smem + cur * 2 * WG_TILE;
    const unsigned char* X = Z + WG_TILE;
#pragma unroll
    for (int ks = 0; ks < WG_BP / 16; ++ks) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) af[a] = tr_frag(Z + ks * 16 * WG_ROW + frag_off + (wm * 64 + a * 32) * 2);
#pragma unroll
      for (int b = 0; b < 2; ++b) bfr[b] = tr_frag(X + ks * 16 * WG_ROW + frag_off + (wn * 64 + b * 32) * 2);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = FX_MFMA_32x32x16(af[a], bfr[b], acc[a][b]);
    }
  };

  const int nsteps = (m_hi - m_lo + WG_BP - 1) / WG_BP;
  load(rzA, rxA, m_lo);
  if (nsteps > 1) load(rzB, rxB, m_lo + WG_BP);
  store(rzA, rxA, 0);
  __syncthreads();
  for (int st = 0; st < nsteps; st += 2) {
    // even step: LDS[0] = tile st, registers B = tile st+1, registers A free
    if (st + 2 < nsteps) load(rzA, rxA, m_lo + (st + 2) * WG_BP);
    compute(0);
    if (st + 1 < nsteps) store(rzB, rxB, 1);
    __syncthreads();
    if (st + 1 >= nsteps) break;
    // odd step: LDS[1] = tile st+1, registers A = tile st+2, registers B free
    if (st + 3 < nsteps) load(rzB, rxB, m_lo + (st + 3) * WG_BP);
    compute(1);
    if (st + 2 < nsteps) store(rzA, rxA, 0);
    __syncthreads();
  }
  if (want_bias) {  // reduce the 16 row-threads of each channel chunk through LDS (the tiles are no longer needed)
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < 8; ++j) red[r0 * 128 + cc * 8 + j] = bsum[j];
    __syncthreads();
    if (tid < 128 && n0 + tid < p.n_store) {
      float t = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += red[r * 128 + tid];
      unsafeAtomicAdd(p.dbias + n0 + tid, t);
    }
  }
  // epilogue: D rows = out channel n, cols = kc; a register index r is one n for 32 consecutive kc -> 128-byte atomics
  const int l32 = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int kcol = kc0 + wn * 64 + b * 32 + l32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (n < p.n_store && kcol < p.k_store) {
          if (p.split_stride) p.dw[(int64_t)by * p.split_stride + (int64_t)n * p.ld_dw + kcol] = acc[a][b][r];
          else unsafeAtomicAdd(p.dw + (int64_t)n * p.ld_dw + kcol, acc[a][b][r]);
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Round 4: the wide-layer form (N % 256 == 0, C % 256 == 0; 1x1, or 3x3 / stride 1 / pad 1) - 256 x 256 output tiles.
//
// Why the kernel above stays near 0.13 of the MFMA peak on these layers (profiles/r04_wgrad_table_before.txt / _after.txt: 320-400 TFLOP/s): (i) a
// 64 x 64 wave tile reads 2 + 2 fragments for 4 MFMAs, and a transposed fragment is TWO ds_read_b64_tr_b16 - 2 LDS instructions
// and 1 KiB of LDS traffic per MFMA, exactly the LDS's 128 B/clk at full MFMA rate, so nothing overlaps; (ii) tiles go global ->
// registers -> LDS one K-step ahead: ~1 us of prefetch distance against a loaded-memory latency of 2-4 us; (iii) the staging
// stores and the im2col arithmetic of all four waves compete with the MFMAs for issue slots.  Here:
//   * four waves, each a 128 x 128 block of the tile (4 x 4 accumulators = 256 AGPRs): 4 + 4 fragments per 16 MFMAs - one LDS
//     instruction and 512 B per MFMA slot;
//   * operands go HBM/L2 -> LDS by LDS-DMA (no registers, no staging stores) into a FOUR-stage ring of 32-pixel steps; the DMA
//     of step s+4 is issued in the middle of step s, waits are counted (s_waitcnt vmcnt(16): the 8 + 8 DMAs of steps s+2,
//     s+3 may still be in flight), one raw s_barrier per step (no fence: the only LDS writers are the DMAs);
//   * LDS layout per step: four half-tiles [32 pixels][128 channels] of 256-byte rows (two of dZ, two of X).  A DMA
//     instruction fills 4 whole rows; rows cannot be padded (the 64 lanes' 16-byte pieces land back to back), so the bank
//     spread of the transpose reads ([4 rows][32 bytes] per 16 lanes) comes from a SOURCE-side swizzle: the 32-byte run at
//     position q of row r holds channels 16 * (q ^ (r & 7)) - the lane that fills position q simply fetches that run;
//   * 3x3: a K tile is one filter tap (C % 256 == 0) - the input pixel of output pixel m is m + (kh-1) W + (kw-1), validity from
//     the (row, column) of m, carried from step to step for the two rows a lane fills (no division in the loop).
// Steps past the end of the pixel range are issued like the others with out-of-range offsets (zeros land, nothing is added) so that
// the counted waits stay uniform.
#define WD_BP 32
#define WD_HT (WD_BP * 256)       // bytes of a half-tile
#define WD_STAGE (4 * WD_HT)
#define WD_STAGES 4

typedef __attribute__((address_space(3))) unsigned char wd_lds_u8;

// LDS-DMA as an asm statement: hipcc tracks the builtin form and puts s_waitcnt vmcnt(0) in front of the next LDS read - here that
// would be the fragments of the CURRENT step, i.e. a full memory round trip per step; the waits are counted by hand instead
// (s_waitcnt vmcnt(16) in the step body).  M0 = LDS byte address of lane 0's piece; one wait state between the M0 write and its use.
typedef __attribute__((ext_vector_type(4))) int wd_v4i;
__device__ __forceinline__ void wd_dma16(wd_v4i rsrc, unsigned lds_addr, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc) : "memory");   // m0 is a RESERVED register for hipcc (listing it as a clobber is rejected with "clobber list contains reserved registers: m0"): the compiler never keeps a value of its own in it across statements - it writes m0 immediately in front of each of its own uses
}

template <int IMM>
__device__ __forceinline__ bf16x8 wd_frag(int lo, int hi) {   // lo / hi: LDS byte addresses of the lane's two transpose reads (+ IMM)
  s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(size_t)(unsigned)(lo + IMM));
  s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(size_t)(unsigned)(hi + IMM));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <bool PW>
__global__ __launch_bounds__(256, 1) void conv_wgrad_dma_kernel(const WgradArgs p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wk = wave & 1;
  const int bid = fx_xcd_remap(blockIdx.x, gridDim.x);
  const int bx = bid % p.tiles, by = bid / p.tiles;
  const int nt = bx / p.nKt, kt = bx % p.nKt;
  const int n0 = nt * 256, kc0 = kt * 256;
  const int m_lo = by * p.mchunk;
  const int m_hi = min(p.M, m_lo + p.mchunk);
  if (m_lo >= m_hi) return;

  // ---- DMA duty of this lane: rows rr and rr + 16 of every step, in all four half-tiles; 16-byte piece dp of the row
  const int dq = lane >> 4, dp = lane & 15;
  const int rr = 4 * wave + dq;
  const int sp = (((dp >> 1) ^ (rr & 7)) << 1) | (dp & 1);      // source piece of LDS position dp (rr + 16 has the same rr & 7)
  // 3x3: tap of this K tile and the (image row, column) of the lane's two rows at the current step
  int tap_kh = 0, tap_kw = 0, cch = kc0, shift = 0;
  int ho[2] = {0, 0}, wo[2] = {0, 0};
  const int q32 = WD_BP / p.W, r32 = WD_BP - q32 * p.W;
  if constexpr (!PW) {
    const int tap = kc0 / p.C;
    cch = kc0 - tap * p.C;
    tap_kh = tap / 3;
    tap_kw = tap - tap_kh * 3;
    shift = (tap_kh - 1) * p.W + (tap_kw - 1);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = m_lo + rr + 16 * j;
      const int rem = m % (p.H * p.W);
      ho[j] = rem / p.W;
      wo[j] = rem - ho[j] * p.W;
    }
  }
  const unsigned zcol = (unsigned)(n0 + sp * 8), xcol = (unsigned)(cch + sp * 8);
  // the same descriptors as plain words for the asm statements: base (48 bits, stride 0), byte count, raw-buffer flags
  auto rsrc_words = [](const void* ptr, unsigned bytes) {
    const unsigned long long a = (unsigned long long)ptr;
    wd_v4i r = {(int)(unsigned)a, (int)(unsigned)((a >> 32) & 0xffffu), (int)bytes, 0x00020000};
    return r;
  };
  const wd_v4i xr4 = rsrc_words(p.x, p.x_bytes), zr4 = rsrc_words(p.dz, p.dz_bytes);
  const unsigned lds_base = (unsigned)(size_t)(wd_lds_u8*)smem;
  auto issue = [&](int step, int stage_off) {   // the 8 DMA instructions of this wave for pixel step `step`
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = m_lo + step * WD_BP + rr + 16 * j;
      const bool ok = m < m_hi;
      bool okx = ok;
      if constexpr (!PW) {
        const int hi_ = ho[j] + tap_kh - 1, wi_ = wo[j] + tap_kw - 1;
        okx = ok && (unsigned)hi_ < (unsigned)p.H && (unsigned)wi_ < (unsigned)p.W;
        // advance to the next step
        int w2 = wo[j] + r32, h2 = ho[j] + q32;
        const bool cw = w2 >= p.W;
        w2 -= cw ? p.W : 0;
        h2 += cw ? 1 : 0;
        h2 -= h2 >= p.H ? p.H : 0;
        wo[j] = w2;
        ho[j] = h2;
      }
      const unsigned zo = ok ? ((unsigned)m * (unsigned)p.lddz + zcol) * 2u : FX_OOB;
      const unsigned xo = okx ? ((unsigned)(m + shift) * (unsigned)p.ldx + xcol) * 2u : FX_OOB;
      const unsigned dst = lds_base + stage_off + (wave + 4 * j) * 1024;
      wd_dma16(zr4, dst, zo);
      wd_dma16(zr4, dst + WD_HT, zo == FX_OOB ? FX_OOB : zo + 256u);
      wd_dma16(xr4, dst + 2 * WD_HT, xo);
      wd_dma16(xr4, dst + 3 * WD_HT, xo == FX_OOB ? FX_OOB : xo + 256u);
    }
  };

  // ---- fragment addressing (header of this file): group g = lane >> 4 -> channels 16 (g & 1).., pixel half g >> 1; t = lane & 15 supplies
  // row t >> 2, 8-byte piece t & 3 of a [4 pixels][16 channels] block.  Block b of a half-tile: run 2b + (g & 1), swizzled by the row.
  const int g = lane >> 4, t = lane & 15;
  const int lds0 = (int)lds_base;
  int alo[4], ahi[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int run = 2 * b + (g & 1), row = (g >> 1) * 8 + (t >> 2);
    alo[b] = lds0 + row * 256 + ((run ^ (t >> 2)) * 32) + (t & 3) * 8;
    ahi[b] = lds0 + (row + 4) * 256 + ((run ^ (t >> 2) ^ 4) * 32) + (t & 3) * 8;
  }
  const int a_base = wn * WD_HT, b_base = (2 + wk) * WD_HT;

  f32x16 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  const int nsteps = (m_hi - m_lo + WD_BP - 1) / WD_BP;
  // Software pipeline over half-steps (16 pixels = one MFMA K-step): the 16 transpose reads of the NEXT half-step are issued between
  // the 16 MFMAs of the current one (sched_group_barrier: one LDS read per MFMA slot), so only the very first reads are exposed.
  // The step's barrier sits between its two halves: by then every wave has read both halves of step s (lgkmcnt(0) in front of
  // it), so buffer s % 4 is refilled with step s+4 right after it, and step s+1 - whose first half is read next - has landed.
  bf16x8 fa[2][4], fb[2][4];
  auto read_frags = [&](bf16x8* a, bf16x8* b, int off) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a[i] = wd_frag<0>(alo[i] + off + a_base, ahi[i] + off + a_base);
      b[i] = wd_frag<0>(alo[i] + off + b_base, ahi[i] + off + b_base);
    }
  };
  auto mfma16 = [&](const bf16x8* a_, const bf16x8* b_) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = FX_MFMA_32x32x16(a_[a], b_[b], acc[a][b]);
  };
  auto interleave = [&]() {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one LDS read
    }
  };
  issue(0, 0);
  issue(1, WD_STAGE);
  issue(2, 2 * WD_STAGE);
  issue(3, 3 * WD_STAGE);
  asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  read_frags(fa[0], fb[0], 0);
  auto step_body = [&](auto ST, int s) {
    constexpr int st = decltype(ST)::value;
    read_frags(fa[1], fb[1], st * WD_STAGE + 16 * 256);            // second half of step s
    mfma16(fa[0], fb[0]);
    interleave();
    asm volatile("s_waitcnt vmcnt(16)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");   // step s+1 has landed (s+2, s+3 may be in flight); my reads of step s are done
    __builtin_amdgcn_s_barrier();
    issue(s + 4, st * WD_STAGE);
    read_frags(fa[0], fb[0], ((st + 1) & 3) * WD_STAGE);          // first half of step s+1
    mfma16(fa[1], fb[1]);
    interleave();
  };
  for (int s0 = 0; s0 < nsteps; s0 += 4) {
    step_body(std::integral_constant<int, 0>{}, s0);
    if (s0 + 1 < nsteps) step_body(std::integral_constant<int, 1>{}, s0 + 1);
    if (s0 + 2 < nsteps) step_body(std::integral_constant<int, 2>{}, s0 + 2);
    if (s0 + 3 < nsteps) step_body(std::integral_constant<int, 3>{}, s0 + 3);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the (zero-filling) DMAs of the steps issued past the end

  // ---- epilogue: this pixel range's partial tile, plain stores: register r of block (a, b) is output channel n for 32 consecutive kc
  const int l32 = lane & 31, lh = lane >> 5;
  float* slab = p.dw + (int64_t)by * p.split_stride;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int kcol = kc0 + wk * 128 + b * 32 + l32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 128 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        slab[(int64_t)n * p.ld_dw + kcol] = acc[a][b][r];
      }
    }
}

// Shape class of the wide-layer form.  Layers below FX_WGRAD_DMA_MIN_GFLOP (default 8) stay on the 128 x 128 kernel: with one or two
// 256 x 256 tiles they are a handful of launch-latency-bound workgroups either way, and the smaller tiles split them finer
// (25600 x 256 x 256 pointwise, 3.4 GFLOP: 26 us there, 31 us here; profiles/r04_wgrad_table_before.txt / _after.txt).
static bool wgrad_dma_shape(int M, int N, int C, int KH, int KW) {
  static const int on = fx_tune("FX_WGRAD_DMA", 1), min_gflop = fx_tune("FX_WGRAD_DMA_MIN_GFLOP", 8);
  return on && N % 256 == 0 && C % 256 == 0 && ((KH == 1 && KW == 1) || (KH == 3 && KW == 3)) &&
         2.0 * M * N * C * KH * KW >= 1e9 * min_gflop;
}

extern "C" int fx_conv2d_wgrad_bias_nhwc_bf16(const void* x, int ldx, const void* dz, int lddz, float* dw, float* dbias, int B, int H, int W,
                                              int C, int Ho, int Wo, int N, int KH, int KW, int stride, int pad, fx_stream_t stream_);

extern "C" int fx_conv2d_wgrad_nhwc_bf16(const void* x, int ldx, const void* dz, int lddz, float* dw, int B, int H, int W, int C, int Ho,
                                         int Wo, int N, int KH, int KW, int stride, int pad, fx_stream_t stream_) {
  return fx_conv2d_wgrad_bias_nhwc_bf16(x, ldx, dz, lddz, dw, nullptr, B, H, W, C, Ho, Wo, N, KH, KW, stride, pad, stream_);
}

// pixel split of one launch: enough workgroups to fill the chip `occ` times over, chunks a multiple of the K-step
static void wgrad_split(int M, int tiles, int wgs, int* mchunk_out, int* splits_out) {
  int want = (wgs + tiles - 1) / tiles;
  int mchunk = (M + want - 1) / want;
  if (mchunk < 512) mchunk = 512;
  mchunk = (mchunk + WG_BP - 1) / WG_BP * WG_BP;
  *mchunk_out = mchunk;
  *splits_out = (M + mchunk - 1) / mchunk;
}

// workgroups the partial-slab form aims for.  Large filters (N x K >= 128 Ki fp32 per slab): one workgroup per CU - fewer pixel splits
// = fewer slabs to write and sum, and the weight gradients run beside the input-gradient chain, which fills the rest of the chip
// (RT-DETR step 588 -> 600 img/s).  Small filters (the narrow layers of STDC / the stem): the slabs are cheap and the pixel range per
// workgroup is what matters - four workgroups per CU as before.
static int wgrad_target_wgs(int N, int Ktot) {
  static const int big = fx_tune("FX_WGRAD_WGS", 256), small = fx_tune("FX_WGRAD_WGS_SMALL", 1024);
  return (int64_t)N * Ktot >= 128 * 1024 ? big : small;
}

// Pixel split of the partial-slab form for one layer shape - ONE rule for both kernels, because the caller sizes the slab workspace from
// fx_conv2d_wgrad_splits(), which does not know the stride: wide layers (wgrad_dma_shape) count 256 x 256 tiles and aim at
// FX_WGRAD_DMA_WGS workgroups (one per CU; a tile's partial is 256 KiB of fp32, so the slab traffic is workgroups x 256 KiB to write and
// again to sum - fewer, longer pixel ranges than CUs are the better trade beside the input-gradient chain that shares the chip).
static void wgrad_partial_plan(int M, int N, int C, int KH, int KW, int* mchunk_out, int* splits_out) {
  const int Ktot = KH * KW * C;
  if (wgrad_dma_shape(M, N, C, KH, KW)) {
    static const int wgs = fx_tune("FX_WGRAD_DMA_WGS", 160);
    const int tiles = (N / 256) * (Ktot / 256);
    int want = (wgs + tiles - 1) / tiles;
    int mchunk = (M + want - 1) / want;
    if (mchunk < 512) mchunk = 512;
    mchunk = (mchunk + WG_BP - 1) / WG_BP * WG_BP;
    *mchunk_out = mchunk;
    *splits_out = (M + mchunk - 1) / mchunk;
    return;
  }
  wgrad_split(M, ((N + 127) / 128) * ((Ktot + 127) / 128), wgrad_target_wgs(N, Ktot), mchunk_out, splits_out);
}

static int wgrad_launch(const void* x, int ldx, const void* dz, int lddz, float* dw, long long split_stride, int expect_splits, float* dbias, int B, int H,
                        int W, int C, int Ho, int Wo, int N, int KH, int KW, int stride, int pad, fx_stream_t stream_, int n_store = 0, int k_store = 0,
                        int ld_dw = 0) {
  FX_CHECK_ARG(x && dz && dw && B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && N > 0 && C > 0);
  FX_CHECK_ARG(C % 8 == 0 && N % 8 == 0 && ldx >= C && lddz >= N && ldx % 8 == 0 && lddz % 8 == 0);
  FX_CHECK_ARG(KH >= 1 && KW >= 1 && stride >= 1 && pad >= 0);
  FX_CHECK_ARG(Ho == (H + 2 * pad - KH) / stride + 1 && Wo == (W + 2 * pad - KW) / stride + 1);
  FX_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)dz % 16) == 0 && ((uintptr_t)dw % 4) == 0);
  const int64_t x_bytes = ((int64_t)B * H * W - 1) * ldx * 2 + (int64_t)C * 2;
  const int64_t dz_bytes = ((int64_t)B * Ho * Wo - 1) * lddz * 2 + (int64_t)N * 2;
  if (x_bytes >= 0xFFFFFFF0ll || dz_bytes >= 0xFFFFFFF0ll || (int64_t)B * Ho * Wo >= (1ll << 31)) return FX_ERR_UNSUPPORTED;
  WgradArgs a;
  a.x = reinterpret_cast<const bf16_t*>(x);
  a.dz = reinterpret_cast<const bf16_t*>(dz);
  a.dw = dw;
  a.dbias = dbias;
  a.B = B; a.H = H; a.W = W; a.C = C; a.ldx = ldx;
  a.Ho = Ho; a.Wo = Wo; a.N = N; a.lddz = lddz;
  a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad;
  a.M = B * Ho * Wo;
  a.Ktot = KH * KW * C;
  a.nKt = (a.Ktot + 127) / 128;
  const int nNt = (N + 127) / 128;
  a.x_bytes = (unsigned)x_bytes;
  a.dz_bytes = (unsigned)dz_bytes;
  const int tiles = nNt * a.nKt;
  // atomics: every split adds one full pass of fp32 atomics over dW, and the L2 atomic units sustain only ~0.6 TB/s - few splits.
  // partial stores: plain coalesced stores (summed later by fx_unpack_conv_wgrad_sum_f32) - more splits, more parallelism.
  int S;
  if (split_stride) wgrad_partial_plan(a.M, N, C, KH, KW, &a.mchunk, &S);
  else wgrad_split(a.M, tiles, 512, &a.mchunk, &S);
  FX_CHECK_ARG(!split_stride || (S == expect_splits && split_stride >= (long long)N * a.Ktot));
  a.split_stride = split_stride;
  a.tiles = tiles;
  a.n_store = n_store > 0 ? n_store : N;
  a.k_store = k_store > 0 ? k_store : a.Ktot;
  a.ld_dw = ld_dw > 0 ? ld_dw : a.Ktot;
  FX_CHECK_ARG(a.n_store <= N && a.k_store <= a.Ktot && a.ld_dw >= a.k_store);
  const bool pw = KH == 1 && KW == 1 && stride == 1 && pad == 0;
  if (split_stride && !dbias && wgrad_dma_shape(a.M, N, C, KH, KW) && (pw || (KH == 3 && stride == 1 && pad == 1)) && ldx % 8 == 0 && a.n_store == N &&
      a.k_store == a.Ktot && (int64_t)a.M + 4 * W + 8 < (1ll << 30)) {
    a.nKt = a.Ktot / 256;
    a.tiles = (N / 256) * a.nKt;
    constexpr int smem = WD_STAGES * WD_STAGE;
    static bool attr_done = false;
    if (!attr_done) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_dma_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_dma_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
        return FX_ERR_RUNTIME;
      attr_done = true;
    }
    if (pw) hipLaunchKernelGGL(conv_wgrad_dma_kernel<true>, dim3(a.tiles * S), dim3(256), smem, reinterpret_cast<hipStream_t>(stream_), a);
    else hipLaunchKernelGGL(conv_wgrad_dma_kernel<false>, dim3(a.tiles * S), dim3(256), smem, reinterpret_cast<hipStream_t>(stream_), a);
    return fx_launch_status();
  }
  if (pw) hipLaunchKernelGGL(conv_wgrad_kernel<true>, dim3(tiles * S), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), a);
  else hipLaunchKernelGGL(conv_wgrad_kernel<false>, dim3(tiles * S), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), a);
  return fx_launch_status();
}

extern "C" int fx_conv2d_wgrad_bias_nhwc_bf16(const void* x, int ldx, const void* dz, int lddz, float* dw, float* dbias, int B, int H, int W,
                                              int C, int Ho, int Wo, int N, int KH, int KW, int stride, int pad, fx_stream_t stream_) {
  return wgrad_launch(x, ldx, dz, lddz, dw, 0, 0, dbias, B, H, W, C, Ho, Wo, N, KH, KW, stride, pad, stream_);
}

// Linear layer whose operands are zero-padded views of a narrower layer (bbox heads N = 4, class heads N = 365, query-pos head K = 4: the
// kernels want K % 32 == 0 and N % 8 == 0): x [R][ldx] with Kp valid-or-zero columns, dz [R][lddz] with Np; ONLY the n_store x k_store
// corner of dW (row stride ld_dw) and the first n_store bias gradients are accumulated - straight into the master gradient, instead of a
// padded staging matrix + slice + add per layer and step.
extern "C" int fx_linear_wgrad_bias_bf16(const void* x, int ldx, const void* dz, int lddz, float* dw, int ld_dw, float* dbias, int R, int Kp, int Np,
                                         int k_store, int n_store, fx_stream_t stream_) {
  FX_CHECK_ARG(k_store > 0 && n_store > 0 && k_store <= Kp && n_store <= Np && ld_dw >= k_store);
  return wgrad_launch(x, ldx, dz, lddz, dw, 0, 0, dbias, 1, 1, R, Kp, 1, R, Np, 1, 1, 1, 0, stream_, n_store, k_store, ld_dw);
}

// Label of the kernel the partial-slab weight gradient of this layer shape runs on (the routing predicate of wgrad_launch itself, for the
// kernel census of tests / profiles): "conv_wgrad_dma<pw|3x3>" or "conv_wgrad<pw|im2col>".
extern "C" int fx_conv2d_wgrad_variant(int B, int Ho, int Wo, int C, int N, int KH, int KW, int stride, int pad, char* out, int cap) {
  FX_CHECK_ARG(out && cap >= 32 && B > 0 && Ho > 0 && Wo > 0 && C > 0 && N > 0);
  const bool pw = KH == 1 && KW == 1 && stride == 1 && pad == 0;
  const bool dma = wgrad_dma_shape(B * Ho * Wo, N, C, KH, KW) && (pw || (KH == 3 && stride == 1 && pad == 1));
  snprintf(out, cap, "%s<%s>", dma ? "conv_wgrad_dma" : "conv_wgrad", pw ? "pw" : (dma ? "3x3" : "im2col"));
  return FX_OK;
}

extern "C" int fx_conv2d_wgrad_splits(int B, int Ho, int Wo, int C, int N, int KH, int KW) {
  if (B <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || N <= 0 || KH <= 0 || KW <= 0) return 0;
  int mchunk, S;
  wgrad_partial_plan(B * Ho * Wo, N, C, KH, KW, &mchunk, &S);
  return S;
}

extern "C" int fx_conv2d_wgrad_partial_nhwc_bf16(const void* x, int ldx, const void* dz, int lddz, float* partials, int64_t split_stride, int splits,
                                                 int B, int H, int W, int C, int Ho, int Wo, int N, int KH, int KW, int stride, int pad,
                                                 fx_stream_t stream_) {
  FX_CHECK_ARG(split_stride > 0 && splits > 0);
  return wgrad_launch(x, ldx, dz, lddz, partials, split_stride, splits, nullptr, B, H, W, C, Ho, Wo, N, KH, KW, stride, pad, stream_);
}
