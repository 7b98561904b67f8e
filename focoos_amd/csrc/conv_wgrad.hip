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

#include "conv_common.h"

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
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
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
  wgrad_split(a.M, tiles, split_stride ? wgrad_target_wgs(N, a.Ktot) : 512, &a.mchunk, &S);
  FX_CHECK_ARG(!split_stride || (S == expect_splits && split_stride >= (long long)N * a.Ktot));
  a.split_stride = split_stride;
  a.tiles = tiles;
  a.n_store = n_store > 0 ? n_store : N;
  a.k_store = k_store > 0 ? k_store : a.Ktot;
  a.ld_dw = ld_dw > 0 ? ld_dw : a.Ktot;
  FX_CHECK_ARG(a.n_store <= N && a.k_store <= a.Ktot && a.ld_dw >= a.k_store);
  const bool pw = KH == 1 && KW == 1 && stride == 1 && pad == 0;
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

extern "C" int fx_conv2d_wgrad_splits(int B, int Ho, int Wo, int C, int N, int KH, int KW) {
  if (B <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || N <= 0 || KH <= 0 || KW <= 0) return 0;
  int mchunk, S;
  wgrad_split(B * Ho * Wo, ((N + 127) / 128) * ((KH * KW * C + 127) / 128), wgrad_target_wgs(N, KH * KW * C), &mchunk, &S);
  return S;
}

extern "C" int fx_conv2d_wgrad_partial_nhwc_bf16(const void* x, int ldx, const void* dz, int lddz, float* partials, int64_t split_stride, int splits,
                                                 int B, int H, int W, int C, int Ho, int Wo, int N, int KH, int KW, int stride, int pad,
                                                 fx_stream_t stream_) {
  FX_CHECK_ARG(split_stride > 0 && splits > 0);
  return wgrad_launch(x, ldx, dz, lddz, partials, split_stride, splits, nullptr, B, H, W, C, Ho, Wo, N, KH, KW, stride, pad, stream_);
}
