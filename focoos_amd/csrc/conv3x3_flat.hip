// 3x3 / stride 1 / pad 1 convolution WITHOUT im2col re-fetching (gfx950).
//
// The implicit-GEMM kernels (conv_igemm*.hip) fetch the pixel tile once per filter tap: nine L2 -> LDS passes over the same
// pixels and a barrier every 32..64 channels.  Here the workgroup owns BM consecutive pixels of the flattened NHWC tensor
// (m = (b*H + y)*W + x); a tap (dy,dx) of pixel m is pixel m + dy*W + dx of the same flat sequence, so ONE contiguous halo
// range [m0 - W - 1, m0 + BM + W + 1) of CC channels serves all nine taps:
//   * a dedicated LOADER wave streams the halo of the next channel chunk into LDS with buffer_load...lds while the consumer
//     waves run 9 x CC/16 MFMA k-steps on the current chunk - 36 k-steps between barriers instead of 1..4; its DMA count
//     lives in its own vmcnt, so the consumers' weight loads are never serialised behind it;
//   * the pixels are the MFMA B operand, read from the halo at row p + (W+1) + dy*W + dx: consecutive lanes = consecutive
//     rows, conflict-free under the row-pair XOR swizzle for ANY row offset; image borders (and neighbouring images - the
//     halo is loaded blindly) are masked in registers with a per-pixel 9-bit tap mask;
//   * the weights are the A operand and never touch LDS: fragment-ordered ([n/32][k/16][lane][8], k = tap*C + c) 1 KiB reads
//     from L2 into a register ring;
//   * epilogue: bias (accumulator init) + activation (+ residual, either side of the activation) in the accumulator layout
//     on an LDS tile, then 16-byte coalesced stores.
#include <type_traits>

#include "pw_common.h"

struct C3Args {
  const bf16_t* x;
  const bf16_t* wp;
  const float* bias;
  const bf16_t* res;
  bf16_t* y;
  int H, W, C, N, ldx, ldy, ldr, M;
  int act, res_after;
  int HLp;  // halo rows per chunk buffer (BM + 2W + 2 rounded up to whole DMA instructions)
  int HW;            // pixels per image
  int64_t y_bstride;  // elements between images of y (0: contiguous)
  unsigned x_bytes, r_bytes;
};

// KT = 9: 3x3 / stride 1 / pad 1, LDS rows of CC = 64 channels, nine taps = nine flat row offsets.
// KT = 1: pointwise (1x1) layer on the same machinery: LDS rows of CC = 256 channels, the "taps" are the CC/64 channel
//         quarters of a row (column offsets instead of row offsets), no halo, no border masks.
// LOADER = 0: no loader wave (layers whose whole K is ONE chunk, e.g. 256-channel pointwise layers: nothing to stream behind the
//         MFMAs) - a 4-wave workgroup at <= 256 VGPRs fits twice on a CU (two waves per SIMD: one workgroup's epilogue and
//         prologue overlap the other's MFMAs), which the 5-wave form cannot.
template <int KT, int CC, int TN, int TM, int WN, int WM, int ACT, int RESMODE, int LOADER = 1>
__global__ __launch_bounds__((WN* WM + LOADER) * 64, 1) void conv3x3_flat_kernel(const C3Args p) {
  constexpr int NW = WN * WM, NT = (NW + LOADER) * 64, NDW = NW + LOADER;   // NDW: waves sharing the first-chunk / residual DMA
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int RL = CC / 8, ROWB = CC * 2;
  constexpr int KJ = 4;                              // k16 steps per tap (64 channels)
  constexpr int NTAP = KT == 9 ? 9 : CC / 64;        // ring cycles per chunk
  constexpr int PF = KJ;  // weight-fragment ring: one slot per k-step of a tap, refilled for the next tap
  constexpr int RLT = BN / 8;
  static_assert(TN <= 2 && (KT == 9 ? CC == 64 : CC % 64 == 0), "tap = 64 channels = 4 k-steps");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_loader = LOADER && wave == NW;
  const int l32 = lane & 31, half = lane >> 5;
  const int wn = wave % WN, wm = wave / WN;
  // 1-D grid, n-tile fastest, each XCD (= one L2) owning a contiguous range of tiles: the N/BN workgroups that share a pixel
  // tile run back to back on the same L2 instead of re-fetching it from HBM once per n-tile
  const int nNt = p.N / BN;
  const int bid = fx_xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / nNt) * BM, n0 = (bid % nNt) * BN;
  const int lo = KT == 9 ? m0 - p.W - 1 : m0;
  const int NCH = p.C / CC;
  const int buf_bytes = p.HLp * ROWB;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  unsigned char* T = smem;  // output tile [BM][BN] bf16 (aliases the halo buffers after the K loop), rows of BN*2 bytes, pw_swz<RLT>

  // waves [first, first + step, ...) of the DMA instructions of chunk cc: the loader alone streams the later chunks behind the
  // MFMAs; the first chunk (nothing to hide behind) and the residual tile are split over all waves - one wave issues
  // LDS-DMA at ~25 GB/s, a 64 KiB tile would take 2.6 us
  auto dma_chunk = [&](int cc, unsigned char* buf, int first, int step) {
    const int ninstr = p.HLp * RL / 64;
    for (int i = first; i < ninstr; i += step) {
      const int q = i * 64 + lane;
      const int r = q / RL, pc = q % RL;
      const int lc = pw_swz<RL>(r, pc);
      const int f = lo + r;
      const bool ok = f >= 0 && f < p.M;
      pw_dma16(xr, buf + i * 1024, ok ? (unsigned)(f * p.ldx + cc * CC + lc * 8) * 2u : FX_OOB);
    }
  };

  auto dma_res = [&](int first) {   // residual tile -> T, instructions first, first + NW + 1, ...
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, p.r_bytes, 0x00020000);
    for (int i = first; i < BM * RLT / 64; i += NDW) {
      const int q = i * 64 + lane;
      const int r = q / RLT, pc = q % RLT;
      const int lc = pw_swz<RLT>(r, pc);
      const int m = m0 + r;
      pw_dma16(rr, T + i * 1024, m < p.M ? (unsigned)(m * p.ldr + n0 + lc * 8) * 2u : FX_OOB);
    }
  };

  // ---- consumer state
  f32x16 acc[TN][TM];
  bf16x8 ar[PF][TN];
  unsigned mask9[TM];
  int rbase[TM];
  const int Cs = p.C >> 4;          // k16 steps per filter tap over all channels
  const bf16_t* wbase = p.wp + (size_t)((n0 >> 5) + wn * TN) * (size_t)(KT * Cs) * 512 + lane * 8;
  // fragment of n-block a, chunk cc, in-chunk step s (tap = s / KJ, j = s % KJ); k = filter_tap * C + channel
  auto a_ptr = [&](int a, int cc, int s) -> const bf16_t* {
    const int ks = KT == 9 ? (s / KJ) * Cs + cc * KJ + (s % KJ) : cc * (CC / 16) + s;
    return wbase + ((size_t)a * (size_t)(KT * Cs) + ks) * 512;
  };
  // The two roles are separate code regions with the SAME barrier sequence (one __syncthreads per chunk, then E1 / [E2] / E3
  // below): keeping them in one loop makes the register allocator carry the 128 accumulator registers through the loader's
  // branch and spill them at every chunk boundary.
  if (is_loader) {
    if (lane < 8) *reinterpret_cast<uint4*>(smem + (NCH > 1 ? 2 : 1) * buf_bytes + lane * 16) = make_uint4(0, 0, 0, 0);  // the zero row (published by the first barrier)
    dma_chunk(0, smem, NW, NDW);
    for (int cc = 0; cc < NCH; ++cc) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // chunk cc has landed; every consumer is done with chunk cc-1 (the other buffer)
      if (cc + 1 < NCH) dma_chunk(cc + 1, smem + ((cc + 1) & 1) * buf_bytes, 0, 1);
    }
    __syncthreads();  // E1: the halo buffers are dead; the output tile T (aliases them) may be written
    if constexpr (RESMODE != 0) {
      dma_res(NW);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // E2: residual tile in T
    }
    __syncthreads();  // E3: output tile complete
  } else {
    if (!LOADER && wave == 0 && lane < 8) *reinterpret_cast<uint4*>(smem + buf_bytes + lane * 16) = make_uint4(0, 0, 0, 0);  // zero row
    dma_chunk(0, smem, wave, NDW);   // this wave's share of the first chunk (hipcc waits for it at the first barrier)
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        // bias == NULL: the input-gradient convolutions of the training path
        const float4 bb = p.bias ? *reinterpret_cast<const float4*>(p.bias + n0 + (wn * TN + a) * 32 + 8 * gq + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int b = 0; b < TM; ++b) {
          acc[a][b][4 * gq] = bb.x; acc[a][b][4 * gq + 1] = bb.y; acc[a][b][4 * gq + 2] = bb.z; acc[a][b][4 * gq + 3] = bb.w;
        }
      }
    const int HW = p.H * p.W;
#pragma unroll
    for (int b = 0; b < TM; ++b) {
      const int pl = (wm * TM + b) * 32 + l32;
      const int m = m0 + pl;
      rbase[b] = KT == 9 ? pl + p.W + 1 : pl;
      const bool ok = m < p.M;
      const int mm = ok ? m : 0;
      const int rem = mm % HW;
      const int yy = rem / p.W, xx = rem - yy * p.W;
      unsigned msk = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int dy = t / 3 - 1, dx = t % 3 - 1;
        if (ok && (unsigned)(yy + dy) < (unsigned)p.H && (unsigned)(xx + dx) < (unsigned)p.W) msk |= 1u << t;
      }
      mask9[b] = KT == 9 ? msk : 0x1ffu;   // pointwise: rows past M are zero-filled by the DMA and never stored
    }
#pragma unroll
    for (int i = 0; i < PF; ++i)
#pragma unroll
      for (int a = 0; a < TN; ++a) {
        ar[i][a] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        c3_ldg_async(ar[i][a], a_ptr(a, 0, i));
      }
    const int zrow = (NCH > 1 ? 2 : 1) * buf_bytes;  // 128 zero bytes behind the halo buffers: where a masked (pixel, tap) reads its operand from
    // per-lane constants of the k16 steps: byte offset of the lane's 16-byte chunk before the row swizzle
    int hc[KJ];
#pragma unroll
    for (int j = 0; j < KJ; ++j) hc[j] = (j * 2 + half) << 4;
    // row swizzle (pw_swz<RL>) as a byte mask on the chunk offset: RL = 8: ((r>>1)&7)<<4 = (r<<3)&0x70; RL >= 16: (r&15)<<4 = (r<<4)&0xf0
    constexpr int SWM = (RL >= 16 ? 0xf0 : 0x70);
    constexpr int SWS = (RL >= 16 ? 4 : 3);
    // Per (tap, block): row byte address `ra` and swizzle `sw`; an invalid (pixel, tap) - image border, or the neighbouring image
    // the blindly loaded halo contains - is redirected to the 128-byte zero row (ra = zrow, sw = 0): two selects per tap and
    // block, nothing to do per k-step or once the data is back.  A fragment address is then ONE v_xad (xor + add).
    auto tap_setup = [&](int t, int toff, int bufo, int (&ra)[TM], int (&sw)[TM]) {
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        const int r = rbase[b] + toff;
        const bool ok = KT == 9 ? ((mask9[b] >> t) & 1u) : true;
        ra[b] = ok ? bufo + r * ROWB : zrow;
        sw[b] = ok ? ((r << SWS) & SWM) : 0;
      }
    };
    for (int cc = 0; cc < NCH; ++cc) {
      if (cc == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the first chunk
      __syncthreads();  // chunk cc has landed; every consumer is done with chunk cc-1 (the other buffer)
      const int bufo = (cc & 1) * buf_bytes;
      const int ccn = cc + 1 < NCH ? cc + 1 : cc;
      int ra[TM], sw[TM];
      tap_setup(0, KT == 9 ? -p.W - 1 : 0, bufo, ra, sw);
      bf16x8 xb[2][TM];
#pragma unroll
      for (int b = 0; b < TM; ++b) xb[0][b] = *reinterpret_cast<const bf16x8*>(smem + ((hc[0] ^ sw[b]) + ra[b]));   // tap 0: column offset 0
      // The tap loop is a real loop (one ring cycle of KJ k-steps per tap): fully unrolled, hipcc hoists 36 steps' worth of
      // addresses and loads and spills hundreds of registers.
#pragma unroll 1
      for (int t = 0; t < NTAP; ++t) {
        const int tn = t + 1;
        const bool last_tap = t == NTAP - 1;
        const int co = KT == 9 ? 0 : t * 128, con = KT == 9 ? 0 : tn * 128;   // pointwise: byte offset of the tap's 64 channels within the row
        // weight fragments to request during this tap: the same k-steps of the next tap (tap 0 of the next chunk after the last
        // tap; last chunk: a harmless re-read) - one base pointer per n-block, the k-step is an immediate offset
        const bf16_t* wnext[TN];
#pragma unroll
        for (int a = 0; a < TN; ++a) wnext[a] = a_ptr(a, last_tap ? ccn : cc, last_tap ? 0 : tn * KJ);
        int ran[TM], swn[TM];
        tap_setup(last_tap ? 0 : tn, (KT != 9 || last_tap) ? 0 : (tn / 3 - 1) * p.W + (tn % 3 - 1), bufo, ran, swn);
        auto kstep = [&](auto jc) {   // j must be a compile-time constant: it is the immediate offset of the asm weight loads
          constexpr int j = decltype(jc)::value;
          constexpr bool tap_end = j + 1 == KJ;
          if constexpr (TN == 1) c3_wait<(KJ - 1) * TN>(ar[j][0]); else c3_wait<(KJ - 1) * TN>(ar[j][0], ar[j][1]);
          // One wave per SIMD issues in order: an MFMA occupies the matrix pipe for 32 cycles but only 4 issue cycles, so the
          // other work of the step is INTERLEAVED between the MFMAs (pinned with sched_barrier: left alone, hipcc groups the 8
          // MFMAs back to back and the wave's address / LDS / load instructions wait behind them):
          //   after MFMA (a=0, b): the next k-step's pixel fragment b (address = one xor + add, then ds_read_b128) - it has
          //   the remaining MFMAs of this step to land; after the last MFMA of n-block a: the refill of its weight-ring slot.
#pragma unroll
          for (int a = 0; a < TN; ++a) {
#pragma unroll
            for (int b = 0; b < TM; ++b) {
              acc[a][b] = FX_MFMA_32x32x16(ar[j][a], xb[j & 1][b], acc[a][b]);
              if (a == 0) {
                if constexpr (!tap_end) {
                  xb[(j + 1) & 1][b] = *reinterpret_cast<const bf16x8*>(smem + (((co + hc[tap_end ? 0 : j + 1]) ^ sw[b]) + ra[b]));
                } else {
                  if (!last_tap) xb[0][b] = *reinterpret_cast<const bf16x8*>(smem + (((con + hc[0]) ^ swn[b]) + ran[b]));
                }
              }
              if (b == TM - 1) c3_ldg_async<j * 1024>(ar[j][a], wnext[a]);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        };
        c3_static_for<KJ>(kstep);
#pragma unroll
        for (int b = 0; b < TM; ++b) {
          ra[b] = ran[b];
          sw[b] = swn[b];
        }
      }
    }
    // drain the hidden loads before their registers are reused by the epilogue
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      if constexpr (TN == 1) c3_wait<0>(ar[i][0]); else c3_wait<0>(ar[i][0], ar[i][1]);
    }
    __syncthreads();  // E1
    if constexpr (RESMODE != 0) {
      dma_res(wave);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // E2
    }
  #pragma unroll
      for (int a = 0; a < TN; ++a)
  #pragma unroll
        for (int b = 0; b < TM; ++b)
  #pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const int row = (wm * TM + b) * 32 + l32;
            const int chunk = (wn * TN + a) * 4 + gq;
            unsigned char* tp = T + row * (BN * 2) + (pw_swz<RLT>(row, chunk) << 4) + half * 8;
            float v0 = acc[a][b][4 * gq], v1 = acc[a][b][4 * gq + 1], v2 = acc[a][b][4 * gq + 2], v3 = acc[a][b][4 * gq + 3];
            float r0 = 0.0f, r1 = 0.0f, r2 = 0.0f, r3 = 0.0f;
            if constexpr (RESMODE != 0) {
              const uint2 rv = *reinterpret_cast<const uint2*>(tp);
              r0 = bf16lo_to_f32(rv.x); r1 = bf16hi_to_f32(rv.x);
              r2 = bf16lo_to_f32(rv.y); r3 = bf16hi_to_f32(rv.y);
            }
            if constexpr (RESMODE == 1) { v0 += r0; v1 += r1; v2 += r2; v3 += r3; }       // act(conv + residual)
            if constexpr (ACT == FX_ACT_RELU) {
              v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); v2 = fmaxf(v2, 0.0f); v3 = fmaxf(v3, 0.0f);
            } else if constexpr (ACT == FX_ACT_SILU) {
              v0 = v0 / (1.0f + __expf(-v0)); v1 = v1 / (1.0f + __expf(-v1)); v2 = v2 / (1.0f + __expf(-v2)); v3 = v3 / (1.0f + __expf(-v3));
            }
            if constexpr (RESMODE == 2) { v0 += r0; v1 += r1; v2 += r2; v3 += r3; }       // act(conv) + residual
            if constexpr (RESMODE == 3) {   // act(conv) * (residual > 0): the ReLU backward of the layer that produced `residual`
              v0 = r0 > 0.0f ? v0 : 0.0f; v1 = r1 > 0.0f ? v1 : 0.0f; v2 = r2 > 0.0f ? v2 : 0.0f; v3 = r3 > 0.0f ? v3 : 0.0f;
            }
            uint2 o;
            o.x = pack_bf16x2(v0, v1);
            o.y = pack_bf16x2(v2, v3);
            *reinterpret_cast<uint2*>(tp) = o;
          }

    __syncthreads();  // E3
  }
  for (int q = tid; q < BM * RLT; q += NT) {
    const int row = q / RLT, lc = q % RLT;
    const int m = m0 + row;
    if (m < p.M) {
      const uint4 v = *reinterpret_cast<const uint4*>(T + row * (BN * 2) + (pw_swz<RLT>(row, lc) << 4));
      size_t yo = (size_t)m * p.ldy;
      if (p.y_bstride) {
        const int bb = m / p.HW;
        yo = (size_t)bb * p.y_bstride + (size_t)(m - bb * p.HW) * p.ldy;
      }
      *reinterpret_cast<uint4*>(p.y + yo + n0 + lc * 8) = v;
    }
  }
}

template <int KT, int CC, int TN, int TM, int WN, int WM, int ACT, int RESMODE, int LOADER = 1>
static int launch_c3(C3Args& a, hipStream_t stream) {
  constexpr int NW = WN * WM, BM = WM * TM * 32, BN = WN * TN * 32, RL = CC / 8;
  constexpr int RPI = 64 / RL > 0 ? 64 / RL : 1;  // rows per DMA instruction
  const int HL = BM + (KT == 9 ? 2 * a.W + 2 : 0);
  a.HLp = (HL + RPI - 1) / RPI * RPI;
  const int halo = (a.C / CC > 1 ? 2 : 1) * a.HLp * CC * 2 + 128, tile = BM * BN * 2;  // chunk buffer(s) + the zero row
  const int smem = halo > tile ? halo : tile;
  if (smem > 160 * 1024) return FX_ERR_UNSUPPORTED;
  if (!LOADER && a.C != CC) return FX_ERR_UNSUPPORTED;
  auto kern = conv3x3_flat_kernel<KT, CC, TN, TM, WN, WM, ACT, RESMODE, LOADER>;
  static int attr_smem = 0;
  if (smem > attr_smem) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) return FX_ERR_RUNTIME;
    attr_smem = smem;
  }
  hipLaunchKernelGGL(kern, dim3(((a.M + BM - 1) / BM) * (a.N / BN)), dim3((NW + LOADER) * 64), smem, stream, a);
  return fx_launch_status();
}

// Epilogue variants: 0 ReLU, 1 SiLU, 2 SiLU + residual after the activation (CSPRepLayer), 3 none, 4 ReLU(conv + residual)
// (bottleneck tail).  -1: not covered.  The 3x3 kernel has 0-3, the pointwise one 0, 1, 3, 4.
int fx_c3_epilogue_mode(int act, bool has_res, int res_after) {
  if (!has_res) return act == FX_ACT_RELU ? 0 : (act == FX_ACT_SILU ? 1 : (act == FX_ACT_NONE ? 3 : -1));
  if (act == FX_ACT_SILU && res_after == 1) return 2;
  if (act == FX_ACT_RELU && res_after == 0) return 4;
  if (act == FX_ACT_NONE && res_after == 2) return 5;   // training: input gradient masked by the saved ReLU output
  if (act == FX_ACT_NONE && res_after == 0) return 6;   // training: input gradient + the shortcut branch's gradient (pointwise only)
  return -1;
}

// 1 iff fx_conv2d_nhwc_bf16 routes this 3x3 shape to the halo kernel (the halo tile must fit the 160 KiB LDS)
extern "C" int fx_conv3x3_flat_supported(int C, int N, int W) {
  if (fx_conv3x3_kplane_supported(C, N, W) || fx_conv3x3_c32_supported(C, N, W, 0)) return 1;
  if (C % 64 != 0 || !(N == 64 || N == 128 || N == 256)) return 0;
  const int BM = N == 256 ? 128 : 256;
  const int HLp = (BM + 2 * W + 2 + 7) / 8 * 8;
  const int halo = (C / 64 > 1 ? 2 : 1) * HLp * 128 + 128, tile = BM * N * 2;
  return (halo > tile ? halo : tile) <= 160 * 1024 ? 1 : 0;
}

static void c3_fill(C3Args& a, const ConvArgs& c, const bf16_t* w_frag) {
  a.x = c.x; a.wp = w_frag; a.bias = c.bias; a.res = c.res; a.y = reinterpret_cast<bf16_t*>(c.y);
  a.H = c.H; a.W = c.W; a.C = c.C; a.N = c.N; a.ldx = c.ldx; a.ldy = c.ldy; a.ldr = c.ldr; a.M = c.M;
  a.act = c.act; a.res_after = c.res_after; a.HLp = 0; a.HW = c.Ho * c.Wo; a.y_bstride = c.y_bstride;
  a.x_bytes = c.x_bytes; a.r_bytes = c.r_bytes;
}

int fx_launch_conv3x3_flat(const ConvArgs& c, const bf16_t* w_frag, hipStream_t stream) {
  // round 3: the k-plane kernel (conv3x3_kplane.hip) wherever its plane fits the LDS; FX_C3_KPLANE=0 keeps the round-2 kernel
  // below for A/B runs.  (What was tried on this kernel before the rewrite - consumers issuing the chunk DMA themselves for
  // 8-wave / 256-register tiles - is in the history of scripts/probes/c3_probe.hip and profiles/r03_c3_probe_*.txt: LDS-DMA and
  // register loads of one wave do not retire in order, so counted vmcnt waits over a mixed queue read stale fragments under load.)
  static const int kplane_on = fx_tune("FX_C3_KPLANE", 1);
  if (c.C == 32) return fx_launch_conv3x3_c32(c, w_frag, stream);
  // round 5: the 64 -> 64 layers (res2 branch2b) on the LDS-resident-filter kernel (conv3x3_c64.hip)
  if (fx_conv3x3_c64_supported(c.C, c.N, fx_c3_epilogue_mode(c.act, c.res != nullptr, c.res_after)) && !c.y_bstride) return fx_launch_conv3x3_c64(c, w_frag, stream);
  if (kplane_on && fx_conv3x3_kplane_supported(c.C, c.N, c.W)) return fx_launch_conv3x3_kplane(c, w_frag, stream);
  C3Args a;
  c3_fill(a, c, w_frag);
  // 4 consumer waves + the loader, one workgroup per CU; 2 x 4 accumulator blocks per wave where N allows (operand economy:
  // 2 weight + 4 pixel fragments per 8 MFMAs).  N = 64: 256 pixels x 64 channels; 128: 256 x 128; 256: 128 pixels x 256 channels.
  const int mode = fx_c3_epilogue_mode(c.act, c.res != nullptr, c.res_after);
#define FX_C3_TILE(ACT_, RM_)                                                      \
  {                                                                                \
    if (c.N == 64) return launch_c3<9, 64, 1, 4, 2, 2, ACT_, RM_>(a, stream);      \
    if (c.N == 128) return launch_c3<9, 64, 2, 4, 2, 2, ACT_, RM_>(a, stream);     \
    return launch_c3<9, 64, 2, 4, 4, 1, ACT_, RM_>(a, stream);                     \
  }
  switch (mode) {
    case 0: FX_C3_TILE(FX_ACT_RELU, 0)
    case 1: FX_C3_TILE(FX_ACT_SILU, 0)
    case 2: FX_C3_TILE(FX_ACT_SILU, 2)
    case 3: FX_C3_TILE(FX_ACT_NONE, 0)
    case 5: FX_C3_TILE(FX_ACT_NONE, 3)
    default: return FX_ERR_UNSUPPORTED;
  }
#undef FX_C3_TILE
}

// Pointwise (1x1) layers with C % 256 == 0 and N % 256 == 0: 128 pixels x 256 channels per workgroup, K in 256-channel chunks
int fx_launch_pw_flat(const ConvArgs& c, const bf16_t* w_frag, hipStream_t stream) {
  // round 3: layers whose reduction fits the LDS (K = 256 / 512) on the resident-tile kernel (conv_pw_kplane.hip); FX_PW_KPLANE=0
  // keeps the round-2 kernel below for A/B runs
  static const int kplane_on = fx_tune("FX_PW_KPLANE", 1);
  if (kplane_on && fx_pw_kplane_supported(c.C, c.N, fx_c3_epilogue_mode(c.act, c.res != nullptr, c.res_after)) && c.N * 4 + 128 * c.C * 2 <= 160 * 1024)
    return fx_launch_pw_kplane(c, w_frag, stream);
  C3Args a;
  c3_fill(a, c, w_frag);
  static const int no_loader = fx_tune("FX_PW_NO_LOADER", 1);
  const int mode = fx_c3_epilogue_mode(c.act, c.res != nullptr, c.res_after);
  if (c.C == 256 && no_loader) {   // K = one chunk: 4-wave workgroups, two per CU
    switch (mode) {
      case 0: return launch_c3<1, 256, 2, 4, 4, 1, FX_ACT_RELU, 0, 0>(a, stream);
      case 1: return launch_c3<1, 256, 2, 4, 4, 1, FX_ACT_SILU, 0, 0>(a, stream);
      case 3: return launch_c3<1, 256, 2, 4, 4, 1, FX_ACT_NONE, 0, 0>(a, stream);
      case 4: return launch_c3<1, 256, 2, 4, 4, 1, FX_ACT_RELU, 1, 0>(a, stream);
      case 5: return launch_c3<1, 256, 2, 4, 4, 1, FX_ACT_NONE, 3, 0>(a, stream);
      case 6: return launch_c3<1, 256, 2, 4, 4, 1, FX_ACT_NONE, 1, 0>(a, stream);
      default: return FX_ERR_UNSUPPORTED;
    }
  }
  // round 6: 64-pixel tiles (TM = 2: 64 KiB of chunk buffers, 127 registers) for the SMALL-M deep-K pointwise layers (the 20x20 level: res5's
  // branch2a / shortcut, the 2048 -> 256 projection; M = 12 800 at bs = 32) - twice the workgroups, two of them per CU.  Measured per launch
  // (profiles/r06_pw_knobs.txt): 92.6 -> 82.7, 86.1 -> 75.8, 43.1 -> 38.0, 42.5 -> 36.1, 40.7 -> 28.5 us; at M = 51 200 (res4's branch2a) no
  // change, so the limit stays FX_PWF_BM64_MAX_M (20 000; 0 = off).
  static const int bm64 = fx_tune("FX_PWF_BM64_MAX_M", 20000);
  if (c.M <= bm64 && (mode == 0 || mode == 3)) {
    if (mode == 0) return launch_c3<1, 256, 2, 2, 4, 1, FX_ACT_RELU, 0>(a, stream);
    return launch_c3<1, 256, 2, 2, 4, 1, FX_ACT_NONE, 0>(a, stream);
  }
  switch (mode) {
    case 0: return launch_c3<1, 256, 2, 4, 4, 1, FX_ACT_RELU, 0>(a, stream);
    case 1: return launch_c3<1, 256, 2, 4, 4, 1, FX_ACT_SILU, 0>(a, stream);
    case 3: return launch_c3<1, 256, 2, 4, 4, 1, FX_ACT_NONE, 0>(a, stream);
    case 4: return launch_c3<1, 256, 2, 4, 4, 1, FX_ACT_RELU, 1>(a, stream);
    case 5: return launch_c3<1, 256, 2, 4, 4, 1, FX_ACT_NONE, 3>(a, stream);
    case 6: return launch_c3<1, 256, 2, 4, 4, 1, FX_ACT_NONE, 1>(a, stream);
    default: return FX_ERR_UNSUPPORTED;
  }
}
