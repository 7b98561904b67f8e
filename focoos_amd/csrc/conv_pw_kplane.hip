// Pointwise (1x1) layers with the WHOLE reduction resident in LDS (gfx950, round 3): K = 256 (+ optional residual) or K = 512.
//
// The round-2 pointwise kernel (conv3x3_flat.hip, KT = 1) launches one workgroup per (128-pixel tile, 256-channel tile): a layer
// such as res4's branch2c (256 -> 1024, + residual, ReLU) fetches every pixel tile four times (from L2), runs 2 us of MFMAs between
// a load phase and a store phase that nothing overlaps, and reaches 2.4 TB/s of algorithmic traffic where the layer is HBM-bound.
// Here a workgroup owns 128 pixels and walks ALL output-channel tiles:
//   * the pixel tile is fetched ONCE (LDS-DMA by all five waves) in the k-plane layout of conv3x3_kplane.hip: the 16-byte piece a lane
//     feeds to k-step j lives at row*16 + (half*KS + j)*PLANE, fragment reads are ds_read_b128 with immediate offsets from ONE
//     address register per 32-pixel block, set once per launch - no address arithmetic in the loop, no swizzle;
//   * weights: MFMA A operand in fragment order from L2 through the 4-slot register ring, one continuous stream over the n-tiles
//     (scalar-base loads, the pointer update is SALU work);
//   * epilogue: bias (accumulator init) + residual / activation in registers, v_permlane32_swap pairs -> 16-byte row stores straight to
//     memory (no output tile in LDS, no barrier);
//   * residual: read in the epilogue with the SAME 16-byte-per-lane addressing as the stores (one row block ahead), and taken back to
//     the accumulator layout by the same v_permlane32_swap pairs (the swap is its own inverse).  (First form of this kernel: a fifth
//     wave DMA-ing a 64 KiB residual tile into LDS per n-tile, two barriers per n-tile, one workgroup per CU - 9.7 us per n-tile where
//     the MFMAs take 2; now the workgroup is four waves and two of them share a CU, one's epilogue beside the other's K loop.)
// Per 128-pixel tile of branch2c: 64 KiB in + 4 x (64 KiB residual + 64 KiB out) against 4 x 2 us of MFMAs: HBM-bound as it should be.
#include <type_traits>

#include "pw_common.h"

#ifndef FX_PWK_RES_DEPTH
#define FX_PWK_RES_DEPTH 3
#endif

int fx_tune(const char* env_name, int default_value);  // conv_igemm.hip

struct PWKArgs {
  const bf16_t* x;
  const bf16_t* wp;
  const float* bias;
  const bf16_t* res;
  bf16_t* y;
  int N, ldx, ldy, ldr, M;
  int ntg;            // n-tiles per workgroup (blockIdx.y walks the groups: small-M layers spread their n-tiles over more workgroups)
  int HW;             // pixels per image
  int64_t y_bstride;  // elements between images of y (0: contiguous)
  unsigned x_bytes, r_bytes;
  int rot;            // 1: workgroup i walks its n-tiles starting at tile i % count (round 6, see the kernel)
};

template <int OFF>
__device__ __forceinline__ void pwk_ldg(bf16x8& dst, unsigned voff, const bf16_t* sbase) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "+v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
typedef __attribute__((address_space(3))) const bf16x8 pwk_lds_frag_t;
template <int IMM>
__device__ __forceinline__ bf16x8 pwk_lds_read(int addr) {
  return *reinterpret_cast<pwk_lds_frag_t*>((size_t)(unsigned)(addr + IMM));
}
__device__ __forceinline__ float pwk_act(float v, std::integral_constant<int, FX_ACT_RELU>) { return fmaxf(v, 0.0f); }
__device__ __forceinline__ float pwk_act(float v, std::integral_constant<int, FX_ACT_SILU>) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ float pwk_act(float v, std::integral_constant<int, FX_ACT_NONE>) { return v; }

// K: reduction length (256 / 512), resident.  4 waves side by side over a 256-channel n-tile (2 x 4 accumulator blocks each:
// 128 pixels x 64 channels).  RESMODE: 0 none, 1 act(conv + residual), 3 act(conv) * (residual > 0).  Four waves at <= 256
// registers, so that TWO workgroups share a CU when the pixel tile is 64 KiB (K = 256): one's epilogue (residual loads, conversion,
// stores - nothing for the matrix cores) runs beside the other's K loop.
// TM: 32-pixel blocks per wave = pixel tile / 32.  4 (128 pixels, two workgroups per CU) or - round 6 - 2 (64 pixels: 32 KiB of LDS at
// K = 256 and <= 168 registers, THREE workgroups per CU: these layers are HBM-bound, and what a CU can keep in flight towards memory is set by
// the number of waves that are in a load / store phase, not by its MFMA rate - at 64 pixels a wave issues one memory instruction per MFMA,
// which would cap an MFMA-bound kernel at half rate and costs nothing here).
// ABL: ablation switches of scripts/probes/pw_probe.hip (1: weight ring never refilled, 2: no output stores, 4: residual not read,
//      8: no MFMAs, 16: pixel tile not fetched); 0 in the product.
template <int K, int ACT, int RESMODE, int TM = 4, int ABL = 0>
__global__ __launch_bounds__(256, TM == 4 ? 2 : 3) void conv_pw_kplane_kernel(const PWKArgs p) {
  constexpr int TN = 2, NW = 4, BM = TM * 32, BN = 256;
  constexpr int NTHR = 256, NDW = NTHR / 64;
  constexpr int KS = K / 16;                 // k16 steps
  constexpr int PLANE = BM * 16, XBYTES = 2 * KS * PLANE;   // = BM * K * 2
  constexpr int PF = 4;
  static_assert(KS % PF == 0 && (KS - 1) * PLANE < 65536, "plane offsets are 16-bit immediates");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // the bias vector, staged once: an ordinary global load inside the K loop would make hipcc drain the (asm-issued) weight ring
  float* biasL = reinterpret_cast<float*>(smem + XBYTES);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, half = lane >> 5;
  const int m0 = fx_xcd_remap(blockIdx.x, gridDim.x) * BM;
  const int nt_first = blockIdx.y * p.ntg;
  const int NT = min(p.N / BN, nt_first + p.ntg);   // this workgroup's n-tiles: [nt_first, NT)
  // Rotation (round 6): the workgroups of a launch start together and would all walk the n-tiles in the same order - at any moment every
  // residual read and every store of the chip falls into the same 512-byte column window of the 2-4 KiB output rows, i.e. onto a fraction of
  // the memory channels.  Workgroup i starts at tile i % count instead.
  const int cnt = NT - nt_first;
  const int rot = p.rot ? (int)(blockIdx.x % (unsigned)cnt) : 0;
  auto tile_of = [&](int k) { const int t = k + rot; return nt_first + (t >= cnt ? t - cnt : t); };

  for (int i = tid; i < p.N; i += NTHR) biasL[i] = p.bias ? p.bias[i] : 0.0f;
  // pixel tile: instruction i = (row block i / (K/8), piece i % (K/8)); lane = row.  The pieces of a row block back to back: the 64
  // lines they share are fetched once.
  {
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    constexpr int NP = K / 8;
    static_assert(BM % 64 == 0, "64-row DMA blocks");
    for (int i = wave; i < (BM / 64) * NP; i += NDW) {
      const int blk = i / NP, c = i % NP;
      const int m = m0 + blk * 64 + lane;
      const int pln = (c & 1) * KS + (c >> 1);
      if constexpr (!(ABL & 16)) pw_dma16(xr, smem + pln * PLANE + blk * 1024, m < p.M ? (unsigned)(m * p.ldx + c * 8) * 2u : FX_OOB);
    }
  }
  f32x16 acc[TN][TM];
  bf16x8 ar[PF][TN];
  const unsigned wvoff = lane * 16;
  // weights: fragment (n-block, k-step) = 512 elements, n-block major: the KS fragments of an n-block are contiguous
  const bf16_t* wwave = p.wp + (size_t)(wave * TN) * KS * 512;     // + nt * 8 n-blocks per n-tile
  auto w_ptr = [&](int a, int nt, int ks) -> const bf16_t* { return wwave + ((size_t)(nt * 8 + a) * KS + ks) * 512; };
#pragma unroll
  for (int a = 0; a < TN; ++a) {
    const bf16_t* w0 = w_ptr(a, tile_of(0), 0);
    c3_static_for<PF>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      ar[i][a] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      pwk_ldg<i * 1024>(ar[i][a], wvoff, w0);
    });
  }
  const int lds0 = (int)(unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
  int addr[TM];
#pragma unroll
  for (int b = 0; b < TM; ++b) addr[b] = lds0 + (b * 32 + l32) * 16 + half * KS * PLANE;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the pixel tile (and the ring)
  __syncthreads();   // S

  for (int kt = 0; kt < cnt; ++kt) {
    const int nt = tile_of(kt);
    const int n0 = nt * BN;
    const int ntn = kt + 1 < cnt ? tile_of(kt + 1) : nt;
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const float4 bb = *reinterpret_cast<const float4*>(biasL + n0 + (wave * TN + a) * 32 + 8 * gq + 4 * half);
#pragma unroll
        for (int b = 0; b < TM; ++b) {
          acc[a][b][4 * gq] = bb.x; acc[a][b][4 * gq + 1] = bb.y; acc[a][b][4 * gq + 2] = bb.z; acc[a][b][4 * gq + 3] = bb.w;
        }
      }
    bf16x8 xb[2][TM];
#pragma unroll
    for (int b = 0; b < TM; ++b) xb[0][b] = pwk_lds_read<0>(addr[b]);
    // KS / 4 ring cycles of 4 k-steps; the refills of a cycle request the same slots of the next cycle (first cycle of the next
    // n-tile behind the last one; last n-tile: a harmless re-read)
#pragma unroll 1
    for (int g = 0; g < KS / PF; ++g) {
      const bool last = g == KS / PF - 1;
      const bf16_t* wnext[TN];
#pragma unroll
      for (int a = 0; a < TN; ++a) wnext[a] = w_ptr(a, last ? ntn : nt, last ? 0 : (g + 1) * PF);
      const int aoff = g * PF * PLANE;   // byte offset of this cycle's first plane
      auto kstep = [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (TN == 1) c3_wait<(PF - 1) * TN>(ar[j][0]); else c3_wait<(PF - 1) * TN>(ar[j][0], ar[j][1]);
#pragma unroll
        for (int a = 0; a < TN; ++a) {
#pragma unroll
          for (int b = 0; b < TM; ++b) {
            if constexpr (!(ABL & 8)) acc[a][b] = FX_MFMA_32x32x16(ar[j][a], xb[j & 1][b], acc[a][b]);
            if (a == 0) {   // fragment b of the next k-step (behind the last one of the n-tile: k-step 0 again, unused)
              if constexpr (j + 1 < PF) xb[(j + 1) & 1][b] = pwk_lds_read<(j + 1) * PLANE>(addr[b] + aoff);
              else xb[0][b] = pwk_lds_read<PF * PLANE>(addr[b] + (last ? -PF * PLANE : aoff));
            }
            if constexpr (!(ABL & 1)) {
              if (b == TM - 1) pwk_ldg<j * 1024>(ar[j][a], wvoff, wnext[a]);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      };
      c3_static_for<PF>(kstep);
    }
    // ---- epilogue of n-tile nt.  Residual: rr[.][a*2+g2] = the 16 bytes this lane will store at (row, a*32 + g2*16 + half*8), one
    // row block ahead of its use.  (Ordinary loads: they are younger than the ring refills in flight, so the compiler's own vmcnt
    // waits stay correct - a wait for them also covers the refills, which the next n-tile needs first thing anyway.)
    // residual prefetch depth (round 6): RD row blocks of the n-tile are requested before the first is consumed.  The cold-buffer probe
    // (profiles/r06_pw_probe_cold.txt) shows this epilogue's reads as the launch's bound: 79.5 us as shipped, 41.9 without the residual,
    // against 39.7 for a copy of the output tensor - one block ahead keeps too few bytes in flight per CU.  Depth 1 / 2 / 3 / 4 in that
    // probe: 77.8 / 74.5 / 68.4 / 70.1 us (M = 51 200) and 44.7 / 42.7 / 39.6 / 39.3 (M = 25 600); depth 3 costs 64 registers of prefetched rows
    // (256 VGPRs and 7 spilled dwords per lane outside the K loop).
    constexpr int RD = (FX_PWK_RES_DEPTH < TM) ? FX_PWK_RES_DEPTH : TM;
    constexpr int RS = RD + 1 > TM ? TM : RD + 1;      // ring slots (block b + RD is requested while block b is being consumed)
    uint4 rr[RS][TN * 2];
    if constexpr (ABL & 4) {
#pragma unroll
      for (int i = 0; i < RS * TN * 2; ++i) (&rr[0][0])[i] = make_uint4(0, 0, 0, 0);
    }
    auto ld_res = [&](int b, uint4* dst) {
      const int m = min(m0 + b * 32 + l32, p.M - 1);
      const bf16_t* rrow = p.res + (size_t)m * p.ldr + n0 + wave * TN * 32 + half * 8;
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) dst[a * 2 + g2] = *reinterpret_cast<const uint4*>(rrow + a * 32 + g2 * 16);
    };
    if constexpr (RESMODE != 0 && !(ABL & 4)) {
#pragma unroll
      for (int b = 0; b < RD && b < TM; ++b) ld_res(b, rr[b % RS]);
    }
#pragma unroll
    for (int b = 0; b < TM; ++b) {
      if constexpr (RESMODE != 0 && !(ABL & 4)) {
        if (b + RD < TM) ld_res(b + RD, rr[(b + RD) % RS]);
      }
      const int row = b * 32 + l32;
      const int m = m0 + row;
      size_t yo = (size_t)m * p.ldy;
      if (p.y_bstride) {
        const int bb = m / p.HW;
        yo = (size_t)bb * p.y_bstride + (size_t)(m - bb * p.HW) * p.ldy;
      }
      bf16_t* yrow = p.y + yo + n0 + wave * TN * 32 + half * 8;
      const bool live = m < p.M;
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
          unsigned pk[2][2], rq[2][2] = {{0u, 0u}, {0u, 0u}};
          if constexpr (RESMODE != 0) {   // store layout -> accumulator layout: the inverse of the swaps below
            const uint4 R = rr[b % RS][a * 2 + g2];
            const auto s0 = __builtin_amdgcn_permlane32_swap(R.x, R.z, false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(R.y, R.w, false, false);
            rq[0][0] = s0[0]; rq[1][0] = s0[1];
            rq[0][1] = s1[0]; rq[1][1] = s1[1];
          }
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int gq = 2 * g2 + q;
            float v[4];
            const float r[4] = {bf16lo_to_f32(rq[q][0]), bf16hi_to_f32(rq[q][0]), bf16lo_to_f32(rq[q][1]),
                                bf16hi_to_f32(rq[q][1])};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[e] = acc[a][b][4 * gq + e];
              if constexpr (RESMODE == 1) v[e] += r[e];
              v[e] = pwk_act(v[e], std::integral_constant<int, ACT>{});
              if constexpr (RESMODE == 3) v[e] = r[e] > 0.0f ? v[e] : 0.0f;
            }
            pk[q][0] = pack_bf16x2(v[0], v[1]);
            pk[q][1] = pack_bf16x2(v[2], v[3]);
          }
#pragma unroll
          for (int w = 0; w < 2; ++w) {   // lanes 32-63 of pk[0] <-> lanes 0-31 of pk[1]: 16 contiguous bytes per lane
            const auto sw = __builtin_amdgcn_permlane32_swap(pk[0][w], pk[1][w], false, false);
            pk[0][w] = sw[0];
            pk[1][w] = sw[1];
          }
          if (live && !((ABL & 2) && p.M > 0)) *reinterpret_cast<uint4*>(yrow + a * 32 + g2 * 16) = make_uint4(pk[0][0], pk[0][1], pk[1][0], pk[1][1]);
        }
    }
  }
  // the ring's last refills are still in flight: let them land before the wave ends (their registers are dead, but an asm load
  // must not outlive its wave's register allocation)
#pragma unroll
  for (int i = 0; i < PF; ++i) {
    if constexpr (TN == 1) c3_wait<0>(ar[i][0]); else c3_wait<0>(ar[i][0], ar[i][1]);
  }
}

template <int K, int ACT, int RESMODE, int TM = 4, int ABL = 0>
static int launch_pwk(PWKArgs& a, hipStream_t stream) {
  static const int one_per_cu = fx_tune("FX_PWK_ONE_PER_CU", 0);   // A/B knob: pad the LDS request so that one workgroup owns a CU
  constexpr int BM = TM * 32, BN = 256;
  int smem = BM * K * 2 + a.N * 4;
  if (smem > 160 * 1024) return FX_ERR_UNSUPPORTED;
  if (one_per_cu && smem < 96 * 1024) smem = 96 * 1024;
  // n-tiles per workgroup: all of them when the pixel tiles alone fill the chip, else spread over more workgroups (two 64 KiB pixel
  // tiles share a CU)
  const int mt = (a.M + BM - 1) / BM, NT = a.N / BN;
  static const int target = fx_tune("FX_PWK_TARGET_WGS", 512);
  const int want = K == 256 ? target : target / 2;
  int groups = mt >= want * 3 / 4 ? 1 : (want + mt - 1) / mt;
  if (groups > NT) groups = NT;
  a.ntg = (NT + groups - 1) / groups;
  groups = (NT + a.ntg - 1) / a.ntg;
  static const int rot = fx_tune("FX_PWK_ROT", 1);
  a.rot = rot;
  auto kern = conv_pw_kplane_kernel<K, ACT, RESMODE, TM, ABL>;
  static int attr_smem = 0;
  if (smem > attr_smem) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) return FX_ERR_RUNTIME;
    attr_smem = smem;
  }
  hipLaunchKernelGGL(kern, dim3(mt, groups), dim3(256), smem, stream, a);
  return fx_launch_status();
}

// 1 iff fx_launch_pw_kplane covers (C, N, epilogue mode): K = 256 / 512 with every pointwise epilogue
bool fx_pw_kplane_supported(int C, int N, int mode) {
  if (N <= 0 || N % 256 != 0 || N > 4096) return false;
  static const int k512_res = fx_tune("FX_PWK_K512_RES", 1);   // A/B knob: K = 512 layers with a residual (else conv3x3_flat.hip's KT = 1 form)
  if (C == 256) return mode == 0 || mode == 1 || mode == 3 || mode == 4 || mode == 5 || mode == 6;
  if (C == 512) return mode == 0 || mode == 1 || mode == 3 || (k512_res && (mode == 4 || mode == 5 || mode == 6));
  return false;
}

int fx_launch_pw_kplane(const ConvArgs& c, const bf16_t* w_frag, hipStream_t stream) {
  const int mode = fx_c3_epilogue_mode(c.act, c.res != nullptr, c.res_after);
  if (!fx_pw_kplane_supported(c.C, c.N, mode)) return FX_ERR_UNSUPPORTED;
  PWKArgs a{};
  a.x = c.x; a.wp = w_frag; a.bias = c.bias; a.res = c.res; a.y = reinterpret_cast<bf16_t*>(c.y);
  a.N = c.N; a.ldx = c.ldx; a.ldy = c.ldy; a.ldr = c.ldr; a.M = c.M;
  a.HW = c.Ho * c.Wo; a.y_bstride = c.y_bstride; a.x_bytes = c.x_bytes; a.r_bytes = c.r_bytes;
  // 64-pixel tiles (three workgroups per CU) for the layers selected by FX_PWK_BM64: bit 0 = K 256 with a residual (res4's branch2c), bit 1 =
  // K 256 without, bit 2 = K 512 with a residual (res5's branch2c), bit 3 = K 512 without
  static const int bm64 = fx_tune("FX_PWK_BM64", 0);
  const bool has_res = mode >= 4;
  if (bm64 & (c.C == 256 ? (has_res ? 1 : 2) : (has_res ? 4 : 8))) {
    if (c.C == 256) {
      switch (mode) {
        case 0: return launch_pwk<256, FX_ACT_RELU, 0, 2>(a, stream);
        case 1: return launch_pwk<256, FX_ACT_SILU, 0, 2>(a, stream);
        case 3: return launch_pwk<256, FX_ACT_NONE, 0, 2>(a, stream);
        case 4: return launch_pwk<256, FX_ACT_RELU, 1, 2>(a, stream);
      }
    } else {
      switch (mode) {
        case 0: return launch_pwk<512, FX_ACT_RELU, 0, 2>(a, stream);
        case 1: return launch_pwk<512, FX_ACT_SILU, 0, 2>(a, stream);
        case 3: return launch_pwk<512, FX_ACT_NONE, 0, 2>(a, stream);
        case 4: return launch_pwk<512, FX_ACT_RELU, 1, 2>(a, stream);
      }
    }
  }
  if (c.C == 256) {
    switch (mode) {
      case 0: return launch_pwk<256, FX_ACT_RELU, 0>(a, stream);
      case 1: return launch_pwk<256, FX_ACT_SILU, 0>(a, stream);
      case 3: return launch_pwk<256, FX_ACT_NONE, 0>(a, stream);
      case 4: return launch_pwk<256, FX_ACT_RELU, 1>(a, stream);
      case 5: return launch_pwk<256, FX_ACT_NONE, 3>(a, stream);
      case 6: return launch_pwk<256, FX_ACT_NONE, 1>(a, stream);   // training: input gradient + the shortcut branch's gradient
    }
  } else {
    switch (mode) {
      case 0: return launch_pwk<512, FX_ACT_RELU, 0>(a, stream);
      case 1: return launch_pwk<512, FX_ACT_SILU, 0>(a, stream);
      case 3: return launch_pwk<512, FX_ACT_NONE, 0>(a, stream);
      case 4: return launch_pwk<512, FX_ACT_RELU, 1>(a, stream);
      case 5: return launch_pwk<512, FX_ACT_NONE, 3>(a, stream);
      case 6: return launch_pwk<512, FX_ACT_NONE, 1>(a, stream);
    }
  }
  return FX_ERR_UNSUPPORTED;
}
