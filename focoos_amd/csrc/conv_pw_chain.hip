// Back-to-back pointwise GEMM chain for the ResNet bottleneck seam (gfx950):
//
//   y1 = act1( [x1 | x2] . W1^T + b1 (+ residual) )          branch2c (+ the variant-d shortcut conv as a K-concatenated
//                                                              second source: no shortcut tensor is written or re-read)
//   y2 = act2( y1 . W2^T + b2 )                               the NEXT block's branch2a
//
// Why: the 1x1 convolutions around the residual add are HBM-bound (K = 64..256 per output byte); run separately, the
// block output y1 is written by `c` and read back by the next `a` (419 MB per res2 block at bs=32).  Here the y1 tile
// never leaves the CU between the two GEMMs: it is written to HBM once (it is the next block's residual) and consumed
// from LDS by the second GEMM.
//
// Structure of one workgroup (256 threads = 4 waves, BM = 64 pixels):
//   * X tile(s) [64][K1A] (+ [64][K1B]) -> LDS by buffer_load...lds DMA (XOR swizzle applied to the source address).
//   * N1 is processed in groups of 256 channels; per group:
//       - the residual tile [64][256] is DMA'd straight into the LDS tile T (same swizzle);
//       - GEMM1: wave w owns channels [64w, 64w+64) of the group: the weights are the MFMA A operand and come straight
//         from L2 into registers - they are pre-packed in FRAGMENT order ([n/32][k/16][lane][8]) so that every wave load is
//         one contiguous 1 KiB read; the pixels (B operand) come from the LDS X tile (conflict-free ds_read_b128);
//       - epilogue in the accumulator layout, IN PLACE on T: every lane reads its 4 residual values (8 B), adds bias +
//         accumulator, applies the activation and writes the bf16 result back to the same 8 bytes - no fp32 staging;
//       - T is now the y1 tile: it is stored to HBM as full 16-byte channel vectors (one 512-byte row per 32 lanes) and
//         is the B operand of GEMM2's K-slice of this group (acc2 += W2[:, group] . T).
//   * y2: accumulators -> bf16 -> LDS (T reused) -> coalesced 16-byte stores.
// Numerics are those of the two separate launches: bf16 operands, fp32 accumulate in the same k order, y1 rounded to bf16
// before it feeds GEMM2.  (With a second source the shortcut sum stays in fp32 instead of being rounded to bf16 first.)
#include "pw_common.h"

struct PwChainArgs {
  const bf16_t* x1;
  const bf16_t* x2;
  const bf16_t* res;
  const bf16_t* w1p;
  const bf16_t* w2p;
  const float* b1;
  const float* b2;
  bf16_t* y1;
  bf16_t* y2;
  int ldx1, ldx2, ldr, ldy1, ldy2;
  int M, N1, act1, act2;
  unsigned x1_bytes, x2_bytes, r_bytes;
  // QUAD form (round 4): the tile is 16 consecutive 2x2 pixel quads instead of 64 consecutive pixels, and pool = AvgPool2d(2,2) of y1
  bf16_t* pool;
  int ldp, H, W;     // image height / width (even) of every operand of this launch
};

// DMA `nrows` tile rows of RL 16-byte chunks whose pixel index comes from `pix(row)` (-1: beyond the tensor): pw_dma_rows with a pixel map
template <int RL, typename F>
__device__ __forceinline__ void pw_dma_rows_map(__amdgpu_buffer_rsrc_t r, unsigned char* tile, int nrows, int ld, int col0, int wave, int lane, F&& pix) {
  const int ninstr = nrows * RL / 64;
  for (int i = wave; i < ninstr; i += 4) {
    const int q = i * 64 + lane;
    const int row = q / RL, pc = q % RL;
    const int lc = pw_swz<RL>(row, pc);
    const int m = pix(row);
    const unsigned off = (m >= 0) ? (unsigned)(m * ld + col0 + lc * 8) * 2u : FX_OOB;
    pw_dma16(r, tile + i * 1024, off);
  }
}

// QUAD: tile row r = pixel (sub-position r & 3 of quad blockIdx.x * 16 + (r >> 2)): any 64 pixels do for a pointwise layer, and with whole 2x2
// quads in the tile the variant-d shortcut's AvgPool2d(2, 2) of the block output (resnet.py:46,95) is the average of four rows of the y1 tile
// that sits in LDS anyway - written as a quarter-size second output, the stand-alone pooling pass over y1 (122 MB read per launch) is gone.
template <int K1A, int K1B, int N2, int BM, bool QUAD = false>
__global__ __launch_bounds__(256, ((N2 <= 64 && K1B == 0) ? 4 : ((N2 <= 128 && (K1B == 0 || N2 <= 64)) ? 3 : 2))) void pw_chain_kernel(const PwChainArgs p) {
  constexpr int K1 = K1A + K1B, KS1 = K1 / 16, KS1A = K1A / 16;
  constexpr int RLA = K1A / 8, RLB = (K1B > 0 ? K1B : 64) / 8;
  constexpr int TM = BM / 32;
  constexpr int XA_BYTES = BM * K1A * 2, XB_BYTES = BM * K1B * 2;
  constexpr int PF1 = KS1 < 4 ? KS1 : 4;
  constexpr bool HAS2 = N2 > 0;
  constexpr int WN2 = !HAS2 ? 1 : (N2 >= 128 ? 4 : 2), WM2 = 4 / WN2;
  constexpr int TN2 = !HAS2 ? 1 : N2 / (32 * WN2), TM2 = BM / (32 * WM2);
  constexpr int RL2 = (HAS2 ? N2 : 64) / 8;
  constexpr int PF2 = 4;
  static_assert(BM == 64, "64-pixel tiles");
  static_assert(K1A % 64 == 0 && K1B % 64 == 0 && PF1 == 4, "K segments in multiples of 64 (= whole blocks of 4 k-steps)");
  static_assert(!HAS2 || (N2 % 64 == 0 && N2 <= 256), "N2");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* XA = smem;
  unsigned char* XB = smem + XA_BYTES;
  unsigned char* T = smem + XA_BYTES + XB_BYTES;  // [BM][256] bf16, 512-byte rows

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, half = lane >> 5;
  const int m0 = blockIdx.x * BM;
  const int NG = p.N1 >> 8;
  const int KS2 = p.N1 >> 4;  // k-steps of GEMM2 over all of N1

  // pixel of tile row `row` (-1: beyond the tensor).  QUAD: a table of the tile's 64 pixels in LDS, filled once by the first wave (round 5).
  // The map costs two runtime divisions, and every DMA instruction and every 16-byte store of the kernel asks for it: ~50 divisions per
  // thread and tile = ~1 700 VALU instructions per wave in front of the addresses, against ~48 MFMAs - the quad launches moved 3.2 TB/s
  // where the flat form of the same layers moves 4.8.
  __shared__ int s_pix[QUAD ? BM : 1];
  if constexpr (QUAD) {
    if (tid < BM) {
      const int W2 = p.W >> 1, QI = (p.H >> 1) * W2;
      const int q = blockIdx.x * (BM / 4) + (tid >> 2), sub = tid & 3;
      int m = -1;
      if (q * 4 < p.M) {
        const int b = q / QI, rem = q - b * QI;
        const int y2 = rem / W2, x2 = rem - y2 * W2;
        m = (b * p.H + 2 * y2 + (sub >> 1)) * p.W + 2 * x2 + (sub & 1);
      }
      s_pix[tid] = m;
    }
    __syncthreads();
  }
  auto pix = [&](int row) -> int {
    if constexpr (!QUAD) {
      const int m = m0 + row;
      return m < p.M ? m : -1;
    } else {
      return s_pix[row];
    }
  };
  const __amdgpu_buffer_rsrc_t x1r = __builtin_amdgcn_make_buffer_rsrc((void*)p.x1, 0, p.x1_bytes, 0x00020000);
  pw_dma_rows_map<RLA>(x1r, XA, BM, p.ldx1, 0, wave, lane, pix);
  if constexpr (K1B > 0) {
    const __amdgpu_buffer_rsrc_t x2r = __builtin_amdgcn_make_buffer_rsrc((void*)p.x2, 0, p.x2_bytes, 0x00020000);
    pw_dma_rows_map<RLB>(x2r, XB, BM, p.ldx2, 0, wave, lane, pix);
  }
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res ? p.res : p.x1), 0, p.res ? p.r_bytes : 0u, 0x00020000);

  const int wn2 = wave % WN2, wm2 = wave / WN2;
  f32x16 acc2[TN2][TM2];
#pragma unroll
  for (int a = 0; a < TN2; ++a)
#pragma unroll
    for (int b = 0; b < TM2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[a][b][r] = 0.0f;

  for (int g = 0; g < NG; ++g) {
    // per-iteration copies the optimiser cannot see through: keeps the (dozens of) LDS / global addresses derived from the lane
    // position from being hoisted out of the group loop into long-lived registers
    int l32g = l32, halfg = half, tidg = tid;
    asm volatile("" : "+v"(l32g), "+v"(halfg), "+v"(tidg));
    if (p.res) pw_dma_rows_map<32>(rr, T, BM, p.ldr, g * 256, wave, lane, pix);
    // ---- first weight fragments of GEMM2's K-slice [g*256, +256): requested now, consumed after the y1 store
    bf16x8 a2[PF2][TN2];
    const bf16_t* w2 = p.w2p + (size_t)((wn2 * TN2) * KS2 + g * 16) * 512 + lane * 8;
    if constexpr (HAS2) {
#pragma unroll
      for (int i = 0; i < PF2; ++i)
#pragma unroll
        for (int a = 0; a < TN2; ++a) a2[i][a] = pw_ldg_frag(w2 + (size_t)(a * KS2 + i) * 512);
    }
    // ---- GEMM1 of this group: wave `wave` -> channels [g*256 + wave*64, +64), as two 32-channel passes (keeps one 32x64
    //      accumulator block pair live at a time: 4 workgroups per CU instead of 2)
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const bf16_t* w1 = p.w1p + (size_t)((g * 8 + wave * 2 + a) * KS1) * 512 + lane * 8;  // n-block (32 rows) = KS1 fragments of 512 elements
      bf16x8 a1[PF1];
#pragma unroll
      for (int i = 0; i < PF1; ++i) a1[i] = pw_ldg_frag(w1 + i * 512);
      // accumulators start from the bias (accumulator layout: register 4*gq + j of a 32-channel block = channel 8*gq + 4*halfg + j)
      f32x16 acc[TM];
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const float4 bb = *reinterpret_cast<const float4*>(p.b1 + g * 256 + wave * 64 + a * 32 + 8 * gq + 4 * halfg);
#pragma unroll
        for (int b = 0; b < TM; ++b) {
          acc[b][4 * gq] = bb.x; acc[b][4 * gq + 1] = bb.y; acc[b][4 * gq + 2] = bb.z; acc[b][4 * gq + 3] = bb.w;
        }
      }
      if (a == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // X tile(s) and this group's residual have landed for every wave
      }
      // k-loop in blocks of PF1 steps (the fragment ring is indexed statically inside a block; the block loop itself is
      // not unrolled - a fully unrolled K = 768 loop costs 80 spilled registers)
#define FX_PW_STEPS(XT_, KROW_, RL_, KL0_)                                                                                  \
  _Pragma("unroll") for (int i = 0; i < PF1; ++i) {                                                                         \
    bf16x8 xb[TM];                                                                                                          \
    _Pragma("unroll") for (int b = 0; b < TM; ++b) {                                                                        \
      const int row = b * 32 + l32k;                                                                                         \
      xb[b] = *reinterpret_cast<const bf16x8*>((XT_) + row * ((KROW_)*2) + (pw_swz<RL_>(row, ((KL0_) + i) * 2 + halfg) << 4)); \
    }                                                                                                                       \
    _Pragma("unroll") for (int b = 0; b < TM; ++b) acc[b] = FX_MFMA_32x32x16(a1[i], xb[b], acc[b]); \
    const int kn = ks0 + i + PF1;                                                                                           \
    a1[i] = pw_ldg_frag(w1 + (kn < KS1 ? kn : KS1 - 1) * 512); /* clamped: the last block re-requests a fragment it never uses */ \
  }
#pragma unroll 1
      for (int ks0 = 0; ks0 < KS1; ks0 += PF1) {
        int l32k = l32g;
        asm volatile("" : "+v"(l32k));  // as above: no hoisting of both segments' fragment addresses out of the k loop
        if (K1B == 0 || ks0 < KS1A) {
          FX_PW_STEPS(XA, K1A, RLA, ks0)
        } else {
          FX_PW_STEPS(XB, (K1B > 0 ? K1B : 64), RLB, ks0 - KS1A)
        }
      }
#undef FX_PW_STEPS
      // ---- epilogue 1, in place on T (accumulator layout: lane = pixel l32g, channels 8*gq + 4*halfg + 0..3 of the 32-block)
#pragma unroll
      for (int b = 0; b < TM; ++b)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int row = b * 32 + l32g;
          const int chunk = wave * 8 + a * 4 + gq;
          unsigned char* tp = T + row * 512 + ((chunk ^ (row & 15)) << 4) + halfg * 8;
          float v0 = acc[b][4 * gq], v1 = acc[b][4 * gq + 1], v2 = acc[b][4 * gq + 2], v3 = acc[b][4 * gq + 3];
          if (p.res) {
            const uint2 r2 = *reinterpret_cast<const uint2*>(tp);
            v0 += bf16lo_to_f32(r2.x);
            v1 += bf16hi_to_f32(r2.x);
            v2 += bf16lo_to_f32(r2.y);
            v3 += bf16hi_to_f32(r2.y);
          }
          if (p.act1 == FX_ACT_RELU) {  // ResNet bottlenecks: ReLU or nothing (keeps erff / expf out of the epilogue)
            v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); v2 = fmaxf(v2, 0.0f); v3 = fmaxf(v3, 0.0f);
          }
          uint2 o;
          o.x = pack_bf16x2(v0, v1);
          o.y = pack_bf16x2(v2, v3);
          *reinterpret_cast<uint2*>(tp) = o;
        }
    }
    __syncthreads();  // T = the finished y1 tile of this group
    // ---- y1 -> HBM: one 512-byte row per 32 lanes
#pragma unroll
    for (int i = 0; i < BM * 32 / 256; ++i) {
      const int q = tidg + i * 256;
      const int row = q >> 5, lc = q & 31;
      const uint4 v = *reinterpret_cast<const uint4*>(T + row * 512 + ((lc ^ (row & 15)) << 4));
      const int m = pix(row);
      if (m >= 0 && p.y1) *reinterpret_cast<uint4*>(p.y1 + (size_t)m * p.ldy1 + g * 256 + lc * 8) = v;   // y1 == nullptr (QUAD form only): the block output lives on through y2 and the pooled tensor alone
    }
    if constexpr (QUAD) {
      // AvgPool2d(2, 2) of the y1 tile: quad j = rows 4j .. 4j+3 in the order (dy, dx) = (0,0), (0,1), (1,0), (1,1) - the summation order of
      // avgpool2_kernel (pixel_ops.hip), so the pooled tensor is bit-identical to the stand-alone pass over the stored bf16 y1
#pragma unroll
      for (int i = 0; i < (BM / 4) * 32 / 256; ++i) {
        const int idx = tidg + i * 256;
        const int j = idx >> 5, lc = idx & 31;
        float s8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * j + r;
          float f[8];
          unpack_bf16x8(*reinterpret_cast<const uint4*>(T + row * 512 + ((lc ^ (row & 15)) << 4)), f);
#pragma unroll
          for (int e = 0; e < 8; ++e) s8[e] += f[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) s8[e] *= 0.25f;
        const int q = blockIdx.x * (BM / 4) + j;
        if (q * 4 < p.M) *reinterpret_cast<uint4*>(p.pool + (size_t)q * p.ldp + g * 256 + lc * 8) = pack_bf16x8(s8);
      }
    }
    // ---- GEMM2, K-slice of this group (B operand = T)
    if constexpr (HAS2) {
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        bf16x8 tb[TM2];
#pragma unroll
        for (int b = 0; b < TM2; ++b) {
          const int row = (wm2 * TM2 + b) * 32 + l32g;
          tb[b] = *reinterpret_cast<const bf16x8*>(T + row * 512 + (((ks * 2 + halfg) ^ (row & 15)) << 4));
        }
#pragma unroll
        for (int a = 0; a < TN2; ++a)
#pragma unroll
          for (int b = 0; b < TM2; ++b) acc2[a][b] = FX_MFMA_32x32x16(a2[ks % PF2][a], tb[b], acc2[a][b]);
        if (ks + PF2 < 16) {
#pragma unroll
          for (int a = 0; a < TN2; ++a) a2[ks % PF2][a] = pw_ldg_frag(w2 + (size_t)(a * KS2 + ks + PF2) * 512);
        }
      }
    }
    __syncthreads();  // everyone is done with T before the next group's residual (or the y2 staging) overwrites it
  }

  if constexpr (HAS2) {
    // ---- epilogue 2: bias + activation -> bf16 tile [BM][N2] in T -> coalesced stores
#pragma unroll
    for (int a = 0; a < TN2; ++a)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int n = (wn2 * TN2 + a) * 32 + 8 * gq + 4 * half;
        const float4 bb = *reinterpret_cast<const float4*>(p.b2 + n);
#pragma unroll
        for (int b = 0; b < TM2; ++b) {
          const int row = (wm2 * TM2 + b) * 32 + l32;
          float v0 = acc2[a][b][4 * gq] + bb.x, v1 = acc2[a][b][4 * gq + 1] + bb.y;
          float v2 = acc2[a][b][4 * gq + 2] + bb.z, v3 = acc2[a][b][4 * gq + 3] + bb.w;
          if (p.act2 == FX_ACT_RELU) {
            v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); v2 = fmaxf(v2, 0.0f); v3 = fmaxf(v3, 0.0f);
          }
          uint2 o;
          o.x = pack_bf16x2(v0, v1);
          o.y = pack_bf16x2(v2, v3);
          *reinterpret_cast<uint2*>(T + row * (N2 * 2) + (pw_swz<RL2>(row, n >> 3) << 4) + half * 8) = o;
        }
      }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < BM * RL2 / 256; ++i) {
      const int q = tid + i * 256;
      const int row = q / RL2, lc = q % RL2;
      const uint4 v = *reinterpret_cast<const uint4*>(T + row * (N2 * 2) + (pw_swz<RL2>(row, lc) << 4));
      const int m = pix(row);
      if (m >= 0) *reinterpret_cast<uint4*>(p.y2 + (size_t)m * p.ldy2 + lc * 8) = v;
    }
  }
}

template <int K1A, int K1B, int N2, bool QUAD = false>
static int launch_pw_chain(const PwChainArgs& a, hipStream_t stream) {
  constexpr int BM = 64;
  constexpr int SMEM = BM * (K1A + K1B) * 2 + BM * 512;
  static bool attr_set = false;
  auto kern = pw_chain_kernel<K1A, K1B, N2, BM, QUAD>;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess)
      return FX_ERR_RUNTIME;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((a.M + BM - 1) / BM), dim3(256), SMEM, stream, a);
  return fx_launch_status();
}

// (K1a, K1b, N2) instances: res2 (64 -> 256), res3 (128 -> 512), res4 (256 -> 1024) seams, with / without the shortcut source
#define FX_PW_INSTANCES(X)                                                                             \
  X(64, 0, 0) X(64, 0, 64) X(64, 0, 128) X(64, 64, 0) X(64, 64, 64) X(64, 64, 128)                     \
  X(128, 0, 0) X(128, 0, 128) X(128, 0, 256) X(128, 256, 0) X(128, 256, 128) X(128, 256, 256)          \
  X(256, 0, 0) X(256, 0, 256) X(256, 512, 0) X(256, 512, 256)

extern "C" int fx_pw_chain_supported(int K1a, int K1b, int N1, int N2) {
  if (N1 <= 0 || N1 % 256 != 0) return 0;
#define FX_PW_CASE(KA, KB, NN2) \
  if (K1a == KA && K1b == KB && N2 == NN2) return 1;
  FX_PW_INSTANCES(FX_PW_CASE)
#undef FX_PW_CASE
  return 0;
}

// (K1a, K1b, N2) instances of the QUAD form: the seams in front of a stride-2 block (res2 -> res3, res3 -> res4)
#define FX_PW_POOL_INSTANCES(X) X(64, 0, 128) X(128, 0, 256)

extern "C" int fx_pw_chain_pool_supported(int K1a, int K1b, int N1, int N2) {
  if (N1 <= 0 || N1 % 256 != 0) return 0;
#define FX_PW_CASE(KA, KB, NN2) \
  if (K1a == KA && K1b == KB && N2 == NN2) return 1;
  FX_PW_POOL_INSTANCES(FX_PW_CASE)
#undef FX_PW_CASE
  return 0;
}

extern "C" int fx_pw_chain_bf16(const fx_pw_chain_desc* d, fx_stream_t stream_) {
  FX_CHECK_ARG(d && d->x1 && d->w1 && d->bias1 && d->M > 0);
  FX_CHECK_ARG(d->y1 || (d->pool && d->N2 > 0));   // y1 may be dropped only where the pooled tensor and y2 carry the block output on (round 5: RT-DETR never reads res2 itself)
  FX_CHECK_ARG(d->K1b == 0 || d->x2);
  FX_CHECK_ARG((d->act1 == FX_ACT_NONE || d->act1 == FX_ACT_RELU) && (d->act2 == FX_ACT_NONE || d->act2 == FX_ACT_RELU));
  FX_CHECK_ARG(d->N2 == 0 || (d->w2 && d->bias2 && d->y2));
  if (!fx_pw_chain_supported(d->K1a, d->K1b, d->N1, d->N2)) return FX_ERR_UNSUPPORTED;
  FX_CHECK_ARG(d->ldx1 >= d->K1a && d->ldx1 % 8 == 0 && (!d->y1 || (d->ldy1 >= d->N1 && d->ldy1 % 8 == 0)));
  FX_CHECK_ARG(d->K1b == 0 || (d->ldx2 >= d->K1b && d->ldx2 % 8 == 0));
  FX_CHECK_ARG(!d->residual || (d->ldr >= d->N1 && d->ldr % 8 == 0));
  FX_CHECK_ARG(d->N2 == 0 || (d->ldy2 >= d->N2 && d->ldy2 % 8 == 0));
  FX_CHECK_ARG(((uintptr_t)d->x1 % 16) == 0 && ((uintptr_t)d->x2 % 16) == 0 && ((uintptr_t)d->residual % 16) == 0);
  FX_CHECK_ARG(((uintptr_t)d->w1 % 16) == 0 && ((uintptr_t)d->w2 % 16) == 0 && ((uintptr_t)d->y1 % 16) == 0 && ((uintptr_t)d->y2 % 16) == 0);
  FX_CHECK_ARG(((uintptr_t)d->bias1 % 16) == 0 && ((uintptr_t)d->bias2 % 16) == 0);
  const int64_t x1_bytes = ((int64_t)d->M - 1) * d->ldx1 * 2 + (int64_t)d->K1a * 2;
  const int64_t x2_bytes = d->K1b ? ((int64_t)d->M - 1) * d->ldx2 * 2 + (int64_t)d->K1b * 2 : 0;
  const int64_t r_bytes = d->residual ? ((int64_t)d->M - 1) * d->ldr * 2 + (int64_t)d->N1 * 2 : 0;
  if (x1_bytes >= 0xFFFFFFF0ll || x2_bytes >= 0xFFFFFFF0ll || r_bytes >= 0xFFFFFFF0ll) return FX_ERR_UNSUPPORTED;
  PwChainArgs a;
  a.x1 = reinterpret_cast<const bf16_t*>(d->x1);
  a.x2 = reinterpret_cast<const bf16_t*>(d->x2);
  a.res = reinterpret_cast<const bf16_t*>(d->residual);
  a.w1p = reinterpret_cast<const bf16_t*>(d->w1);
  a.w2p = reinterpret_cast<const bf16_t*>(d->w2);
  a.b1 = d->bias1;
  a.b2 = d->bias2;
  a.y1 = reinterpret_cast<bf16_t*>(d->y1);
  a.y2 = reinterpret_cast<bf16_t*>(d->y2);
  a.ldx1 = d->ldx1; a.ldx2 = d->ldx2; a.ldr = d->ldr; a.ldy1 = d->ldy1; a.ldy2 = d->ldy2;
  a.M = d->M; a.N1 = d->N1; a.act1 = d->act1; a.act2 = d->act2;
  a.x1_bytes = (unsigned)x1_bytes; a.x2_bytes = (unsigned)x2_bytes; a.r_bytes = (unsigned)r_bytes;
  a.pool = reinterpret_cast<bf16_t*>(d->pool); a.ldp = d->ldp; a.H = d->img_h; a.W = d->img_w;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (d->pool) {
    FX_CHECK_ARG(d->img_h > 0 && d->img_w > 0 && d->img_h % 2 == 0 && d->img_w % 2 == 0 && d->M % (d->img_h * d->img_w) == 0);
    FX_CHECK_ARG(d->ldp >= d->N1 && d->ldp % 8 == 0 && ((uintptr_t)d->pool % 16) == 0);
    if (!fx_pw_chain_pool_supported(d->K1a, d->K1b, d->N1, d->N2)) return FX_ERR_UNSUPPORTED;
#define FX_PW_CASE(KA, KB, NN2) \
  if (d->K1a == KA && d->K1b == KB && d->N2 == NN2) return launch_pw_chain<KA, KB, NN2, true>(a, stream);
    FX_PW_POOL_INSTANCES(FX_PW_CASE)
#undef FX_PW_CASE
    return FX_ERR_UNSUPPORTED;
  }
#define FX_PW_CASE(KA, KB, NN2) \
  if (d->K1a == KA && d->K1b == KB && d->N2 == NN2) return launch_pw_chain<KA, KB, NN2>(a, stream);
  FX_PW_INSTANCES(FX_PW_CASE)
#undef FX_PW_CASE
  return FX_ERR_UNSUPPORTED;
}
