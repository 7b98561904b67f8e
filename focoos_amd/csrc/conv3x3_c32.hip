// 3x3 / stride 1 / pad 1 convolution with 32 INPUT channels (gfx950, round 3): the ResNet-vd stem layers conv1_2 (32 -> 32) and conv1_3
// (32 -> 64) at 320 x 320 - 1.6 M output pixels per 16-image part, 288-deep reduction, HBM-bound layers (105 + 105 / 210 MB per part
// against 30 / 60 GFLOP) that the implicit-GEMM kernel ran at 2x their traffic floor (nine L2 -> LDS passes over the same pixels).
// Same scheme as conv3x3_kplane.hip - flat pixel range + halo fetched ONCE, k-plane LDS layout with immediate-offset fragment reads,
// weights from L2 through a register ring, direct 16-byte row stores - cut for K = 288:
//   * a tap is 32 channels = 2 k-steps; the ring holds 6 fragments per weight block = 3 taps, so the 9 taps are 3 ring cycles and the 18
//     fragments of a weight block are ONE contiguous 18 KiB stream (k = tap * 32 + channel);
//   * no loader wave: the whole reduction is one chunk, so the four waves fetch the halo themselves and the workgroup is 4 waves at
//     <= 256 registers - two workgroups share a CU (2 x 78 KiB of LDS at W = 320) and one's halo fetch / store phase runs beside the
//     other's MFMAs: the layer is a stream of 512-pixel tiles whose 2-5 k cycles of MFMA work hide behind the memory phases.
// Tile: 512 pixels x N channels (N = 32: one weight block, N = 64: two), 4 waves over the pixels (4 x 4 pixel blocks of 32).
#include <type_traits>

#include "pw_common.h"

struct C32Args {
  const bf16_t* x;
  const bf16_t* wp;
  const float* bias;
  bf16_t* y;
  int H, W, ldx, ldy, M;
  int HLp;            // halo rows of this launch rounded up to whole 64-row DMA blocks (<= HLP)
  unsigned x_bytes;
};

template <int OFF>
__device__ __forceinline__ void c32_ldg(bf16x8& dst, unsigned voff, const bf16_t* sbase) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "+v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
typedef __attribute__((address_space(3))) const bf16x8 c32_lds_frag_t;
template <int IMM>
__device__ __forceinline__ bf16x8 c32_lds_read(int addr) {
  return *reinterpret_cast<c32_lds_frag_t*>((size_t)(unsigned)(addr + IMM));
}

template <int TN, int HLP, int ACT>
__global__ __launch_bounds__(256, 2) void conv3x3_c32_kernel(const C32Args p) {
  constexpr int TM = 4, NW = 4, BM = NW * TM * 32, BN = TN * 32;
  constexpr int KJ = 2, G = 3, PF = KJ * G;                    // k-steps per tap, taps per ring cycle, ring slots per weight block
  constexpr int PLANE = (HLP + 1) * 16;                         // plane = half * 2 + j; the zero row of a plane at row index HLP
  static_assert(HLP % 64 == 0 && PLANE < 65536, "k-step offsets are 16-bit immediates");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, half = lane >> 5;
  const int m0 = fx_xcd_remap(blockIdx.x, gridDim.x) * BM;
  const int lo = m0 - p.W - 1;

  // ---- prologue: the weight ring first (first touch: the longest way), then the halo, then what only needs registers
  bf16x8 ar[PF][TN];
  const unsigned wvoff = lane * 16;
  const bf16_t* wbase[TN];
#pragma unroll
  for (int a = 0; a < TN; ++a) {
    wbase[a] = p.wp + (size_t)a * 18 * 512;     // weight block a: 18 fragments of 512 elements
    c3_static_for<PF>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      ar[i][a] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      c32_ldg<(i & 3) * 1024>(ar[i][a], wvoff, wbase[a] + (i >> 2) * 2048);
    });
  }
  if (wave == 0 && lane < 4) *reinterpret_cast<uint4*>(smem + lane * PLANE + HLP * 16) = make_uint4(0, 0, 0, 0);   // zero rows
  {
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const int ninstr = (p.HLp >> 6) * 4;   // instruction i = (row block i / 4, 8-channel piece i % 4); lane = row
    for (int i = wave; i < ninstr; i += NW) {
      const int blk = i >> 2, c = i & 3;
      const int f = lo + blk * 64 + lane;
      const bool ok = f >= 0 && f < p.M;
      pw_dma16(xr, smem + ((c & 1) * 2 + (c >> 1)) * PLANE + blk * 1024, ok ? (unsigned)(f * p.ldx + c * 8) * 2u : FX_OOB);
    }
  }
  const int lds0 = (int)(unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
  const int HWp = p.H * p.W;
  unsigned mask9[TM];
  int row0[TM];
#pragma unroll
  for (int b = 0; b < TM; ++b) {
    const int pl = (wave * TM + b) * 32 + l32;
    const int m = m0 + pl;
    row0[b] = lds0 + (pl + p.W + 1) * 16 + half * 2 * PLANE;
    const bool ok = m < p.M;
    const int mm = ok ? m : 0;
    const int rem = mm % HWp;
    const int yy = rem / p.W, xx = rem - yy * p.W;
    unsigned msk = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int dy = t / 3 - 1, dx = t % 3 - 1;
      if (ok && (unsigned)(yy + dy) < (unsigned)p.H && (unsigned)(xx + dx) < (unsigned)p.W) msk |= 1u << t;
    }
    mask9[b] = msk;
  }
  const int zaddr = lds0 + half * 2 * PLANE + HLP * 16;
  f32x16 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const float4 bb = p.bias ? *reinterpret_cast<const float4*>(p.bias + a * 32 + 8 * gq + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        acc[a][b][4 * gq] = bb.x; acc[a][b][4 * gq + 1] = bb.y; acc[a][b][4 * gq + 2] = bb.z; acc[a][b][4 * gq + 3] = bb.w;
      }
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the halo (and the ring)
  __syncthreads();

  // ---- 3 ring cycles of 3 taps x 2 k-steps
  auto tap_off = [&](int t) { return ((t / 3 - 1) * p.W + (t % 3 - 1)) * 16; };
  int addr[G][TM], addrn[G][TM];
#pragma unroll
  for (int u = 0; u < G; ++u)
#pragma unroll
    for (int b = 0; b < TM; ++b) {
      addr[u][b] = (mask9[b] & (1u << u)) ? row0[b] + tap_off(u) : zaddr;
      addrn[u][b] = addr[u][b];
    }
  bf16x8 xb[2][TM];
#pragma unroll
  for (int b = 0; b < TM; ++b) xb[0][b] = c32_lds_read<0>(addr[0][b]);
#pragma unroll 1
  for (int g = 0; g < 3; ++g) {
    const int gn = g < 2 ? g + 1 : 0;    // taps of the NEXT cycle (behind the last one: cycle 0 again - reads that nothing uses)
    int offn[G];
    unsigned bitn[G];
#pragma unroll
    for (int u = 0; u < G; ++u) {
      offn[u] = tap_off(gn * G + u);
      bitn[u] = 1u << (gn * G + u);
    }
    const bf16_t* wnext[TN];
#pragma unroll
    for (int a = 0; a < TN; ++a) wnext[a] = wbase[a] + (size_t)(gn * PF) * 512;
    auto kstep = [&](auto sc) {
      constexpr int s = decltype(sc)::value;
      constexpr int u = s / KJ, j = s % KJ;
      if constexpr (TN == 1) c3_wait<(PF - 1) * TN>(ar[s][0]); else c3_wait<(PF - 1) * TN>(ar[s][0], ar[s][1]);
#pragma unroll
      for (int a = 0; a < TN; ++a) {
#pragma unroll
        for (int b = 0; b < TM; ++b) {
          acc[a][b] = FX_MFMA_32x32x16(ar[s][a], xb[s & 1][b], acc[a][b]);
          if (a == 0) {   // fragment b of the next k-step
            if constexpr (s + 1 < PF) xb[(s + 1) & 1][b] = c32_lds_read<((s + 1) % KJ) * PLANE>(addr[(s + 1) / KJ][b]);
            else xb[0][b] = c32_lds_read<0>(addrn[0][b]);
          }
          if constexpr (s < G) {   // address of block b at tap s of the next cycle
            if (a == TN - 1) addrn[s][b] = (mask9[b] & bitn[s]) ? row0[b] + offn[s] : zaddr;
          }
          if (b == TM - 1) c32_ldg<(s & 3) * 1024>(ar[s][a], wvoff, wnext[a] + (s >> 2) * 2048);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      (void)u; (void)j;
    };
    c3_static_for<PF>(kstep);
#pragma unroll
    for (int u = 0; u < G; ++u)
#pragma unroll
      for (int b = 0; b < TM; ++b) addr[u][b] = addrn[u][b];
  }
  // drain the ring (its last refills are unused re-reads of cycle 0)
#pragma unroll
  for (int i = 0; i < PF; ++i) {
    if constexpr (TN == 1) c3_wait<0>(ar[i][0]); else c3_wait<0>(ar[i][0], ar[i][1]);
  }
  // ---- epilogue: activation, v_permlane32_swap pairs -> 16 contiguous bytes per lane, row stores
#pragma unroll
  for (int b = 0; b < TM; ++b) {
    const int m = m0 + (wave * TM + b) * 32 + l32;
    bf16_t* yrow = p.y + (size_t)m * p.ldy + half * 8;
    const bool live = m < p.M;
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int g2 = 0; g2 < 2; ++g2) {
        unsigned pk[2][2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = acc[a][b][4 * (2 * g2 + q) + e];
            if constexpr (ACT == FX_ACT_RELU) v[e] = fmaxf(v[e], 0.0f);
          }
          pk[q][0] = pack_bf16x2(v[0], v[1]);
          pk[q][1] = pack_bf16x2(v[2], v[3]);
        }
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          const auto sw = __builtin_amdgcn_permlane32_swap(pk[0][w], pk[1][w], false, false);
          pk[0][w] = sw[0];
          pk[1][w] = sw[1];
        }
        if (live) *reinterpret_cast<uint4*>(yrow + a * 32 + g2 * 16) = make_uint4(pk[0][0], pk[0][1], pk[1][0], pk[1][1]);
      }
  }
}

template <int TN, int HLP, int ACT>
static int launch_c32(C32Args& a, hipStream_t stream) {
  constexpr int BM = 512, PLANE = (HLP + 1) * 16;
  const int smem = 4 * PLANE;
  auto kern = conv3x3_c32_kernel<TN, HLP, ACT>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) return FX_ERR_RUNTIME;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((a.M + BM - 1) / BM), dim3(256), smem, stream, a);
  return fx_launch_status();
}

// C = 32, N = 32 / 64, no residual, ReLU or no activation, contiguous batch; W up to 415 (HLP 1344: MaskFormer's 400-wide stem)
bool fx_conv3x3_c32_supported(int C, int N, int W, int mode) {
  return C == 32 && (N == 32 || N == 64) && (mode == 0 || mode == 3) && 512 + 2 * W + 2 <= 1344;
}

int fx_launch_conv3x3_c32(const ConvArgs& c, const bf16_t* w_frag, hipStream_t stream) {
  const int mode = fx_c3_epilogue_mode(c.act, c.res != nullptr, c.res_after);
  if (!fx_conv3x3_c32_supported(c.C, c.N, c.W, mode) || c.y_bstride) return FX_ERR_UNSUPPORTED;
  C32Args a{};
  a.x = c.x; a.wp = w_frag; a.bias = c.bias; a.y = reinterpret_cast<bf16_t*>(c.y);
  a.H = c.H; a.W = c.W; a.ldx = c.ldx; a.ldy = c.ldy; a.M = c.M; a.x_bytes = c.x_bytes;
  a.HLp = (512 + 2 * c.W + 2 + 63) / 64 * 64;
  const bool small = a.HLp <= 1216;
#define FX_C32(TN_, ACT_) return small ? launch_c32<TN_, 1216, ACT_>(a, stream) : launch_c32<TN_, 1344, ACT_>(a, stream)
  if (c.N == 32) {
    if (mode == 0) FX_C32(1, FX_ACT_RELU);
    FX_C32(1, FX_ACT_NONE);
  }
  if (mode == 0) FX_C32(2, FX_ACT_RELU);
  FX_C32(2, FX_ACT_NONE);
#undef FX_C32
}
