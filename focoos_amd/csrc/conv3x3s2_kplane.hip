// 3x3 / STRIDE 2 / pad 1 convolution on the k-plane machinery of conv3x3_kplane.hip (gfx950, round 4).
//
// The three stride-2 layers of a ResNet-vd (branch2b of res3/res4/res5 block 0: focoos/nn/backbone/resnet.py:72-121, stride on
// branch2b in variant d) ran on the implicit-GEMM tiles at 305-455 TFLOP/s (profiles/r03z_per_op_hipevent.txt: 80 + 66 + 99 us per
// 16-image part) because an im2col row of a stride-2 layer has no contiguous halo range to share between taps.  It has one in
// the SPACE-TO-DEPTH view of the input: with the four parity planes P[py][px](i, j) = x(2i + py, 2j + px), each of size Ho x Wo,
//     y(oy, ox) = sum over taps (dy, dx) of  W[dy][dx] . P[py][px](oy + ry, ox + rx),
//     dy = 0 -> (py = 1, ry = -1), dy = 1 -> (py = 0, ry = 0), dy = 2 -> (py = 1, ry = 0)      (the same for dx),
// i.e. a stride-1 correlation over the flat output-pixel range with tap offsets in {-Wo-1, -Wo, -1, 0} - the flat-range + halo scheme
// of the stride-1 kernel with a SMALLER halo (Wo + 1 rows on one side only).  Nothing is re-laid-out in memory: a K chunk of this
// kernel is (parity plane, 64 input channels), the loader's LDS-DMA gathers the chunk's rows from the pixels (2i + py, 2j + px) of the
// NHWC tensor - 128 contiguous bytes per pixel, exactly one cache line per (pixel, chunk), every line of the input fetched once
// (+ halo) - and the chunk's tap list is that of its plane: 1, 2, 2 or 4 taps (9 per channel chunk in total: no wasted MFMA work).
//
// What differs from the stride-1 kernel:
//   * a chunk feeds only 2.25 taps on average (9 for stride 1), so the loader must deliver a chunk per ~2 300 MFMA cycles instead
//     of per ~9 200: THREE halo buffers (the loader runs two chunks ahead; counted vmcnt waits - its DMAs retire in order) instead
//     of two, so that the 1-tap chunks do not stall on their successor's fetch;
//   * per-lane source offsets of the halo rows (row -> image, oy, ox: two integer divisions) are computed once per launch, the
//     chunk only adds its plane's offset and channel base;
//   * border masks: only the taps with ry = -1 / rx = -1 can leave the image (top / left); 4 mask bits per pixel block;
//   * no residual form (branch2b has none): always the direct accumulator -> memory epilogue.
// Measured (profiles/r04_s2_kernel.txt): 80 / 66 / 99 us -> 62 / 47 / 58 us per 16-image part for the res3 / res4 / res5 layer; without the chunk
// DMA the same launches take 51 / 40 / 45 us and the stride-1 kernel needs 43 / 32 / 37 us for the same FLOPs: a (plane, 64-channel) chunk
// is a quarter of the stride-1 chunk's work behind each barrier, and a second loader wave (NLD = 2) does not change the time.
// Weights: the same fragment-ordered copy as the stride-1 kernel (k = tap * C + channel), read through the 4-slot register ring.
#include <type_traits>

#include "pw_common.h"

struct C3S2Args {
  const bf16_t* x;
  const bf16_t* wp;
  const float* bias;
  bf16_t* y;
  int Ho, Wo, C, N, ldx, ldy, M;   // M = B * Ho * Wo (output pixels); the input is [B, 2Ho, 2Wo, C] with pixel stride ldx
  int HLp;            // halo rows of this launch (BM + Wo + 1) rounded up to a multiple of 32 (<= HLP)
  int HW;             // Ho * Wo
  int64_t y_bstride;  // elements between images of y (0: contiguous)
  unsigned x_bytes;
};

template <int OFF>
__device__ __forceinline__ void s2k_ldg(bf16x8& dst, unsigned voff, const bf16_t* sbase) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "+v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
typedef __attribute__((address_space(3))) const bf16x8 s2k_lds_frag_t;
template <int IMM>
__device__ __forceinline__ bf16x8 s2k_lds_read(int addr) {
  return *reinterpret_cast<s2k_lds_frag_t*>((size_t)(unsigned)(addr + IMM));
}
__device__ __forceinline__ float s2k_act(float v, std::integral_constant<int, FX_ACT_RELU>) { return fmaxf(v, 0.0f); }
__device__ __forceinline__ float s2k_act(float v, std::integral_constant<int, FX_ACT_SILU>) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ float s2k_act(float v, std::integral_constant<int, FX_ACT_NONE>) { return v; }

// tap `ti` of parity plane (py, px): original filter tap t9 = dy * 3 + dx, LDS row offset (ry * Wo + rx) in bytes, border-mask bit
struct S2Tap { int t9, off, bit; };
__device__ __forceinline__ S2Tap s2k_tap(int py, int px, int ti, int Wo) {
  const int iy = px ? (ti >> 1) : ti, ix = px ? (ti & 1) : 0;
  const int ry = py ? iy - 1 : 0, rx = px ? ix - 1 : 0;
  const int dy = py ? 2 * iy : 1, dx = px ? 2 * ix : 1;
  S2Tap t;
  t.t9 = dy * 3 + dx;
  t.off = (ry * Wo + rx) * 16;
  t.bit = 1 << ((ry + 1) * 2 + (rx + 1));
  return t;
}

template <int N>
__device__ __forceinline__ void s2k_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// TN x TM 32x32 accumulator blocks per wave, WN x WM consumer waves + the loader wave; HLP rows per plane (multiple of 64).
template <int TN, int TM, int WN, int WM, int HLP, int ACT, int NLD>
__global__ __launch_bounds__((WN* WM + NLD) * 64, 1) void conv3x3s2_kplane_kernel(const C3S2Args p) {
  constexpr int NW = WN * WM;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int KJ = 4, PF = KJ;
  constexpr int PLANE = (HLP + 1) * 16, BUF = 8 * PLANE, NBUF = 3;
  constexpr int NBLK = HLP / 64;
  static_assert(TN <= 2 && HLP % 64 == 0 && 3 * PLANE < 65536 && NBLK * 8 <= 63, "k-step offsets are 16-bit immediates; vmcnt is 6 bits");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_loader = wave >= NW;   // NLD loader waves: loader l fetches the pieces c = l, l + NLD, ... of every row block
  const int l32 = lane & 31, half = lane >> 5;
  const int wn = wave % WN, wm = wave / WN;
  const int nNt = p.N / BN;
  const int bid = fx_xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / nNt) * BM, n0 = (bid % nNt) * BN;
  const int lo = m0 - p.Wo - 1;
  const int NCHC = p.C >> 6, NCH = 4 * NCHC;   // chunk cc = (channel chunk cc >> 2, parity plane cc & 3): taps 1, 2, 2, 4
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const int nact = (p.HLp + 63) >> 6;          // 64-row DMA blocks in use

  // element offset of the plane-(0,0) pixel behind halo row r = blk * 64 + lane (row r <-> flat s2d pixel lo + r), or -1
  int so[NBLK];
#pragma unroll
  for (int blk = 0; blk < NBLK; ++blk) {
    const int r = blk * 64 + lane;
    const int f = lo + r;
    const bool ok = f >= 0 && f < p.M && r < p.HLp;
    const int ff = ok ? f : 0;
    const int bi = ff / p.HW, rem = ff - bi * p.HW;
    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    so[blk] = ok ? ((bi * 2 * p.Ho + 2 * oy) * (2 * p.Wo) + 2 * ox) * p.ldx : -1;
  }
  // pieces c = first, first + step, ... of every active row block of chunk cc -> buf
  auto dma_chunk = [&](int cc, unsigned char* buf, int first, int step) {
    const int plane = cc & 3, cch = cc >> 2;
    const int poff = ((plane >> 1) * 2 * p.Wo + (plane & 1)) * p.ldx + cch * 64;
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
      if (blk < nact) {
        for (int c = first; c < 8; c += step) {
          const int pln = (c & 1) * 4 + (c >> 1);   // piece c = channels [8c, 8c+8) = k-step c/2, half c%2
          if (blk * 64 + lane < p.HLp) pw_dma16(xr, buf + pln * PLANE + blk * 1024, so[blk] >= 0 ? (unsigned)(so[blk] + poff + c * 8) * 2u : FX_OOB);
        }
      }
    }
  };

  if (is_loader) {
    // the zero row of every plane of the three buffers (published by the first barrier; the DMA never touches row HLP)
    const int ld = wave - NW;
    if (ld == 0 && lane < 24) *reinterpret_cast<uint4*>(smem + (lane >> 3) * BUF + (lane & 7) * PLANE + HLP * 16) = make_uint4(0, 0, 0, 0);
    if (NCH > 1) dma_chunk(1, smem + BUF, ld, NLD);
    int bsel = 2;   // buffer of chunk cc + 2
    for (int cc = 0; cc < NCH; ++cc) {
      // this loader's share of chunk cc must have landed; its share of chunk cc + 1 (younger, nact * 8 / NLD instructions) may be in flight
      if (cc + 1 < NCH) {
        switch (nact) {
          case 1: s2k_vmcnt<8 / NLD>(); break;
          case 2: s2k_vmcnt<16 / NLD>(); break;
          case 3: s2k_vmcnt<24 / NLD>(); break;
          case 4: s2k_vmcnt<32 / NLD>(); break;
          case 5: s2k_vmcnt<40 / NLD>(); break;
          case 6: s2k_vmcnt<48 / NLD>(); break;
          default: s2k_vmcnt<0>(); break;
        }
      } else {
        s2k_vmcnt<0>();
      }
      __syncthreads();  // chunk cc is visible; every consumer is done with chunk cc - 1, whose buffer chunk cc + 2 reuses
      if (cc + 2 < NCH) dma_chunk(cc + 2, smem + bsel * BUF, ld, NLD);
      bsel = bsel == 2 ? 0 : bsel + 1;
    }
  } else {
    f32x16 acc[TN][TM];
    bf16x8 ar[PF][TN];
    unsigned mask4[TM];
    int row0[TM];      // byte address of the lane's piece of pixel block b at tap offset 0, k-step 0, buffer 0
    const int Cs = p.C >> 4;
    const unsigned wvoff = lane * 16;
    const bf16_t* wbase[TN];
#pragma unroll
    for (int a = 0; a < TN; ++a) wbase[a] = p.wp + (size_t)((n0 >> 5) + wn * TN + a) * (size_t)(9 * Cs) * 512;
    auto w_ptr = [&](int a, int cch, int t9) -> const bf16_t* { return wbase[a] + (size_t)(t9 * Cs + cch * KJ) * 512; };
#pragma unroll
    for (int a = 0; a < TN; ++a) {
      const bf16_t* w0 = w_ptr(a, 0, 4);   // chunk 0 = plane (0,0): its one tap is the centre tap (dy, dx) = (1, 1)
      c3_static_for<PF>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        ar[i][a] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        s2k_ldg<i * 1024>(ar[i][a], wvoff, w0);
      });
    }
    dma_chunk(0, smem, wave, NW);   // this wave's share of the first chunk
    const int lds0 = (int)(unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
#pragma unroll
    for (int b = 0; b < TM; ++b) {
      const int pl = (wm * TM + b) * 32 + l32;
      const int m = m0 + pl;
      row0[b] = lds0 + (pl + p.Wo + 1) * 16 + half * 4 * PLANE;
      const bool ok = m < p.M;
      const int mm = ok ? m : 0;
      const int rem = mm % p.HW;
      const int yy = rem / p.Wo, xx = rem - yy * p.Wo;
      unsigned msk = 0;
      if (ok) msk = 8u | (xx >= 1 ? 4u : 0u) | (yy >= 1 ? 2u : 0u) | ((yy >= 1 && xx >= 1) ? 1u : 0u);
      mask4[b] = msk;
    }
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const float4 bb = p.bias ? *reinterpret_cast<const float4*>(p.bias + n0 + (wn * TN + a) * 32 + 8 * gq + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int b = 0; b < TM; ++b) {
          acc[a][b][4 * gq] = bb.x; acc[a][b][4 * gq + 1] = bb.y; acc[a][b][4 * gq + 2] = bb.z; acc[a][b][4 * gq + 3] = bb.w;
        }
      }
    const int zhalf = lds0 + half * 4 * PLANE + HLP * 16;   // the lane's zero row (buffer 0)

    int bsel = 0;
    for (int cc = 0; cc < NCH; ++cc) {
      if (cc == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the first chunk (and the ring)
      __syncthreads();  // chunk cc has landed; every consumer is done with chunk cc - 1
      const int bufo = bsel * BUF;
      bsel = bsel == 2 ? 0 : bsel + 1;
      const int plane = cc & 3, cch = cc >> 2;
      const int py = plane >> 1, px = plane & 1;
      const int ntap = (1 + py) * (1 + px);
      const int ccn = cc + 1 < NCH ? cc + 1 : cc;
      const int pyn = (ccn & 3) >> 1, pxn = ccn & 1, cchn = ccn >> 2;
      const int t9n_first = s2k_tap(pyn, pxn, 0, p.Wo).t9;
      const int zaddr = zhalf + bufo;
      int addr[TM], addrn[TM];
      {
        const S2Tap t0 = s2k_tap(py, px, 0, p.Wo);
#pragma unroll
        for (int b = 0; b < TM; ++b) {
          addr[b] = (mask4[b] & (unsigned)t0.bit) ? row0[b] + bufo + t0.off : zaddr;
          addrn[b] = addr[b];
        }
      }
      bf16x8 xb[2][TM];
#pragma unroll
      for (int b = 0; b < TM; ++b) xb[0][b] = s2k_lds_read<0>(addr[b]);
#pragma unroll 1
      for (int ti = 0; ti < ntap; ++ti) {
        const bool last_tap = ti == ntap - 1;
        // scalar state of the NEXT tap (behind the chunk's last tap: its tap 0 again - harmless unused fragment reads, as in the
        // stride-1 kernel: the next chunk re-reads after its barrier)
        const S2Tap tn = s2k_tap(py, px, last_tap ? 0 : ti + 1, p.Wo);
        const int offn = bufo + tn.off;
        const unsigned bitn = (unsigned)tn.bit;
        const bf16_t* wnext[TN];
#pragma unroll
        for (int a = 0; a < TN; ++a) wnext[a] = last_tap ? w_ptr(a, cchn, t9n_first) : w_ptr(a, cch, tn.t9);
        auto kstep = [&](auto jc) {
          constexpr int j = decltype(jc)::value;
          if constexpr (TN == 1) c3_wait<(KJ - 1) * TN>(ar[j][0]); else c3_wait<(KJ - 1) * TN>(ar[j][0], ar[j][1]);
#pragma unroll
          for (int a = 0; a < TN; ++a) {
#pragma unroll
            for (int b = 0; b < TM; ++b) {
              acc[a][b] = FX_MFMA_32x32x16(ar[j][a], xb[j & 1][b], acc[a][b]);
              if (a == 0) {   // fragment b of the next k-step (k-step 0 of the next tap behind the last one)
                if constexpr (j + 1 < KJ) xb[(j + 1) & 1][b] = s2k_lds_read<(j + 1) * PLANE>(addr[b]);
                else xb[0][b] = s2k_lds_read<0>(addrn[b]);
              }
              if constexpr (j == 0) {
                if (a == TN - 1) addrn[b] = (mask4[b] & bitn) ? row0[b] + offn : zaddr;
              }
              if (b == TM - 1) s2k_ldg<j * 1024>(ar[j][a], wvoff, wnext[a]);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        };
        c3_static_for<KJ>(kstep);
#pragma unroll
        for (int b = 0; b < TM; ++b) addr[b] = addrn[b];
      }
    }
    // drain the hidden loads before their registers are reused by the epilogue
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      if constexpr (TN == 1) c3_wait<0>(ar[i][0]); else c3_wait<0>(ar[i][0], ar[i][1]);
    }
    // straight from the accumulators to memory (conv3x3_kplane.hip, RESMODE 0): v_permlane32_swap pairs -> 16-byte row stores
#pragma unroll
    for (int b = 0; b < TM; ++b) {
      const int m = m0 + (wm * TM + b) * 32 + l32;
      size_t yo = (size_t)m * p.ldy;
      if (p.y_bstride) {
        const int bb = m / p.HW;
        yo = (size_t)bb * p.y_bstride + (size_t)(m - bb * p.HW) * p.ldy;
      }
      bf16_t* yrow = p.y + yo + n0 + wn * TN * 32 + half * 8;
      const bool live = m < p.M;
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          unsigned pk[2][2];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = s2k_act(acc[a][b][4 * (2 * g + q) + e], std::integral_constant<int, ACT>{});
            pk[q][0] = pack_bf16x2(v[0], v[1]);
            pk[q][1] = pack_bf16x2(v[2], v[3]);
          }
#pragma unroll
          for (int w = 0; w < 2; ++w) {   // lanes 32-63 of pk[0] <-> lanes 0-31 of pk[1]
            const auto sw = __builtin_amdgcn_permlane32_swap(pk[0][w], pk[1][w], false, false);
            pk[0][w] = sw[0];
            pk[1][w] = sw[1];
          }
          if (live) *reinterpret_cast<uint4*>(yrow + a * 32 + g * 16) = make_uint4(pk[0][0], pk[0][1], pk[1][0], pk[1][1]);
        }
    }
  }
}

template <int TN, int TM, int WN, int WM, int HLP, int ACT, int NLD>
static int launch_c3s2(C3S2Args& a, hipStream_t stream) {
  constexpr int NW = WN * WM, BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int PLANE = (HLP + 1) * 16, BUF = 8 * PLANE;
  const int HL = BM + a.Wo + 1;
  a.HLp = (HL + 31) / 32 * 32;
  if (a.HLp > HLP || a.C % 64 != 0 || a.N % BN != 0) return FX_ERR_UNSUPPORTED;
  const int smem = 3 * BUF;
  if (smem > 160 * 1024) return FX_ERR_UNSUPPORTED;
  auto kern = conv3x3s2_kplane_kernel<TN, TM, WN, WM, HLP, ACT, NLD>;
  static int attr_smem = 0;
  if (smem > attr_smem) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) return FX_ERR_RUNTIME;
    attr_smem = smem;
  }
  hipLaunchKernelGGL(kern, dim3(((a.M + BM - 1) / BM) * (a.N / BN)), dim3((NW + NLD) * 64), smem, stream, a);
  return fx_launch_status();
}

// Tiles as in the stride-1 kernel: N % 256 == 0: 128 pixels x 256 channels (64 pixels for small M: twice the workgroups for the
// 20x20-level layer), N = 128: 256 x 128.  Returns the plane height HLP of the instance that covers (C, N, Wo, M), 0: none.
static int c3s2_plan(int C, int N, int Wo, int M) {
  if (C % 64 != 0 || Wo < 1) return 0;
  static const int small_thr = fx_tune("FX_C3K_SMALL_M", 16000);
  if (N > 0 && N % 256 == 0) {
    if (M <= small_thr && 64 + Wo + 1 <= 128) return 128;
    if (128 + Wo + 1 <= 192) return 192;
    if (128 + Wo + 1 <= 256) return 256;
    return 0;
  }
  if (N == 128 && 256 + Wo + 1 <= 384) return 384;
  return 0;
}

bool fx_conv3x3s2_kplane_supported(int C, int N, int Wo, int M) { return c3s2_plan(C, N, Wo, M) != 0; }

int fx_launch_conv3x3s2_kplane(const ConvArgs& c, const bf16_t* w_frag, hipStream_t stream) {
  const int hlp = c3s2_plan(c.C, c.N, c.Wo, c.M);
  if (!hlp || c.res || c.H != 2 * c.Ho || c.W != 2 * c.Wo || c.stride != 2 || c.KH != 3 || c.KW != 3 || c.pad != 1) return FX_ERR_UNSUPPORTED;
  C3S2Args a{};
  a.x = c.x; a.wp = w_frag; a.bias = c.bias; a.y = reinterpret_cast<bf16_t*>(c.y);
  a.Ho = c.Ho; a.Wo = c.Wo; a.C = c.C; a.N = c.N; a.ldx = c.ldx; a.ldy = c.ldy; a.M = c.M;
  a.HW = c.Ho * c.Wo; a.y_bstride = c.y_bstride; a.x_bytes = c.x_bytes;
  const int mode = fx_c3_epilogue_mode(c.act, false, 0);
#define FX_C3S2_TILE(ACT_)                                                             \
  {                                                                                     \
    if (c.N == 128) return launch_c3s2<2, 4, 2, 2, 384, ACT_, 1>(a, stream);            \
    if (hlp == 128) return launch_c3s2<2, 2, 4, 1, 128, ACT_, 1>(a, stream);            \
    if (hlp == 192) return launch_c3s2<2, 4, 4, 1, 192, ACT_, 1>(a, stream);            \
    return launch_c3s2<2, 4, 4, 1, 256, ACT_, 1>(a, stream);                            \
  }
  switch (mode) {
    case 0: FX_C3S2_TILE(FX_ACT_RELU)
    case 1: FX_C3S2_TILE(FX_ACT_SILU)
    case 3: FX_C3S2_TILE(FX_ACT_NONE)
    default: return FX_ERR_UNSUPPORTED;
  }
#undef FX_C3S2_TILE
}
