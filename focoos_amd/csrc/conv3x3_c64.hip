// 3x3 / stride 1 / pad 1 convolution with 64 input and 64 output channels (+ bias, optional ReLU, no residual) - the branch2b layers of res2
// (ResNet-vd BottleNeck, focoos/nn/backbone/resnet.py:72-121) at 160 x 160 (RT-DETR) / 200 x 200 (MaskFormer) - round 5, gfx950.
//
// Built like the round-5 stem kernels (stem_pool.hip, stem12.hip): 2-D tiles with everything a tile needs in LDS, no register ring, no asm, the
// compiler schedules.  The whole filter is 72 KiB, so it is LDS-RESIDENT next to a 16-row x 32-column tile of the input (18 x 34 positions x 64
// channels in the k-plane layout [plane = half * 4 + j][position][8 channels], zeros outside the image by OOB LDS-DMA: no border masks in the K loop):
// 149 KiB, one workgroup of eight waves per CU.  A wave owns two tile rows x 32 columns x both 32-channel blocks (2 x 2 accumulator blocks): per
// k-step two weight fragments and two pixel fragments feed four MFMAs - 1.0 LDS reads per MFMA, the CU's LDS port and its matrix pipes in balance -
// software-pipelined one k-step ahead.  The round-3 form of this layer (conv3x3_kplane<2,2,1,4,608,...,LD=0>: 256-pixel flat tiles, weights from L2
// through a register ring per wave) runs at 2 x the layer's byte bound (50 us per 16-image part for 105 MB); every wave of every workgroup streams the
// 72 KiB filter from L2 again.
// Accumulation order per output: bias, then k-steps 0..35 (k = tap * 64 + channel) - the same as the k-plane kernel's, so results are bit-identical.
#include "pw_common.h"

struct C64Args {
  const bf16_t* x;
  const bf16_t* wp;     // fragment order [2 channel blocks][36 k-steps][64 lanes][8]
  const float* bias;
  bf16_t* y;
  int H, W, ldx, ldy;
  int nbands, nstrips;
  unsigned x_bytes;
};

#define C64_TW 34
#define C64_TH 18
#define C64_POS (C64_TW * C64_TH)            // 612
#define C64_PLANE (C64_POS * 16)
#define C64_NDMA ((8 * C64_POS + 63) / 64)   // 77 instructions over the 4896 (plane, position) slots
#define C64_WOFF (C64_NDMA * 1024)           // 78 848
#define C64_SMEM (C64_WOFF + 72 * 1024)      // 152 576 bytes

template <int ACT>
__global__ __launch_bounds__(512, 2) void conv3x3_c64_kernel(const C64Args p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, half = lane >> 5;
  int bid = fx_xcd_remap(blockIdx.x, gridDim.x);
  const int strip = bid % p.nstrips;
  bid /= p.nstrips;
  const int band = bid % p.nbands, b = bid / p.nbands;
  const int Y0 = 16 * band - 1, X0 = 32 * strip - 1;   // image coordinates of tile position (0, 0)

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, 72 * 1024, 0x00020000);
  // ---- input tile first (its misses go to HBM), then the filter (L2-resident): slot S = plane * 612 + position, lane = slot;
  // plane pl holds piece c = (pl & 3) * 2 + (pl >> 2) of a pixel's 128 bytes (plane = half * 4 + j <-> channels 16 j + 8 half .. + 7)
  for (int i = wave; i < C64_NDMA; i += 8) {
    const int S = i * 64 + lane;
    const int pl = S / C64_POS, t = S - pl * C64_POS;
    const int c = (pl & 3) * 2 + (pl >> 2);
    const int ty = t / C64_TW, tx = t - ty * C64_TW;
    const int gy = Y0 + ty, gx = X0 + tx;
    const bool ok = pl < 8 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    pw_dma16(xr, smem + i * 1024, ok ? (unsigned)(((b * p.H + gy) * p.W + gx) * p.ldx + c * 8) * 2u : FX_OOB);
  }
  for (int i = wave; i < 72; i += 8) pw_dma16(wr, smem + C64_WOFF + i * 1024, (unsigned)(i * 1024 + lane * 16));

  const int lds0 = (int)(unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
  int row0[2];   // LDS address of (tile row 2 wave + 1 + bb, column l32 + 1) in plane half * 4
#pragma unroll
  for (int bb = 0; bb < 2; ++bb) row0[bb] = lds0 + half * 4 * C64_PLANE + ((2 * wave + 1 + bb) * C64_TW + (l32 + 1)) * 16;
  const int waddr = lds0 + C64_WOFF + lane * 16;
  f32x16 acc[2][2];   // [channel block][row]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const float4 bb4 = p.bias ? *reinterpret_cast<const float4*>(p.bias + a * 32 + 8 * gq + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        acc[a][bb][4 * gq] = bb4.x; acc[a][bb][4 * gq + 1] = bb4.y; acc[a][bb][4 * gq + 2] = bb4.z; acc[a][bb][4 * gq + 3] = bb4.w;
      }
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  typedef __attribute__((address_space(3))) const bf16x8 lds_frag_t;
  // k-step s = tap * 4 + j: channels 16 j .. 16 j + 15 of tap (dy, dx); fragments of step s + 1 requested before the MFMAs of step s
  auto toff = [&](int s) { return (((s >> 2) / 3 - 1) * C64_TW + ((s >> 2) % 3 - 1)) * 16 + (s & 3) * C64_PLANE; };
  bf16x8 xb[2][2], af[2][2];
#pragma unroll
  for (int bb = 0; bb < 2; ++bb) xb[0][bb] = *reinterpret_cast<lds_frag_t*>((size_t)(unsigned)(row0[bb] + toff(0)));
  af[0][0] = *reinterpret_cast<lds_frag_t*>((size_t)(unsigned)(waddr));
  af[0][1] = *reinterpret_cast<lds_frag_t*>((size_t)(unsigned)(waddr + 36 * 1024));
#pragma unroll
  for (int s = 0; s < 36; ++s) {
    const int cur = s & 1, nxt = cur ^ 1;
    if (s + 1 < 36) {
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) xb[nxt][bb] = *reinterpret_cast<lds_frag_t*>((size_t)(unsigned)(row0[bb] + toff(s + 1)));
      af[nxt][0] = *reinterpret_cast<lds_frag_t*>((size_t)(unsigned)(waddr + (s + 1) * 1024));
      af[nxt][1] = *reinterpret_cast<lds_frag_t*>((size_t)(unsigned)(waddr + (36 + s + 1) * 1024));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
      acc[0][bb] = FX_MFMA_32x32x16(af[cur][0], xb[cur][bb], acc[0][bb]);
      acc[1][bb] = FX_MFMA_32x32x16(af[cur][1], xb[cur][bb], acc[1][bb]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // ---- epilogue: activation, v_permlane32_swap pairs -> 16 contiguous bytes per lane, row stores
  const int xc = 32 * strip + l32;
#pragma unroll
  for (int bb = 0; bb < 2; ++bb) {
    const int yr = 16 * band + 2 * wave + bb;
    const bool live = yr < p.H && xc < p.W;
    bf16_t* yrow = p.y + ((size_t)(b * p.H + yr) * p.W + xc) * p.ldy + half * 8;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int g2 = 0; g2 < 2; ++g2) {
        unsigned q[2][2];
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = acc[a][bb][4 * (2 * g2 + q2) + e];
            if constexpr (ACT == FX_ACT_RELU) v[e] = fmaxf(v[e], 0.0f);
          }
          q[q2][0] = pack_bf16x2(v[0], v[1]);
          q[q2][1] = pack_bf16x2(v[2], v[3]);
        }
#pragma unroll
        for (int w2 = 0; w2 < 2; ++w2) {
          const auto sw = __builtin_amdgcn_permlane32_swap(q[0][w2], q[1][w2], false, false);
          q[0][w2] = sw[0];
          q[1][w2] = sw[1];
        }
        if (live) *reinterpret_cast<uint4*>(yrow + a * 32 + g2 * 16) = make_uint4(q[0][0], q[0][1], q[1][0], q[1][1]);
      }
  }
}

// C = N = 64, 3x3 / s1 / p1, no residual, ReLU or no activation, contiguous batch
bool fx_conv3x3_c64_supported(int C, int N, int mode) {
  static const int on = fx_tune("FX_C3_C64", 0);   // OFF: measured 55-57 us per res2 layer against 52 us on conv3x3_kplane<2,2,1,4,608,LD=0> (profiles/r05_c64_ab.txt): one 149 KiB workgroup per CU = two waves per SIMD and a 152 KiB fetch in front of every tile
  return on && C == 64 && N == 64 && (mode == 0 || mode == 3);
}

int fx_launch_conv3x3_c64(const ConvArgs& c, const bf16_t* w_frag, hipStream_t stream) {
  const int mode = fx_c3_epilogue_mode(c.act, c.res != nullptr, c.res_after);
  if (!fx_conv3x3_c64_supported(c.C, c.N, mode) || c.y_bstride || c.stride != 1 || c.KH != 3 || c.KW != 3 || c.pad != 1) return FX_ERR_UNSUPPORTED;
  C64Args a{};
  a.x = c.x; a.wp = w_frag; a.bias = c.bias; a.y = reinterpret_cast<bf16_t*>(c.y);
  a.H = c.H; a.W = c.W; a.ldx = c.ldx; a.ldy = c.ldy; a.x_bytes = c.x_bytes;
  a.nbands = (c.H + 15) / 16;
  a.nstrips = (c.W + 31) / 32;
  const int64_t grid = (int64_t)c.B * a.nbands * a.nstrips;
  if (grid >= (1ll << 31)) return FX_ERR_UNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c64_kernel<FX_ACT_RELU>), hipFuncAttributeMaxDynamicSharedMemorySize, C64_SMEM) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c64_kernel<FX_ACT_NONE>), hipFuncAttributeMaxDynamicSharedMemorySize, C64_SMEM) != hipSuccess)
      return FX_ERR_RUNTIME;
    attr_set = true;
  }
  if (mode == 0) hipLaunchKernelGGL(conv3x3_c64_kernel<FX_ACT_RELU>, dim3((int)grid), dim3(512), C64_SMEM, stream, a);
  else hipLaunchKernelGGL(conv3x3_c64_kernel<FX_ACT_NONE>, dim3((int)grid), dim3(512), C64_SMEM, stream, a);
  return fx_launch_status();
}
