// EXPERIMENT (measured negative on MI355X, not built into libfocoos_amd.so; kept as a probe): an LDS-free, register-resident-W
// streaming kernel for 1x1 layers with K <= 256.  Correct (passed tests/test_gpu_kernels.py) but 1.04-1.9x SLOWER than the tiled
// kernel on every RT-DETR shape: X fragments are fetched as 32-byte pieces per row (poor TA coalescing) and re-fetched by every
// channel-block wave, and the layers it targets already sit at the ~4.5 TB/s read+write plateau of this part.

// Pointwise (1x1, stride 1) convolution / linear layer for SHORT reductions (K = 64, 128, 256) on gfx950: the
// HBM-bound third of the network (ResNet "c"/"a" convs, shortcut convs, value / K / V projections).
//
// Why a third conv kernel: at K <= 256 a 128x128 (or 256x256) output tile is one to four K-steps deep, so a tiled
// kernel spends its time in prologue (first loads) and epilogue (LDS transpose + stores) with nothing to overlap
// them with - measured 1.4-3 TB/s and 120-330 TFLOP/s, neither roofline.  These layers are streaming: per output row
// K*2 bytes in, N*2 out (+ N*2 residual in); what matters is bytes in flight, not tile reuse.  So:
//   * W is REGISTER-resident: a wave owns CW = 64 output channels and keeps their [64 x K] weight block as MFMA
//     A-operand fragments (K/16 x 2 x 4 VGPRs) for its whole life; bias likewise;
//   * X streams straight from global memory into the MFMA B operand (lane = pixel, 16 bytes at k = 16*kk + 8*(lane>>5)):
//     no LDS, no barriers, every wave independent; a wave walks 32-pixel tiles in a grid-stride loop and prefetches
//     the next tile's X fragments and residual while the current tile computes;
//   * the accumulator (column = pixel, rows = 4 consecutive channels x 4 groups) is turned into 8 consecutive channels per
//     lane with v_permlane32_swap (lanes j and j+32 hold the same pixel), so bias/residual/activation are applied and
//     stored as 16-byte vectors directly from registers.
// The 4 waves of a workgroup take neighbouring channel blocks of the same pixel tile (X hits in L1).
#include "conv_common.h"

struct PwArgs {
  const bf16_t* x;
  const bf16_t* w;
  const float* bias;
  const bf16_t* res;
  void* y;
  int M, N, Nstore, ldx, ldy, ldr;
  int act, out_f32, res_after;
  int nCb, nTiles, HoWo;
  unsigned x_bytes, r_bytes;
  int64_t y_bstride;
};

template <int K>
__global__ __launch_bounds__(256) void conv_pw_stream_kernel(const PwArgs p) {
  constexpr int KK = K / 16, CW = 64, CT = CW / 32;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int wid = blockIdx.x * 4 + wave, TW = gridDim.x * 4;  // TW % nCb == 0 (launcher): a wave keeps its channel block
  const int cb = wid % p.nCb;
  const int c0 = cb * CW;
  const int tstride = TW / p.nCb;
  int tile = wid / p.nCb;
  if (tile >= p.nTiles) return;

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res ? p.res : p.x), 0, p.res ? p.r_bytes : 0u, 0x00020000);

  // resident weight fragments + bias (weights / bias are padded to a multiple of 128 rows >= c0 + 64)
  bf16x8 wf[CT][KK];
#pragma unroll
  for (int a = 0; a < CT; ++a)
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
      wf[a][kk] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(p.w + (int64_t)(c0 + 32 * a + j) * K + 16 * kk + 8 * h));
  float bs[CT][16];
#pragma unroll
  for (int a = 0; a < CT; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) bs[a][r] = p.bias ? p.bias[c0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h] : 0.0f;

  uint4 xn[KK], rn[CT][2];
#define FX_PW_LOAD(T_)                                                                                       \
  {                                                                                                          \
    const int m_ = (T_)*32 + j;                                                                              \
    const unsigned xo_ = m_ < p.M ? (unsigned)m_ * (unsigned)(p.ldx * 2) + 16u * h : FX_OOB;                 \
    _Pragma("unroll") for (int kk = 0; kk < KK; ++kk) xn[kk] = buf_load16(xr, m_ < p.M ? xo_ + 32u * kk : FX_OOB); \
    if (p.res) {                                                                                             \
      _Pragma("unroll") for (int a = 0; a < CT; ++a) _Pragma("unroll") for (int q = 0; q < 2; ++q) {          \
        const int ch_ = c0 + 32 * a + 16 * q + 8 * h;                                                        \
        rn[a][q] = buf_load16(rr, (m_ < p.M && ch_ < p.Nstore) ? ((unsigned)m_ * (unsigned)p.ldr + ch_) * 2u : FX_OOB); \
      }                                                                                                      \
    }                                                                                                        \
  }
  FX_PW_LOAD(tile);
  for (; tile < p.nTiles; tile += tstride) {
    uint4 xc[KK], rc[CT][2];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) xc[kk] = xn[kk];
#pragma unroll
    for (int a = 0; a < CT; ++a) rc[a][0] = rn[a][0], rc[a][1] = rn[a][1];
    if (tile + tstride < p.nTiles) FX_PW_LOAD(tile + tstride);  // next tile's bytes fly under this tile's math and stores
    f32x16 acc[CT];
#pragma unroll
    for (int a = 0; a < CT; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][r] = bs[a][r];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
      for (int a = 0; a < CT; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[a][kk], __builtin_bit_cast(bf16x8, xc[kk]), acc[a], 0, 0, 0);
    const int m = tile * 32 + j;
    int64_t yrow;
    if (p.y_bstride) {
      const int bb = m / p.HoWo;
      yrow = (int64_t)bb * p.y_bstride + (int64_t)(m - bb * p.HoWo) * p.ldy;
    } else {
      yrow = (int64_t)m * p.ldy;
    }
#pragma unroll
    for (int a = 0; a < CT; ++a)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        // rows 8q..8q+3 (channels 16q + 4h + 0..3) and 8q+4..8q+7 (channels 16q + 8 + 4h + 0..3) of this pixel:
        // swapping the upper half-wave of the first with the lower half-wave of the second leaves lane (j,h) with the 8
        // consecutive channels 16q + 8h + 0..7.
        float v[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[a][8 * q + i]), __float_as_uint(acc[a][8 * q + 4 + i]), false, false);
          v[i] = __uint_as_float(s[0]);
          v[4 + i] = __uint_as_float(s[1]);
        }
        const int ch = c0 + 32 * a + 16 * q + 8 * h;
        float rf[8];
        if (p.res) {
          unpack_bf16x8(rc[a][q], rf);
          if (!p.res_after) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += rf[i];
          }
        }
        if (p.act != FX_ACT_NONE) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = fx_act(v[i], p.act);
        }
        if (p.res && p.res_after) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] += rf[i];
        }
        if (m < p.M && ch < p.Nstore) {
          if (p.out_f32) {
            float* dst = reinterpret_cast<float*>(p.y) + yrow + ch;
            *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
          } else {
            *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.y) + yrow + ch) = pack_bf16x8(v);
          }
        }
      }
  }
#undef FX_PW_LOAD
}

#ifndef FX_PW_MIN_M
#define FX_PW_MIN_M 16384
#endif

bool fx_conv_pw_eligible(const ConvArgs& a) {
  static const int on = fx_tune("FX_PW_STREAM", 1), min_m = fx_tune("FX_PW_MIN_M", FX_PW_MIN_M);
  if (!on || a.KH != 1 || a.KW != 1 || a.stride != 1 || a.pad != 0) return false;
  if (a.C != 64 && a.C != 128 && a.C != 256) return false;
  return a.M >= min_m;
}

int fx_launch_conv_pw(const ConvArgs& a, hipStream_t stream) {
  PwArgs p;
  p.x = a.x; p.w = a.w; p.bias = a.bias; p.res = a.res; p.y = a.y;
  p.M = a.M; p.N = a.N; p.Nstore = a.Nstore; p.ldx = a.ldx; p.ldy = a.ldy; p.ldr = a.ldr;
  p.act = a.act; p.out_f32 = a.out_f32; p.res_after = a.res_after;
  p.nCb = (a.Nstore + 63) / 64;
  p.nTiles = (a.M + 31) / 32;
  p.HoWo = a.Ho * a.Wo;
  p.x_bytes = a.x_bytes; p.r_bytes = a.r_bytes;
  p.y_bstride = a.y_bstride;
  // persistent grid: ~8 waves per CU, total wave count a multiple of the channel-block count
  static const int wpc = fx_tune("FX_PW_WAVES_PER_CU", 8);
  const int64_t items = (int64_t)p.nTiles * p.nCb;
  int64_t waves = (int64_t)256 * wpc;
  if (waves > items) waves = items;
  int64_t blocks = (waves + 3) / 4;
  // 4 * blocks must be a multiple of nCb: round blocks up to a multiple of nCb / gcd(nCb, 4)
  int g = p.nCb % 4 == 0 ? 4 : (p.nCb % 2 == 0 ? 2 : 1);
  const int step = p.nCb / g;
  blocks = (blocks + step - 1) / step * step;
  dim3 grid((unsigned)blocks), block(256);
  if (a.C == 64) hipLaunchKernelGGL(conv_pw_stream_kernel<64>, grid, block, 0, stream, p);
  else if (a.C == 128) hipLaunchKernelGGL(conv_pw_stream_kernel<128>, grid, block, 0, stream, p);
  else hipLaunchKernelGGL(conv_pw_stream_kernel<256>, grid, block, 0, stream, p);
  return fx_launch_status();
}
