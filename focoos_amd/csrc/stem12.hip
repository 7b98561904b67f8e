// The first two stem layers in ONE launch (round 5, gfx950): normalise + conv1_1 (3x3 / s2 / p1, 3 -> 32, BN folded, ReLU) + conv1_2 (3x3 / s1 / p1,
// 32 -> 32, BN folded, ReLU) straight from the uint8 image.  Reference: FAIDetr.forward's (images - mean) / std (fai_detr/modelling.py:1349) and
// ResNet.conv1 = conv1_1, conv1_2, conv1_3 (focoos/nn/backbone/resnet.py:184-196).
//
// Why: unfused, the [B,320,320,32] conv1_1 activation is written (105 MB per 16-image part) and read back by conv1_2 for 3 GFLOP of work on a stage
// that runs at its byte bound on both queues (fx_stem_conv3x3s2 58 us + conv3x3_c32<32> 62 us per part).  conv1_1 is cheap enough to recompute
// for a halo: fused, the pair reads the 20 MB uint8 image ~1.3 times and writes conv1_2's 105 MB.
//
// Same construction as stem_pool.hip (2-D tiles, everything a tile needs in LDS, eight waves per workgroup at <= 128 registers, two 77 KiB workgroups
// per CU = four waves per SIMD, no register ring, no asm).  A workgroup owns one image, 8 conv1_2 rows and 64 conv1_2 columns:
//   phase 0  the 21 x 133 input pixels the tile depends on, NORMALISED ((u8 - mean) * inv_std, rounded to bf16, zero outside the image: conv1_1's zero
//            padding applies to the normalised image) -> LDS [21][400] bf16;
//   phase 1  conv1_1 on the 10 x 66 positions conv1_2 needs (tile + 1 halo) with fx_stem_conv3x3s2's MFMA form: K = 27 taps padded to 32 = two
//            v_mfma_f32_32x32x16 per 32 positions, weights as the A operand, the SAME assignment of taps to K slots ([row 0 bytes 0-7 | row 2 bytes
//            0-7], then [row 1 bytes 0-7 | byte 8 of rows 0, 1, 2 + zeros]) and the same operand arithmetic, so its outputs are bit-identical to
//            the stand-alone kernel's; ReLU, ZERO outside the conv1_1 image (conv1_2's padding), bf16 -> LDS in conv3x3_c32's k-plane layout;
//   phase 2  conv1_2 from those planes (tap (dy, dx) of position t = position t + dy * 66 + dx, no border masks), weights = the layer's fragment-order
//            image (18 KiB) in LDS, a wave = one tile row = two 32-column blocks; bias-initialised accumulators, k-steps 0..17 in conv3x3_c32's order:
//            bit-identical to it; ReLU, v_permlane32_swap pairs -> 16-byte row stores.
#include "pw_common.h"

struct Stem12Args {
  const unsigned char* x;   // uint8 HWC images [B,H,W,3]
  const float* w1;          // conv1_1 weights (BN folded) [27][32]: ((ky * 3 + kx) * 3 + c) * 32 + n  (fx_stem_conv3x3s2's table)
  const float* b1;          // [32]
  const float* mean;        // [3]
  const float* inv_std;     // [3]
  const bf16_t* w2;         // conv1_2 weights in fragment order [18][64][8]
  const float* b2;          // [32]
  bf16_t* y;                // [B,H1,W1,32] (pixel stride ldy)
  int H, W, H1, W1, ldy;
  int nbands, nstrips;
  unsigned x_bytes;
};

#define S12_TW 66                       // conv1_1 positions per tile row: 64 + 1 halo each side
#define S12_TH 10                       // 8 + 1 halo each side
#define S12_POS (S12_TW * S12_TH)       // 660
#define S12_PLANE (S12_POS * 16)
#define S12_IN_ROWS 21                  // input pixel rows: 2 * 10 + 1
#define S12_IN_COLS 133                 // input pixel columns: 2 * 66 + 1
#define S12_IN_PITCH 800                // bytes per input row in LDS: 133 * 3 = 399 bf16 (+ 1 pad)
#define S12_PL_OFF 0
#define S12_IN_OFF (4 * S12_PLANE)                        // 42 240
#define S12_W_OFF (S12_IN_OFF + S12_IN_ROWS * S12_IN_PITCH)   // 59 040
#define S12_SMEM (S12_W_OFF + 18 * 1024)                  // 77 472 bytes

__global__ __launch_bounds__(512, 4) void stem12_kernel(const Stem12Args p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, half = lane >> 5;
  int bid = fx_xcd_remap(blockIdx.x, gridDim.x);
  const int strip = bid % p.nstrips;
  bid /= p.nstrips;
  const int band = bid % p.nbands, b = bid / p.nbands;
  const int R0 = 8 * band, C0 = 64 * strip;                 // first conv1_2 row / column of the tile
  const int GY0 = 2 * R0 - 3, GX0 = 2 * C0 - 3;             // image coordinates of input-tile pixel (0, 0)

  // ---- conv1_2 weights by LDS-DMA (18 instructions over the eight waves)
  {
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, 18 * 1024, 0x00020000);
    for (int i = wave; i < 18; i += 8) pw_dma16(wr, smem + S12_W_OFF + i * 1024, (unsigned)(i * 1024 + lane * 16));
  }
  // ---- phase 0: normalised input tile -> LDS (element e of a row = pixel e / 3, channel e % 3).  Thread = one element column e of all 21 rows: the
  // pixel / channel split, the column validity and the normalisation constants are per-thread constants, a row costs one byte load (all 21 issued
  // before the first is consumed) + convert + store.  (First form: a flat index per element - two integer divisions and 64-bit address arithmetic
  // per byte, and load -> convert -> store per iteration: 111 us per part, VALU- and latency-bound in this phase.)
  {
    bf16_t* tin = reinterpret_cast<bf16_t*>(smem + S12_IN_OFF);
    const __amdgpu_buffer_rsrc_t ir = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const int e = tid;                                   // 400 of the 512 threads carry a column
    const int px = e / 3, c = e - px * 3;
    const int gx = GX0 + px;
    const bool col_ok = e < 399 && (unsigned)gx < (unsigned)p.W;
    const float mc = c == 0 ? p.mean[0] : (c == 1 ? p.mean[1] : p.mean[2]), sc = c == 0 ? p.inv_std[0] : (c == 1 ? p.inv_std[1] : p.inv_std[2]);
    const unsigned coff = (unsigned)(gx * 3 + c);
    unsigned raw[S12_IN_ROWS];
#pragma unroll
    for (int r = 0; r < S12_IN_ROWS; ++r) {
      const int gy = GY0 + r;
      const bool ok = col_ok && (unsigned)gy < (unsigned)p.H;
      raw[r] = (unsigned)(unsigned char)__builtin_amdgcn_raw_buffer_load_b8(ir, ok ? (unsigned)((b * p.H + gy) * p.W * 3) + coff : FX_OOB, 0, 0);
    }
    if (e < 400) {
#pragma unroll
      for (int r = 0; r < S12_IN_ROWS; ++r) {
        const bool ok = col_ok && (unsigned)(GY0 + r) < (unsigned)p.H;
        tin[r * 400 + e] = f32_to_bf16(ok ? ((float)raw[r] - mc) * sc : 0.0f);
      }
    }
  }
  // ---- A fragments of conv1_1 (as stem_mfma_kernel builds them): channel = l32; slot j of the lane's 8 K values -> (image row, byte)
  bf16x8 a1, a2;
  float bs1[16];
  {
    float f1[8], f2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int rr1 = half ? 2 : 0;                        // first MFMA: row 0 / row 2, byte j
      f1[j] = p.w1[((rr1 * 3 + j / 3) * 3 + j % 3) * 32 + l32];
      if (half == 0) f2[j] = p.w1[((1 * 3 + j / 3) * 3 + j % 3) * 32 + l32];          // second MFMA, low half: row 1, byte j
      else f2[j] = j < 3 ? p.w1[((j * 3 + 2) * 3 + 2) * 32 + l32] : 0.0f;             // high half: byte 8 (kx = 2, c = 2) of rows 0, 1, 2
    }
    a1 = __builtin_bit_cast(bf16x8, pack_bf16x8(f1));
    a2 = __builtin_bit_cast(bf16x8, pack_bf16x8(f2));
#pragma unroll
    for (int r = 0; r < 16; ++r) bs1[r] = p.b1[(r & 3) + 8 * (r >> 2) + 4 * half];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- phase 1: conv1_1 on the 660 tile positions (21 blocks of 32 over the eight waves) -> k-plane tile
  {
    const unsigned char* tin = smem + S12_IN_OFF;
    for (int blk = wave; blk < (S12_POS + 31) / 32; blk += 8) {
      const int pos = blk * 32 + l32;
      const int pc = pos < S12_POS ? pos : S12_POS - 1;
      const int ty = pc / S12_TW, tx = pc - ty * S12_TW;
      const int base = (2 * ty) * S12_IN_PITCH + 12 * tx;       // byte address of (input row 2 ty, element 6 tx)
      // MFMA 1: half 0 = row 0 bytes 0-7, half 1 = row 2 bytes 0-7; MFMA 2: half 0 = row 1 bytes 0-7, half 1 = byte 8 of rows 0, 1, 2
      const int ra = base + (half ? 2 : 0) * S12_IN_PITCH;
      uint4 u1, u2;
      u1.x = *reinterpret_cast<const unsigned*>(tin + ra);
      u1.y = *reinterpret_cast<const unsigned*>(tin + ra + 4);
      u1.z = *reinterpret_cast<const unsigned*>(tin + ra + 8);
      u1.w = *reinterpret_cast<const unsigned*>(tin + ra + 12);
      const int rb = base + S12_IN_PITCH;
      const unsigned r1x = *reinterpret_cast<const unsigned*>(tin + rb), r1y = *reinterpret_cast<const unsigned*>(tin + rb + 4),
                     r1z = *reinterpret_cast<const unsigned*>(tin + rb + 8), r1w = *reinterpret_cast<const unsigned*>(tin + rb + 12);
      const unsigned e0 = *reinterpret_cast<const unsigned short*>(tin + base + 16), e1 = *reinterpret_cast<const unsigned short*>(tin + rb + 16),
                     e2 = *reinterpret_cast<const unsigned short*>(tin + base + 2 * S12_IN_PITCH + 16);
      u2.x = half ? (e0 | (e1 << 16)) : r1x;
      u2.y = half ? e2 : r1y;
      u2.z = half ? 0u : r1z;
      u2.w = half ? 0u : r1w;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = bs1[r];
      acc = FX_MFMA_32x32x16(a1, __builtin_bit_cast(bf16x8, u1), acc);
      acc = FX_MFMA_32x32x16(a2, __builtin_bit_cast(bf16x8, u2), acc);
      const int y1 = R0 - 1 + ty, x1 = C0 - 1 + tx;
      const bool ok = (unsigned)y1 < (unsigned)p.H1 && (unsigned)x1 < (unsigned)p.W1;
      unsigned pk[4][2];
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
#pragma unroll
        for (int w2 = 0; w2 < 2; ++w2) {
          const float v0 = ok ? fmaxf(acc[4 * gq + 2 * w2], 0.0f) : 0.0f, v1 = ok ? fmaxf(acc[4 * gq + 2 * w2 + 1], 0.0f) : 0.0f;
          pk[gq][w2] = pack_bf16x2(v0, v1);
        }
#pragma unroll
      for (int g2 = 0; g2 < 2; ++g2) {
        unsigned q0[2] = {pk[2 * g2][0], pk[2 * g2][1]}, q1[2] = {pk[2 * g2 + 1][0], pk[2 * g2 + 1][1]};
#pragma unroll
        for (int w2 = 0; w2 < 2; ++w2) {   // half 0 ends up with channels 16 g2 + 0..7 (piece 2 g2), half 1 with 16 g2 + 8..15 (piece 2 g2 + 1)
          const auto sw = __builtin_amdgcn_permlane32_swap(q0[w2], q1[w2], false, false);
          q0[w2] = sw[0];
          q1[w2] = sw[1];
        }
        // piece c lives in plane (c & 1) * 2 + (c >> 1): half 0 -> plane g2, half 1 -> plane 2 + g2
        if (pos < S12_POS) *reinterpret_cast<uint4*>(smem + S12_PL_OFF + (half * 2 + g2) * S12_PLANE + pos * 16) = make_uint4(q0[0], q0[1], q1[0], q1[1]);
      }
    }
  }
  __syncthreads();

  // ---- phase 2: conv1_2, wave = tile row `wave`, two 32-column blocks
  {
    const int lds0 = (int)(unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
    int row0[2];
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) row0[bb] = lds0 + S12_PL_OFF + half * 2 * S12_PLANE + ((wave + 1) * S12_TW + (32 * bb + l32 + 1)) * 16;
    const int waddr = lds0 + S12_W_OFF + lane * 16;
    f32x16 acc[2];
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const float4 bb4 = *reinterpret_cast<const float4*>(p.b2 + 8 * gq + 4 * half);
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        acc[bb][4 * gq] = bb4.x; acc[bb][4 * gq + 1] = bb4.y; acc[bb][4 * gq + 2] = bb4.z; acc[bb][4 * gq + 3] = bb4.w;
      }
    }
    typedef __attribute__((address_space(3))) const bf16x8 lds_frag_t;
#pragma unroll
    for (int s = 0; s < 18; ++s) {   // k-step s = tap * 2 + j
      const int toff = (((s >> 1) / 3 - 1) * S12_TW + ((s >> 1) % 3 - 1)) * 16 + (s & 1) * S12_PLANE;
      const bf16x8 af = *reinterpret_cast<lds_frag_t*>((size_t)(unsigned)(waddr + s * 1024));
      bf16x8 xf[2];
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) xf[bb] = *reinterpret_cast<lds_frag_t*>((size_t)(unsigned)(row0[bb] + toff));
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) acc[bb] = FX_MFMA_32x32x16(af, xf[bb], acc[bb]);
    }
    const int yr = R0 + wave;
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
      const int xc = C0 + 32 * bb + l32;
      const bool live = yr < p.H1 && xc < p.W1;
      bf16_t* yrow = p.y + ((size_t)(b * p.H1 + yr) * p.W1 + xc) * p.ldy + half * 8;
#pragma unroll
      for (int g2 = 0; g2 < 2; ++g2) {
        unsigned q[2][2];
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
          const int e = 4 * (2 * g2 + q2);
          q[q2][0] = pack_bf16x2(fmaxf(acc[bb][e], 0.0f), fmaxf(acc[bb][e + 1], 0.0f));
          q[q2][1] = pack_bf16x2(fmaxf(acc[bb][e + 2], 0.0f), fmaxf(acc[bb][e + 3], 0.0f));
        }
#pragma unroll
        for (int w2 = 0; w2 < 2; ++w2) {
          const auto sw = __builtin_amdgcn_permlane32_swap(q[0][w2], q[1][w2], false, false);
          q[0][w2] = sw[0];
          q[1][w2] = sw[1];
        }
        if (live) *reinterpret_cast<uint4*>(yrow + g2 * 16) = make_uint4(q[0][0], q[0][1], q[1][0], q[1][1]);
      }
    }
  }
}

extern "C" int fx_stem_conv12_u8_bf16(const void* x_u8, const float* w1, const float* b1, const float* mean, const float* inv_std, const void* w2_frag,
                                      const float* b2, void* y, int ldy, int B, int H, int W, fx_stream_t stream_) {
  FX_CHECK_ARG(x_u8 && w1 && b1 && mean && inv_std && w2_frag && b2 && y && B > 0 && H >= 2 && W >= 2 && ldy >= 32 && ldy % 8 == 0);
  FX_CHECK_ARG(((uintptr_t)w2_frag % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)b2 % 16) == 0);
  Stem12Args a{};
  a.x = reinterpret_cast<const unsigned char*>(x_u8);
  a.w1 = w1; a.b1 = b1; a.mean = mean; a.inv_std = inv_std;
  a.w2 = reinterpret_cast<const bf16_t*>(w2_frag);
  a.b2 = b2;
  a.y = reinterpret_cast<bf16_t*>(y);
  a.H = H; a.W = W; a.ldy = ldy;
  a.H1 = (H - 1) / 2 + 1; a.W1 = (W - 1) / 2 + 1;
  a.nbands = (a.H1 + 7) / 8;
  a.nstrips = (a.W1 + 63) / 64;
  const int64_t grid = (int64_t)B * a.nbands * a.nstrips;
  if (grid >= (1ll << 31) || (int64_t)B * H * W * 3 >= (1ll << 31)) return FX_ERR_UNSUPPORTED;
  a.x_bytes = (unsigned)((int64_t)B * H * W * 3);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(stem12_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, S12_SMEM) != hipSuccess) return FX_ERR_RUNTIME;
    attr_set = true;
  }
  hipLaunchKernelGGL(stem12_kernel, dim3((int)grid), dim3(512), S12_SMEM, reinterpret_cast<hipStream_t>(stream_), a);
  return fx_launch_status();
}
