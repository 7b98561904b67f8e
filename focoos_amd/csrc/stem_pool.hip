// conv1_3 (3x3 / s1 / p1, 32 -> 64 channels, folded BatchNorm + ReLU) FUSED with the stem's max-pool (3x3 / s2 / p1) - round 5 (gfx950).
// Reference: ResNet.forward  x = self.conv1(x); x = F.max_pool2d(x, 3, 2, 1)  (focoos/nn/backbone/resnet.py:184-196, 252-256).
//
// Why: unfused, the [B,320,320,64] conv1_3 activation is written once (210 MB per 16-image part) and read back 1.0-1.5 times by the pool
// (318 MB measured) to produce a tensor a quarter of its size - 0.53 of the stem chain's 1.0 GB per part, on a stage that runs AT its byte
// bound on both queues.  Fused, the layer reads its 105 MB input ~1.35 times and writes 52 MB.
//
// Tiling.  The pool needs conv rows 2p-1 .. 2p+1 for pooled row p, so the tile is 2-D: a workgroup (4 waves) owns one image, a band of 8
// pooled rows and a strip of 15 pooled columns; wave w owns pooled rows 2w, 2w+1 of the band = 5 conv rows x 32 conv columns (columns
// 2 i0 - 1 .. 2 i0 + 30 for the strip's first pooled column i0): five 32-pixel MFMA blocks whose lanes are 32 CONSECUTIVE COLUMNS of one
// conv row, so
//   * the vertical 3-maximum of a pooled row is elementwise over three of the wave's own accumulators (no data movement),
//   * the horizontal 3-maximum (conv columns 2k-1+.., i.e. lanes 2k, 2k+1, 2k+2) is two ds_bpermute per packed register, valid at the even lanes
//     0..28 = 15 pooled columns per 32 conv columns (6.7 % recompute), and a conv row shared by two pooled rows of different waves /
//     bands is recomputed (5 rows per 4: 25 %).  Executed flops = 1.37 x the layer's: MFMA work of ~35 us per 16-image part at peak against the
//     conv (94 us) + pool (59 us) launches it replaces.
// The input tile (19 rows x 34 columns of 32 channels, zero outside the image: OOB buffer loads) sits in LDS in the k-plane layout of
// conv3x3_c32.hip ([plane = half * 2 + j][position][8 channels]), fetched by buffer_load ... lds (each 64-byte pixel is four 16-byte pieces, one per
// plane); a conv output at tile position t reads tap (dy, dx) at t + dy * 34 + dx - no border masks in the K loop (the padding IS zeros in the
// tile).  Weights: the SAME fragment-order image conv3x3_c32 uses ([2 blocks][18 k-steps][64 lanes][8]), all 36 KiB of it in LDS (an A
// fragment feeds five MFMAs, a B fragment both channel blocks: 126 LDS reads per 180 MFMAs of a wave) - no register ring, no asm: two
// workgroups per CU (77 KiB each) cover each other's fetch / store phases.  (First form: one 18 KiB weight block at a time, two passes over
// the tile with a reload + two barriers in between and every B fragment read twice - 116 us per part, LDS-bound; the four planes are packed
// at their exact 646-position pitch, filled by 41 DMA instructions over the concatenated slot space, to make room for both blocks.)
// Exactness: per output the accumulation order is bias, then k-steps 0..17 - conv3x3_c32's order - and max commutes with the monotonic
// bf16 rounding, so the result is bit-identical to the two launches it replaces.  Conv positions outside the image are forced to 0 before the
// maximum: every value is >= 0 after the ReLU and every window holds at least one real pixel, so 0 stands in for max_pool2d's -inf padding.
#include "pw_common.h"

struct StemPoolArgs {
  const bf16_t* x;
  const bf16_t* wp;
  const float* bias;
  bf16_t* y;
  int H, W, Ho, Wo, ldx, ldy;
  int nbands, nstrips;
  unsigned x_bytes;
};

#define SP_TW 34                      // tile columns: 32 conv columns + 1 halo column each side
#define SP_TH 19                      // tile rows: 17 conv rows (8 pooled rows) + 1 halo row each side
#define SP_POS (SP_TW * SP_TH)        // 646 positions per plane, planes packed back to back
#define SP_PLANE (SP_POS * 16)
#define SP_NDMA ((4 * SP_POS + 63) / 64)   // 41 DMA instructions over the 2584 (plane, position) slots; the last one's 40 spare lanes land in the pad
#define SP_WOFF (SP_NDMA * 1024)      // both weight blocks behind the planes (+ pad)
#define SP_SMEM (SP_WOFF + 36 * 1024)

typedef __attribute__((ext_vector_type(2))) unsigned short sp_u16x2;

__device__ __forceinline__ unsigned sp_max_u16x2(unsigned a, unsigned b) {   // packed maximum of two NON-NEGATIVE 16-bit floats (order = integer order)
  const sp_u16x2 va = __builtin_bit_cast(sp_u16x2, a), vb = __builtin_bit_cast(sp_u16x2, b);
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(va, vb));
}

// DMA of one input tile (plane-packed, 41 instructions) by the calling wave(s): instruction i = first, first + step, ...
__device__ __forceinline__ void sp_dma_tile(const StemPoolArgs& p, __amdgpu_buffer_rsrc_t xr, unsigned char* tile, int b, int band, int strip, int lane,
                                            int first, int step) {
  const int Y0 = 16 * band - 2, X0 = 30 * strip - 2;       // image coordinates of tile position (0, 0)
  // slot S = plane * 646 + position, 64 slots per instruction, lane = slot; plane pl holds piece c = (pl & 1) * 2 + (pl >> 1) of a pixel's
  // 64 bytes (plane = half * 2 + j <-> channels 16 j + 8 half .. + 7)
  for (int i = first; i < SP_NDMA; i += step) {
    const int S = i * 64 + lane;
    const int pl = S / SP_POS, t = S - pl * SP_POS;
    const int c = (pl & 1) * 2 + (pl >> 1);
    const int ty = t / SP_TW, tx = t - ty * SP_TW;
    const int gy = Y0 + ty, gx = X0 + tx;
    const bool ok = pl < 4 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    const unsigned off = ok ? (unsigned)(((b * p.H + gy) * p.W + gx) * p.ldx + c * 8) * 2u : FX_OOB;
    pw_dma16(xr, tile + i * 1024, off);
  }
}

// One tile by the four compute waves: `tile_lds` / `w_lds` = LDS byte addresses of the input tile and of the weights
__device__ __forceinline__ void sp_compute_tile(const StemPoolArgs& p, int tile_lds, int w_lds, int b, int band, int strip, int wave, int lane) {
  const int l32 = lane & 31, half = lane >> 5;
  int row0[5];      // LDS address of (conv row bb of this wave, conv column l32) in plane half * 2
#pragma unroll
  for (int bb = 0; bb < 5; ++bb) row0[bb] = tile_lds + half * 2 * SP_PLANE + ((4 * wave + 1 + bb) * SP_TW + (l32 + 1)) * 16;
  const int cx = 30 * strip - 1 + l32;                      // conv column of this lane
  const bool col_ok = (unsigned)cx < (unsigned)p.W;
  const int cy0 = 16 * band + 4 * wave - 1;                  // conv row of block 0
  const int prow0 = 8 * band + 2 * wave;                     // first pooled row of this wave
  const int pcol = 15 * strip + (l32 >> 1);                  // pooled column held by an even lane after the horizontal maximum
  const bool store_lane = (l32 & 1) == 0 && l32 <= 28 && pcol < p.Wo;
  const int waddr = w_lds + lane * 16;
  f32x16 acc[2][5];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const float4 bb4 = *reinterpret_cast<const float4*>(p.bias + a * 32 + 8 * gq + 4 * half);
#pragma unroll
      for (int bb = 0; bb < 5; ++bb) {
        acc[a][bb][4 * gq] = bb4.x; acc[a][bb][4 * gq + 1] = bb4.y; acc[a][bb][4 * gq + 2] = bb4.z; acc[a][bb][4 * gq + 3] = bb4.w;
      }
    }
  typedef __attribute__((address_space(3))) const bf16x8 lds_frag_t;
  // k-step s = tap * 2 + j: channels 16 j .. 16 j + 15 of tap (dy, dx).  Software pipeline: the seven fragments of step s + 1 (five pixel
  // blocks, two weight blocks) are requested BEFORE the ten MFMAs of step s, into the other half of a register double buffer - an LDS read
  // takes ~130 cycles, ten MFMAs 320; left to itself hipcc issued each read right in front of the MFMA pair that needs it (first form:
  // `ds_read; s_waitcnt lgkmcnt(0); 2 x v_mfma`, 104 us per part).  sched_barrier keeps the steps from being interleaved again.
  auto toff = [&](int s) { return (((s >> 1) / 3 - 1) * SP_TW + ((s >> 1) % 3 - 1)) * 16 + (s & 1) * SP_PLANE; };
  bf16x8 xb[2][5], af[2][2];
#pragma unroll
  for (int bb = 0; bb < 5; ++bb) xb[0][bb] = *reinterpret_cast<lds_frag_t*>((size_t)(unsigned)(row0[bb] + toff(0)));
  af[0][0] = *reinterpret_cast<lds_frag_t*>((size_t)(unsigned)(waddr));
  af[0][1] = *reinterpret_cast<lds_frag_t*>((size_t)(unsigned)(waddr + 18 * 1024));
#pragma unroll
  for (int s = 0; s < 18; ++s) {
    const int cur = s & 1, nxt = cur ^ 1;
    if (s + 1 < 18) {
#pragma unroll
      for (int bb = 0; bb < 5; ++bb) xb[nxt][bb] = *reinterpret_cast<lds_frag_t*>((size_t)(unsigned)(row0[bb] + toff(s + 1)));
      af[nxt][0] = *reinterpret_cast<lds_frag_t*>((size_t)(unsigned)(waddr + (s + 1) * 1024));
      af[nxt][1] = *reinterpret_cast<lds_frag_t*>((size_t)(unsigned)(waddr + (18 + s + 1) * 1024));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int bb = 0; bb < 5; ++bb) {
      acc[0][bb] = FX_MFMA_32x32x16(af[cur][0], xb[cur][bb], acc[0][bb]);
      acc[1][bb] = FX_MFMA_32x32x16(af[cur][1], xb[cur][bb], acc[1][bb]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    // ---- ReLU, zero outside the image, vertical maximum, pack, horizontal maximum, 16-byte stores
#pragma unroll
    for (int bb = 0; bb < 5; ++bb) {
      const bool ok = col_ok && (unsigned)(cy0 + bb) < (unsigned)p.H;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][bb][e] = (ok && acc[a][bb][e] > 0.0f) ? acc[a][bb][e] : 0.0f;   // (never -0.0: the packed maximum below orders BIT PATTERNS)
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      unsigned pk[4][2];   // [accumulator quad gq][register pair]: channels a*32 + 8 gq + 4 half + (0,1) / (2,3)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
#pragma unroll
        for (int w2 = 0; w2 < 2; ++w2) {
          const int e = 4 * gq + 2 * w2;
          const float v0 = fmaxf(fmaxf(acc[a][2 * r][e], acc[a][2 * r + 1][e]), acc[a][2 * r + 2][e]);
          const float v1 = fmaxf(fmaxf(acc[a][2 * r][e + 1], acc[a][2 * r + 1][e + 1]), acc[a][2 * r + 2][e + 1]);
          unsigned d = pack_bf16x2(v0, v1);
          const unsigned d1 = (unsigned)__builtin_amdgcn_ds_bpermute(((lane + 1) & 63) << 2, (int)d);
          const unsigned d2 = (unsigned)__builtin_amdgcn_ds_bpermute(((lane + 2) & 63) << 2, (int)d);
          pk[gq][w2] = sp_max_u16x2(sp_max_u16x2(d, d1), d2);
        }
      const int prow = prow0 + r;
      bf16_t* yrow = p.y + ((size_t)(b * p.Ho + prow) * p.Wo + pcol) * p.ldy + a * 32 + half * 8;
      const bool live = store_lane && prow < p.Ho;
#pragma unroll
      for (int g2 = 0; g2 < 2; ++g2) {
        unsigned q0[2] = {pk[2 * g2][0], pk[2 * g2][1]}, q1[2] = {pk[2 * g2 + 1][0], pk[2 * g2 + 1][1]};
#pragma unroll
        for (int w2 = 0; w2 < 2; ++w2) {   // half 0 keeps quad 2 g2 of both halves (8 consecutive channels), half 1 quad 2 g2 + 1
          const auto sw = __builtin_amdgcn_permlane32_swap(q0[w2], q1[w2], false, false);
          q0[w2] = sw[0];
          q1[w2] = sw[1];
        }
        if (live) *reinterpret_cast<uint4*>(yrow + g2 * 16) = make_uint4(q0[0], q0[1], q1[0], q1[1]);
      }
    }
  }
}

// Form 1: one workgroup (4 waves) per tile, two workgroups per CU (77 KiB each) covering each other's fetch / store phases.
__global__ __launch_bounds__(256, 2) void stem_c3_pool_kernel(const StemPoolArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware order: an XCD (own L2) takes a contiguous run of (image, band, strip) tiles - neighbours share halo rows / columns
  int bid = fx_xcd_remap(blockIdx.x, gridDim.x);
  const int strip = bid % p.nstrips;
  bid /= p.nstrips;
  const int band = bid % p.nbands, b = bid / p.nbands;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, 2 * 18 * 1024, 0x00020000);
  for (int i = wave; i < 36; i += 4) pw_dma16(wr, smem + SP_WOFF + i * 1024, (unsigned)(i * 1024 + lane * 16));
  sp_dma_tile(p, xr, smem, b, band, strip, lane, wave, 4);
  const int lds0 = (int)(unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  sp_compute_tile(p, lds0, lds0 + SP_WOFF, b, band, strip, wave, lane);
}

// Form 2 (round 5, measured and REMOVED: profiles/r05_stem_pool_persistent_ab.txt): persistent workgroups, one per CU, four compute waves + a
// loader wave double-buffering the tiles, weights fetched once per workgroup - 155 us per part against form 1's 101: with ONE wave per SIMD
// nothing covers a wave's LDS / epilogue latencies; the kernel is bound by how many waves share a SIMD, not by the fetch at its start.
//
// Form 3: eight waves per workgroup - waves 0-3 compute channel block 0 of the tile, waves 4-7 block 1, from the same input tile - each with
// ONE accumulator set (80 registers instead of 160): 128 registers per wave, so that two 77 KiB workgroups per CU are FOUR waves per SIMD.
__device__ __forceinline__ void sp_compute_tile_half(const StemPoolArgs& p, int tile_lds, int w_lds, int a, int b, int band, int strip, int wave,
                                                     int lane) {
  const int l32 = lane & 31, half = lane >> 5;
  int row0[5];
#pragma unroll
  for (int bb = 0; bb < 5; ++bb) row0[bb] = tile_lds + half * 2 * SP_PLANE + ((4 * wave + 1 + bb) * SP_TW + (l32 + 1)) * 16;
  const int cx = 30 * strip - 1 + l32;
  const bool col_ok = (unsigned)cx < (unsigned)p.W;
  const int cy0 = 16 * band + 4 * wave - 1;
  const int prow0 = 8 * band + 2 * wave;
  const int pcol = 15 * strip + (l32 >> 1);
  const bool store_lane = (l32 & 1) == 0 && l32 <= 28 && pcol < p.Wo;
  const int waddr = w_lds + a * 18 * 1024 + lane * 16;
  f32x16 acc[5];
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    const float4 bb4 = *reinterpret_cast<const float4*>(p.bias + a * 32 + 8 * gq + 4 * half);
#pragma unroll
    for (int bb = 0; bb < 5; ++bb) {
      acc[bb][4 * gq] = bb4.x; acc[bb][4 * gq + 1] = bb4.y; acc[bb][4 * gq + 2] = bb4.z; acc[bb][4 * gq + 3] = bb4.w;
    }
  }
  typedef __attribute__((address_space(3))) const bf16x8 lds_frag_t;
#pragma unroll
  for (int s = 0; s < 18; ++s) {
    const int toff = (((s >> 1) / 3 - 1) * SP_TW + ((s >> 1) % 3 - 1)) * 16 + (s & 1) * SP_PLANE;
    const bf16x8 af = *reinterpret_cast<lds_frag_t*>((size_t)(unsigned)(waddr + s * 1024));
    bf16x8 xf[5];
#pragma unroll
    for (int bb = 0; bb < 5; ++bb) xf[bb] = *reinterpret_cast<lds_frag_t*>((size_t)(unsigned)(row0[bb] + toff));
#pragma unroll
    for (int bb = 0; bb < 5; ++bb) acc[bb] = FX_MFMA_32x32x16(af, xf[bb], acc[bb]);
    __builtin_amdgcn_sched_barrier(0);   // one k-step's six reads and five MFMAs at a time: four waves per SIMD cover the latency, not a deep
                                         // per-wave prefetch (left free, the scheduler hoisted reads until 194 registers spilled at the 128 cap)
  }
#pragma unroll
  for (int bb = 0; bb < 5; ++bb) {
    const bool ok = col_ok && (unsigned)(cy0 + bb) < (unsigned)p.H;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[bb][e] = (ok && acc[bb][e] > 0.0f) ? acc[bb][e] : 0.0f;
  }
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    unsigned pk[4][2];
#pragma unroll
    for (int gq = 0; gq < 4; ++gq)
#pragma unroll
      for (int w2 = 0; w2 < 2; ++w2) {
        const int e = 4 * gq + 2 * w2;
        const float v0 = fmaxf(fmaxf(acc[2 * r][e], acc[2 * r + 1][e]), acc[2 * r + 2][e]);
        const float v1 = fmaxf(fmaxf(acc[2 * r][e + 1], acc[2 * r + 1][e + 1]), acc[2 * r + 2][e + 1]);
        unsigned d = pack_bf16x2(v0, v1);
        const unsigned d1 = (unsigned)__builtin_amdgcn_ds_bpermute(((lane + 1) & 63) << 2, (int)d);
        const unsigned d2 = (unsigned)__builtin_amdgcn_ds_bpermute(((lane + 2) & 63) << 2, (int)d);
        pk[gq][w2] = sp_max_u16x2(sp_max_u16x2(d, d1), d2);
      }
    const int prow = prow0 + r;
    bf16_t* yrow = p.y + ((size_t)(b * p.Ho + prow) * p.Wo + pcol) * p.ldy + a * 32 + half * 8;
    const bool live = store_lane && prow < p.Ho;
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
      unsigned q0[2] = {pk[2 * g2][0], pk[2 * g2][1]}, q1[2] = {pk[2 * g2 + 1][0], pk[2 * g2 + 1][1]};
#pragma unroll
      for (int w2 = 0; w2 < 2; ++w2) {
        const auto sw = __builtin_amdgcn_permlane32_swap(q0[w2], q1[w2], false, false);
        q0[w2] = sw[0];
        q1[w2] = sw[1];
      }
      if (live) *reinterpret_cast<uint4*>(yrow + g2 * 16) = make_uint4(q0[0], q0[1], q1[0], q1[1]);
    }
  }
}

__global__ __launch_bounds__(512, 4) void stem_c3_pool8_kernel(const StemPoolArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = fx_xcd_remap(blockIdx.x, gridDim.x);
  const int strip = bid % p.nstrips;
  bid /= p.nstrips;
  const int band = bid % p.nbands, b = bid / p.nbands;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, 2 * 18 * 1024, 0x00020000);
  for (int i = wave; i < 36; i += 8) pw_dma16(wr, smem + SP_WOFF + i * 1024, (unsigned)(i * 1024 + lane * 16));
  sp_dma_tile(p, xr, smem, b, band, strip, lane, wave, 8);
  const int lds0 = (int)(unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  sp_compute_tile_half(p, lds0, lds0 + SP_WOFF, wave >> 2, b, band, strip, wave & 3, lane);
}

// 1 iff the fused launch covers the layer pair: conv 3x3 / s1 / p1 with 32 input and 64 output channels + ReLU, then max-pool 3x3 / s2 / p1
extern "C" int fx_stem_conv_pool_supported(int C, int N, int H, int W) { return C == 32 && N == 64 && H >= 2 && W >= 2; }

extern "C" int fx_stem_conv3x3_relu_maxpool_bf16(const void* x, int ldx, const void* w_frag, const float* bias, void* y, int ldy, int B, int H, int W,
                                                 fx_stream_t stream_) {
  FX_CHECK_ARG(x && w_frag && bias && y && B > 0 && H >= 2 && W >= 2 && ldx >= 32 && ldx % 8 == 0 && ldy >= 64 && ldy % 8 == 0);
  FX_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)w_frag % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)bias % 16) == 0);
  const int64_t x_bytes = ((int64_t)B * H * W - 1) * ldx * 2 + 64;
  if (x_bytes >= 0xFFFFFFF0ll) return FX_ERR_UNSUPPORTED;
  StemPoolArgs a{};
  a.x = reinterpret_cast<const bf16_t*>(x);
  a.wp = reinterpret_cast<const bf16_t*>(w_frag);
  a.bias = bias;
  a.y = reinterpret_cast<bf16_t*>(y);
  a.H = H; a.W = W; a.ldx = ldx; a.ldy = ldy;
  a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1;
  a.nbands = (a.Ho + 7) / 8;
  a.nstrips = (a.Wo + 14) / 15;
  a.x_bytes = (unsigned)x_bytes;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(stem_c3_pool_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SP_SMEM) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(stem_c3_pool8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SP_SMEM) != hipSuccess)
      return FX_ERR_RUNTIME;
    attr_set = true;
  }
  const int64_t grid = (int64_t)B * a.nbands * a.nstrips;
  if (grid >= (1ll << 31)) return FX_ERR_UNSUPPORTED;
  static const int eight = fx_tune("FX_STEM_POOL_8WAVE", 1);
  if (eight) {
    hipLaunchKernelGGL(stem_c3_pool8_kernel, dim3((int)grid), dim3(512), SP_SMEM, reinterpret_cast<hipStream_t>(stream_), a);
    return fx_launch_status();
  }
  hipLaunchKernelGGL(stem_c3_pool_kernel, dim3((int)grid), dim3(256), SP_SMEM, reinterpret_cast<hipStream_t>(stream_), a);
  return fx_launch_status();
}
