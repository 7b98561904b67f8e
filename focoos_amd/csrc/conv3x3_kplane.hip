// 3x3 / stride 1 / pad 1 convolution on a flat pixel range + halo, K-PLANE LDS layout (gfx950).  Round-3 successor of the 3x3
// path of conv3x3_flat.hip (same idea: a workgroup owns BM consecutive pixels of the flattened NHWC tensor, ONE contiguous halo
// range [m0 - W - 1, m0 + BM + W + 1) of 64 channels serves all nine taps, a dedicated loader wave streams the next channel
// chunk with LDS-DMA, weights are the MFMA A operand in fragment order straight from L2 through a register ring).
//
// What the s_memtime stamps of the round-2 kernel showed (scripts/probes/c3_probe.hip, profiles/r03_c3_probe_*.txt): a
// workgroup spent 57.5k of its 75k cycles in the K loop whose MFMAs need 36.9k, and 13.4k in the epilogue.  With one wave per
// SIMD the wave's instruction stream is issued in order, and every memory instruction in it (ds_read_b128, global_load) costs
// ~20 issue cycles next to an MFMA that leaves 32: four pixel-fragment reads, each preceded by two address VALU ops, behind
// the first four MFMAs of a k-step overflowed those slots (+21 cycles per read, +20 per weight load), and the SiLU epilogue
// spent ~10 VALU per value on a full-precision division.  Hence, here:
//   * LDS layout [plane = half*4 + j][row][8 channels]: the 16-byte piece a lane feeds to k-step j of a tap lives at
//     row*16 + (half*4 + j)*PLANE, so consecutive lanes read consecutive 16-byte pieces (conflict-free without a swizzle) and
//     the k-step is an IMMEDIATE offset of ds_read_b128: one address register per (32-pixel block, tap), set up by 4 VALU ops
//     that sit behind MFMAs of the previous tap; no address arithmetic inside the k-steps.  The loader's DMA gathers 16 bytes
//     per lane from 64 consecutive pixels (the eight planes of a row block back to back, so the 64 lines stay in L1);
//   * masked (pixel, tap) pairs read a zero row that every plane carries at row index HLP (an address select, as before);
//   * per k-step the memory instructions are spread one per MFMA slot: fragment reads behind the MFMAs of weight block 0, ring
//     refills behind the last MFMA of each weight block, next-tap address setup behind the MFMAs of the last weight block;
//   * weight loads use the scalar-base form (global_load ... v_off, s[base]) - the per-tap pointer update is SALU work;
//   * SiLU = x * rcp(1 + exp(-x)) with the hardware reciprocal (1 ulp in fp32, far below the bf16 rounding of the output).
#include <type_traits>

#include "pw_common.h"

struct C3KArgs {
  const bf16_t* x;
  const bf16_t* wp;
  const float* bias;
  const bf16_t* res;
  bf16_t* y;
  int H, W, C, N, ldx, ldy, ldr, M;
  int HLp;            // halo rows of this launch rounded up to whole 64-row DMA blocks (<= HLP)
  int HW;             // pixels per image
  int64_t y_bstride;  // elements between images of y (0: contiguous)
  unsigned x_bytes, r_bytes;
  unsigned long long* dbg;   // probe builds only (ABL & 16)
};

template <int OFF>
__device__ __forceinline__ void c3k_ldg(bf16x8& dst, unsigned voff, const bf16_t* sbase) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "+v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}

typedef __attribute__((address_space(3))) const bf16x8 c3k_lds_frag_t;
// fragment read at an integer LDS byte address (+ immediate): no "base + offset" VALU add per read
template <int IMM>
__device__ __forceinline__ bf16x8 c3k_lds_read(int addr) {
  return *reinterpret_cast<c3k_lds_frag_t*>((size_t)(unsigned)(addr + IMM));
}

__device__ __forceinline__ float c3k_act(float v, std::integral_constant<int, FX_ACT_RELU>) { return fmaxf(v, 0.0f); }
__device__ __forceinline__ float c3k_act(float v, std::integral_constant<int, FX_ACT_SILU>) {
  return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
}
__device__ __forceinline__ float c3k_act(float v, std::integral_constant<int, FX_ACT_NONE>) { return v; }

// TN x TM 32x32 accumulator blocks per wave (TN weight blocks x TM pixel blocks), WN x WM consumer waves + the loader wave.
// HLP: rows per plane (compile time: the k-step offsets are immediates); a launch needs BM + 2W + 2 <= HLP.
// ABL: ablation / instrumentation switches of scripts/probes/c3_probe.hip (1: ring never refilled, 2: fragments read once,
//      4: no DMA after the first chunk, 8: no output stores, 16: s_memtime stamps of workgroup 0); 0 in the product.
// LD:  1 = loader wave (default).  0 = none: one-chunk layers (C = 64) without a residual have nothing to stream after the first chunk,
//      which the consumer waves fetch themselves anyway - the workgroup is then WN x WM waves at <= 256 registers and TWO of them
//      share a CU when their LDS allows (res2's 3x3 layers at W = 160: 2 x 78 KiB), one's halo fetch / stores beside the other's MFMAs.
template <int TN, int TM, int WN, int WM, int HLP, int ACT, int RESMODE, int ABL = 0, int LD = 1>
__global__ __launch_bounds__((WN* WM + LD) * 64, LD ? 1 : 2) void conv3x3_kplane_kernel(const C3KArgs p) {
  constexpr int NW = WN * WM, NT = (NW + LD) * 64, NDW = NW + LD;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int KJ = 4, PF = KJ;
  constexpr int PLANE = (HLP + 1) * 16, BUF = 8 * PLANE;
  constexpr int RLT = BN / 8;
  static_assert(TN <= 2 && HLP % 32 == 0 && 3 * PLANE < 65536, "k-step offsets are 16-bit immediates");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_loader = LD && wave == NW;
  const int l32 = lane & 31, half = lane >> 5;
  const int wn = wave % WN, wm = wave / WN;
  const int nNt = p.N / BN;
  const int bid = fx_xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / nNt) * BM, n0 = (bid % nNt) * BN;
  const int lo = m0 - p.W - 1;
  const int NCH = p.C >> 6;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  unsigned char* T = smem;  // output tile [BM][BN] bf16 (aliases the halo buffers after the K loop), rows of BN*2 bytes, pw_swz<RLT>

  // chunk cc -> buf: instructions first, first + step, ...; instruction i = (row block i / 8, channel piece i % 8): the eight
  // pieces of a row block are issued back to back, so the 64 lines they share are fetched once
  auto dma_chunk = [&](int cc, unsigned char* buf, int first, int step) {
    const int ninstr = ((p.HLp + 63) >> 6) * 8;
    for (int i = first; i < ninstr; i += step) {
      const int blk = i >> 3, c = i & 7;
      const int r = blk * 64 + lane;
      const int f = lo + r;
      const bool ok = f >= 0 && f < p.M;
      const int pln = (c & 1) * 4 + (c >> 1);   // piece c = channels [8c, 8c+8) = k-step c/2, half c%2
      // (p.HLp is a multiple of 32: the upper half of the last instruction may lie behind the plane - those lanes stay out of it)
      if (r < p.HLp) pw_dma16(xr, buf + pln * PLANE + blk * 1024, ok ? (unsigned)(f * p.ldx + cc * 64 + c * 8) * 2u : FX_OOB);
    }
  };
  auto dma_res = [&](int first) {   // residual tile -> T
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, p.r_bytes, 0x00020000);
    for (int i = first; i < BM * RLT / 64; i += NDW) {
      const int q = i * 64 + lane;
      const int r = q / RLT, pc = q % RLT;
      const int lc = pw_swz<RLT>(r, pc);
      const int m = m0 + r;
      pw_dma16(rr, T + i * 1024, m < p.M ? (unsigned)(m * p.ldr + n0 + lc * 8) * 2u : FX_OOB);
    }
  };
  int dbg_slot = 0;
  auto stamp = [&]() {
    if constexpr (ABL & 16) {
      if (blockIdx.x == 0 && lane == 0) p.dbg[wave * 16 + dbg_slot] = __builtin_amdgcn_s_memtime();
      ++dbg_slot;
    }
  };
  stamp();
  int dbg2 = 0;
  auto stamp2 = [&]() {   // prologue detail (ABL & 64): slots 9..15
    if constexpr (ABL & 64) {
      if (blockIdx.x == 0 && lane == 0) p.dbg[wave * 16 + 9 + dbg2] = __builtin_amdgcn_s_memtime();
      ++dbg2;
    }
  };

  if constexpr (!LD && (ABL & (32 | 128))) {
    // Two workgroups of this form share a CU (two waves per SIMD).  Started together and both MFMA-bound they advance in lockstep - prologue
    // beside prologue, K loop beside K loop at half speed each, epilogue beside epilogue - and the matrix pipe idles exactly where it idles
    // for one workgroup.  A fixed issue priority breaks the symmetry: the favoured wave of a SIMD runs its K loop at full rate, the other
    // one fills its gaps and owns the pipe during the favoured one's prologue / epilogue.  ABL & 32: by the wave's hardware slot
    // (HW_ID.wave_id parity); ABL & 128: by the parity of blockIdx.x / 256 (the dispatcher fills every CU once before the second slot).
    // Measured (profiles/r06_c3_duo_probe.txt): +-1 % either way - the two workgroups are NOT held back by lockstep; the form's 3-4 % over the
    // loader-wave kernel at M >= 51 200 is all the co-residency gives, and at M = 25 600 (200 tiles: one workgroup per CU) it is 6 % slower.
    if constexpr (ABL & 32) fx_prio_by_hw_slot();
    else if ((blockIdx.x >> 8) & 1u) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(3);
  }
  if (is_loader) {
    // the zero row of every plane of both buffers (published by the first barrier; the DMA never touches row HLP)
    if (lane < (NCH > 1 ? 16 : 8)) *reinterpret_cast<uint4*>(smem + (lane >> 3) * BUF + (lane & 7) * PLANE + HLP * 16) = make_uint4(0, 0, 0, 0);
    stamp2();
    stamp2();
    for (int cc = 0; cc < NCH; ++cc) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (cc == 0) stamp2();
      __syncthreads();  // chunk cc has landed; every consumer is done with chunk cc-1 (the other buffer)
      if (cc == 0) stamp2();
      if constexpr (!(ABL & 4)) {
        if (cc + 1 < NCH) dma_chunk(cc + 1, smem + ((cc + 1) & 1) * BUF, 0, 1);
      }
    }
    if constexpr (RESMODE != 0) {
      __syncthreads();  // E1: the halo buffers are dead; the output tile T (aliases them) may be written
      dma_res(NW);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // E2: residual tile in T
      __syncthreads();  // E3: output tile complete
    }
  } else {
    f32x16 acc[TN][TM];
    bf16x8 ar[PF][TN];
    unsigned mask9[TM];
    int row0[TM];      // byte address of the lane's piece of pixel block b at tap offset 0, k-step 0, buffer 0
    if (!LD && wave == 0 && lane < 8) *reinterpret_cast<uint4*>(smem + lane * PLANE + HLP * 16) = make_uint4(0, 0, 0, 0);   // zero rows
    // first the requests with the longest way to go (weights: first touch of this launch, L2 / HBM), then the DMA share of the
    // first chunk; the bias loads and the mask arithmetic below run while both are in flight
    // weights: fragment (n-block, k16 step) = 512 elements; k = tap * C + channel, so the four k-steps of a (tap, chunk)
    // are 4 KiB contiguous.  wbase[a]: scalar pointer of n-block a of this wave; the lane's 16 bytes at lane * 16.
    const int Cs = p.C >> 4;
    const unsigned wvoff = lane * 16;
    const bf16_t* wbase[TN];
#pragma unroll
    for (int a = 0; a < TN; ++a) wbase[a] = p.wp + (size_t)((n0 >> 5) + wn * TN + a) * (size_t)(9 * Cs) * 512;
    auto w_ptr = [&](int a, int cc, int t) -> const bf16_t* { return wbase[a] + (size_t)(t * Cs + cc * KJ) * 512; };
#pragma unroll
    for (int a = 0; a < TN; ++a) {
      const bf16_t* w0 = w_ptr(a, 0, 0);
      c3_static_for<PF>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        ar[i][a] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        c3k_ldg<i * 1024>(ar[i][a], wvoff, w0);
      });
    }
    stamp2();
    dma_chunk(0, smem, wave, NW);   // this wave's share of the first chunk (the loader, second wave on its SIMD, is slow to start)
    stamp2();
    const int HWp = p.H * p.W;
    const int lds0 = (int)(unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);   // LDS address of the dynamic segment
#pragma unroll
    for (int b = 0; b < TM; ++b) {
      const int pl = (wm * TM + b) * 32 + l32;
      const int m = m0 + pl;
      row0[b] = lds0 + (pl + p.W + 1) * 16 + half * 4 * PLANE;
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
    stamp2();
    // accumulators start at the bias (an ordinary load: hipcc waits for everything in flight before its first use, so it comes
    // last in the prologue)
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
    const int zhalf = lds0 + half * 4 * PLANE + HLP * 16;   // the lane's zero row (buffer 0)
    stamp2();

    for (int cc = 0; cc < NCH; ++cc) {
      if (cc == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the first chunk (and the ring)
        stamp2();
      } else if constexpr (!LD) {
        // loader-less multi-chunk form (round 6): ONE halo buffer, every consumer fetches its share of the next chunk once all of them
        // are done with the current one; the fetch's round trip is exposed to THIS workgroup and covered by the second workgroup of the
        // CU (two waves per SIMD).  vmcnt(0) also lands the ring's first fragments of the chunk, so the counted waits of the tap loop
        // never see an LDS-DMA in the queue (a zero-fill DMA may retire ahead of an older weight load: DESIGN_HISTORY, round 3).
        __syncthreads();
        dma_chunk(cc, smem, wave, NW);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();  // chunk cc has landed; every consumer is done with chunk cc-1 (the other buffer)
      stamp();
      const int bufo = LD ? (cc & 1) * BUF : 0;
      const int ccn = cc + 1 < NCH ? cc + 1 : cc;
      const int zaddr = zhalf + bufo;
      int addr[TM], addrn[TM];
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        addr[b] = (mask9[b] & 1u) ? row0[b] + bufo + (-p.W - 1) * 16 : zaddr;
        addrn[b] = addr[b];
      }
      bf16x8 xb[2][TM];
#pragma unroll
      for (int b = 0; b < TM; ++b) xb[0][b] = c3k_lds_read<0>(addr[b]);
      if constexpr (ABL & 2) {
#pragma unroll
        for (int b = 0; b < TM; ++b) xb[1][b] = xb[0][b];
      }
      // The tap loop is a real loop (one ring cycle of KJ k-steps per tap).
#pragma unroll 1
      for (int t = 0; t < 9; ++t) {
        const bool last_tap = t == 8;
        const int tn = last_tap ? 0 : t + 1;
        // scalar state of the NEXT tap: row offset (bytes, incl. the buffer) and mask bit.  After the last tap of a chunk the
        // "next tap" is tap 0 of the SAME buffer: its fragment reads are harmless and unused (the next chunk re-reads after
        // its barrier), which keeps the k-steps free of branches.
        const int offn = bufo + ((tn / 3 - 1) * p.W + (tn % 3 - 1)) * 16;
        const unsigned bitn = 1u << tn;
        const bf16_t* wnext[TN];
#pragma unroll
        for (int a = 0; a < TN; ++a) wnext[a] = w_ptr(a, last_tap ? ccn : cc, tn);
        auto kstep = [&](auto jc) {   // j is a compile-time constant: immediate offsets of the fragment reads and weight loads
          constexpr int j = decltype(jc)::value;
          if constexpr (!(ABL & 1)) {
            if constexpr (TN == 1) c3_wait<(KJ - 1) * TN>(ar[j][0]); else c3_wait<(KJ - 1) * TN>(ar[j][0], ar[j][1]);
          }
#pragma unroll
          for (int a = 0; a < TN; ++a) {
#pragma unroll
            for (int b = 0; b < TM; ++b) {
              acc[a][b] = FX_MFMA_32x32x16(ar[j][a], xb[j & 1][b], acc[a][b]);
              if constexpr (!(ABL & 2)) {
                if (a == 0) {   // fragment b of the next k-step (k-step 0 of the next tap behind the last one)
                  if constexpr (j + 1 < KJ) xb[(j + 1) & 1][b] = c3k_lds_read<(j + 1) * PLANE>(addr[b]);
                  else xb[0][b] = c3k_lds_read<0>(addrn[b]);
                }
              }
              if constexpr (j == 0) {   // address of block b at the next tap: 4 VALU ops behind an MFMA of the last weight block
                if (a == TN - 1) addrn[b] = (mask9[b] & bitn) ? row0[b] + offn : zaddr;
              }
              if constexpr (!(ABL & 1)) {
                if (b == TM - 1) c3k_ldg<j * 1024>(ar[j][a], wvoff, wnext[a]);
              }
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        };
        c3_static_for<KJ>(kstep);
#pragma unroll
        for (int b = 0; b < TM; ++b) addr[b] = addrn[b];
      }
    }
    stamp();
    // drain the hidden loads before their registers are reused by the epilogue
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      if constexpr (TN == 1) c3_wait<0>(ar[i][0]); else c3_wait<0>(ar[i][0], ar[i][1]);
    }
    if constexpr (RESMODE == 0) {
      // No residual: straight from the accumulators to memory.  A lane owns 4 consecutive channels (8 bytes) of pixel l32 per
      // (n-block, 8-channel group gq); v_permlane32_swap pairs group 2g of the lower half-wave with group 2g+1 of the upper one,
      // after which every lane holds 16 contiguous bytes of its pixel row: 16 dwordx4 stores per wave, no LDS tile, no barrier.
      stamp();
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        const int m = m0 + (wm * TM + b) * 32 + l32;
        size_t yo = (size_t)m * p.ldy;
        if (p.y_bstride) {
          const int bb = m / p.HW;
          yo = (size_t)bb * p.y_bstride + (size_t)(m - bb * p.HW) * p.ldy;
        }
        bf16_t* yrow = p.y + yo + n0 + wn * TN * 32 + half * 8;
        const bool live = m < p.M && !((ABL & 8) && p.H > 0);
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            unsigned pk[2][2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              float v[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = c3k_act(acc[a][b][4 * (2 * g + q) + e], std::integral_constant<int, ACT>{});
              pk[q][0] = pack_bf16x2(v[0], v[1]);
              pk[q][1] = pack_bf16x2(v[2], v[3]);
            }
            // lower lanes keep group 2g and receive the partner's group 2g (channels +4); upper lanes receive the partner's
            // group 2g+1 and keep their own
#pragma unroll
            for (int w = 0; w < 2; ++w) {   // lanes 32-63 of pk[0] <-> lanes 0-31 of pk[1]
              const auto sw = __builtin_amdgcn_permlane32_swap(pk[0][w], pk[1][w], false, false);
              pk[0][w] = sw[0];
              pk[1][w] = sw[1];
            }
            if (live) *reinterpret_cast<uint4*>(yrow + a * 32 + g * 16) = make_uint4(pk[0][0], pk[0][1], pk[1][0], pk[1][1]);
          }
      }
      stamp();
    } else {
    __syncthreads();  // E1
    {
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
          float v[4] = {acc[a][b][4 * gq], acc[a][b][4 * gq + 1], acc[a][b][4 * gq + 2], acc[a][b][4 * gq + 3]};
          float r[4] = {0.f, 0.f, 0.f, 0.f};
          if constexpr (RESMODE != 0) {
            const uint2 rv = *reinterpret_cast<const uint2*>(tp);
            r[0] = bf16lo_to_f32(rv.x); r[1] = bf16hi_to_f32(rv.x);
            r[2] = bf16lo_to_f32(rv.y); r[3] = bf16hi_to_f32(rv.y);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if constexpr (RESMODE == 1) v[e] += r[e];                                  // act(conv + residual)
            v[e] = c3k_act(v[e], std::integral_constant<int, ACT>{});
            if constexpr (RESMODE == 2) v[e] += r[e];                                  // act(conv) + residual
            if constexpr (RESMODE == 3) v[e] = r[e] > 0.0f ? v[e] : 0.0f;              // act(conv) * (residual > 0): ReLU backward
          }
          uint2 o;
          o.x = pack_bf16x2(v[0], v[1]);
          o.y = pack_bf16x2(v[2], v[3]);
          *reinterpret_cast<uint2*>(tp) = o;
        }
    __syncthreads();  // E3
    stamp();
    }
  }
  if constexpr (RESMODE != 0)
  for (int q = tid; q < BM * RLT; q += NT) {
    const int row = q / RLT, lc = q % RLT;
    const int m = m0 + row;
    if (m < p.M && !((ABL & 8) && p.H > 0)) {
      const uint4 v = *reinterpret_cast<const uint4*>(T + row * (BN * 2) + (pw_swz<RLT>(row, lc) << 4));
      size_t yo = (size_t)m * p.ldy;
      if (p.y_bstride) {
        const int bb = m / p.HW;
        yo = (size_t)bb * p.y_bstride + (size_t)(m - bb * p.HW) * p.ldy;
      }
      *reinterpret_cast<uint4*>(p.y + yo + n0 + lc * 8) = v;
    }
  }
  if constexpr (ABL & 16) {
    __syncthreads();
    stamp();
  }
}

template <int TN, int TM, int WN, int WM, int HLP, int ACT, int RESMODE, int ABL = 0, int LD = 1>
static int launch_c3k(C3KArgs& a, hipStream_t stream) {
  constexpr int NW = WN * WM, BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int PLANE = (HLP + 1) * 16, BUF = 8 * PLANE;
  const int HL = BM + 2 * a.W + 2;
  a.HLp = (HL + 31) / 32 * 32;
  if (a.HLp > HLP || a.C % 64 != 0 || a.N % BN != 0) return FX_ERR_UNSUPPORTED;
  const int halo = ((a.C > 64 && LD) ? 2 : 1) * BUF, tile = RESMODE ? BM * BN * 2 : 0;
  const int smem = halo > tile ? halo : tile;
  if (smem > 160 * 1024) return FX_ERR_UNSUPPORTED;
  auto kern = conv3x3_kplane_kernel<TN, TM, WN, WM, HLP, ACT, RESMODE, ABL, LD>;
  static int attr_smem = 0;
  if (smem > attr_smem) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) return FX_ERR_RUNTIME;
    attr_smem = smem;
  }
  hipLaunchKernelGGL(kern, dim3(((a.M + BM - 1) / BM) * (a.N / BN)), dim3((NW + LD) * 64), smem, stream, a);
  return fx_launch_status();
}

// ---- product entry points (declared in conv_common.h)
// Tiles: N a multiple of 256: 128 pixels x 256 channels (4 waves side by side over N); N = 128: 256 x 128 (2 x 2 waves);
// N = 64: 512 x 64 (4 waves over M, all four reading the same weight fragments - the whole filter is 72 KiB, L1/L2 resident).
// Plane heights: the smallest instantiated HLP >= BM + 2W + 2 whose buffers fit the 160 KiB LDS.
// small-M form of the N % 256 tile: 64 pixels x 256 channels (2 x 2 accumulator blocks per wave) - twice the workgroups for the
// 20x20-level layers (M = 6 400 per 16-image part: 50 -> 100 pixel tiles), each with half the MFMA work
static bool c3k_small_m(int M, int N, int W, int C) {
  static const int thr = fx_tune("FX_C3K_SMALL_M", 16000);
  // round 6: for C <= 256 (the hybrid encoder's 20x20 level: four chunks per tile) the limit can be set apart (FX_C3K_SMALL_M_C256; default = the
  // general one) - under three batches in flight the 128-pixel tiles were +1 % for RT-DETR at M = 12 800 while BiSeNetFormer's 512-channel
  // layers of the same M lost 15 % on them
  static const int thr256 = fx_tune("FX_C3K_SMALL_M_C256", thr);
  return N % 256 == 0 && M <= (C <= 256 ? thr256 : thr) && 64 + 2 * W + 2 <= 192;
}

static int c3k_plan(int C, int N, int W) {   // 0: not covered, else HLP
  if (C % 64 != 0) return 0;
  const int nbuf = C > 64 ? 2 : 1;
  static const int wide_n = fx_tune("FX_C3K_WIDE_N", 1);   // 0: only N = 256 (A/B knob for the N = 512 layers of res5)
  const bool n256 = N == 256 || (wide_n && N > 0 && N % 256 == 0);
  const int BM = n256 ? 128 : (N == 128 ? 256 : (N == 64 ? 512 : 0));
  if (!BM) return 0;
  const int need = (BM + 2 * W + 2 + 63) / 64 * 64;
  static const int opts256[] = {320, 576}, opts128[] = {512}, opts64[] = {960};
  const int* opts = n256 ? opts256 : (N == 128 ? opts128 : opts64);
  const int nopt = n256 ? 2 : 1;
  for (int i = 0; i < nopt; ++i)
    if (opts[i] >= need && nbuf * 8 * (opts[i] + 1) * 16 <= 160 * 1024) return opts[i];
  return 0;
}

extern "C" int fx_conv3x3_kplane_supported(int C, int N, int W) { return c3k_plan(C, N, W) ? 1 : 0; }

int fx_launch_conv3x3_kplane(const ConvArgs& c, const bf16_t* w_frag, hipStream_t stream) {
  const int hlp = c3k_plan(c.C, c.N, c.W);
  if (!hlp) return FX_ERR_UNSUPPORTED;
  C3KArgs a{};
  a.x = c.x; a.wp = w_frag; a.bias = c.bias; a.res = c.res; a.y = reinterpret_cast<bf16_t*>(c.y);
  a.H = c.H; a.W = c.W; a.C = c.C; a.N = c.N; a.ldx = c.ldx; a.ldy = c.ldy; a.ldr = c.ldr; a.M = c.M;
  a.HW = c.Ho * c.Wo; a.y_bstride = c.y_bstride; a.x_bytes = c.x_bytes; a.r_bytes = c.r_bytes; a.dbg = nullptr;
  const int mode = fx_c3_epilogue_mode(c.act, c.res != nullptr, c.res_after);
  const bool small = c3k_small_m(c.M, c.N, c.W, c.C);
  // res2's 3x3 layers (64 -> 64, one chunk, no residual): 256-pixel tiles on four waves, no loader, two workgroups per CU
  static const int duo_on = fx_tune("FX_C3K_DUO", 1);
  if (duo_on && c.N == 64 && c.C == 64 && !c.res && 256 + 2 * c.W + 2 <= 608) {
    if (mode == 0) return launch_c3k<2, 2, 1, 4, 608, FX_ACT_RELU, 0, 0, 0>(a, stream);
    if (mode == 1) return launch_c3k<2, 2, 1, 4, 608, FX_ACT_SILU, 0, 0, 0>(a, stream);
    if (mode == 3) return launch_c3k<2, 2, 1, 4, 608, FX_ACT_NONE, 0, 0, 0>(a, stream);
  }
  // round 6: the loader-less multi-chunk form (two workgroups per CU) for N % 256 == 0 layers without a residual from FX_C3K_DUO256_MIN_M pixels on
  // (default 40 000; 0 = never): +3-4 % per launch at M >= 51 200 in the stand-alone probe (profiles/r06_c3_duo_probe.txt; -6 % at M = 25 600,
  // where 200 tiles leave one workgroup per CU), +1 % on the three-lane headline in one call (5 099 / 5 099 -> 5 142 / 5 159 img/s, `roofline.frac`
  // 0.408 -> 0.423)
  static const int duo256_min_m = fx_tune("FX_C3K_DUO256_MIN_M", 40000);
  static const int duo256_res = fx_tune("FX_C3K_DUO256_RES", 1);   // also the layers WITH a residual (the output tile goes through LDS: 64 KiB per workgroup)
  const bool duo256 = duo256_min_m > 0 && c.M >= duo256_min_m && (!c.res || duo256_res) && c.N % 256 == 0;   // hlp 320 (W <= 95) and 576 (MaskFormer's 200-wide level: one 74 KiB buffer, two per CU)
  static const int duo128_on = fx_tune("FX_C3K_DUO128", 0);   // N = 128 (res3's branch2b): 256 x 128 tiles, one 66 KiB buffer
  const bool duo128 = duo128_on && duo256_min_m > 0 && c.M >= duo256_min_m && c.N == 128 && (!c.res || duo256_res);
#define FX_C3K_TILE(ACT_, RM_)                                                             \
  {                                                                                        \
    if (duo128) return launch_c3k<2, 4, 2, 2, 512, ACT_, RM_, 0, 0>(a, stream);            \
    if (duo256 && hlp == 320) return launch_c3k<2, 4, 4, 1, 320, ACT_, RM_, 0, 0>(a, stream); \
    if (duo256) return launch_c3k<2, 4, 4, 1, 576, ACT_, RM_, 0, 0>(a, stream);            \
    if (small) return launch_c3k<2, 2, 4, 1, 192, ACT_, RM_>(a, stream);                   \
    if (c.N == 64) return launch_c3k<2, 4, 1, 4, 960, ACT_, RM_>(a, stream);               \
    if (c.N == 128) return launch_c3k<2, 4, 2, 2, 512, ACT_, RM_>(a, stream);              \
    if (hlp == 320) return launch_c3k<2, 4, 4, 1, 320, ACT_, RM_>(a, stream);              \
    return launch_c3k<2, 4, 4, 1, 576, ACT_, RM_>(a, stream);                              \
  }
  switch (mode) {
    case 0: FX_C3K_TILE(FX_ACT_RELU, 0)
    case 1: FX_C3K_TILE(FX_ACT_SILU, 0)
    case 2: FX_C3K_TILE(FX_ACT_SILU, 2)
    case 3: FX_C3K_TILE(FX_ACT_NONE, 0)
    case 5: FX_C3K_TILE(FX_ACT_NONE, 3)
    default: return FX_ERR_UNSUPPORTED;
  }
#undef FX_C3K_TILE
}
