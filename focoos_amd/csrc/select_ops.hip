// Selection / head kernels of the RT-DETR path (gfx950): row max, exact top-k (threshold search + rank
// sort in LDS), row gathers, the tiny K=4 / N=4 linear layers fused with the box update, the output
// head and the device-side DETRProcessor.postprocess.  Index results are int32 and must be bit-exact
// with the reference's int64 indices: top-k order is (value descending, index ascending).
#include <stdlib.h>

#include "common.h"

int fx_tune(const char* env_name, int default_value);  // conv_igemm.hip

// Per-kernel bisection of the two-queue failure (scripts/dev/pk_bisect.sh kernels): when this unit is compiled WITH packed-fp32
// instructions and -DFX_SELECT_PK_MASK=<bits>, only the kernels whose bit is set keep them.  In the product build (packed fp32 off for the
// whole unit) FX_SEL_PK expands to nothing.
#ifdef FX_SELECT_PK_MASK
#define FX_SEL_NOPK __attribute__((target("no-packed-fp32-ops")))
#if (FX_SELECT_PK_MASK >> 0) & 1
#define FX_SEL_PK_0
#else
#define FX_SEL_PK_0 FX_SEL_NOPK
#endif
#if (FX_SELECT_PK_MASK >> 1) & 1
#define FX_SEL_PK_1
#else
#define FX_SEL_PK_1 FX_SEL_NOPK
#endif
#if (FX_SELECT_PK_MASK >> 2) & 1
#define FX_SEL_PK_2
#else
#define FX_SEL_PK_2 FX_SEL_NOPK
#endif
#if (FX_SELECT_PK_MASK >> 3) & 1
#define FX_SEL_PK_3
#else
#define FX_SEL_PK_3 FX_SEL_NOPK
#endif
#if (FX_SELECT_PK_MASK >> 4) & 1
#define FX_SEL_PK_4
#else
#define FX_SEL_PK_4 FX_SEL_NOPK
#endif
#if (FX_SELECT_PK_MASK >> 5) & 1
#define FX_SEL_PK_5
#else
#define FX_SEL_PK_5 FX_SEL_NOPK
#endif
#if (FX_SELECT_PK_MASK >> 6) & 1
#define FX_SEL_PK_6
#else
#define FX_SEL_PK_6 FX_SEL_NOPK
#endif
#if (FX_SELECT_PK_MASK >> 7) & 1
#define FX_SEL_PK_7
#else
#define FX_SEL_PK_7 FX_SEL_NOPK
#endif
#define FX_SEL_PK(i) FX_SEL_PK_##i
#else
#define FX_SEL_PK(i)
#endif


// ------------------------------------------------------------------------------------------------
__global__ FX_SEL_PK(0) __launch_bounds__(256) void rowmax_kernel(const float* __restrict__ x, int ldx, float* __restrict__ out, int rows, int cols) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (int64_t)row * ldx;
  float m = -INFINITY;
  for (int c = lane; c < cols; c += 64) m = fmaxf(m, xr[c]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if (lane == 0) out[row] = m;
}

extern "C" int fx_rowmax_f32(const float* x, int ldx, float* out, int rows, int cols, fx_stream_t stream_) {
  FX_CHECK_ARG(x && out && rows > 0 && cols > 0 && ldx >= cols);
  hipLaunchKernelGGL(rowmax_kernel, dim3((rows + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), x, ldx, out, rows, cols);
  return fx_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Exact top-k per row.  One 1024-thread workgroup per row:
//  1. the key T of the k-th largest element and how many elements equal to T are needed: rows of <= 12 288 elements keep their keys in
//     registers and build T three bits per round from compare-and-count (round 5); longer rows stream through a 4-pass MSB-first radix
//     select (8-bit digits, LDS histogram);
//  2. every element with key > T plus the needed number of key == T elements (lowest indices first,
//     ordered block scan only when there are surplus ties) are collected into LDS;
//  3. the <= 1024 (key, ~index) entries are ordered (value desc, index asc) by rank: place = number of greater entries.
// Keys are the order-preserving unsigned image of the float bits.
#define TOPK_THREADS 1024
#define TOPK_MAXK 1024

__device__ __forceinline__ uint32_t f32_key(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_f32(uint32_t k) {
  uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

// Virtual rows: blockIdx.x = image * nchunk + chunk; the row is elements [chunk * chunk_len, ...) of the image's `n_total` scores
// (nchunk = 1: the plain per-image form).  `src_idx` (optional, [images][n_total]): original index of every element - the second level of
// the two-level form, whose elements are candidates of the first; the sort key and the output use the ORIGINAL index, so ties order
// exactly as in a single pass (candidates of equal value sit in ascending original-index order: chunks are index ranges, and each
// chunk's list is (value desc, index asc)).  Slots beyond the row's own length are padded (-inf, INT_MAX).
// NPT > 0 (round 5): the row's keys live in REGISTERS, NPT per thread (rows of <= 1024 * NPT elements), loaded once with all loads in flight
// together; the four radix passes, the collection and the tie scan then run on registers and LDS only.  The streaming form (NPT = 0, any
// length) re-reads the row from L2 in every pass, one dependent load per 1024 elements and pass: 36 us for the encoder's 8 400 scores per
// image, almost all of it load latency in a single workgroup.
template <int NPT>
__global__ FX_SEL_PK(1) __launch_bounds__(TOPK_THREADS) void topk_kernel(const float* __restrict__ scores, int ld, int n_total, int chunk_len, int nchunk, int k_out,
                                                             const int32_t* __restrict__ src_idx, float* __restrict__ out_val,
                                                             int32_t* __restrict__ out_idx, unsigned long long* dbg) {
#define TOPK_STAMP(slot) do { if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dbg[slot] = __builtin_amdgcn_s_memtime(); } while (0)
  __shared__ uint32_t hist[256];
  __shared__ unsigned long long sel[TOPK_MAXK];
  __shared__ uint32_t s_prefix, s_remaining, s_count, s_base, s_wave_tot[16], s_cnt[2][TOPK_THREADS / 64], s_rank[TOPK_MAXK];
  const int tid = threadIdx.x;
  const int img = blockIdx.x / nchunk, ch = blockIdx.x - img * nchunk;
  const int base = ch * chunk_len;
  const int n = min(chunk_len, n_total - base);
  const int k = min(k_out, n);
  const float* row = scores + (int64_t)img * ld + base;
  const int32_t* sidx = src_idx ? src_idx + (int64_t)img * ld + base : nullptr;

  TOPK_STAMP(0);
  uint32_t kreg[NPT > 0 ? NPT : 1];
  if constexpr (NPT > 0) {
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
      const int i = j * TOPK_THREADS + tid;
      kreg[j] = i < n ? f32_key(row[i]) : 0u;
    }
  }
  const int iters = NPT > 0 ? NPT : (n + TOPK_THREADS - 1) / TOPK_THREADS;
  // key of element j * 1024 + tid (callers test i < n themselves)
#define TOPK_KEY(j, i) (NPT > 0 ? kreg[(NPT > 0) ? (j) : 0] : ((i) < n ? f32_key(row[(i)]) : 0u))
#define TOPK_FOR_ELEMENTS(j) _Pragma("unroll") for (int j = 0; j < (NPT > 0 ? NPT : iters); ++j)

  uint32_t T, need_eq, count_eq;   // key of the k-th largest element, how many elements equal to it are needed (1..count_eq), how many there are
  if constexpr (NPT > 0) {
    // Keys in registers: T is built bit by bit from the top - T | bit stays iff at least k keys are >= it - with one v_cmp + s_bcnt1 per key
    // and wave, one LDS word per wave and ONE barrier per bit (the 16 partial counts alternate between two rows, so a wave that is a round
    // ahead never overwrites what a slower one still reads).  s_memtime stamps (scripts/dev/topk_stamps.py, profiles/r05_topk_ab.txt): a
    // round costs ~60 ticks per key register (four waves per SIMD issuing the compare + count) + ~250 for the exchange, 26 000 ticks for
    // 8 400 scores; three bits per round (63 compares, 11 rounds) measured 42 600 - the compares are the cost, not the barriers - and the
    // 8-bit radix passes of the streaming form (histogram atomics, one ballot round per distinct bin) 4 x ~8 us.
    const int lane = tid & 63, wv = tid >> 6;
    auto block_sum = [&](uint32_t wave_total, int buf) -> uint32_t {
      if (lane == 0) s_cnt[buf][wv] = wave_total;
      __syncthreads();
      uint32_t tot = 0;
#pragma unroll
      for (int w = 0; w < TOPK_THREADS / 64; ++w) tot += s_cnt[buf][w];
      return tot;
    };
    uint32_t t = 0;
    if (dbg) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // diagnostic only: the stamp below is taken with the keys loaded
    TOPK_STAMP(1);
#pragma unroll 2
    for (int bit = 31; bit >= 0; --bit) {
      const uint32_t cand = t | (1u << bit);   // >= 1: the zero keys of the padding slots are never counted
      uint32_t c = 0;
#pragma unroll
      for (int j = 0; j < NPT; ++j) c += (uint32_t)__popcll(__ballot(kreg[j] >= cand));
      if (block_sum(c, bit & 1) >= (uint32_t)k) t = cand;
    }
    uint32_t cg = 0, ce = 0;
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
      const bool in = j * TOPK_THREADS + tid < n;
      cg += (uint32_t)__popcll(__ballot(in && kreg[j] > t));
      ce += (uint32_t)__popcll(__ballot(in && kreg[j] == t));
    }
    // row 1: the round of bit 0 used row 0 and its readers may still be at it; row 1 was last read before that round's barrier.
    // cg <= k <= 1024, ce <= 12 288: no carry between the two halves
    const uint32_t both = block_sum((cg << 16) | ce, 1);
    T = t;
    need_eq = (uint32_t)k - (both >> 16);
    count_eq = both & 0xffffu;
  } else {
    uint32_t prefix = 0, mask = 0;
    if (tid == 0) s_remaining = (uint32_t)k;
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
      TOPK_FOR_ELEMENTS(j) {
        const int i = j * TOPK_THREADS + tid;
        const bool in = i < n;
        const uint32_t key = TOPK_KEY(j, i);
        bool live = in && (key & mask) == prefix;
        const uint32_t bin = (key >> shift) & 255u;
        // scores crowd into one or two bins per digit: one atomic per (wave, bin) for the first few distinct bins of the wave, plain
        // atomics for whatever is left (same-address LDS atomics serialise per lane otherwise)
  #pragma unroll 1
        for (int it = 0; it < 4; ++it) {
          const unsigned long long act = __ballot(live);
          if (!act) break;
          const uint32_t lead = __builtin_amdgcn_readlane(bin, (int)__builtin_ctzll(act));
          const unsigned long long same = __ballot(live && bin == lead);
          if ((tid & 63) == (int)__builtin_ctzll(act)) atomicAdd(&hist[lead], (uint32_t)__popcll(same));
          if (bin == lead) live = false;
        }
        if (live) atomicAdd(&hist[bin], 1u);
      }
      __syncthreads();
      if (tid < 64) {   // wave 0: lane l owns bins 4l..4l+3; suffix sums over the lanes locate the bin where the count from the top reaches `rem`
        const uint32_t rem = s_remaining;
        const uint32_t h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
        const uint32_t mine = h0 + h1 + h2 + h3;
        uint32_t suf = mine;   // inclusive suffix sum: bins of lanes >= tid
  #pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t v = __shfl_down(suf, o, 64);
          if (tid + o < 64) suf += v;
        }
        const uint32_t above = suf - mine;   // count in bins above this lane's
        if (above < rem && suf >= rem) {     // exactly one lane
          uint32_t c = above;
          int bin;
          uint32_t hb;
          if (c + h3 >= rem) { bin = 4 * tid + 3; hb = h3; }
          else if ((c += h3) + h2 >= rem) { bin = 4 * tid + 2; hb = h2; }
          else if ((c += h2) + h1 >= rem) { bin = 4 * tid + 1; hb = h1; }
          else { c += h1; bin = 4 * tid; hb = h0; }
          s_remaining = rem - c;  // still needed among keys whose digits so far equal prefix|bin
          s_prefix = prefix | ((uint32_t)bin << shift);
          s_count = hb;           // after the last pass: number of elements with key == T
        }
      }
      __syncthreads();
      prefix = s_prefix;
      mask |= 0xffu << shift;
    }
    T = prefix;
    need_eq = s_remaining;  // 1..count_eq
    count_eq = s_count;
  }
  __syncthreads();
  TOPK_STAMP(2);
  if (tid == 0) {
    s_count = 0;
    s_base = 0;
  }
  for (int i = tid; i < TOPK_MAXK; i += TOPK_THREADS) sel[i] = 0ull;
  __syncthreads();
  // strictly greater: all selected (there are exactly k - need_eq of them)
  const bool all_eq = (count_eq == need_eq);
  TOPK_FOR_ELEMENTS(j) {
    const int i = j * TOPK_THREADS + tid;
    const uint32_t key = TOPK_KEY(j, i);
    if (i < n && (key > T || (all_eq && key == T))) {
      uint32_t pos = atomicAdd(&s_count, 1u);
      if (pos < TOPK_MAXK) sel[pos] = ((unsigned long long)key << 32) | (uint32_t)(0xffffffffu - (uint32_t)(sidx ? sidx[i] : base + i));
    }
  }
  __syncthreads();
  if (!all_eq) {
    // surplus ties: take the need_eq lowest indices among key == T (ordered block scan)
    const int lane = tid & 63, wv = tid >> 6;
    TOPK_FOR_ELEMENTS(j) {
      const int i = j * TOPK_THREADS + tid;
      if (j * TOPK_THREADS >= n) break;   // uniform
      bool flag = (i < n) && (TOPK_KEY(j, i) == T);
      unsigned long long bal = __ballot(flag);
      uint32_t wexcl = __popcll(bal & ((1ull << lane) - 1ull));
      if (lane == 0) s_wave_tot[wv] = __popcll(bal);
      __syncthreads();
      uint32_t woff = 0, tot = 0;
      for (int w = 0; w < 16; ++w) {
        uint32_t tw = s_wave_tot[w];
        if (w < wv) woff += tw;
        tot += tw;
      }
      uint32_t rank = s_base + woff + wexcl;
      if (flag && rank < need_eq) {
        uint32_t pos = (uint32_t)(k - (int)need_eq) + rank;
        if (pos < TOPK_MAXK) sel[pos] = ((unsigned long long)T << 32) | (uint32_t)(0xffffffffu - (uint32_t)(sidx ? sidx[i] : base + i));
      }
      __syncthreads();
      if (tid == 0) s_base += tot;
      __syncthreads();
      if (s_base >= need_eq) break;
    }
    __syncthreads();
  }
  TOPK_STAMP(3);
  // Order (value descending, index ascending) = descending order of the 64-bit entries, which are distinct for real elements (the index is
  // part of them): an entry's place is the number of entries greater than it (+ the equal ones in front of it, see the loop).  One thread per (entry, slice of the list) - floor(1024 / k) slices, every
  // thread on a real entry - and the slice is read with wave-uniform addresses where a wave lies inside one slice (LDS broadcast); nothing
  // in the loop depends on a previous LDS access: 100 reads + compares per thread for k = 300.  (Until round 5 a bitonic network sorted
  // the list in LDS: 45 dependent read-compare-write steps of ~500 ticks each, 22 000 of the launch's 55 000 ticks.)
  const int parts = TOPK_THREADS / k, part = tid / k, e = tid - part * k;
  const int len = (k + parts - 1) / parts;
  s_rank[tid] = 0;
  __syncthreads();
  const unsigned long long me = sel[e];
  if (part < parts) {
    const int j0 = part * len, j1 = min(k, j0 + len);
    uint32_t r = 0;
#pragma unroll 4
    for (int j = j0; j < j1; ++j) r += (sel[j] > me || (sel[j] == me && j < e)) ? 1u : 0u;   // equal entries (the two-level form's (-inf, INT_MAX) pads): LDS position breaks the tie, the places stay a permutation
    if (r) atomicAdd(&s_rank[e], r);
  }
  __syncthreads();
  TOPK_STAMP(4);
  if (part == 0) {
    const uint32_t rank = s_rank[e];
    out_val[(int64_t)blockIdx.x * k_out + rank] = key_f32((uint32_t)(me >> 32));
    out_idx[(int64_t)blockIdx.x * k_out + rank] = (int32_t)(0xffffffffu - (uint32_t)(me & 0xffffffffull));
  }
  for (int i = k + tid; i < k_out; i += TOPK_THREADS) {   // a row shorter than k_out: padded
    out_val[(int64_t)blockIdx.x * k_out + i] = -INFINITY;
    out_idx[(int64_t)blockIdx.x * k_out + i] = 0x7fffffff;
  }
  TOPK_STAMP(5);
#undef TOPK_STAMP
}

#undef TOPK_KEY
#undef TOPK_FOR_ELEMENTS

// rows (chunks) of <= 5 120 / 9 216 / 12 288 elements keep their keys in registers; longer ones stream (FX_TOPK_REGS=0: always stream)
static void topk_launch(int grid, hipStream_t stream, const float* scores, int ld, int n_total, int chunk_len, int nchunk, int k_out,
                        const int32_t* src_idx, float* out_val, int32_t* out_idx) {
  static const int regs = fx_tune("FX_TOPK_REGS", 1);
  // diagnostic (scripts/dev/topk_stamps.py): FX_TOPK_DBG = address of 8 x u64 that receive workgroup 0's s_memtime stamps (start, keys loaded,
  // threshold found, candidates collected, sorted, written)
  static unsigned long long* const dbg = reinterpret_cast<unsigned long long*>((uintptr_t)strtoull(getenv("FX_TOPK_DBG") ? getenv("FX_TOPK_DBG") : "0", nullptr, 0));
  const int len = chunk_len < n_total ? chunk_len : n_total;
  const int npt = !regs ? 0 : (len <= 5 * TOPK_THREADS ? 5 : (len <= 9 * TOPK_THREADS ? 9 : (len <= 12 * TOPK_THREADS ? 12 : 0)));
#define TOPK_GO(N) hipLaunchKernelGGL(topk_kernel<N>, dim3(grid), dim3(TOPK_THREADS), 0, stream, scores, ld, n_total, chunk_len, nchunk, k_out, src_idx, out_val, out_idx, dbg)
  if (npt == 5) TOPK_GO(5);
  else if (npt == 9) TOPK_GO(9);
  else if (npt == 12) TOPK_GO(12);
  else TOPK_GO(0);
#undef TOPK_GO
}

extern "C" int fx_topk_rows_f32(const float* scores, int ld, int B, int n, int k, float* out_val, int32_t* out_idx, fx_stream_t stream_) {
  FX_CHECK_ARG(scores && out_val && out_idx && B > 0 && n > 0 && k > 0 && k <= n && ld >= n);
  if (k > TOPK_MAXK) return FX_ERR_UNSUPPORTED;
  topk_launch(B, reinterpret_cast<hipStream_t>(stream_), scores, ld, n, n, 1, k, nullptr, out_val, out_idx);
  return fx_launch_status();
}

// Two-level form for long rows (the post-process top-k over Q*K = 109 500 scores per image runs as 32 workgroups in the one-level
// form): level 1 = exact top-k of every 8192-element chunk (B * ceil(n / 8192) workgroups), level 2 = exact top-k of the <= 14 * k
// candidates per image.  The global top-k is a subset of the union of the chunk top-ks, and tie order is preserved (see topk_kernel).
#define TOPK_CHUNK 8192
extern "C" size_t fx_topk_rows_workspace_bytes(int B, int n, int k) {
  if (B <= 0 || n <= 0 || k <= 0) return 0;
  const int nchunk = (n + TOPK_CHUNK - 1) / TOPK_CHUNK;
  return nchunk > 1 ? (size_t)B * nchunk * k * 8 : 0;
}

extern "C" int fx_topk_rows_ws_f32(const float* scores, int ld, int B, int n, int k, float* out_val, int32_t* out_idx, void* workspace,
                                   size_t workspace_bytes, fx_stream_t stream_) {
  const size_t need = fx_topk_rows_workspace_bytes(B, n, k);
  if (need == 0 || k > TOPK_CHUNK) return fx_topk_rows_f32(scores, ld, B, n, k, out_val, out_idx, stream_);
  FX_CHECK_ARG(scores && out_val && out_idx && workspace && workspace_bytes >= need && k <= n && ld >= n && ((uintptr_t)workspace % 4) == 0);
  if (k > TOPK_MAXK) return FX_ERR_UNSUPPORTED;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const int nchunk = (n + TOPK_CHUNK - 1) / TOPK_CHUNK;
  float* cval = reinterpret_cast<float*>(workspace);
  int32_t* cidx = reinterpret_cast<int32_t*>(cval + (size_t)B * nchunk * k);
  topk_launch(B * nchunk, stream, scores, ld, n, TOPK_CHUNK, nchunk, k, nullptr, cval, cidx);
  topk_launch(B, stream, (const float*)cval, nchunk * k, nchunk * k, nchunk * k, 1, k, (const int32_t*)cidx, out_val, out_idx);
  return fx_launch_status();
}

// ------------------------------------------------------------------------------------------------
__global__ FX_SEL_PK(2) __launch_bounds__(256) void gather_rows_kernel(const bf16_t* __restrict__ src, int lds, int rpb, const int32_t* __restrict__ idx,
                                                           int k, bf16_t* __restrict__ out, int ldo, int B, int C8) {
  int64_t total = (int64_t)B * k * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int c8 = (int)(i % C8);
    int64_t r = i / C8;
    int b = (int)(r / k);
    int s = idx[r];
    *reinterpret_cast<uint4*>(out + r * ldo + c8 * 8) = *reinterpret_cast<const uint4*>(src + ((int64_t)b * rpb + s) * lds + c8 * 8);
  }
}

extern "C" int fx_gather_rows_bf16(const void* src, int lds, int rows_per_batch, const int32_t* idx, int k, void* out, int ldo, int B, int cols,
                                   fx_stream_t stream_) {
  FX_CHECK_ARG(src && idx && out && B > 0 && k > 0 && cols > 0 && cols % 8 == 0 && lds >= cols && ldo >= cols && lds % 8 == 0 && ldo % 8 == 0);
  int64_t total = (int64_t)B * k * (cols / 8);
  int grid = (int)((total + 255) / 256);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), (const bf16_t*)src, lds,
                     rows_per_batch, idx, k, (bf16_t*)out, ldo, B, cols / 8);
  return fx_launch_status();
}

__global__ FX_SEL_PK(3) __launch_bounds__(256) void fill_rows_kernel(bf16_t* __restrict__ x, int ldx, int rpb, const int32_t* __restrict__ rows_idx, int n_idx,
                                                         const bf16_t* __restrict__ rowv, int B, int C8) {
  int64_t total = (int64_t)B * n_idx * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int c8 = (int)(i % C8);
    int64_t r = i / C8;
    int b = (int)(r / n_idx), j = (int)(r % n_idx);
    *reinterpret_cast<uint4*>(x + ((int64_t)b * rpb + rows_idx[j]) * ldx + c8 * 8) = *reinterpret_cast<const uint4*>(rowv + c8 * 8);
  }
}

extern "C" int fx_fill_rows_bf16(void* x, int ldx, int rows_per_batch, const int32_t* rows_idx, int n_idx, const void* row_bf16, int B, int cols,
                                 fx_stream_t stream_) {
  FX_CHECK_ARG(x && row_bf16 && B > 0 && cols > 0 && cols % 8 == 0 && ldx >= cols && ldx % 8 == 0 && n_idx >= 0);
  if (n_idx == 0) return FX_OK;
  FX_CHECK_ARG(rows_idx != nullptr);
  int64_t total = (int64_t)B * n_idx * (cols / 8);
  int grid = (int)((total + 255) / 256);
  hipLaunchKernelGGL(fill_rows_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), (bf16_t*)x, ldx, rows_per_batch,
                     rows_idx, n_idx, (const bf16_t*)row_bf16, B, cols / 8);
  return fx_launch_status();
}

// ------------------------------------------------------------------------------------------------
// relu(ref[r,0:4] @ W^T + b): K = 4 is too thin for MFMA; 8 outputs per lane, fully coalesced stores.
__global__ FX_SEL_PK(4) __launch_bounds__(256) void linear_k4_relu_kernel(const float* __restrict__ ref, const float* __restrict__ w,
                                                              const float* __restrict__ bias, bf16_t* __restrict__ out, int ldo, int rows, int N8) {
  int64_t total = (int64_t)rows * N8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int n8 = (int)(i % N8);
    int r = (int)(i / N8);
    float4 x = *reinterpret_cast<const float4*>(ref + (int64_t)r * 4);
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float4 wv = *reinterpret_cast<const float4*>(w + (int64_t)(n8 * 8 + j) * 4);
      o[j] = fmaxf(bias[n8 * 8 + j] + x.x * wv.x + x.y * wv.y + x.z * wv.z + x.w * wv.w, 0.0f);
    }
    *reinterpret_cast<uint4*>(out + (int64_t)r * ldo + n8 * 8) = pack_bf16x8(o);
  }
}

extern "C" int fx_linear_k4_relu(const float* ref, const float* w, const float* b, void* out, int ldo, int rows, int N, fx_stream_t stream_) {
  FX_CHECK_ARG(ref && w && b && out && rows > 0 && N > 0 && N % 8 == 0 && ldo >= N && ldo % 8 == 0);
  int64_t total = (int64_t)rows * (N / 8);
  int grid = (int)((total + 255) / 256);
  hipLaunchKernelGGL(linear_k4_relu_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), ref, w, b, (bf16_t*)out, ldo,
                     rows, N / 8);
  return fx_launch_status();
}

// N = 4 projection (one wave per row) fused with the reference-box update.
__device__ __forceinline__ float inv_sigmoid(float x) {
  x = fminf(fmaxf(x, 0.0f), 1.0f);
  return __logf(fmaxf(x, 1e-5f) / fmaxf(1.0f - x, 1e-5f));
}

__global__ FX_SEL_PK(5) __launch_bounds__(256) void bbox_head_kernel(const bf16_t* __restrict__ h, int ldh, const float* __restrict__ w,
                                                         const float* __restrict__ bias, const float* __restrict__ ref,
                                                         const float* __restrict__ anchors, const int32_t* __restrict__ idx, int rpb, int mode,
                                                         float* __restrict__ new_ref, float* __restrict__ unact_out, int rows, int K) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c = lane * 4; c < K; c += 256) {
    uint2 hv = *reinterpret_cast<const uint2*>(h + (int64_t)row * ldh + c);
    float x[4] = {bf16lo_to_f32(hv.x), bf16hi_to_f32(hv.x), bf16lo_to_f32(hv.y),
                  bf16hi_to_f32(hv.y)};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float4 wv = *reinterpret_cast<const float4*>(w + (int64_t)j * K + c);
      acc[j] += x[0] * wv.x + x[1] * wv.y + x[2] * wv.z + x[3] * wv.w;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc[j] += __shfl_xor(acc[j], o, 64);
  if (lane < 4) {
    float v = (lane == 0 ? acc[0] : lane == 1 ? acc[1] : lane == 2 ? acc[2] : acc[3]) + bias[lane];
    float u;
    if (mode == 0) {
      u = v + inv_sigmoid(ref[(int64_t)row * 4 + lane]);
    } else {
      (void)rpb;
      u = v + anchors[(int64_t)idx[row] * 4 + lane];
    }
    if (unact_out) unact_out[(int64_t)row * 4 + lane] = u;
    new_ref[(int64_t)row * 4 + lane] = 1.0f / (1.0f + __expf(-u));
  }
}

extern "C" int fx_bbox_head(const void* h, int ldh, const float* w, const float* b, const float* ref, const float* anchors, const int32_t* idx,
                            int rows_per_batch, int mode, float* new_ref, float* unact_out, int rows, int K, fx_stream_t stream_) {
  FX_CHECK_ARG(h && w && b && new_ref && rows > 0 && K > 0 && K % 4 == 0 && ldh >= K && ldh % 4 == 0);
  FX_CHECK_ARG((mode == 0 && ref) || (mode == 1 && anchors && idx));
  hipLaunchKernelGGL(bbox_head_kernel, dim3((rows + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), (const bf16_t*)h, ldh, w, b,
                     ref, anchors, idx, rows_per_batch, mode, new_ref, unact_out, rows, K);
  return fx_launch_status();
}

// ------------------------------------------------------------------------------------------------
__global__ FX_SEL_PK(6) __launch_bounds__(256) void detr_head_out_kernel(const float* __restrict__ logits, int ldl, const float* __restrict__ ref,
                                                             float* __restrict__ probs, float* __restrict__ boxes, int rows, int K) {
  int64_t total = (int64_t)rows * K;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int c = (int)(i % K);
    int64_t r = i / K;
    float x = logits[r * ldl + c];
    probs[i] = 1.0f / (1.0f + expf(-x));
    if (c < 4) {
      float cx = ref[r * 4 + 0], cy = ref[r * 4 + 1], w = ref[r * 4 + 2], hh = ref[r * 4 + 3];
      float v = c == 0 ? cx - 0.5f * w : c == 1 ? cy - 0.5f * hh : c == 2 ? cx + 0.5f * w : cy + 0.5f * hh;
      boxes[r * 4 + c] = v;
    }
  }
}

extern "C" int fx_detr_head_out(const float* logits, int ldl, const float* ref_cxcywh, float* probs, float* boxes_xyxy, int rows, int K,
                                fx_stream_t stream_) {
  FX_CHECK_ARG(logits && ref_cxcywh && probs && boxes_xyxy && rows > 0 && K >= 4 && ldl >= K);
  int64_t total = (int64_t)rows * K;
  int64_t grid = (total + 255) / 256;
  if (grid > 256 * 16) grid = 256 * 16;
  hipLaunchKernelGGL(detr_head_out_kernel, dim3((int)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), logits, ldl, ref_cxcywh,
                     probs, boxes_xyxy, rows, K);
  return fx_launch_status();
}

// DETRProcessor.postprocess tail: label/query split, box scale + round-half-even (torch.round) -> int32,
// count of scores above the threshold (scores are sorted, so the kept ones form a prefix).
__global__ FX_SEL_PK(7) __launch_bounds__(256) void detr_postprocess_kernel(const float* __restrict__ tv, const int32_t* __restrict__ ti,
                                                                const float* __restrict__ boxes, const int32_t* __restrict__ sizes, int Q, int K,
                                                                int top_k, float thr, int32_t* __restrict__ labels, int32_t* __restrict__ queries,
                                                                int32_t* __restrict__ obox, int32_t* __restrict__ count) {
  const int b = blockIdx.x;
  __shared__ int s_cnt;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  const float Hs = (float)sizes[2 * b], Ws = (float)sizes[2 * b + 1];
  int local = 0;
  for (int i = threadIdx.x; i < top_k; i += 256) {
    int64_t o = (int64_t)b * top_k + i;
    int id = ti[o];
    int lab = id % K, qi = id / K;
    labels[o] = lab;
    queries[o] = qi;
    const float* bx = boxes + ((int64_t)b * Q + qi) * 4;
    obox[o * 4 + 0] = (int32_t)rintf(bx[0] * Ws);
    obox[o * 4 + 1] = (int32_t)rintf(bx[1] * Hs);
    obox[o * 4 + 2] = (int32_t)rintf(bx[2] * Ws);
    obox[o * 4 + 3] = (int32_t)rintf(bx[3] * Hs);
    if (tv[o] > thr) ++local;
  }
  atomicAdd(&s_cnt, local);
  __syncthreads();
  if (threadIdx.x == 0) count[b] = s_cnt;
}

extern "C" int fx_detr_postprocess(const float* topk_val, const int32_t* topk_idx, const float* boxes_xyxy, const int32_t* sizes, int B, int Q, int K,
                                   int top_k, float threshold, int32_t* labels, int32_t* queries, int32_t* boxes_i32, int32_t* count,
                                   fx_stream_t stream_) {
  FX_CHECK_ARG(topk_val && topk_idx && boxes_xyxy && sizes && labels && queries && boxes_i32 && count && B > 0 && Q > 0 && K > 0 && top_k > 0);
  hipLaunchKernelGGL(detr_postprocess_kernel, dim3(B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), topk_val, topk_idx, boxes_xyxy, sizes,
                     Q, K, top_k, threshold, labels, queries, boxes_i32, count);
  return fx_launch_status();
}
