// Row-local layer chains of the transformer decoder as ONE launch per chain (gfx950).
//
// A decoder layer of RT-DETR (fai_detr/modelling.py:924-1020) is ~19 launches in the one-kernel-per-op form, most of them
// 2-12 us GEMMs / LayerNorms / adds over the same [B*300, 256] rows, each paying a launch boundary and an HBM round trip of
// its tiny activation.  Everything except the two attention cores (self-attention mixes the 300 rows of an image, deformable
// attention gathers from the memory) is ROW-LOCAL: a workgroup that owns 32 rows can run the whole chain with the activations in
// LDS.  This kernel is an interpreter for such chains: the host hands it a small program (fx_rc_stage[]), every stage reads
// and writes [32][K] bf16 LDS buffers (XOR-swizzled, rows of K*2 bytes) and optionally loads / stores global rows:
//   RC_LOAD     global bf16 rows -> LDS
//   RC_GEMM     Y = act(X . W^T + b), N a multiple of 32, to LDS and / or global (bf16 or f32)
//   RC_GEMM_LN  Y = LayerNorm_256(X . W^T + b + residual) * gamma + beta  (statistics on the fp32 accumulators)
//   RC_ADD      Y = X1 + X2
//   RC_K4       Y = relu(ref[.,4] . W^T + b)            (query_pos_head layer 0, modelling.py:996)
//   RC_BBOX     ref' = sigmoid(X . W4^T + b + inverse_sigmoid(ref))   (bbox head tail + refinement, modelling.py:1000-1003)
//   RC_LN       Y = LayerNorm_256(X) * gamma + beta               (the pre-norm layers of the masked-attention decoders, round 4)
//   RC_FFN_LN   Y = LayerNorm_256(relu(X . W1^T + b1) . W2^T + b2 + residual) * gamma + beta   (round 5: the whole feed-forward block;
//               the hidden layer never exists as a [32][N] buffer - it is produced in 256-column chunks into two ping-pong [32][256]
//               slots while the second GEMM accumulates in registers across the chunks, so a decoder chain needs four 16 KiB slots
//               instead of four + a 64 KiB one: 77.5 KiB of LDS, TWO workgroups per CU)
// Round 5 also: RC_GEMM can read a K = 512 operand from two [32][256] slots (flags bit 2: columns >= 256 from `aux`), RC_K4 can write
// its N = 512 result into two such slots (flags bit 2: channels >= 256 to the slot at `ld2`) - the query-position MLP without the wide slot.
// GEMMs: 8 waves, a wave owns 32 output channels per pass (one 32x32 accumulator block over the 32 rows), weights = MFMA A
// operand in fragment order straight from L2 through an 8-deep register ring (loads hidden from hipcc's waitcnt bookkeeping, see
// pw_common.h), rows = B operand from LDS.  With 32 rows a fragment is used by one MFMA only: the chain is bound by the
// weight stream from L2 (~2 MB per workgroup and layer), not by the matrix cores - which is still 5-10x less time than the
// launch-bound form.
#include <stdlib.h>

#include "pw_common.h"

enum { RC_LOAD = 0, RC_GEMM = 1, RC_GEMM_LN = 2, RC_ADD = 3, RC_K4 = 4, RC_BBOX = 5, RC_LN = 6, RC_FFN_LN = 7 };

__device__ __forceinline__ float rc_inv_sigmoid(float x) {
  x = fminf(fmaxf(x, 0.0f), 1.0f);
  return __logf(fmaxf(x, 1e-5f) / fmaxf(1.0f - x, 1e-5f));
}

// byte offset of 16-byte chunk `chunk` of row `row` in a [32][K] buffer (K >= 128: 16 chunks per 256-byte bank window)
__device__ __forceinline__ int rc_off(int row, int chunk, int rowb) { return row * rowb + ((chunk ^ (row & 15)) << 4); }

// Round 5 (two workgroups per CU = 128 registers per wave): lean addressing of the GEMM stages.
// (1) A lane's B-operand fragment of k-step ks in a [32][K] slot (K a power of two, slot offset a multiple of the row pitch) sits at
//     rc_off(l32, 2 ks + half, 2K) = PRE ^ (ks << 5) with the per-lane constant PRE = slot + l32 * 2K + ((half ^ (l32 & 15)) << 4): the
//     row base has zeros where ks << 5 and the swizzle live, so the XOR swizzle and the k-step commute - one v_xor per LDS read
//     instead of one address register per unrolled step.
__device__ __forceinline__ int rc_xpre(int slot, int l32, int half, int rowb) { return slot + l32 * rowb + ((half ^ (l32 & 15)) << 4); }
__device__ __forceinline__ bf16x8 rc_xread(const unsigned char* smem, int pre, int ks) { return *reinterpret_cast<const bf16x8*>(smem + (pre ^ (ks << 5))); }
// (2) Weight fragments by (scalar base pointer, 32-bit lane offset + wave-uniform fragment offset): one VGPR per load instead of a
//     64-bit pointer per ring slot.  Same contract as c3_ldg_async (asm load on a read-write operand, counted waits).
__device__ __forceinline__ void rc_ldg_async(bf16x8& dst, const void* sbase, unsigned voff) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
// the stream form: a wave's fragments are consecutive 1 KiB blocks, the offset register runs along (one v_add per load; precomputed
// per-step offsets cost a scalar or vector register each and were what spilled under the 128-register budget)
__device__ __forceinline__ void rc_ldg_next(bf16x8& dst, const void* sbase, unsigned& voff) {
  asm volatile("global_load_dwordx4 %0, %1, %2\n\tv_add_u32 %1, 0x400, %1" : "+v"(dst), "+v"(voff) : "s"(sbase) : "memory");
}


// Tail of the LayerNorm-fused GEMM stages: acc = this wave's 32 channels [32*wave, +32) of the 32 rows (lane = row l32, channels
// 8*gq + 4*half + 0..3 per accumulator quad); + residual (bf16 LDS [32][256], aux >= 0), two-pass statistics over the row's 256 channels
// (2 lanes x 8 waves, through the `red` scratch), affine, bf16 to the LDS slot `dst` and optionally to global rows.
__device__ __forceinline__ void rc_ln_tail(f32x16& acc, const fx_rc_stage& st, unsigned char* smem, float* red, int wave, int l32, int half, int m0, int M) {
  if (st.aux >= 0) {
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int n = wave * 32 + 8 * gq + 4 * half;
      const uint2 rv = *reinterpret_cast<const uint2*>(smem + st.aux + rc_off(l32, n >> 3, 512) + half * 8);
      acc[4 * gq] += bf16lo_to_f32(rv.x);
      acc[4 * gq + 1] += bf16hi_to_f32(rv.x);
      acc[4 * gq + 2] += bf16lo_to_f32(rv.y);
      acc[4 * gq + 3] += bf16hi_to_f32(rv.y);
    }
  }
  float s = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += acc[r];
  s += __shfl_xor(s, 32);
  __syncthreads();   // every wave is past its reads of src / aux and of `red`
  if (half == 0) red[wave * 32 + l32] = s;
  __syncthreads();
  float mean = 0.0f;
#pragma unroll
  for (int wv = 0; wv < 8; ++wv) mean += red[wv * 32 + l32];
  mean *= (1.0f / 256.0f);
  s = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float d = acc[r] - mean;
    s += d * d;
  }
  s += __shfl_xor(s, 32);
  __syncthreads();
  if (half == 0) red[wave * 32 + l32] = s;
  __syncthreads();
  float var = 0.0f;
#pragma unroll
  for (int wv = 0; wv < 8; ++wv) var += red[wv * 32 + l32];
  const float rstd = rsqrtf(var * (1.0f / 256.0f) + 1e-5f);
  const int m = m0 + l32;
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    const int n = wave * 32 + 8 * gq + 4 * half;
    const float4 gg = *reinterpret_cast<const float4*>(st.gamma + n), be = *reinterpret_cast<const float4*>(st.beta + n);
    uint2 o;
    o.x = pack_bf16x2((acc[4 * gq] - mean) * rstd * gg.x + be.x, (acc[4 * gq + 1] - mean) * rstd * gg.y + be.y);
    o.y = pack_bf16x2((acc[4 * gq + 2] - mean) * rstd * gg.z + be.z, (acc[4 * gq + 3] - mean) * rstd * gg.w + be.w);
    *reinterpret_cast<uint2*>(smem + st.dst + rc_off(l32, n >> 3, 512) + half * 8) = o;
    if (st.g0 && m < M) *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(st.g0) + (size_t)m * st.ld + n) = o;
  }
}

// bias_off: LDS byte offset of a 12 KiB scratch behind the program's own buffers (round 3).  Per-stage s_memtime stamps of the post-MSDA
// chain (scripts/dev/rc_stage_stamps.py; 136k cycles): the GEMM stages stream 2 MB of weights per workgroup at ~8.7 TB/s of aggregate L2
// bandwidth (150 workgroups pulling the same fragments - that is their bound: a 16-deep ring changed nothing), but the two VALU stages
// K4 and BBOX took 19k cycles EACH for ~130 kFLOP: every thread walked its fp32 weights with dependent global loads.  Both now copy
// their weights into the scratch with one coalesced pass and compute from LDS (K4 with lane = row, so that a wave reads two
// addresses per instruction: broadcasts).  A GEMM stage keeps its bias vector there (no global load per 32-channel pass).
#define RC_SCRATCH 12288
#define RC_RING 8    // weight fragments in flight per wave (16 measured the same: the stages are not waiting on this stream's round trips)
__global__ __launch_bounds__(512, 4) void row_chain_kernel(const fx_rc_stage* __restrict__ prog, int nstages, int M, int bias_off, unsigned long long* dbg,
                                                              int prefetch) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* biasL = reinterpret_cast<float*>(smem + bias_off);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m0 = blockIdx.x * 32;

  // Weight prefetch into this XCD's L2 (round 5).  scripts/dev/rc_scaling_probe.py: ONE 32-row workgroup alone on an idle chip takes the same 37 us
  // for the post-MSDA chain as 150 together (41 us) - the chain is not bound by anything the workgroups share but by its own dependency chain:
  // a 128 KiB GEMM stage is two ring-fulls per wave = two exposed round trips to wherever the weights live, and a decoder's 14 MB of weights do
  // not live in a 4 MiB L2 from one launch to the next (5.4-6.7 k cycles per such stage for 0.5 k cycles of MFMA work).  So at kernel start the
  // workgroups of an XCD (blockIdx % 8: they share one L2) split ALL GEMM weights of the program among themselves and touch them - ~100 KiB each at
  // 150 workgroups - into one throw-away register, wait once, and every stage's ring then fills from L2.
  if (prefetch) {
    const int lane = threadIdx.x & 63;
    const int wg_in_xcd = blockIdx.x >> 3, n_in_xcd = ((int)gridDim.x - (int)(blockIdx.x & 7) + 7) >> 3;
    bf16x8 sink = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    for (int si = 0; si < nstages; ++si) {
      const fx_rc_stage st = prog[si];
      int chunks = 0;   // 1 KiB pieces (one wave instruction each) of the stage's fragment-order weights
      if (st.type == RC_GEMM) chunks = (st.N * st.K) >> 9;
      else if (st.type == RC_GEMM_LN) chunks = (256 * st.K) >> 9;
      else if (st.type == RC_FFN_LN) chunks = (st.N * 256) >> 9;
      for (int c = wg_in_xcd + n_in_xcd * wave; c < chunks; c += n_in_xcd * 8) {
        rc_ldg_async(sink, st.w, (unsigned)(c * 1024 + lane * 16));
        if (st.type == RC_FFN_LN) rc_ldg_async(sink, st.g1, (unsigned)(c * 1024 + lane * 16));
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink));
  }
  if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dbg[0] = __builtin_amdgcn_s_memtime();
  for (int si = 0; si < nstages; ++si) {
    // The lane id is made opaque PER STAGE: every per-lane address constant of a stage is derived from it inside the loop body, so the
    // optimiser cannot hoist the stages' invariants out of the stage loop and keep ALL of them live at once (round 5: 218 registers for
    // the union of stages that need 14-110 each; the two-workgroups-per-CU form has 128).
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, l32 = lane & 31, half = lane >> 5;
    if (dbg && blockIdx.x == 0 && tid == 0) dbg[1 + si] = __builtin_amdgcn_s_memtime();
    const fx_rc_stage st = prog[si];   // wave-uniform: scalar loads
    const int K = st.K, N = st.N;
    if (st.type == RC_LOAD) {
      const bf16_t* g = reinterpret_cast<const bf16_t*>(st.g0);
      const int cpr = K >> 3;
      for (int q = tid; q < 32 * cpr; q += 512) {
        const int row = q / cpr, c = q - row * cpr;
        const int m = m0 + row;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (m < M) v = *reinterpret_cast<const uint4*>(g + (size_t)m * st.ld + c * 8);
        *reinterpret_cast<uint4*>(smem + st.dst + rc_off(row, c, K * 2)) = v;
      }
    } else if (st.type == RC_ADD) {
      const int cpr = K >> 3;
      for (int q = tid; q < 32 * cpr; q += 512) {
        const int o = q << 4;   // same swizzle in all three buffers: element-wise on the physical layout
        const uint4 a = *reinterpret_cast<const uint4*>(smem + st.src + o), b = *reinterpret_cast<const uint4*>(smem + st.aux + o);
        float fa[8], fb[8];
        unpack_bf16x8(a, fa);
        unpack_bf16x8(b, fb);
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[i] += fb[i];
        const uint4 pk = pack_bf16x8(fa);
        *reinterpret_cast<uint4*>(smem + st.dst + o) = pk;
        if (st.g0) {   // also to global (bf16 rows of stride ld): undo the swizzle of the physical chunk index
          const int row = q / cpr, pc = q - row * cpr;
          const int c = (pc & ~15) | ((pc ^ (row & 15)) & 15);
          const int m = m0 + row;
          if (m < M) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(st.g0) + (size_t)m * st.ld + c * 8) = pk;
        }
      }
    } else if (st.type == RC_LN) {
      // 16 threads per row, 16 channels each (two 16-byte chunks); fp32 statistics, two passes, the row's partial sums meet by xor-shuffles
      const int row = tid >> 4, t16 = tid & 15;
      float v[16];
#pragma unroll
      for (int c = 0; c < 2; ++c) unpack_bf16x8(*reinterpret_cast<const uint4*>(smem + st.src + rc_off(row, 2 * t16 + c, 512)), v + 8 * c);
      float sum = 0.0f;
#pragma unroll
      for (int i = 0; i < 16; ++i) sum += v[i];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) sum += __shfl_xor(sum, o, 64);
      const float mean = sum * (1.0f / 256.0f);
      float var = 0.0f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        v[i] -= mean;
        var += v[i] * v[i];
      }
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) var += __shfl_xor(var, o, 64);
      const float rstd = rsqrtf(var * (1.0f / 256.0f) + 1e-5f);
      const int m = m0 + row;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int n = (2 * t16 + c) * 8;
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = v[8 * c + i] * rstd * st.gamma[n + i] + st.beta[n + i];
        const uint4 pk = pack_bf16x8(o);
        if (st.dst >= 0) *reinterpret_cast<uint4*>(smem + st.dst + rc_off(row, 2 * t16 + c, 512)) = pk;
        if (st.g0 && m < M) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(st.g0) + (size_t)m * st.ld + n) = pk;
      }
    } else if (st.type == RC_K4) {
      // Y[row][n] = relu(b[n] + sum_k ref[row][k] * W[n][k]); ref from the LDS hand-over area (aux >= 0) or from global.  N <= 512.
      float4* wL = reinterpret_cast<float4*>(smem + bias_off);            // [N] rows of 4 weights
      float* bL = reinterpret_cast<float*>(smem + bias_off + 8192);       // [N]
      for (int n = tid; n < N; n += 512) {
        wL[n] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(st.w) + (size_t)n * 4);
        bL[n] = st.bias[n];
      }
      const int m = m0 + l32;
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
      if (st.aux >= 0) r = *reinterpret_cast<const float4*>(smem + st.aux + l32 * 16);
      else if (m < M) r = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(st.g0) + (size_t)m * 4);
      __syncthreads();
      // lane = row; (wave, half, it) -> 8-channel chunk c: the weights of a chunk are the same for all 32 rows (two addresses per wave)
      const int cpr = N >> 3;
      for (int c = wave * 2 + half; c < cpr; c += 16) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float4 wv = wL[c * 8 + i];
          v[i] = fmaxf(bL[c * 8 + i] + r.x * wv.x + r.y * wv.y + r.z * wv.z + r.w * wv.w, 0.0f);
        }
        if (st.flags & 4) *reinterpret_cast<uint4*>(smem + (c < 32 ? st.dst : st.ld2) + rc_off(l32, c & 31, 512)) = pack_bf16x8(v);   // two [32][256] slots
        else *reinterpret_cast<uint4*>(smem + st.dst + rc_off(l32, c, N * 2)) = pack_bf16x8(v);
      }
    } else if (st.type == RC_BBOX) {
      // 32 rows x 4 outputs, K = 256: 8 threads per row, 32 channels each
      float* wLb = reinterpret_cast<float*>(smem + bias_off);   // [4][K] fp32, K = 256
      for (int q = tid; q < K; q += 512) *reinterpret_cast<float4*>(wLb + q * 4) = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(st.w) + q * 4);
      __syncthreads();
      const float* w = wLb;
      const int row = (tid >> 3) & 31, part = tid & 7;   // threads 256..511 recompute rows 0..31 and store nothing
      float acc4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const int c = part * 4 + cc;
        float x[8];
        unpack_bf16x8(*reinterpret_cast<const uint4*>(smem + st.src + rc_off(row, c, K * 2)), x);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 w0 = *reinterpret_cast<const float4*>(w + (size_t)j * K + c * 8), w1 = *reinterpret_cast<const float4*>(w + (size_t)j * K + c * 8 + 4);
          acc4[j] += x[0] * w0.x + x[1] * w0.y + x[2] * w0.z + x[3] * w0.w + x[4] * w1.x + x[5] * w1.y + x[6] * w1.z + x[7] * w1.w;
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc4[j] += __shfl_xor(acc4[j], 1);
        acc4[j] += __shfl_xor(acc4[j], 2);
        acc4[j] += __shfl_xor(acc4[j], 4);
      }
      const int m = m0 + row;
      if (part < 4 && tid < 256) {
        const float v = (part == 0 ? acc4[0] : part == 1 ? acc4[1] : part == 2 ? acc4[2] : acc4[3]) + st.bias[part];
        float rin = 0.5f;
        if (m < M) rin = reinterpret_cast<const float*>(st.g0)[(size_t)m * 4 + part];
        const float u = v + rc_inv_sigmoid(rin);
        const float nr = 1.0f / (1.0f + __expf(-u));
        if (m < M) reinterpret_cast<float*>(st.g1)[(size_t)m * 4 + part] = nr;
        if (st.aux >= 0) reinterpret_cast<float*>(smem + st.aux)[row * 4 + part] = nr;
      }
    } else if (st.type == RC_GEMM) {
      // wave w computes the 32-channel blocks nb = w, w + 8, ...; its weight fragments form ONE stream over all its passes
      // (index i -> block i / KS, k-step i % KS) fed through an 8-deep register ring: 8 KiB in flight per wave, 64 KiB per CU -
      // the chain is bound by this stream from L2, so depth is what buys time
      const int nblk = N >> 5, KS = K >> 4, ksh = 31 - __clz(KS);
      const bool split = (st.flags & 4) != 0;   // K = 512 operand in two [32][256] slots: columns < 256 in src, the rest in aux
      const int rowb = split ? 512 : K * 2;
      // Every wave runs at least one pass (round 5): a wave beyond the last block (N < 256) recomputes block wave % nblk and stores
      // nothing - the stage is then ONE straight path for all waves; with the ring set up and used inside `if (npass > 0)` the optimiser
      // carried a phantom copy of the eight ring registers around the whole stage loop (32 of the 128 registers).
      const bool store_ok = wave < nblk;
      const int wv = store_ok ? wave : wave % nblk;
      const int npass = store_ok ? (nblk - wave + 7) >> 3 : 1;
      const unsigned lane16 = lane * 16;
      // stream of this wave: pass ps = fragments (wave + 8 ps) * KS + 0 .. KS-1, consecutive 1 KiB blocks; between passes the offset
      // jumps over the other seven waves' blocks.  `voff` = offset of the NEXT fragment to request.
      unsigned voff = lane16 + (unsigned)((wv * KS) << 10);
      const unsigned pass_jump = (unsigned)((7 * KS) << 10);
      (void)ksh;
      bf16x8 ar[RC_RING];
      // the ring's first fragments go out before the bias copy: one L2 round trip covers both
#pragma unroll
      for (int i = 0; i < RC_RING; ++i) {
        ar[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        rc_ldg_next(ar[i], st.w, voff);   // (KS >= 8: the first eight fragments lie in pass 0)
      }
      for (int n = tid; n < N; n += 512) biasL[n] = st.bias ? st.bias[n] : 0.0f;
      __syncthreads();
      {
        f32x16 acc;
        auto init_acc = [&](int nb) {
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const float4 bb = *reinterpret_cast<const float4*>(biasL + nb * 32 + 8 * gq + 4 * half);
            acc[4 * gq] = bb.x; acc[4 * gq + 1] = bb.y; acc[4 * gq + 2] = bb.z; acc[4 * gq + 3] = bb.w;
          }
        };
        auto epilogue = [&](int nb) {   // lane = row l32, channels nb*32 + 8*gq + 4*half + 0..3
          const int m = m0 + l32;
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            float v0 = acc[4 * gq], v1 = acc[4 * gq + 1], v2 = acc[4 * gq + 2], v3 = acc[4 * gq + 3];
            if (st.act == FX_ACT_RELU) {
              v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); v2 = fmaxf(v2, 0.0f); v3 = fmaxf(v3, 0.0f);
            }
            const int n = nb * 32 + 8 * gq + 4 * half;
            uint2 o;
            o.x = pack_bf16x2(v0, v1);
            o.y = pack_bf16x2(v2, v3);
            if (st.dst >= 0 && store_ok) *reinterpret_cast<uint2*>(smem + st.dst + rc_off(l32, n >> 3, N * 2) + half * 8) = o;
            if (st.g0 && m < M && store_ok) {
              if (st.flags & 1) *reinterpret_cast<float4*>(reinterpret_cast<float*>(st.g0) + (size_t)m * st.ld + n) = make_float4(v0, v1, v2, v3);
              else *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(st.g0) + (size_t)m * st.ld + n) = o;
            }
          }
        };
        // The stream's LAST group of eight fragments is peeled off the loops: nothing is left to request behind it, so its waits count
        // down (loads return in order) and the stage ends without draining eight throw-away re-reads - one L2 round trip per stage.
        // (Peeled, not branched: two paths that both rewrite the ring registers would meet in a phi, and a register copy of a fragment
        // whose load is still in flight reads garbage - hipcc does not know about the asm loads.)
        // activation fragment of k-step ks (k-steps wrap: the read issued behind a pass's last MFMA is the next pass's first)
        const int pre0 = rc_xpre(st.src, l32, half, rowb), pre1 = split ? rc_xpre(st.aux, l32, half, rowb) : pre0;
        auto xfrag = [&](int ks) {
          ks &= KS - 1;
          return (split && ks >= 16) ? rc_xread(smem, pre1, ks & 15) : rc_xread(smem, pre0, ks);
        };
        bf16x8 xb[2];
        xb[0] = xfrag(0);
        for (int ps = 0; ps < npass; ++ps) {
          const int nb = wv + ps * 8;
          init_acc(nb);
          const int kend = ps == npass - 1 ? KS - RC_RING : KS;
#pragma unroll 1
          for (int ks0 = 0; ks0 < kend; ks0 += RC_RING) {
            if (ks0 + RC_RING == KS) voff += pass_jump;   // this group's refills are the next pass's first eight fragments
            auto step = [&](auto ic) {
              constexpr int i = decltype(ic)::value;
              xb[(i + 1) & 1] = xfrag(ks0 + i + 1);   // next k-step's rows (wraps to k-step 0 for the next pass), one MFMA ahead of its use
              c3_wait<RC_RING - 1>(ar[i]);
              acc = FX_MFMA_32x32x16(ar[i], xb[i & 1], acc);
              rc_ldg_next(ar[i], st.w, voff);
            };
            c3_static_for<RC_RING>(step);
          }
          if (ps < npass - 1) epilogue(nb);
        }
        {
          auto step_tail = [&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (i + 1 < RC_RING) xb[(i + 1) & 1] = xfrag(KS - RC_RING + i + 1);
            c3_wait<RC_RING - 1 - i>(ar[i]);
            acc = FX_MFMA_32x32x16(ar[i], xb[i & 1], acc);
          };
          c3_static_for<RC_RING>(step_tail);
          epilogue(wv + (npass - 1) * 8);
        }
      }
    } else if (st.type == RC_GEMM_LN) {
      // N = 256: wave -> channels [32*wave, +32); LayerNorm over the row on the fp32 accumulators
      const int KS = K >> 4, rowb = K * 2;
      float* red = reinterpret_cast<float*>(smem + st.ld2);   // [8 waves][32 rows] reduction scratch (LDS byte offset in ld2)
      unsigned voff = lane * 16 + (unsigned)((wave * KS) << 10);
      bf16x8 ar[RC_RING];
#pragma unroll
      for (int i = 0; i < RC_RING; ++i) {
        ar[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        rc_ldg_next(ar[i], st.w, voff);
      }
      f32x16 acc;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const float4 bb = *reinterpret_cast<const float4*>(st.bias + wave * 32 + 8 * gq + 4 * half);
        acc[4 * gq] = bb.x; acc[4 * gq + 1] = bb.y; acc[4 * gq + 2] = bb.z; acc[4 * gq + 3] = bb.w;
      }
      const int pre = rc_xpre(st.src, l32, half, rowb);
      auto xfrag = [&](int ks) { return rc_xread(smem, pre, ks); };
      bf16x8 xb[2];
      xb[0] = xfrag(0);
#pragma unroll 1
      for (int ks0 = 0; ks0 < KS - RC_RING; ks0 += RC_RING) {
        auto step = [&](auto ic) {
          constexpr int i = decltype(ic)::value;
          xb[(i + 1) & 1] = xfrag(ks0 + i + 1);
          c3_wait<RC_RING - 1>(ar[i]);
          acc = FX_MFMA_32x32x16(ar[i], xb[i & 1], acc);
          rc_ldg_next(ar[i], st.w, voff);
        };
        c3_static_for<RC_RING>(step);
      }
      {   // last eight fragments: counted-down waits, no refills, no drain (see RC_GEMM)
        auto step_tail = [&](auto ic) {
          constexpr int i = decltype(ic)::value;
          if constexpr (i + 1 < RC_RING) xb[(i + 1) & 1] = xfrag(KS - RC_RING + i + 1);
          c3_wait<RC_RING - 1 - i>(ar[i]);
          acc = FX_MFMA_32x32x16(ar[i], xb[i & 1], acc);
        };
        c3_static_for<RC_RING>(step_tail);
      }
      rc_ln_tail(acc, st, smem, red, wave, l32, half, m0, M);
    } else if (st.type == RC_FFN_LN) {
      // K = 256, N = hidden width (a multiple of 256); w = W1 in fragment order [N/32][16][64][8], g1 = W2 in fragment order
      // [8][N/16][64][8]; bias = [b1 (N) | b2 (256)]; act / flags = LDS offsets of the two chunk slots; aux = residual slot; ld2 = `red`.
      // Per wave ONE weight stream through the ring: chunk c = 16 fragments of W1 (hidden block 8c + wave), then 16 of W2 (output block
      // `wave`, k-steps 16c..16c+15); fragment j sits in ring slot j % 8 and is refilled with fragment j + 8 right after its MFMA.
      const int NC = N >> 8, KS2 = N >> 4;
      float* red = reinterpret_cast<float*>(smem + st.ld2);
      const unsigned lane16 = lane * 16;
      // fragment j of the stream: r = j % 32 < 16 -> W1, else W2 (compile-time per ring step: j0 is a multiple of 8, so r < 16 is decided
      // by the group)
      // W1: chunk c = fragments (8c + wave) * 16 + 0..15 (consecutive; + 112 KiB to the next chunk's); W2: wave * KS2 + 0 .. KS2-1, consecutive
      // over all chunks.  voff1 / voff2 = offset of the next fragment to request of either matrix.
      unsigned voff1 = lane16 + (unsigned)((wave * 16) << 10), voff2 = lane16 + (unsigned)((wave * KS2) << 10);
      bf16x8 ar[RC_RING];
      bf16x8 xb[2];
#pragma unroll
      for (int i = 0; i < RC_RING; ++i) {
        ar[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        rc_ldg_next(ar[i], st.w, voff1);
      }
      for (int n = tid; n < N + 256; n += 512) biasL[n] = st.bias[n];
      __syncthreads();
      f32x16 acc2, acc1;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const float4 bb = *reinterpret_cast<const float4*>(biasL + N + wave * 32 + 8 * gq + 4 * half);
        acc2[4 * gq] = bb.x; acc2[4 * gq + 1] = bb.y; acc2[4 * gq + 2] = bb.z; acc2[4 * gq + 3] = bb.w;
      }
      // eight ring steps of one GEMM: k-steps ks0 .. ks0 + 7 of the [32][256] operand whose lane constant is `pre`; what the ring is
      // refilled with (the fragment eight positions ahead in the stream): NEXT = 1 - W1 fragments ks0 + 8 .. of chunk c (the same GEMM goes
      // on), 2 - W2 fragments 0..7 of chunk c (the second GEMM follows), 3 - W2 fragments 8..15 of chunk c, 4 - W1 fragments 0..7 of chunk
      // c + 1, 0 - nothing (the stream's last eight: counted-down waits)
      auto group = [&](f32x16& acc, int pre, int ks0, auto next) {
        constexpr int NEXT = decltype(next)::value;
        if constexpr (NEXT == 4) voff1 += 112u << 10;   // on to the next chunk's W1 block
        auto step = [&](auto ic) {
          constexpr int i = decltype(ic)::value;
          xb[(i + 1) & 1] = rc_xread(smem, pre, (ks0 + i + 1) & 15);
          if constexpr (NEXT == 0) c3_wait<RC_RING - 1 - i>(ar[i]);
          else c3_wait<RC_RING - 1>(ar[i]);
          acc = FX_MFMA_32x32x16(ar[i], xb[i & 1], acc);
          if constexpr (NEXT == 1 || NEXT == 4) rc_ldg_next(ar[i], st.w, voff1);
          if constexpr (NEXT == 2 || NEXT == 3) rc_ldg_next(ar[i], st.g1, voff2);
        };
        c3_static_for<RC_RING>(step);
      };
      const int preX = rc_xpre(st.src, l32, half, 512);
      auto first_gemm = [&](int c) {   // hidden chunk c: relu(X W1_c^T + b1_c) -> chunk slot c & 1; returns the slot's lane constant
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const float4 bb = *reinterpret_cast<const float4*>(biasL + (c * 8 + wave) * 32 + 8 * gq + 4 * half);
          acc1[4 * gq] = bb.x; acc1[4 * gq + 1] = bb.y; acc1[4 * gq + 2] = bb.z; acc1[4 * gq + 3] = bb.w;
        }
        xb[0] = rc_xread(smem, preX, 0);
        group(acc1, preX, 0, std::integral_constant<int, 1>{});
        group(acc1, preX, 8, std::integral_constant<int, 2>{});
        const int slot = (c & 1) ? st.flags : st.act;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int n = wave * 32 + 8 * gq + 4 * half;
          uint2 o;
          o.x = pack_bf16x2(fmaxf(acc1[4 * gq], 0.0f), fmaxf(acc1[4 * gq + 1], 0.0f));
          o.y = pack_bf16x2(fmaxf(acc1[4 * gq + 2], 0.0f), fmaxf(acc1[4 * gq + 3], 0.0f));
          *reinterpret_cast<uint2*>(smem + slot + rc_off(l32, n >> 3, 512) + half * 8) = o;
        }
        // ONE barrier per chunk: it also orders this chunk's writes of slot c & 1 behind every wave's reads of it two chunks ago
        // (each wave finished its second GEMM of chunk c - 2 before it entered the barrier of chunk c - 1)
        __syncthreads();
        const int pre = rc_xpre(slot, l32, half, 512);
        xb[0] = rc_xread(smem, pre, 0);
        return pre;
      };
#pragma unroll 1
      for (int c = 0; c < NC - 1; ++c) {
        const int pre = first_gemm(c);
        group(acc2, pre, 0, std::integral_constant<int, 3>{});
        group(acc2, pre, 8, std::integral_constant<int, 4>{});
      }
      {   // last chunk peeled (straight-line: the ring registers must not meet in a phi behind a branch, see RC_GEMM)
        const int c = NC - 1;
        const int pre = first_gemm(c);
        group(acc2, pre, 0, std::integral_constant<int, 3>{});
        group(acc2, pre, 8, std::integral_constant<int, 0>{});
      }
      rc_ln_tail(acc2, st, smem, red, wave, l32, half, m0, M);
    }
    __syncthreads();   // stage boundary: the next stage reads what this one wrote
  }
  if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dbg[1 + nstages] = __builtin_amdgcn_s_memtime();
}

extern "C" int fx_row_chain(const fx_rc_stage* program_device, int n_stages, int rows, int lds_bytes, fx_stream_t stream_) {
  FX_CHECK_ARG(program_device && n_stages > 0 && n_stages <= 64 && rows > 0 && lds_bytes > 0 && lds_bytes % 16 == 0 && lds_bytes + RC_SCRATCH <= 160 * 1024);
  static int attr_smem = 0;
  if (lds_bytes + RC_SCRATCH > attr_smem) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(row_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes + RC_SCRATCH) != hipSuccess)
      return FX_ERR_RUNTIME;
    attr_smem = lds_bytes + RC_SCRATCH;
  }
  // diagnostic (scripts/dev/rc_stage_stamps.py): FX_RC_DBG = address of a device buffer of 64 x u64 that receives the s_memtime stamps of
  // workgroup 0 at every stage boundary; unset (the product): null, the kernel skips the stamps
  static unsigned long long* const dbg = reinterpret_cast<unsigned long long*>((uintptr_t)strtoull(getenv("FX_RC_DBG") ? getenv("FX_RC_DBG") : "0", nullptr, 0));
  static const int prefetch = fx_tune("FX_RC_PREFETCH", 0);   // OFF: serial 0.92 -> 0.81 ms per RT-DETR step but +0.3 % on the two-queue headline and -2 % on BiSeNetFormer (profiles/r05_rc_prefetch_ab.txt)
  hipLaunchKernelGGL(row_chain_kernel, dim3((rows + 31) / 32), dim3(512), lds_bytes + RC_SCRATCH, reinterpret_cast<hipStream_t>(stream_), program_device, n_stages,
                     rows, lds_bytes, dbg, prefetch);
  return fx_launch_status();
}
