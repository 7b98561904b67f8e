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
// GEMMs: 8 waves, a wave owns 32 output channels per pass (one 32x32 accumulator block over the 32 rows), weights = MFMA A
// operand in fragment order straight from L2 through an 8-deep register ring (loads hidden from hipcc's waitcnt bookkeeping, see
// pw_common.h), rows = B operand from LDS.  With 32 rows a fragment is used by one MFMA only: the chain is bound by the
// weight stream from L2 (~2 MB per workgroup and layer), not by the matrix cores - which is still 5-10x less time than the
// launch-bound form.
#include <stdlib.h>

#include "pw_common.h"

enum { RC_LOAD = 0, RC_GEMM = 1, RC_GEMM_LN = 2, RC_ADD = 3, RC_K4 = 4, RC_BBOX = 5, RC_LN = 6 };

__device__ __forceinline__ float rc_inv_sigmoid(float x) {
  x = fminf(fmaxf(x, 0.0f), 1.0f);
  return __logf(fmaxf(x, 1e-5f) / fmaxf(1.0f - x, 1e-5f));
}

// byte offset of 16-byte chunk `chunk` of row `row` in a [32][K] buffer (K >= 128: 16 chunks per 256-byte bank window)
__device__ __forceinline__ int rc_off(int row, int chunk, int rowb) { return row * rowb + ((chunk ^ (row & 15)) << 4); }

// bias_off: LDS byte offset of a 12 KiB scratch behind the program's own buffers (round 3).  Per-stage s_memtime stamps of the post-MSDA
// chain (scripts/dev/rc_stage_stamps.py; 136k cycles): the GEMM stages stream 2 MB of weights per workgroup at ~8.7 TB/s of aggregate L2
// bandwidth (150 workgroups pulling the same fragments - that is their bound: a 16-deep ring changed nothing), but the two VALU stages
// K4 and BBOX took 19k cycles EACH for ~130 kFLOP: every thread walked its fp32 weights with dependent global loads.  Both now copy
// their weights into the scratch with one coalesced pass and compute from LDS (K4 with lane = row, so that a wave reads two
// addresses per instruction: broadcasts).  A GEMM stage keeps its bias vector there (no global load per 32-channel pass).
#define RC_SCRATCH 12288
#define RC_RING 8    // weight fragments in flight per wave (16 measured the same: the stages are not waiting on this stream's round trips)
__global__ __launch_bounds__(512, 1) void row_chain_kernel(const fx_rc_stage* __restrict__ prog, int nstages, int M, int bias_off, unsigned long long* dbg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* biasL = reinterpret_cast<float*>(smem + bias_off);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, half = lane >> 5;
  const int m0 = blockIdx.x * 32;

  if (dbg && blockIdx.x == 0 && tid == 0) dbg[0] = __builtin_amdgcn_s_memtime();
  for (int si = 0; si < nstages; ++si) {
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
        *reinterpret_cast<uint4*>(smem + st.dst + rc_off(l32, c, N * 2)) = pack_bf16x8(v);
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
      const int rowb = K * 2;
      const int npass = (nblk - wave + 7) >> 3;           // passes of this wave (<= 0: none)
      const int total = npass * KS;
      const bf16_t* wb = reinterpret_cast<const bf16_t*>(st.w) + lane * 8;
      auto frag = [&](int i) -> const bf16_t* {
        const int ic = i < total ? i : total - 1;
        return wb + ((size_t)(wave + ((ic >> ksh) << 3)) * KS + (ic & (KS - 1))) * 512;
      };
      bf16x8 ar[RC_RING];
      if (npass > 0) {   // the ring's first fragments go out before the bias copy: one L2 round trip covers both
#pragma unroll
        for (int i = 0; i < RC_RING; ++i) {
          ar[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
          c3_ldg_async(ar[i], frag(i));
        }
      }
      for (int n = tid; n < N; n += 512) biasL[n] = st.bias ? st.bias[n] : 0.0f;
      __syncthreads();
      if (npass > 0) {
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
            if (st.dst >= 0) *reinterpret_cast<uint2*>(smem + st.dst + rc_off(l32, n >> 3, N * 2) + half * 8) = o;
            if (st.g0 && m < M) {
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
        auto xfrag = [&](int ks) { return *reinterpret_cast<const bf16x8*>(smem + st.src + rc_off(l32, (ks & (KS - 1)) * 2 + half, rowb)); };
        bf16x8 xb[2];
        xb[0] = xfrag(0);
        for (int ps = 0; ps < npass; ++ps) {
          const int nb = wave + ps * 8;
          init_acc(nb);
          const int kend = ps == npass - 1 ? KS - RC_RING : KS;
#pragma unroll 1
          for (int ks0 = 0; ks0 < kend; ks0 += RC_RING) {
            auto step = [&](auto ic) {
              constexpr int i = decltype(ic)::value;
              xb[(i + 1) & 1] = xfrag(ks0 + i + 1);   // next k-step's rows (wraps to k-step 0 for the next pass), one MFMA ahead of its use
              c3_wait<RC_RING - 1>(ar[i]);
              acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[i], xb[i & 1], acc, 0, 0, 0);
              c3_ldg_async(ar[i], frag(ps * KS + ks0 + i + RC_RING));
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
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[i], xb[i & 1], acc, 0, 0, 0);
          };
          c3_static_for<RC_RING>(step_tail);
          epilogue(wave + (npass - 1) * 8);
        }
      }
    } else if (st.type == RC_GEMM_LN) {
      // N = 256: wave -> channels [32*wave, +32); LayerNorm over the row on the fp32 accumulators
      const int KS = K >> 4, rowb = K * 2;
      float* red = reinterpret_cast<float*>(smem + st.ld2);   // [8 waves][32 rows] reduction scratch (LDS byte offset in ld2)
      const bf16_t* w = reinterpret_cast<const bf16_t*>(st.w) + (size_t)wave * KS * 512 + lane * 8;
      bf16x8 ar[RC_RING];
#pragma unroll
      for (int i = 0; i < RC_RING; ++i) {
        ar[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        c3_ldg_async(ar[i], w + (size_t)i * 512);
      }
      f32x16 acc;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const float4 bb = *reinterpret_cast<const float4*>(st.bias + wave * 32 + 8 * gq + 4 * half);
        acc[4 * gq] = bb.x; acc[4 * gq + 1] = bb.y; acc[4 * gq + 2] = bb.z; acc[4 * gq + 3] = bb.w;
      }
      auto xfrag = [&](int ks) { return *reinterpret_cast<const bf16x8*>(smem + st.src + rc_off(l32, ks * 2 + half, rowb)); };
      bf16x8 xb[2];
      xb[0] = xfrag(0);
#pragma unroll 1
      for (int ks0 = 0; ks0 < KS - RC_RING; ks0 += RC_RING) {
        auto step = [&](auto ic) {
          constexpr int i = decltype(ic)::value;
          xb[(i + 1) & 1] = xfrag(ks0 + i + 1);
          c3_wait<RC_RING - 1>(ar[i]);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[i], xb[i & 1], acc, 0, 0, 0);
          c3_ldg_async(ar[i], w + (size_t)(ks0 + i + RC_RING) * 512);
        };
        c3_static_for<RC_RING>(step);
      }
      {   // last eight fragments: counted-down waits, no refills, no drain (see RC_GEMM)
        auto step_tail = [&](auto ic) {
          constexpr int i = decltype(ic)::value;
          if constexpr (i + 1 < RC_RING) xb[(i + 1) & 1] = xfrag(KS - RC_RING + i + 1);
          c3_wait<RC_RING - 1 - i>(ar[i]);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[i], xb[i & 1], acc, 0, 0, 0);
        };
        c3_static_for<RC_RING>(step_tail);
      }
      if (st.aux >= 0) {   // + residual (bf16 LDS buffer [32][256])
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int n = wave * 32 + 8 * gq + 4 * half;
          const uint2 rv = *reinterpret_cast<const uint2*>(smem + st.aux + rc_off(l32, n >> 3, 512) + half * 8);
          acc[4 * gq] += __uint_as_float(rv.x << 16);
          acc[4 * gq + 1] += __uint_as_float(rv.x & 0xffff0000u);
          acc[4 * gq + 2] += __uint_as_float(rv.y << 16);
          acc[4 * gq + 3] += __uint_as_float(rv.y & 0xffff0000u);
        }
      }
      // two-pass statistics: a row's 256 channels are spread over 2 lanes (half) x 8 waves
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
    __syncthreads();   // stage boundary: the next stage reads what this one wrote
  }
  if (dbg && blockIdx.x == 0 && tid == 0) dbg[1 + nstages] = __builtin_amdgcn_s_memtime();
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
  hipLaunchKernelGGL(row_chain_kernel, dim3((rows + 31) / 32), dim3(512), lds_bytes + RC_SCRATCH, reinterpret_cast<hipStream_t>(stream_), program_device, n_stages,
                     rows, lds_bytes, dbg);
  return fx_launch_status();
}
