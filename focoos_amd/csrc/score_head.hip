// Encoder score head of the RT-DETR query selection as one back-to-back GEMM kernel (gfx950):
//
//   output_memory = LayerNorm( Linear_256->256( valid_mask * memory ) )         (fai_detr/modelling.py:1202-1208)
//   scores        = max_c ( Linear_256->K( output_memory ) )                     (modelling.py:1209-1214: enc_score_classifier, .max(-1))
//
// Separate launches write the [B*8400, 256] Linear output, re-read it for the LayerNorm, write output_memory, re-read it for
// the class GEMM and write [B*8400, K] fp32 logits (392 MB at bs=32, K=365) that are only ever reduced to their row maxima.
// Here a 64-token tile goes memory -> GEMM1 -> LayerNorm on the fp32 ACCUMULATORS (the Linear output is never rounded)
// -> bf16 output_memory tile in LDS (stored once: the decoder gathers its 300 rows from it) -> GEMM2 -> row max -> 4 bytes per token.
// Index-critical path: everything between the memory tile and the score is fp32 except the bf16 rounding of
// output_memory itself, which is also what the decoder consumes.
//
// Structure as in conv_pw_chain.hip: 4 waves, weights = MFMA A operands in fragment order straight from L2, tokens = B
// operand from a swizzled LDS tile filled by buffer_load...lds; invalid anchors (modelling.py:1183-1189) are DMA'd as zero rows,
// which makes their output_memory LayerNorm(bias) exactly as in the reference.
#include "pw_common.h"

struct ScoreHeadArgs {
  const bf16_t* mem;
  const uint8_t* valid;  // [S] 1 = valid anchor, or NULL
  const bf16_t* w1p;
  const bf16_t* w2p;
  const float* b1;
  const float* gamma;
  const float* beta;
  const float* b2;  // [128*TN2], padding classes hold -3e38
  bf16_t* om;
  float* scores;
  int ldm, ldo, S, M;
  float eps;
  unsigned mem_bytes;
};

template <int TN2>
__global__ __launch_bounds__(256, 2) void score_head_kernel(const ScoreHeadArgs p) {
  constexpr int BM = 64, TM = 2, KS = 16, PF = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* T = smem;                                       // [64][256] bf16: memory tile, then output_memory tile
  float* red = reinterpret_cast<float*>(smem + BM * 512);        // [4 waves][64 rows]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, half = lane >> 5;
  const int m0 = blockIdx.x * BM;

  const __amdgpu_buffer_rsrc_t mr = __builtin_amdgcn_make_buffer_rsrc((void*)p.mem, 0, p.mem_bytes, 0x00020000);
  for (int i = wave; i < BM * 32 / 64; i += 4) {
    const int q = i * 64 + lane;
    const int row = q >> 5, pc = q & 31;
    const int lc = pc ^ (row & 15);
    const int m = m0 + row;
    bool ok = m < p.M;
    if (ok && p.valid) ok = p.valid[m % p.S] != 0;
    pw_dma16(mr, T + i * 1024, ok ? (unsigned)(m * p.ldm + lc * 8) * 2u : FX_OOB);
  }

  // ---- GEMM1: wave -> channels [64*wave, +64); accumulators start from the bias
  f32x16 acc[2][TM];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const float4 bb = *reinterpret_cast<const float4*>(p.b1 + wave * 64 + a * 32 + 8 * gq + 4 * half);
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        acc[a][b][4 * gq] = bb.x; acc[a][b][4 * gq + 1] = bb.y; acc[a][b][4 * gq + 2] = bb.z; acc[a][b][4 * gq + 3] = bb.w;
      }
    }
  const bf16_t* w1 = p.w1p + (size_t)(wave * 2 * KS) * 512 + lane * 8;
  bf16x8 a1[PF][2];
#pragma unroll
  for (int i = 0; i < PF; ++i) {
    a1[i][0] = pw_ldg_frag(w1 + i * 512);
    a1[i][1] = pw_ldg_frag(w1 + (KS + i) * 512);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll 1
  for (int ks0 = 0; ks0 < KS; ks0 += PF) {
    int l32k = l32;
    asm volatile("" : "+v"(l32k));
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      bf16x8 xb[TM];
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        const int row = b * 32 + l32k;
        xb[b] = *reinterpret_cast<const bf16x8*>(T + row * 512 + ((((ks0 + i) * 2 + half) ^ (row & 15)) << 4));
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = FX_MFMA_32x32x16(a1[i][a], xb[b], acc[a][b]);
      const int kn = ks0 + i + PF;
      const int kc = kn < KS ? kn : KS - 1;
      a1[i][0] = pw_ldg_frag(w1 + kc * 512);
      a1[i][1] = pw_ldg_frag(w1 + (KS + kc) * 512);
    }
  }
  // first weight fragments of GEMM2 (requested before the LayerNorm reductions)
  const bf16_t* w2 = p.w2p + (size_t)(wave * TN2 * KS) * 512 + lane * 8;
  bf16x8 a2[PF][TN2];
#pragma unroll
  for (int i = 0; i < PF; ++i)
#pragma unroll
    for (int a = 0; a < TN2; ++a) a2[i][a] = pw_ldg_frag(w2 + (size_t)(a * KS + i) * 512);

  // ---- LayerNorm over the 256 channels of each token, two-pass (mean, then centred second moment) in fp32.
  //      A token's channels are spread over 2 lanes (half) x 4 waves: shuffle across the halves, LDS across the waves.
  float mean[TM], rstd[TM];
  {
    float s[TM];
#pragma unroll
    for (int b = 0; b < TM; ++b) {
      float t = 0.0f;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[a][b][r];
      t += __shfl_xor(t, 32);
      s[b] = t;
    }
    __syncthreads();  // every wave is done reading the memory tile (T is overwritten below) and `red` is free
    if (half == 0) {
#pragma unroll
      for (int b = 0; b < TM; ++b) red[wave * 64 + b * 32 + l32] = s[b];
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < TM; ++b) {
      const int r = b * 32 + l32;
      mean[b] = (red[r] + red[64 + r] + red[128 + r] + red[192 + r]) * (1.0f / 256.0f);
    }
#pragma unroll
    for (int b = 0; b < TM; ++b) {
      float t = 0.0f;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float d = acc[a][b][r] - mean[b];
          t += d * d;
        }
      t += __shfl_xor(t, 32);
      s[b] = t;
    }
    __syncthreads();
    if (half == 0) {
#pragma unroll
      for (int b = 0; b < TM; ++b) red[wave * 64 + b * 32 + l32] = s[b];
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < TM; ++b) {
      const int r = b * 32 + l32;
      rstd[b] = rsqrtf((red[r] + red[64 + r] + red[128 + r] + red[192 + r]) * (1.0f / 256.0f) + p.eps);
    }
  }
  // ---- output_memory tile (bf16) into T, accumulator layout -> 8-byte pieces
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int n = wave * 64 + a * 32 + 8 * gq + 4 * half;
      const float4 gg = *reinterpret_cast<const float4*>(p.gamma + n);
      const float4 be = *reinterpret_cast<const float4*>(p.beta + n);
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        const int row = b * 32 + l32;
        const float v0 = (acc[a][b][4 * gq] - mean[b]) * rstd[b] * gg.x + be.x;
        const float v1 = (acc[a][b][4 * gq + 1] - mean[b]) * rstd[b] * gg.y + be.y;
        const float v2 = (acc[a][b][4 * gq + 2] - mean[b]) * rstd[b] * gg.z + be.z;
        const float v3 = (acc[a][b][4 * gq + 3] - mean[b]) * rstd[b] * gg.w + be.w;
        uint2 o;
        o.x = pack_bf16x2(v0, v1);
        o.y = pack_bf16x2(v2, v3);
        *reinterpret_cast<uint2*>(T + row * 512 + (((n >> 3) ^ (row & 15)) << 4) + half * 8) = o;
      }
    }
  __syncthreads();
  // ---- output_memory -> HBM
#pragma unroll
  for (int i = 0; i < BM * 32 / 256; ++i) {
    const int q = tid + i * 256;
    const int row = q >> 5, lc = q & 31;
    const uint4 v = *reinterpret_cast<const uint4*>(T + row * 512 + ((lc ^ (row & 15)) << 4));
    const int m = m0 + row;
    if (m < p.M) *reinterpret_cast<uint4*>(p.om + (size_t)m * p.ldo + lc * 8) = v;
  }
  // ---- GEMM2: class logits, wave -> classes [32*TN2*wave, +32*TN2); accumulators start from the bias (padding classes: -3e38)
  f32x16 acc2[TN2][TM];
#pragma unroll
  for (int a = 0; a < TN2; ++a)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const float4 bb = *reinterpret_cast<const float4*>(p.b2 + (wave * TN2 + a) * 32 + 8 * gq + 4 * half);
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        acc2[a][b][4 * gq] = bb.x; acc2[a][b][4 * gq + 1] = bb.y; acc2[a][b][4 * gq + 2] = bb.z; acc2[a][b][4 * gq + 3] = bb.w;
      }
    }
#pragma unroll 1
  for (int ks0 = 0; ks0 < KS; ks0 += PF) {
    int l32k = l32;
    asm volatile("" : "+v"(l32k));
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      bf16x8 tb[TM];
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        const int row = b * 32 + l32k;
        tb[b] = *reinterpret_cast<const bf16x8*>(T + row * 512 + ((((ks0 + i) * 2 + half) ^ (row & 15)) << 4));
      }
#pragma unroll
      for (int a = 0; a < TN2; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc2[a][b] = FX_MFMA_32x32x16(a2[i][a], tb[b], acc2[a][b]);
      const int kn = ks0 + i + PF;
      const int kc = kn < KS ? kn : KS - 1;
#pragma unroll
      for (int a = 0; a < TN2; ++a) a2[i][a] = pw_ldg_frag(w2 + (size_t)(a * KS + kc) * 512);
    }
  }
  // ---- row max over the classes
  float mx[TM];
#pragma unroll
  for (int b = 0; b < TM; ++b) {
    float t = acc2[0][b][0];
#pragma unroll
    for (int a = 0; a < TN2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) t = fmaxf(t, acc2[a][b][r]);
    t = fmaxf(t, __shfl_xor(t, 32));
    mx[b] = t;
  }
  __syncthreads();  // `red` was last read for rstd by every wave before this point
  if (half == 0) {
#pragma unroll
    for (int b = 0; b < TM; ++b) red[wave * 64 + b * 32 + l32] = mx[b];
  }
  __syncthreads();
  if (tid < BM && m0 + tid < p.M) p.scores[m0 + tid] = fmaxf(fmaxf(red[tid], red[64 + tid]), fmaxf(red[128 + tid], red[192 + tid]));
}

template <int TN2>
static int launch_score_head(const ScoreHeadArgs& a, hipStream_t stream) {
  constexpr int SMEM = 64 * 512 + 4 * 64 * 4;
  static bool attr_set = false;
  auto kern = score_head_kernel<TN2>;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess)
      return FX_ERR_RUNTIME;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((a.M + 63) / 64), dim3(256), SMEM, stream, a);
  return fx_launch_status();
}

extern "C" int fx_enc_score_head_bf16(const void* memory, int ldm, const uint8_t* valid, int S, const void* w1, const float* b1,
                                      const float* gamma, const float* beta, float eps, const void* w2, const float* b2, int n2_pad,
                                      void* output_memory, int ldo, float* scores, int M, fx_stream_t stream_) {
  FX_CHECK_ARG(memory && w1 && b1 && gamma && beta && w2 && b2 && output_memory && scores && M > 0 && S > 0);
  FX_CHECK_ARG(ldm >= 256 && ldm % 8 == 0 && ldo >= 256 && ldo % 8 == 0);
  FX_CHECK_ARG(((uintptr_t)memory % 16) == 0 && ((uintptr_t)w1 % 16) == 0 && ((uintptr_t)w2 % 16) == 0 && ((uintptr_t)output_memory % 16) == 0);
  FX_CHECK_ARG(((uintptr_t)b1 % 16) == 0 && ((uintptr_t)b2 % 16) == 0 && ((uintptr_t)gamma % 16) == 0 && ((uintptr_t)beta % 16) == 0);
  if (n2_pad != 128 && n2_pad != 256 && n2_pad != 384) return FX_ERR_UNSUPPORTED;
  const int64_t mem_bytes = ((int64_t)M - 1) * ldm * 2 + 512;
  if (mem_bytes >= 0xFFFFFFF0ll) return FX_ERR_UNSUPPORTED;
  ScoreHeadArgs a;
  a.mem = reinterpret_cast<const bf16_t*>(memory);
  a.valid = valid;
  a.w1p = reinterpret_cast<const bf16_t*>(w1);
  a.w2p = reinterpret_cast<const bf16_t*>(w2);
  a.b1 = b1; a.gamma = gamma; a.beta = beta; a.b2 = b2;
  a.om = reinterpret_cast<bf16_t*>(output_memory);
  a.scores = scores;
  a.ldm = ldm; a.ldo = ldo; a.S = S; a.M = M; a.eps = eps;
  a.mem_bytes = (unsigned)mem_bytes;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (n2_pad == 128) return launch_score_head<1>(a, stream);
  if (n2_pad == 256) return launch_score_head<2>(a, stream);
  return launch_score_head<3>(a, stream);
}
