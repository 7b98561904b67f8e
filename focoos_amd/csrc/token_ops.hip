// Token-space kernels of the RT-DETR encoder/decoder (gfx950): row add, LayerNorm(+residual),
// softmax attention (head_dim 32), multi-scale deformable-attention sampling.
// All are HBM/L2-bound wavefront kernels; rows are 256 channels = one 512-byte line per wave.
#include "common.h"

int fx_tune(const char* env_name, int default_value);  // conv_igemm.hip

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void add_rows_kernel(const bf16_t* __restrict__ x, int ldx, const bf16_t* __restrict__ y, int ldy,
                                                        int y_rows, bf16_t* __restrict__ out, int ldo, int rows, int C8) {
  int64_t total = (int64_t)rows * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int c8 = (int)(i % C8);
    int r = (int)(i / C8);
    float a[8], b[8];
    unpack_bf16x8(*reinterpret_cast<const uint4*>(x + (int64_t)r * ldx + c8 * 8), a);
    unpack_bf16x8(*reinterpret_cast<const uint4*>(y + (int64_t)(r % y_rows) * ldy + c8 * 8), b);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] += b[j];
    *reinterpret_cast<uint4*>(out + (int64_t)r * ldo + c8 * 8) = pack_bf16x8(a);
  }
}

extern "C" int fx_add_rows_bf16(const void* x, int ldx, const void* y, int ldy_, int y_rows, void* out, int ldo, int rows, int cols,
                                fx_stream_t stream_) {
  FX_CHECK_ARG(x && y && out && rows > 0 && cols > 0 && cols % 8 == 0 && y_rows > 0);
  FX_CHECK_ARG(ldx >= cols && ldy_ >= cols && ldo >= cols && ldx % 8 == 0 && ldy_ % 8 == 0 && ldo % 8 == 0);
  int64_t total = (int64_t)rows * (cols / 8);
  int64_t grid = (total + 255) / 256;
  if (grid > 256 * 16) grid = 256 * 16;
  hipLaunchKernelGGL(add_rows_kernel, dim3((int)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), (const bf16_t*)x, ldx,
                     (const bf16_t*)y, ldy_, y_rows, (bf16_t*)out, ldo, rows, cols / 8);
  return fx_launch_status();
}

// ------------------------------------------------------------------------------------------------
// LayerNorm over 256 channels, one wave per row (4 channels per lane), fp32 statistics
// (two-pass mean / variance in registers, biased variance, eps inside the sqrt like nn.LayerNorm).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(256) void layernorm256_kernel(const bf16_t* __restrict__ x, int ldx, const bf16_t* __restrict__ res, int ldr,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            bf16_t* __restrict__ out, int ldo, int rows) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  uint2 xv = *reinterpret_cast<const uint2*>(x + (int64_t)row * ldx + lane * 4);
  float v[4] = {bf16lo_to_f32(xv.x), bf16hi_to_f32(xv.x), bf16lo_to_f32(xv.y),
                bf16hi_to_f32(xv.y)};
  if (res) {
    uint2 rv = *reinterpret_cast<const uint2*>(res + (int64_t)row * ldr + lane * 4);
    v[0] += bf16lo_to_f32(rv.x);
    v[1] += bf16hi_to_f32(rv.x);
    v[2] += bf16lo_to_f32(rv.y);
    v[3] += bf16hi_to_f32(rv.y);
  }
  float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.0f / 256.0f);
  float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
  float var = wave_sum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.0f / 256.0f);
  float rstd = rsqrtf(var + 1e-5f);
  float4 g = *reinterpret_cast<const float4*>(gamma + lane * 4), bt = *reinterpret_cast<const float4*>(beta + lane * 4);
  uint2 o;
  o.x = pack_bf16x2(d0 * rstd * g.x + bt.x, d1 * rstd * g.y + bt.y);
  o.y = pack_bf16x2(d2 * rstd * g.z + bt.z, d3 * rstd * g.w + bt.w);
  *reinterpret_cast<uint2*>(out + (int64_t)row * ldo + lane * 4) = o;
}

// The same over 128 channels (2 per lane): the 128-channel pixel-decoder encoders of fai-mf-{m,s}-coco-ins.
__global__ __launch_bounds__(256) void layernorm128_kernel(const bf16_t* __restrict__ x, int ldx, const bf16_t* __restrict__ res, int ldr,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            bf16_t* __restrict__ out, int ldo, int rows) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const unsigned xv = *reinterpret_cast<const unsigned*>(x + (int64_t)row * ldx + lane * 2);
  float v0 = bf16lo_to_f32(xv), v1 = bf16hi_to_f32(xv);
  if (res) {
    const unsigned rv = *reinterpret_cast<const unsigned*>(res + (int64_t)row * ldr + lane * 2);
    v0 += bf16lo_to_f32(rv);
    v1 += bf16hi_to_f32(rv);
  }
  const float mean = wave_sum(v0 + v1) * (1.0f / 128.0f);
  const float d0 = v0 - mean, d1 = v1 - mean;
  const float var = wave_sum(d0 * d0 + d1 * d1) * (1.0f / 128.0f);
  const float rstd = rsqrtf(var + 1e-5f);
  const float2 g = *reinterpret_cast<const float2*>(gamma + lane * 2), bt = *reinterpret_cast<const float2*>(beta + lane * 2);
  *reinterpret_cast<unsigned*>(out + (int64_t)row * ldo + lane * 2) = pack_bf16x2(d0 * rstd * g.x + bt.x, d1 * rstd * g.y + bt.y);
}

extern "C" int fx_layernorm_bf16(const void* x, int ldx, const void* residual, int ldr, const float* gamma, const float* beta, void* out,
                                 int ldo, int rows, int cols, fx_stream_t stream_) {
  FX_CHECK_ARG(x && gamma && beta && out && rows > 0);
  if (cols != 256 && cols != 128) return FX_ERR_UNSUPPORTED;
  FX_CHECK_ARG(ldx >= cols && ldo >= cols && ldx % 4 == 0 && ldo % 4 == 0 && (!residual || (ldr >= cols && ldr % 4 == 0)));
  if (cols == 128)
    hipLaunchKernelGGL(layernorm128_kernel, dim3((rows + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), (const bf16_t*)x,
                       ldx, (const bf16_t*)residual, ldr, gamma, beta, (bf16_t*)out, ldo, rows);
  else
    hipLaunchKernelGGL(layernorm256_kernel, dim3((rows + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), (const bf16_t*)x,
                       ldx, (const bf16_t*)residual, ldr, gamma, beta, (bf16_t*)out, ldo, rows);
  return fx_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Softmax attention, head_dim 32, on the MFMA matrix cores (flash-style, scores never leave registers).
// Workgroup = 4 waves = 128 query rows of one (batch, head); the head's K ([Lk,32], XOR-swizzled rows) and
// V^T ([tile][d][32 keys], XOR-swizzled 8-byte pieces) are staged in LDS.  Per 32-key tile a wave does
//   S^T[key][q] = K_tile . Q^T          2 x v_mfma_f32_32x32x16_bf16   (lane = one query column, 16 keys)
//   online softmax in registers         (row max needs ONE cross-lane exchange: lane ^ 32)
//   O^T[d][q]  += V_tile^T . P          2 x v_mfma_f32_32x32x16_bf16
// The accumulator layout of S^T hands each lane exactly the P fragment the second MFMA wants as its B operand,
// because the reduction index (key) order inside an MFMA is free as long as A (V^T) uses the same order:
// k-slot (half h, j) <-> key (j&3) + 8*(j>>2) + 4h, i.e. V^T pieces h and h+2 of the 8-byte pieces of a row.
// Keys/values are streamed through LDS in chunks of CT 32-key tiles, so Lk is unbounded (the MaskFormer decoder attends
// over up to (H/8)*(W/8) keys).  MASKED adds the boolean attention mask of MultiScaleMaskedTransformerDecoder
// (fai_mf/modelling.py:509-523): bit (key & 31) of word mask[(b*Lq+q)*ldm + key/32] set = key not allowed; a query whose
// mask forbids every key attends everywhere, so both softmaxes are accumulated in the same pass and selected per query.
// SPLIT (flash-decoding): few queries x many keys leaves most CUs idle, so blockIdx.z takes a slice of `tps` key tiles and
// writes un-normalised partials (o[32], m, l per query row; masked and unmasked variants) that mha32_combine_kernel merges.
#define FX_MHA_PART 34  // floats per partial row: o[0..31], m, l
template <int CT, bool MASKED, bool SPLIT>
__global__ __launch_bounds__(256) void mha32_mfma_kernel(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ k, int ldk,
                                                          const bf16_t* __restrict__ v, int ldv, bf16_t* __restrict__ out, int ldo, int Lq,
                                                          int Lk, int heads, const uint32_t* __restrict__ mask, int ldm,
                                                          float* __restrict__ part, int tps) {
  __shared__ __attribute__((aligned(16))) unsigned char ks[CT * 32 * 64];
  __shared__ __attribute__((aligned(16))) unsigned char vt[CT * 32 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.y, b = bh / heads, hd = bh % heads;
  const int T = (Lk + 31) >> 5;
  const bf16_t* kb = k + (int64_t)b * Lk * ldk + hd * 32;
  const bf16_t* vb = v + (int64_t)b * Lk * ldv + hd * 32;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const bool active = q0 < Lq;  // inactive waves still take part in the cooperative loads and barriers
  const int j = lane & 31, h = lane >> 5;
  const int qi = q0 + j;
  bf16x8 qf0, qf1;
  {
    uint4 a = make_uint4(0, 0, 0, 0), c = make_uint4(0, 0, 0, 0);
    if (qi < Lq) {
      const bf16_t* qp = q + ((int64_t)b * Lq + qi) * ldq + hd * 32;
      a = *reinterpret_cast<const uint4*>(qp + 8 * h);
      c = *reinterpret_cast<const uint4*>(qp + 16 + 8 * h);
    }
    qf0 = __builtin_bit_cast(bf16x8, a);
    qf1 = __builtin_bit_cast(bf16x8, c);
  }
  const uint32_t* mrow = MASKED ? mask + ((int64_t)b * Lq + (qi < Lq ? qi : 0)) * ldm : nullptr;
  const float scale = 0.17677669529663687f * 1.4426950408889634f;  // 1/sqrt(32) * log2(e): softmax in the exp2 domain
  f32x16 o, o2;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.0f, o2[r] = 0.0f;
  float m = -INFINITY, l = 0.0f, m2 = -INFINITY, l2 = 0.0f;
  const int vsw = (j >> 2) & 7;  // V^T swizzle of row d = j
  const int tb = SPLIT ? (int)blockIdx.z * tps : 0;
  const int te = SPLIT ? (tb + tps < T ? tb + tps : T) : T;
  for (int c0 = tb; c0 < te; c0 += CT) {
    const int nt = (te - c0) < CT ? (te - c0) : CT;
    if (c0 != tb) __syncthreads();
    for (int i = tid; i < nt * 32 * 4; i += 256) {
      const int lrow = i >> 2, c = i & 3;
      const int row = c0 * 32 + lrow;
      uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
      if (row < Lk) {
        kv = *reinterpret_cast<const uint4*>(kb + (int64_t)row * ldk + c * 8);
        vv = *reinterpret_cast<const uint4*>(vb + (int64_t)row * ldv + c * 8);
      }
      *reinterpret_cast<uint4*>(ks + lrow * 64 + ((c ^ ((lrow >> 2) & 3)) << 4)) = kv;
      const int tile = lrow >> 5, kk = lrow & 31;
      const uint32_t w4[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int d = c * 8 + e;
        const bf16_t val = (bf16_t)((e & 1) ? (w4[e >> 1] >> 16) : (w4[e >> 1] & 0xffffu));
        *reinterpret_cast<bf16_t*>(vt + tile * 2048 + d * 64 + ((((kk >> 2) ^ ((d >> 2) & 7))) << 3) + (kk & 3) * 2) = val;
      }
    }
    __syncthreads();
    if (!active) continue;
    for (int t = 0; t < nt; ++t) {
      const int krow = t * 32 + j;
      const unsigned char* kr = ks + krow * 64;
      const int ksw = (krow >> 2) & 3;
      bf16x8 kf0 = *reinterpret_cast<const bf16x8*>(kr + ((h ^ ksw) << 4));
      bf16x8 kf1 = *reinterpret_cast<const bf16x8*>(kr + (((2 + h) ^ ksw) << 4));
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.0f;
      s = FX_MFMA_32x32x16(kf0, qf0, s);
      s = FX_MFMA_32x32x16(kf1, qf1, s);
      uint32_t mw = 0;
      if (MASKED) mw = mrow[c0 + t];
      float mx = -INFINITY, mx2 = -INFINITY;
      float s2[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kofs = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int key = (c0 + t) * 32 + kofs;
        s[r] = key < Lk ? s[r] * scale : -INFINITY;
        mx = fmaxf(mx, s[r]);
        if (MASKED) {
          s2[r] = ((mw >> kofs) & 1u) ? -INFINITY : s[r];
          mx2 = fmaxf(mx2, s2[r]);
        }
      }
      const unsigned char* vr = vt + t * 2048 + j * 64;
      uint2 va = *reinterpret_cast<const uint2*>(vr + (((h) ^ vsw) << 3));
      uint2 vb2 = *reinterpret_cast<const uint2*>(vr + (((h + 2) ^ vsw) << 3));
      uint2 vc = *reinterpret_cast<const uint2*>(vr + (((h + 4) ^ vsw) << 3));
      uint2 vd = *reinterpret_cast<const uint2*>(vr + (((h + 6) ^ vsw) << 3));
      bf16x8 vf0 = __builtin_bit_cast(bf16x8, make_uint4(va.x, va.y, vb2.x, vb2.y));
      bf16x8 vf1 = __builtin_bit_cast(bf16x8, make_uint4(vc.x, vc.y, vd.x, vd.y));
      {
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m, mx);  // finite: every tile holds at least one real key
        const float alpha = __builtin_amdgcn_exp2f(m - mn);
        m = mn;
        float psum = 0.0f;
        float pf[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          pf[r] = __builtin_amdgcn_exp2f(s[r] - mn);
          psum += pf[r];
        }
        l = l * alpha + psum;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] *= alpha;
        uint4 p0 = pack_bf16x8(pf), p1 = pack_bf16x8(pf + 8);
        o = FX_MFMA_32x32x16(vf0, __builtin_bit_cast(bf16x8, p0), o);
        o = FX_MFMA_32x32x16(vf1, __builtin_bit_cast(bf16x8, p1), o);
      }
      if (MASKED) {
        mx2 = fmaxf(mx2, __shfl_xor(mx2, 32, 64));
        const float mn = fmaxf(m2, mx2);
        const float ms = mn == -INFINITY ? 0.0f : mn;  // nothing allowed so far: keep l2 = 0 without producing NaNs
        const float alpha = __builtin_amdgcn_exp2f(m2 - ms);
        m2 = mn;
        float psum = 0.0f;
        float pf[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          pf[r] = __builtin_amdgcn_exp2f(s2[r] - ms);
          psum += pf[r];
        }
        l2 = l2 * alpha + psum;
#pragma unroll
        for (int r = 0; r < 16; ++r) o2[r] *= alpha;
        uint4 p0 = pack_bf16x8(pf), p1 = pack_bf16x8(pf + 8);
        o2 = FX_MFMA_32x32x16(vf0, __builtin_bit_cast(bf16x8, p0), o2);
        o2 = FX_MFMA_32x32x16(vf1, __builtin_bit_cast(bf16x8, p1), o2);
      }
    }
  }
  if (!active) return;
  l += __shfl_xor(l, 32, 64);
  if (SPLIT) {
    if (qi < Lq) {
      const int nsplit = gridDim.z;
      const int64_t vstride = (int64_t)gridDim.y * nsplit * Lq * FX_MHA_PART;
      float* wp = part + (((int64_t)bh * nsplit + blockIdx.z) * Lq + qi) * FX_MHA_PART;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        wp[8 * g + 4 * h + 0] = o[4 * g]; wp[8 * g + 4 * h + 1] = o[4 * g + 1];
        wp[8 * g + 4 * h + 2] = o[4 * g + 2]; wp[8 * g + 4 * h + 3] = o[4 * g + 3];
      }
      if (h == 0) wp[32] = m, wp[33] = l;
      if (MASKED) {
        l2 += __shfl_xor(l2, 32, 64);
        wp += vstride;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          wp[8 * g + 4 * h + 0] = o2[4 * g]; wp[8 * g + 4 * h + 1] = o2[4 * g + 1];
          wp[8 * g + 4 * h + 2] = o2[4 * g + 2]; wp[8 * g + 4 * h + 3] = o2[4 * g + 3];
        }
        if (h == 0) wp[32] = m2, wp[33] = l2;
      }
    }
    return;
  }
  float inv = 1.0f / l;
  if (MASKED) {
    l2 += __shfl_xor(l2, 32, 64);
    if (l2 > 0.0f) {  // at least one allowed key: the masked softmax is the answer
      inv = 1.0f / l2;
      o = o2;
    }
  }
  if (qi < Lq) {
    bf16_t* op = out + ((int64_t)b * Lq + qi) * ldo + hd * 32 + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint2 w;
      w.x = pack_bf16x2(o[4 * g] * inv, o[4 * g + 1] * inv);
      w.y = pack_bf16x2(o[4 * g + 2] * inv, o[4 * g + 3] * inv);
      *reinterpret_cast<uint2*>(op + 8 * g) = w;
    }
  }
}

// Merge the per-slice partials: thread = (query, channel d of the head).
__global__ __launch_bounds__(256) void mha32_combine_kernel(const float* __restrict__ part, int nsplit, int masked, bf16_t* __restrict__ out,
                                                             int ldo, int Lq, int heads) {
  const int bh = blockIdx.y, b = bh / heads, hd = bh % heads;
  const int qi = blockIdx.x * 8 + (threadIdx.x >> 5), d = threadIdx.x & 31;
  if (qi >= Lq) return;
  const int64_t vstride = (int64_t)gridDim.y * nsplit * Lq * FX_MHA_PART;
  float res = 0.0f;
  bool done = false;
  for (int var = masked ? 1 : 0; var >= 0 && !done; --var) {
    const float* base = part + var * vstride + ((int64_t)bh * nsplit * Lq + qi) * FX_MHA_PART;
    float M = -INFINITY;
    for (int s = 0; s < nsplit; ++s) M = fmaxf(M, base[(int64_t)s * Lq * FX_MHA_PART + 32]);
    if (M == -INFINITY) continue;  // masked variant with no allowed key anywhere: fall through to the unmasked one
    float l = 0.0f, o = 0.0f;
    for (int s = 0; s < nsplit; ++s) {
      const float* r = base + (int64_t)s * Lq * FX_MHA_PART;
      const float ms = r[32];
      const float w = ms == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(ms - M);
      l += w * r[33];
      o += w * r[d];
    }
    if (l > 0.0f) {
      res = o / l;
      done = true;
    }
  }
  out[((int64_t)b * Lq + qi) * ldo + hd * 32 + d] = f32_to_bf16(res);
}

// key-slice policy shared by fx_mha_workspace_bytes and the launcher (a pure function of the shapes)
static inline void fx_mha_split(int B, int Lq, int Lk, int heads, int* nsplit, int* tps) {
  const int T = (Lk + 31) / 32;
  const int base = ((Lq + 127) / 128) * B * heads;
  int ns = 1;
  if (T >= 32) {
    ns = (T + 15) / 16;
    const int want = 1024 / (base > 0 ? base : 1);
    if (ns > want) ns = want;
    if (ns > 16) ns = 16;
    if (ns < 1) ns = 1;
  }
  *tps = (T + ns - 1) / ns;
  *nsplit = (T + *tps - 1) / *tps;
}

extern "C" int fx_mha_workspace_bytes(int B, int Lq, int Lk, int heads, int masked) {
  if (B <= 0 || Lq <= 0 || Lk <= 0 || heads <= 0) return 0;
  int ns, tps;
  fx_mha_split(B, Lq, Lk, heads, &ns, &tps);
  if (ns <= 1) return 0;
  return (int)((size_t)(masked ? 2 : 1) * B * heads * ns * Lq * FX_MHA_PART * sizeof(float));
}

extern "C" int fx_mha_masked_bf16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* out, int ldo, int B, int Lq,
                                  int Lk, int heads, const uint32_t* mask_bits, int ld_mask_words, void* workspace, size_t workspace_bytes,
                                  fx_stream_t stream_) {
  FX_CHECK_ARG(q && k && v && out && B > 0 && Lq > 0 && Lk > 0 && heads > 0);
  FX_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0 && ldq >= heads * 32 && ldo >= heads * 32);
  FX_CHECK_ARG(!mask_bits || ld_mask_words >= (Lk + 31) / 32);
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  dim3 grid((Lq + 127) / 128, B * heads), block(256);
  int ns = 1, tps = 0;
  const size_t need = (size_t)fx_mha_workspace_bytes(B, Lq, Lk, heads, mask_bits != nullptr);
  if (workspace && need > 0 && workspace_bytes >= need) fx_mha_split(B, Lq, Lk, heads, &ns, &tps);
#define FX_MHA(MASKED, SPLIT)                                                                                                           \
  hipLaunchKernelGGL((mha32_mfma_kernel<14, MASKED, SPLIT>), grid, block, 0, stream, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk,      \
                     (const bf16_t*)v, ldv, (bf16_t*)out, ldo, Lq, Lk, heads, mask_bits, ld_mask_words, (float*)workspace, tps)
  if (ns > 1) {
    grid.z = ns;
    if (mask_bits) FX_MHA(true, true);
    else FX_MHA(false, true);
    hipLaunchKernelGGL(mha32_combine_kernel, dim3((Lq + 7) / 8, B * heads), dim3(256), 0, stream, (const float*)workspace, ns,
                       mask_bits ? 1 : 0, (bf16_t*)out, ldo, Lq, heads);
  } else if (mask_bits) {
    FX_MHA(true, false);
  } else {
    FX_MHA(false, false);
  }
#undef FX_MHA
  return fx_launch_status();
}

extern "C" int fx_mha_bf16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* out, int ldo, int B, int Lq, int Lk,
                           int heads, fx_stream_t stream_) {
  return fx_mha_masked_bf16(q, ldq, k, ldk, v, ldv, out, ldo, B, Lq, Lk, heads, nullptr, 0, nullptr, 0, stream_);
}

// ------------------------------------------------------------------------------------------------
// Multi-scale deformable attention sampling (M*D = 256): one wave per (batch, query); lane = (head =
// lane>>3, 4 channels = (lane&7)*4).  Each tap is an 8-byte load; the 8 lanes of a head read one
// contiguous 64-byte head slice of a value token, the wave touches 8 such slices per tap.  The value
// map of one image (8400 x 256 bf16 = 4.3 MB) stays L2/MALL resident across its 300 queries.
// Bilinear taps follow F.grid_sample(align_corners=False, padding_mode="zeros"):
//   ix = ((2*loc_x - 1 + 1) * W - 1) / 2, taps outside the map contribute 0.
template <int MODE>
__global__ __launch_bounds__(256) void msda_kernel(const bf16_t* __restrict__ value, int ldv, const int32_t* __restrict__ shapes,
                                                    const int32_t* __restrict__ lstart, int L, int P, const float* __restrict__ loc,
                                                    int ld_loc, const float* __restrict__ attn, int ld_attn, const float* __restrict__ ref,
                                                    bf16_t* __restrict__ out, int ldo, int B, int S, int Q, int M) {
  const int lane = threadIdx.x & 63;
  const int bq = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bq >= B * Q) return;
  const int b = bq / Q;
  const int h = lane >> 3, cg = lane & 7;
  const int LP = L * P;
  const bf16_t* vb = value + (int64_t)b * S * ldv + h * 32 + cg * 4;
  const float* locp = loc + (int64_t)bq * ld_loc + h * LP * 2;
  const float* attp = attn + (int64_t)bq * ld_attn + h * LP;
  float rcx = 0.f, rcy = 0.f, rw = 0.f, rh = 0.f, amax = 0.f, ainv = 1.f;
  if (MODE == 1) {
    const float* rp = ref + (int64_t)bq * 4;
    rcx = rp[0]; rcy = rp[1]; rw = rp[2]; rh = rp[3];
    amax = -INFINITY;
    for (int i = 0; i < LP; ++i) amax = fmaxf(amax, attp[i]);
    float s = 0.f;
    for (int i = 0; i < LP; ++i) s += __expf(attp[i] - amax);
    ainv = 1.0f / s;
  }
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int l = 0; l < L; ++l) {
    const int Hl = shapes[2 * l], Wl = shapes[2 * l + 1];
    const bf16_t* vl = vb + (int64_t)lstart[l] * ldv;
    for (int pt = 0; pt < P; ++pt) {
      const int i = l * P + pt;
      float lx = locp[2 * i], ly = locp[2 * i + 1], aw = attp[i];
      if (MODE == 1) {
        lx = rcx + lx / (float)P * rw * 0.5f;
        ly = rcy + ly / (float)P * rh * 0.5f;
        aw = __expf(aw - amax) * ainv;
      }
      float gx = 2.0f * lx - 1.0f, gy = 2.0f * ly - 1.0f;
      float ix = ((gx + 1.0f) * (float)Wl - 1.0f) * 0.5f, iy = ((gy + 1.0f) * (float)Hl - 1.0f) * 0.5f;
      float fx0 = floorf(ix), fy0 = floorf(iy);
      int x0 = (int)fx0, y0 = (int)fy0;
      float tx = ix - fx0, ty = iy - fy0;
      float w00 = (1.f - tx) * (1.f - ty), w01 = tx * (1.f - ty), w10 = (1.f - tx) * ty, w11 = tx * ty;
      bool xin0 = (unsigned)x0 < (unsigned)Wl, xin1 = (unsigned)(x0 + 1) < (unsigned)Wl;
      bool yin0 = (unsigned)y0 < (unsigned)Hl, yin1 = (unsigned)(y0 + 1) < (unsigned)Hl;
      float s[4] = {0.f, 0.f, 0.f, 0.f};
      auto tap = [&](bool ok, int yy, int xx, float w) {
        if (ok) {
          uint2 t = *reinterpret_cast<const uint2*>(vl + (int64_t)(yy * Wl + xx) * ldv);
          s[0] += w * bf16lo_to_f32(t.x);
          s[1] += w * bf16hi_to_f32(t.x);
          s[2] += w * bf16lo_to_f32(t.y);
          s[3] += w * bf16hi_to_f32(t.y);
        }
      };
      tap(yin0 && xin0, y0, x0, w00);
      tap(yin0 && xin1, y0, x0 + 1, w01);
      tap(yin1 && xin0, y0 + 1, x0, w10);
      tap(yin1 && xin1, y0 + 1, x0 + 1, w11);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += aw * s[j];
    }
  }
  uint2 o;
  o.x = pack_bf16x2(acc[0], acc[1]);
  o.y = pack_bf16x2(acc[2], acc[3]);
  *reinterpret_cast<uint2*>(out + (int64_t)bq * ldo + h * 32 + cg * 4) = o;
}

// L = 3 levels x P = 4 points (every deformable layer of the three model families): the loops unrolled, a level's 16 taps requested
// together (clamped address, zero weight for taps outside the map - no branch between the loads), offsets / logits read as float4.
// The generic kernel above walks the 12 points one after another, each waiting for its own four taps: ~12 dependent L2 round trips
// per wave with under five waves per SIMD to hide them (32 us for 16 x 300 queries); here a wave waits three times.
template <int MODE>
__global__ __launch_bounds__(256) void msda_l3p4_kernel(const bf16_t* __restrict__ value, int ldv, const int32_t* __restrict__ shapes,
                                                         const int32_t* __restrict__ lstart, const float* __restrict__ loc, int ld_loc,
                                                         const float* __restrict__ attn, int ld_attn, const float* __restrict__ ref,
                                                         bf16_t* __restrict__ out, int ldo, int B, int S, int Q) {
  constexpr int L = 3, P = 4, LP = 12;
  const int lane = threadIdx.x & 63;
  const int bq = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bq >= B * Q) return;
  const int b = bq / Q;
  const int h = lane >> 3, cg = lane & 7;
  const bf16_t* vb = value + (int64_t)b * S * ldv + h * 32 + cg * 4;
  const float4* locp = reinterpret_cast<const float4*>(loc + (int64_t)bq * ld_loc + h * LP * 2);
  const float4* attp = reinterpret_cast<const float4*>(attn + (int64_t)bq * ld_attn + h * LP);
  float lxy[LP * 2], aw[LP];
#pragma unroll
  for (int i = 0; i < LP / 2; ++i) {
    const float4 v = locp[i];
    lxy[4 * i] = v.x; lxy[4 * i + 1] = v.y; lxy[4 * i + 2] = v.z; lxy[4 * i + 3] = v.w;
  }
#pragma unroll
  for (int i = 0; i < LP / 4; ++i) {
    const float4 v = attp[i];
    aw[4 * i] = v.x; aw[4 * i + 1] = v.y; aw[4 * i + 2] = v.z; aw[4 * i + 3] = v.w;
  }
  int Hs[L], Ws[L], st[L];
#pragma unroll
  for (int l = 0; l < L; ++l) { Hs[l] = shapes[2 * l]; Ws[l] = shapes[2 * l + 1]; st[l] = lstart[l]; }
  if (MODE == 1) {
    const float4 rp = *reinterpret_cast<const float4*>(ref + (int64_t)bq * 4);
    float amax = -INFINITY;
#pragma unroll
    for (int i = 0; i < LP; ++i) amax = fmaxf(amax, aw[i]);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < LP; ++i) { aw[i] = __expf(aw[i] - amax); sum += aw[i]; }
    const float ainv = 1.0f / sum;
#pragma unroll
    for (int i = 0; i < LP; ++i) {
      aw[i] *= ainv;
      lxy[2 * i] = rp.x + lxy[2 * i] / (float)P * rp.z * 0.5f;
      lxy[2 * i + 1] = rp.y + lxy[2 * i + 1] / (float)P * rp.w * 0.5f;
    }
  }
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const int Hl = Hs[l], Wl = Ws[l];
    const bf16_t* vl = vb + (int64_t)st[l] * ldv;
    uint2 t[P][4];
    float w[P][4];
#pragma unroll
    for (int pt = 0; pt < P; ++pt) {
      const int i = l * P + pt;
      const float gx = 2.0f * lxy[2 * i] - 1.0f, gy = 2.0f * lxy[2 * i + 1] - 1.0f;
      const float ix = ((gx + 1.0f) * (float)Wl - 1.0f) * 0.5f, iy = ((gy + 1.0f) * (float)Hl - 1.0f) * 0.5f;
      const float fx0 = floorf(ix), fy0 = floorf(iy);
      const int x0 = (int)fx0, y0 = (int)fy0;
      const float tx = ix - fx0, ty = iy - fy0;
      const bool xin0 = (unsigned)x0 < (unsigned)Wl, xin1 = (unsigned)(x0 + 1) < (unsigned)Wl;
      const bool yin0 = (unsigned)y0 < (unsigned)Hl, yin1 = (unsigned)(y0 + 1) < (unsigned)Hl;
      w[pt][0] = (yin0 && xin0) ? aw[i] * (1.f - tx) * (1.f - ty) : 0.f;
      w[pt][1] = (yin0 && xin1) ? aw[i] * tx * (1.f - ty) : 0.f;
      w[pt][2] = (yin1 && xin0) ? aw[i] * (1.f - tx) * ty : 0.f;
      w[pt][3] = (yin1 && xin1) ? aw[i] * tx * ty : 0.f;
      const int xc0 = min(max(x0, 0), Wl - 1), xc1 = min(max(x0 + 1, 0), Wl - 1);
      const int yc0 = min(max(y0, 0), Hl - 1), yc1 = min(max(y0 + 1, 0), Hl - 1);
      t[pt][0] = *reinterpret_cast<const uint2*>(vl + (int64_t)(yc0 * Wl + xc0) * ldv);
      t[pt][1] = *reinterpret_cast<const uint2*>(vl + (int64_t)(yc0 * Wl + xc1) * ldv);
      t[pt][2] = *reinterpret_cast<const uint2*>(vl + (int64_t)(yc1 * Wl + xc0) * ldv);
      t[pt][3] = *reinterpret_cast<const uint2*>(vl + (int64_t)(yc1 * Wl + xc1) * ldv);
    }
    // same summation order as the generic kernel: the four taps of a point first (s), then acc += aw * s - here with aw folded into
    // the tap weights, which changes the rounding by an ulp; parity tests hold both to the same tolerance
#pragma unroll
    for (int pt = 0; pt < P; ++pt) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        acc[0] += w[pt][k] * bf16lo_to_f32(t[pt][k].x);
        acc[1] += w[pt][k] * bf16hi_to_f32(t[pt][k].x);
        acc[2] += w[pt][k] * bf16lo_to_f32(t[pt][k].y);
        acc[3] += w[pt][k] * bf16hi_to_f32(t[pt][k].y);
      }
    }
  }
  uint2 o;
  o.x = pack_bf16x2(acc[0], acc[1]);
  o.y = pack_bf16x2(acc[2], acc[3]);
  *reinterpret_cast<uint2*>(out + (int64_t)bq * ldo + h * 32 + cg * 4) = o;
}

extern "C" int fx_msda_bf16(const void* value, int ldv, const int32_t* spatial_shapes, const int32_t* level_start, int L, int P,
                            const float* loc, int ld_loc, const float* attn, int ld_attn, const float* ref, int mode, void* out, int ldo,
                            int B, int S, int Q, int M, fx_stream_t stream_) {
  FX_CHECK_ARG(value && spatial_shapes && level_start && loc && attn && out && B > 0 && S > 0 && Q > 0 && L > 0 && P > 0);
  if (M != 8) return FX_ERR_UNSUPPORTED;  // M*D = 256 = 64 lanes x 4 channels
  FX_CHECK_ARG(ldv >= 256 && ldo >= 256 && ldv % 4 == 0 && ldo % 4 == 0);
  FX_CHECK_ARG(mode == 0 || (mode == 1 && ref));
  FX_CHECK_ARG(ld_loc >= M * L * P * 2 && ld_attn >= M * L * P);
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  dim3 grid((B * Q + 3) / 4), block(256);
  static const int fast = fx_tune("FX_MSDA_L3P4", 1);
  // the unrolled form needs 16-byte rows of offsets / logits (float4 loads); NaN values in the map are the one thing it treats
  // differently from the generic kernel (a zero-weight clamped tap reads them): value projections are finite
  if (fast && L == 3 && P == 4 && ld_loc % 4 == 0 && ld_attn % 4 == 0 && ((uintptr_t)loc % 16) == 0 && ((uintptr_t)attn % 16) == 0 &&
      (mode == 0 || ((uintptr_t)ref % 16) == 0)) {
    if (mode == 0)
      hipLaunchKernelGGL(msda_l3p4_kernel<0>, grid, block, 0, stream, (const bf16_t*)value, ldv, spatial_shapes, level_start, loc, ld_loc, attn, ld_attn,
                         ref, (bf16_t*)out, ldo, B, S, Q);
    else
      hipLaunchKernelGGL(msda_l3p4_kernel<1>, grid, block, 0, stream, (const bf16_t*)value, ldv, spatial_shapes, level_start, loc, ld_loc, attn, ld_attn,
                         ref, (bf16_t*)out, ldo, B, S, Q);
    return fx_launch_status();
  }
  if (mode == 0)
    hipLaunchKernelGGL(msda_kernel<0>, grid, block, 0, stream, (const bf16_t*)value, ldv, spatial_shapes, level_start, L, P, loc, ld_loc,
                       attn, ld_attn, ref, (bf16_t*)out, ldo, B, S, Q, M);
  else
    hipLaunchKernelGGL(msda_kernel<1>, grid, block, 0, stream, (const bf16_t*)value, ldv, spatial_shapes, level_start, L, P, loc, ld_loc,
                       attn, ld_attn, ref, (bf16_t*)out, ldo, B, S, Q, M);
  return fx_launch_status();
}
