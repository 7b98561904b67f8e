// MaskFormer-specific kernels (gfx950): FPN nearest-upsample + lateral add, query x pixel mask logits on MFMA
// (attention-mask bits / sigmoid mask probabilities), class softmax head, and the device side of
// MaskFormerProcessor.postprocess (bilinear x4 upsample of the mask probabilities fused with threshold, mask score,
// bounding box and bit-packed binary masks).  Reference: focoos/models/fai_mf/modelling.py, fai_mf/processor.py.
#include "common.h"

// ------------------------------------------------------------------------------------------------
// y = lateral + nearest_upsample(top)   (TransformerFPN.forward_features, fai_mf/modelling.py:364;
// F.interpolate(mode="nearest"): src = floor(dst * in/out)).  8 channels (16 B) per thread.
// ATen's nearest_idx (UpSample.h): identity / exact x2 by shift, otherwise floorf(dst * (float)in / out) clamped - in float, which is
// NOT floor(dst * in / out) for every ratio (in 14, out 20, dst 10: 6, not 7).
__device__ __forceinline__ int nearest_src(int dst, int in_size, int out_size) {
  if (out_size == in_size) return dst;
  if (out_size == 2 * in_size) return dst >> 1;
  const float scale = (float)in_size / (float)out_size;
  return min((int)floorf((float)dst * scale), in_size - 1);
}

__global__ __launch_bounds__(256) void upsample_nearest_add_kernel(const bf16_t* __restrict__ lat, int ldl, const bf16_t* __restrict__ top,
                                                                   int ldt, bf16_t* __restrict__ out, int ldo, int B, int H, int W, int Hs,
                                                                   int Ws, int C8) {
  const int64_t total = (int64_t)B * H * W * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    int64_t p = i / C8;
    const int x = (int)(p % W);
    p /= W;
    const int y = (int)(p % H);
    const int b = (int)(p / H);
    const int ys = nearest_src(y, Hs, H), xs = nearest_src(x, Ws, W);
    float a[8], t[8];
    unpack_bf16x8(*reinterpret_cast<const uint4*>(lat + (((int64_t)b * H + y) * W + x) * ldl + c8 * 8), a);
    unpack_bf16x8(*reinterpret_cast<const uint4*>(top + (((int64_t)b * Hs + ys) * Ws + xs) * ldt + c8 * 8), t);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] += t[j];
    *reinterpret_cast<uint4*>(out + (((int64_t)b * H + y) * W + x) * ldo + c8 * 8) = pack_bf16x8(a);
  }
}

extern "C" int fx_upsample_nearest_add_nhwc_bf16(const void* lateral, int ldl, const void* top, int ldt, void* out, int ldo, int B, int H,
                                                 int W, int Hs, int Ws, int C, fx_stream_t stream_) {
  FX_CHECK_ARG(lateral && top && out && B > 0 && H > 0 && W > 0 && Hs > 0 && Ws > 0 && C > 0 && C % 8 == 0);
  FX_CHECK_ARG(ldl >= C && ldt >= C && ldo >= C && ldl % 8 == 0 && ldt % 8 == 0 && ldo % 8 == 0);
  int64_t total = (int64_t)B * H * W * (C / 8);
  int64_t grid = (total + 255) / 256;
  if (grid > 256 * 32) grid = 256 * 32;
  hipLaunchKernelGGL(upsample_nearest_add_kernel, dim3((int)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_),
                     (const bf16_t*)lateral, ldl, (const bf16_t*)top, ldt, (bf16_t*)out, ldo, B, H, W, Hs, Ws, C / 8);
  return fx_launch_status();
}

// Adjoint of the up-sampling half of the kernel above, in gather form (deterministic): dtop[b,ys,xs,:] = sum of dy over the output
// pixels whose nearest source is (ys, xs).  The map dst -> src is monotone, so a source's pre-image is a short contiguous run per axis
// (2 for the x2 case, 1-2 for ceil sizes); it is found by testing the candidates around src * out / in with the forward's own formula.
__global__ __launch_bounds__(256) void upsample_nearest_bwd_kernel(const bf16_t* __restrict__ dy, int lddy, bf16_t* __restrict__ dtop, int lddt, int B,
                                                                   int H, int W, int Hs, int Ws, int C8) {
  const int64_t total = (int64_t)B * Hs * Ws * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    int64_t p = i / C8;
    const int xs = (int)(p % Ws);
    p /= Ws;
    const int ys = (int)(p % Hs);
    const int b = (int)(p / Hs);
    const int y_lo = max(0, (int)(((int64_t)ys * H) / Hs) - 1), y_hi = min(H - 1, (int)(((int64_t)(ys + 1) * H + Hs - 1) / Hs) + 1);
    const int x_lo = max(0, (int)(((int64_t)xs * W) / Ws) - 1), x_hi = min(W - 1, (int)(((int64_t)(xs + 1) * W + Ws - 1) / Ws) + 1);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int y = y_lo; y <= y_hi; ++y) {
      if (nearest_src(y, Hs, H) != ys) continue;
      for (int x = x_lo; x <= x_hi; ++x) {
        if (nearest_src(x, Ws, W) != xs) continue;
        float f[8];
        unpack_bf16x8(*reinterpret_cast<const uint4*>(dy + (((int64_t)b * H + y) * W + x) * lddy + c8 * 8), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += f[j];
      }
    }
    *reinterpret_cast<uint4*>(dtop + (((int64_t)b * Hs + ys) * Ws + xs) * lddt + c8 * 8) = pack_bf16x8(acc);
  }
}

extern "C" int fx_upsample_nearest_bwd_nhwc_bf16(const void* dy, int lddy, void* dtop, int lddt, int B, int H, int W, int Hs, int Ws, int C,
                                                 fx_stream_t stream_) {
  FX_CHECK_ARG(dy && dtop && B > 0 && H > 0 && W > 0 && Hs > 0 && Ws > 0 && C > 0 && C % 8 == 0);
  FX_CHECK_ARG(lddy >= C && lddt >= C && lddy % 8 == 0 && lddt % 8 == 0);
  int64_t total = (int64_t)B * Hs * Ws * (C / 8);
  int64_t grid = (total + 255) / 256;
  if (grid > 256 * 32) grid = 256 * 32;
  hipLaunchKernelGGL(upsample_nearest_bwd_kernel, dim3((int)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), (const bf16_t*)dy, lddy,
                     (bf16_t*)dtop, lddt, B, H, W, Hs, Ws, C / 8);
  return fx_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Query x pixel logits: out[b, q, p] = sum_c embed[b, q, c] * feat[b, p, c], C = 256, Q <= 128
// (PredictionHeads.forward: torch.einsum("bqc,bchw->bqhw", mask_embed, mask_features), fai_mf/modelling.py:88).
// One workgroup = one image's query embeddings staged in LDS (128 x 512 B, 16-byte chunks XOR-swizzled by the row)
// x 256 pixels (each wave two 32-pixel tiles).  Pixel rows go from global memory straight into the MFMA B operand
// (lane = pixel lane&31, 8 channels at 16*kk + 8*(lane>>5)); A = 32 queries from LDS; the 32x32 accumulator has
// column = pixel, row = query, so stores along a query row are 128-byte contiguous and a wave ballot of (logit < 0)
// IS the 32-pixel word of the attention-mask bitmap.
//   MODE 0: f32 logits        out_f32[(b*Q+q)*ldo + p]
//   MODE 1: f32 sigmoid(logit)
//   MODE 2: attention-mask bits (fai_mf/modelling.py:104: mask = resized logit < 0; the bilinear resize of the logits is
//           applied to `feat` beforehand, which is the same linear map): bits[(b*Q+q)*ldw + p/32], bit p&31; pixels
//           beyond P read as masked.
template <int MODE, int CH>   // CH = channels of the mask embedding (256: fai-mf, 128: bisenetformer)
__global__ __launch_bounds__(256) void query_pixel_logits_kernel(const bf16_t* __restrict__ embed, int lde, const bf16_t* __restrict__ feat,
                                                                 int ldf, float* __restrict__ out, int ldo, uint32_t* __restrict__ bits,
                                                                 int ldw, int Q, int P) {
  constexpr int NCH = CH / 8;      // 16-byte chunks per embedding row
  constexpr int KS = CH / 16;      // MFMA k-steps
  constexpr int ROWB = CH * 2;     // bytes per LDS row
  extern __shared__ __attribute__((aligned(16))) unsigned char es[];  // [128][NCH chunks of 16 B], chunk index XOR-swizzled by the row
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const bf16_t* eb = embed + (int64_t)b * Q * lde;
  for (int i = tid; i < 128 * NCH; i += 256) {
    const int row = i / NCH, c = i % NCH;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < Q) v = *reinterpret_cast<const uint4*>(eb + (int64_t)row * lde + c * 8);
    *reinterpret_cast<uint4*>(es + row * ROWB + ((c ^ (row & (NCH - 1))) << 4)) = v;
  }
  __syncthreads();
  const int j = lane & 31, h = lane >> 5;
  const bf16_t* fb = feat + (int64_t)b * P * ldf;
  const int nqt = (Q + 31) >> 5;
#pragma unroll 1
  for (int tt = 0; tt < 2; ++tt) {
    const int p0 = blockIdx.x * 256 + (wave * 2 + tt) * 32;
    if (p0 >= P) break;
    const int p = p0 + j;
    bf16x8 kf[KS];
    {
      const bf16_t* fp = fb + (int64_t)(p < P ? p : P - 1) * ldf + 8 * h;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) kf[kk] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(fp + kk * 16));
    }
#pragma unroll 1
    for (int qt = 0; qt < nqt; ++qt) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
      const int row = qt * 32 + j;
      const unsigned char* er = es + row * ROWB;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        bf16x8 a = *reinterpret_cast<const bf16x8*>(er + (((kk * 2 + h) ^ (row & (NCH - 1))) << 4));
        acc = FX_MFMA_32x32x16(a, kf[kk], acc);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qq = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (MODE == 2) {
          const unsigned long long bal = __ballot(acc[r] < 0.0f || p >= P);
          // low word: query row (h = 0), high word: query row + 4 (h = 1)
          const int ql = qt * 32 + (r & 3) + 8 * (r >> 2);
          if (lane == 0 && ql < Q) bits[((int64_t)b * Q + ql) * ldw + (p0 >> 5)] = (uint32_t)bal;
          if (lane == 32 && ql + 4 < Q) bits[((int64_t)b * Q + ql + 4) * ldw + (p0 >> 5)] = (uint32_t)(bal >> 32);
        } else if (qq < Q && p < P) {
          out[((int64_t)b * Q + qq) * ldo + p] = MODE == 1 ? 1.0f / (1.0f + __expf(-acc[r])) : acc[r];
        }
      }
    }
  }
}

extern "C" int fx_query_pixel_logits_bf16(const void* embed, int lde, const void* feat, int ldf, int mode, float* out, int ldo,
                                          uint32_t* bits, int ld_words, int B, int Q, int P, int C, fx_stream_t stream_) {
  FX_CHECK_ARG(embed && feat && B > 0 && Q > 0 && P > 0);
  if ((C != 256 && C != 128) || Q > 128) return FX_ERR_UNSUPPORTED;
  FX_CHECK_ARG(lde >= C && ldf >= C && lde % 8 == 0 && ldf % 8 == 0);
  FX_CHECK_ARG(mode == 2 ? (bits && ld_words >= (P + 31) / 32) : ((mode == 0 || mode == 1) && out && ldo >= P));
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  dim3 grid((P + 255) / 256, B), block(256);
  const size_t lds = (size_t)128 * C * 2;
#define FX_QPL(M, CH)                                                                                                                  \
  hipLaunchKernelGGL((query_pixel_logits_kernel<M, CH>), grid, block, lds, stream, (const bf16_t*)embed, lde, (const bf16_t*)feat, ldf, out, \
                     ldo, bits, ld_words, Q, P)
  if (C == 256) {
    if (mode == 0) FX_QPL(0, 256);
    else if (mode == 1) FX_QPL(1, 256);
    else FX_QPL(2, 256);
  } else {
    if (mode == 0) FX_QPL(0, 128);
    else if (mode == 1) FX_QPL(1, 128);
    else FX_QPL(2, 128);
  }
#undef FX_QPL
  return fx_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Class head tail (MaskFormerHead.forward, fai_mf/modelling.py:610-613): softmax over K+1 logits with the
// no-object column dropped (or sigmoid when cls_sigmoid), plus the per-query max / argmax that
// MaskFormerProcessor.postprocess takes first (processor.py:212).  One wave per query row, K+1 <= 256.
__global__ __launch_bounds__(256) void mf_class_head_kernel(const float* __restrict__ logits, int ldl, float* __restrict__ probs,
                                                            float* __restrict__ score, int32_t* __restrict__ label, int rows, int K,
                                                            int use_sigmoid) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* lp = logits + (int64_t)row * ldl;
  float v[4], mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    v[i] = c <= K ? lp[c] : -INFINITY;
    mx = fmaxf(mx, v[i]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float e[4], sum = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    e[i] = (lane + 64 * i) <= K ? __expf(v[i] - mx) : 0.0f;
    sum += e[i];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float inv = 1.0f / sum;
  float best = -INFINITY;
  int bi = 0x7fffffff;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    if (c < K) {
      const float pr = use_sigmoid ? 1.0f / (1.0f + __expf(-v[i])) : e[i] * inv;
      probs[(int64_t)row * K + c] = pr;
      if (pr > best) best = pr, bi = c;  // ascending c per lane: keeps the lowest index among equal values
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) best = ob, bi = oi;
  }
  if (lane == 0) {
    score[row] = best;
    label[row] = bi;
  }
}

extern "C" int fx_mf_class_head(const float* logits, int ldl, float* probs, float* score, int32_t* label, int rows, int K,
                                int use_sigmoid, fx_stream_t stream_) {
  FX_CHECK_ARG(logits && probs && score && label && rows > 0 && K > 0 && ldl >= K + 1);
  if (K + 1 > 256) return FX_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(mf_class_head_kernel, dim3((rows + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), logits, ldl, probs,
                     score, label, rows, K, use_sigmoid);
  return fx_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Bilinear upsample of the mask probabilities, align_corners=False (FAIMaskFormer.forward, fai_mf/modelling.py:723):
// src = max((dst + 0.5) * in/out - 0.5, 0), i0 = floor(src), i1 = min(i0 + 1, in - 1), lambda = src - i0.
struct LerpAxis {
  int i0, i1;
  float w0, w1;
};

__device__ __forceinline__ LerpAxis lerp_axis(int dst, float scale, int in) {
  float src = ((float)dst + 0.5f) * scale - 0.5f;
  src = src < 0.0f ? 0.0f : src;
  LerpAxis a;
  a.i0 = (int)src;
  if (a.i0 > in - 1) a.i0 = in - 1;
  a.i1 = a.i0 + (a.i0 < in - 1 ? 1 : 0);
  a.w1 = src - (float)a.i0;
  a.w0 = 1.0f - a.w1;
  return a;
}

// Unfused multiply/add in the order of PyTorch's CPU kernel (UpSampleKernel.cpp, linear interpolate): every kernel
// below that thresholds these values must round identically, so FMA contraction is switched off here.
__device__ __forceinline__ float lerp_taps(float p00, float p01, float p10, float p11, float wy0, float wy1, float wx0, float wx1) {
#pragma clang fp contract(off)
  const float top = wx0 * p00 + wx1 * p01;
  const float bot = wx0 * p10 + wx1 * p11;
  return wy0 * top + wy1 * bot;
}

// the two halves of lerp_taps, for callers that reuse the x-interpolated rows across output rows
__device__ __forceinline__ float lerp_row(const float* __restrict__ row, const LerpAxis& ax) {
#pragma clang fp contract(off)
  return ax.w0 * row[ax.i0] + ax.w1 * row[ax.i1];
}
__device__ __forceinline__ float lerp_col(float top, float bot, const LerpAxis& ay) {
#pragma clang fp contract(off)
  return ay.w0 * top + ay.w1 * bot;
}

__device__ __forceinline__ float lerp2(const float* __restrict__ p, int w, const LerpAxis& ay, const LerpAxis& ax) {
  const float* r0 = p + (int64_t)ay.i0 * w;
  const float* r1 = p + (int64_t)ay.i1 * w;
  return lerp_taps(r0[ax.i0], r0[ax.i1], r1[ax.i0], r1[ax.i1], ay.w0, ay.w1, ax.w0, ax.w1);
}

// masks[b,q,y,x] f32 — the reference's `masks` output tensor (B x Q x H x W x 4 bytes: pure HBM write).
template <typename OutT>
__global__ __launch_bounds__(256) void mf_upsample_probs_kernel(const float* __restrict__ lo, int h, int w, OutT* __restrict__ out, int H,
                                                                int W, float sy, float sx) {
  const int bq = blockIdx.y;
  const float* p = lo + (int64_t)bq * h * w;
  OutT* o = out + (int64_t)bq * H * W;
  const int rows_per_block = 8;
  const int y0 = blockIdx.x * rows_per_block;
  for (int x = threadIdx.x; x < W; x += 256) {
    const LerpAxis ax = lerp_axis(x, sx, w);
    for (int y = y0; y < y0 + rows_per_block && y < H; ++y) {
      const LerpAxis ay = lerp_axis(y, sy, h);
      if constexpr (sizeof(OutT) == 4) o[(int64_t)y * W + x] = lerp2(p, w, ay, ax);
      else o[(int64_t)y * W + x] = f32_to_bf16(lerp2(p, w, ay, ax));
    }
  }
}

template <typename OutT>
static int launch_mf_upsample(const float* lowres, int h, int w, OutT* out, int H, int W, int BQ, hipStream_t stream) {
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  // (a form with eight pixels per thread and 16- / 32-byte stores was measured SLOWER, 1 354 vs 1 480 img/s on the full-masks leg: it recomputes the
  // column taps per row where this one keeps them in registers over its eight rows; the launch is bound by its tap arithmetic, not by the stores)
  hipLaunchKernelGGL(mf_upsample_probs_kernel<OutT>, dim3((H + 7) / 8, BQ), dim3(256), 0, stream, lowres, h, w, out, H, W, sy, sx);
  return fx_launch_status();
}

extern "C" int fx_mf_upsample_probs_f32(const float* lowres, int h, int w, float* out, int H, int W, int BQ, fx_stream_t stream_) {
  FX_CHECK_ARG(lowres && out && h > 0 && w > 0 && H > 0 && W > 0 && BQ > 0);
  return launch_mf_upsample<float>(lowres, h, w, out, H, W, BQ, reinterpret_cast<hipStream_t>(stream_));
}

extern "C" int fx_mf_upsample_probs_bf16(const float* lowres, int h, int w, void* out, int H, int W, int BQ, fx_stream_t stream_) {
  FX_CHECK_ARG(lowres && out && h > 0 && w > 0 && H > 0 && W > 0 && BQ > 0);
  return launch_mf_upsample<bf16_t>(lowres, h, w, reinterpret_cast<bf16_t*>(out), H, W, BQ, reinterpret_cast<hipStream_t>(stream_));
}

// Per (image, query, band of 32 output rows): number of pixels with upsampled probability >= mask_threshold, the sum of
// those probabilities, and the extent of the binary mask (processor.py:229-232, 246-252; utils/vision.py:344-370).
// Partials are written per band and reduced in a fixed order by mf_select_kernel (deterministic, no float atomics).
#define FX_MF_BAND 32
struct MfPartial {
  int32_t cnt;
  float sum;
  int32_t x0, x1, y0, y1;
};

// A query whose class score is <= threshold can never be kept (its final score is the class score times a mean
// probability <= 1), so its masks are not evaluated at all: with trained weights most of the Q queries exit here.
// X4: exact x4 upsample (H = 4h, W = 4w, the MaskFormer case): one thread per quarter-resolution cell produces its 4x4
// output pixels from the 3x3 neighbourhood (9 loads per 16 pixels) with the same tap arithmetic as the generic path.
// NOTE the summation order over pixels differs between the two paths (sum is a float reduction).
// BITS (X4 with w % 8 == 0 only): the pass also writes the binary mask of every EVALUATED query bit-packed (allbits[(b*Q + q)][y][W/32],
// bit x & 31) - the 16 flags of a cell are 4 bits of 4 output rows, eight neighbouring lanes make a word (three xor-shuffles) - so that the
// masks of the kept detections are a plane copy afterwards (mf_compact_masks_kernel) instead of a second interpolation pass over
// H x W pixels per detection (mf_pack_masks_kernel: 465 us against the 282 us of this kernel at 8 x 100 x 800^2).
template <bool X4, bool BITS = false>
__global__ __launch_bounds__(256) void mf_mask_stats_kernel(const float* __restrict__ lo, int h, int w, int H, int W, float sy, float sx,
                                                            float thr, const float* __restrict__ score, float score_thr,
                                                            MfPartial* __restrict__ part, int nband, uint32_t* __restrict__ allbits = nullptr) {
  const int bq = blockIdx.y, band = blockIdx.x;
  if (score_thr > 0.0f && !(score[bq] > score_thr)) {
    if (threadIdx.x == 0) part[(int64_t)bq * nband + band] = MfPartial{0, 0.0f, 0x7fffffff, -1, 0x7fffffff, -1};
    return;
  }
  const float* p = lo + (int64_t)bq * h * w;
  const int ya = band * FX_MF_BAND, yb = min(H, ya + FX_MF_BAND);
  int cnt = 0, x0 = 0x7fffffff, x1 = -1, y0 = 0x7fffffff, y1 = -1;
  float sum = 0.0f;
  if (X4) {
    const int ia = ya >> 2, ib = yb >> 2;  // FX_MF_BAND % 4 == 0
    const int ncell = (ib - ia) * w;
    for (int c0 = 0; c0 < ncell; c0 += 256) {      // uniform trip count: the BITS form exchanges flags between lanes
      const int c = c0 + (int)threadIdx.x;
      unsigned flags = 0;                          // bit 4 r + q: output pixel (4 i + r, 4 j + q) is inside the mask
      const int i = ia + c / w, j = c % w;
      if (c >= ncell) {
      } else if (i >= 1 && i <= h - 2 && j >= 1 && j <= w - 2) {
        float v[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b) v[a][b] = p[(int64_t)(i - 1 + a) * w + (j - 1 + b)];
        // output offset r in 0..3: src = cell + (r - 1.5)/4 -> taps (cell-1, cell) with lambda .625/.875 for r < 2 and
        // (cell, cell+1) with lambda .125/.375 for r >= 2: exactly what lerp_axis yields for interior cells (all values
        // are exact in fp32), so this path is bit-identical to the generic one.
        const float lam[4] = {0.625f, 0.875f, 0.125f, 0.375f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int a0 = r < 2 ? 0 : 1;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int b0 = q < 2 ? 0 : 1;
            const float val = lerp_taps(v[a0][b0], v[a0][b0 + 1], v[a0 + 1][b0], v[a0 + 1][b0 + 1], 1.0f - lam[r], lam[r], 1.0f - lam[q], lam[q]);
            if (val >= thr) {
              const int x = 4 * j + q, y = 4 * i + r;
              if (BITS) flags |= 1u << (4 * r + q);
              ++cnt;
              sum += val;
              x0 = min(x0, x); x1 = max(x1, x);
              y0 = min(y0, y); y1 = max(y1, y);
            }
          }
        }
      } else {  // border cells: clamped taps, generic arithmetic
        for (int r = 0; r < 4; ++r) {
          const int y = 4 * i + r;
          const LerpAxis ay = lerp_axis(y, sy, h);
          for (int q = 0; q < 4; ++q) {
            const int x = 4 * j + q;
            const float val = lerp2(p, w, ay, lerp_axis(x, sx, w));
            if (val >= thr) {
              if (BITS) flags |= 1u << (4 * r + q);
              ++cnt;
              sum += val;
              x0 = min(x0, x); x1 = max(x1, x);
              y0 = min(y0, y); y1 = max(y1, y);
            }
          }
        }
      }
      if constexpr (BITS) {   // w % 8 == 0: the eight lanes 8g .. 8g+7 hold cells j0 .. j0+7 (j0 % 8 == 0) of one cell row = one 32-bit word per output row
        const int l8 = threadIdx.x & 7;
        uint32_t* wo = allbits + ((int64_t)bq * H + 4 * i) * (W >> 5) + (j >> 3);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          unsigned word = ((flags >> (4 * r)) & 0xFu) << (4 * l8);
          word |= __shfl_xor(word, 1, 64);
          word |= __shfl_xor(word, 2, 64);
          word |= __shfl_xor(word, 4, 64);
          if (l8 == 0 && c < ncell) wo[(int64_t)r * (W >> 5)] = word;
        }
      }
    }
  } else {
    for (int x = threadIdx.x; x < W; x += 256) {
      const LerpAxis ax = lerp_axis(x, sx, w);
      for (int y = ya; y < yb; ++y) {
        const LerpAxis ay = lerp_axis(y, sy, h);
        const float v = lerp2(p, w, ay, ax);
        if (v >= thr) {
          ++cnt;
          sum += v;
          x0 = min(x0, x); x1 = max(x1, x);
          y0 = min(y0, y); y1 = max(y1, y);
        }
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    cnt += __shfl_xor(cnt, o, 64);
    sum += __shfl_xor(sum, o, 64);
    x0 = min(x0, __shfl_xor(x0, o, 64)); x1 = max(x1, __shfl_xor(x1, o, 64));
    y0 = min(y0, __shfl_xor(y0, o, 64)); y1 = max(y1, __shfl_xor(y1, o, 64));
  }
  __shared__ MfPartial red[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = MfPartial{cnt, sum, x0, x1, y0, y1};
  __syncthreads();
  if (threadIdx.x == 0) {
    MfPartial r = red[0];
    for (int i = 1; i < 4; ++i) {
      r.cnt += red[i].cnt; r.sum += red[i].sum;
      r.x0 = min(r.x0, red[i].x0); r.x1 = max(r.x1, red[i].x1);
      r.y0 = min(r.y0, red[i].y0); r.y1 = max(r.y1, red[i].y1);
    }
    part[(int64_t)bq * nband + band] = r;
  }
}

// Selection (processor.py:229-262): keep queries whose binary mask has more than one pixel; score = class score x
// mean probability inside the mask (with the reference's 1e-3 scaling and +1e-5 in the denominator); keep score > threshold
// (all non-empty masks when threshold <= 0).  Survivors are compacted in query order, like nonzero().
__global__ __launch_bounds__(128) void mf_select_kernel(const MfPartial* __restrict__ part, int nband, const float* __restrict__ score,
                                                        const int32_t* __restrict__ label, int Q, float thr, int use_mask_score,
                                                        int32_t* __restrict__ det_count, int32_t* __restrict__ det_query,
                                                        float* __restrict__ det_score, int32_t* __restrict__ det_label,
                                                        int32_t* __restrict__ det_box, int32_t* __restrict__ mask_area) {
  const int b = blockIdx.x, q = threadIdx.x;
  MfPartial r{0, 0.0f, 0x7fffffff, -1, 0x7fffffff, -1};
  if (q < Q)
    for (int i = 0; i < nband; ++i) {
      const MfPartial t = part[((int64_t)b * Q + q) * nband + i];
      r.cnt += t.cnt; r.sum += t.sum;
      r.x0 = min(r.x0, t.x0); r.x1 = max(r.x1, t.x1);
      r.y0 = min(r.y0, t.y0); r.y1 = max(r.y1, t.y1);
    }
  float s = q < Q ? score[b * Q + q] : 0.0f;
  bool keep = q < Q && r.cnt > 1;
  if (keep && use_mask_score) s *= (1e-3f * r.sum) / (1e-3f * (float)r.cnt + 1e-5f);
  if (thr > 0.0f) keep = keep && s > thr;
  __shared__ int wcount[2];
  const unsigned long long bal = __ballot(keep);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wcount[wave] = __popcll(bal);
  __syncthreads();
  const int pos = (wave ? wcount[0] : 0) + __popcll(bal & ((1ull << lane) - 1ull));
  if (keep) {
    const int o = b * Q + pos;
    det_query[o] = q;
    det_score[o] = s;
    det_label[o] = label[b * Q + q];
    det_box[o * 4 + 0] = r.x0; det_box[o * 4 + 1] = r.y0; det_box[o * 4 + 2] = r.x1; det_box[o * 4 + 3] = r.y1;
    mask_area[o] = r.cnt;
  }
  if (threadIdx.x == 0) det_count[b] = wcount[0] + wcount[1];
}

// Bit-packed binary masks of the kept detections: words[((b*Q + slot)*H + y)*ceil(W/32) + x/32], bit x&31 (bits >= W are 0).
__global__ __launch_bounds__(256) void mf_pack_masks_kernel(const float* __restrict__ lo, int h, int w, int H, int W, float sy, float sx,
                                                            float thr, const int32_t* __restrict__ det_count,
                                                            const int32_t* __restrict__ det_query, int Q, uint32_t* __restrict__ words) {
  const int slot = blockIdx.y, b = blockIdx.z;
  if (slot >= det_count[b]) return;
  const int q = det_query[b * Q + slot];
  const float* p = lo + ((int64_t)b * Q + q) * h * w;
  const int W32 = (W + 31) >> 5;
  uint32_t* wo = words + ((int64_t)b * Q + slot) * H * W32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ya = blockIdx.x * FX_MF_BAND, yb = min(H, ya + FX_MF_BAND);
  // each wave owns 8 consecutive output rows: consecutive rows share their two source rows, so the x-interpolated
  // source rows are carried in registers (about half a load per output pixel instead of four; same arithmetic)
  const int y0w = ya + wave * (FX_MF_BAND / 4), y1w = min(yb, y0w + FX_MF_BAND / 4);
  for (int xb = 0; xb < W; xb += 64) {
    const int x = xb + lane;
    const LerpAxis ax = lerp_axis(x < W ? x : W - 1, sx, w);
    int c0 = -1, c1 = -1;
    float top = 0.0f, bot = 0.0f;
    for (int y = y0w; y < y1w; ++y) {
      const LerpAxis ay = lerp_axis(y, sy, h);
      if (ay.i0 != c0) {
        top = ay.i0 == c1 ? bot : lerp_row(p + (int64_t)ay.i0 * w, ax);
        c0 = ay.i0;
      }
      if (ay.i1 != c1) {
        bot = ay.i1 == c0 ? top : lerp_row(p + (int64_t)ay.i1 * w, ax);
        c1 = ay.i1;
      }
      const bool on = x < W && lerp_col(top, bot, ay) >= thr;
      const unsigned long long bal = __ballot(on);
      if (lane == 0) wo[(int64_t)y * W32 + (xb >> 5)] = (uint32_t)bal;
      if (lane == 32 && xb + 32 < W) wo[(int64_t)y * W32 + (xb >> 5) + 1] = (uint32_t)(bal >> 32);
    }
  }
}

// Masks of the kept detections from the planes mf_mask_stats_kernel<true, true> wrote for every evaluated query: slot j <- query det_query[j].
__global__ __launch_bounds__(256) void mf_compact_masks_kernel(const uint32_t* __restrict__ allbits, int64_t plane_words, const int32_t* __restrict__ det_count,
                                                               const int32_t* __restrict__ det_query, int Q, uint32_t* __restrict__ words) {
  const int slot = blockIdx.y, b = blockIdx.z;
  if (slot >= det_count[b]) return;
  const int q = det_query[b * Q + slot];
  const uint4* src = reinterpret_cast<const uint4*>(allbits + ((int64_t)b * Q + q) * plane_words);
  uint4* dst = reinterpret_cast<uint4*>(words + ((int64_t)b * Q + slot) * plane_words);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < plane_words / 4; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}

// Workspace that also holds the bit planes of the fused form (x4 upsample with w % 8 == 0 and H * W/32 a multiple of 4): pass this many
// bytes to fx_mf_postprocess and the masks of the kept detections are compacted from them; with the smaller
// fx_mf_postprocess_workspace_bytes() buffer (or any other size ratio) they are interpolated a second time (mf_pack_masks_kernel).
extern "C" size_t fx_mf_postprocess_workspace_bytes_fused(int B, int Q, int h, int w, int H, int W) {
  if (B <= 0 || Q <= 0 || H <= 0 || W <= 0) return 0;
  const size_t base = ((size_t)B * Q * ((H + FX_MF_BAND - 1) / FX_MF_BAND) * sizeof(MfPartial) + 15) / 16 * 16;
  if (H != 4 * h || W != 4 * w || w % 8 != 0 || ((int64_t)H * (W / 32)) % 4 != 0) return base;
  return base + (size_t)B * Q * H * (W / 32) * 4;
}

extern "C" int fx_mf_postprocess_workspace_bytes(int B, int Q, int H) {
  if (B <= 0 || Q <= 0 || H <= 0) return 0;
  return (int)((size_t)B * Q * ((H + FX_MF_BAND - 1) / FX_MF_BAND) * sizeof(MfPartial));
}

extern "C" int fx_mf_postprocess(const float* mask_probs_lowres, int h, int w, int H, int W, const float* score, const int32_t* label, int B,
                                 int Q, float mask_threshold, float threshold, int use_mask_score, void* workspace, size_t workspace_bytes,
                                 int32_t* det_count, int32_t* det_query, float* det_score, int32_t* det_label, int32_t* det_box,
                                 int32_t* det_area, uint32_t* mask_words, fx_stream_t stream_) {
  FX_CHECK_ARG(mask_probs_lowres && score && label && workspace && det_count && det_query && det_score && det_label && det_box && det_area);
  FX_CHECK_ARG(B > 0 && Q > 0 && h > 0 && w > 0 && H > 0 && W > 0);
  if (Q > 128) return FX_ERR_UNSUPPORTED;
  FX_CHECK_ARG(workspace_bytes >= (size_t)fx_mf_postprocess_workspace_bytes(B, Q, H));
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const int nband = (H + FX_MF_BAND - 1) / FX_MF_BAND;
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  MfPartial* part = reinterpret_cast<MfPartial*>(workspace);
  const size_t base = ((size_t)fx_mf_postprocess_workspace_bytes(B, Q, H) + 15) / 16 * 16;
  const bool fused = mask_words && H == 4 * h && W == 4 * w && w % 8 == 0 && ((int64_t)H * (W / 32)) % 4 == 0 && ((uintptr_t)workspace % 16) == 0 &&
                     ((uintptr_t)mask_words % 16) == 0 && workspace_bytes >= base + (size_t)B * Q * H * (W / 32) * 4;
  uint32_t* allbits = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(workspace) + base);
  if (fused)
    hipLaunchKernelGGL((mf_mask_stats_kernel<true, true>), dim3(nband, B * Q), dim3(256), 0, stream, mask_probs_lowres, h, w, H, W, sy, sx,
                       mask_threshold, score, threshold, part, nband, allbits);
  else if (H == 4 * h && W == 4 * w)
    hipLaunchKernelGGL(mf_mask_stats_kernel<true>, dim3(nband, B * Q), dim3(256), 0, stream, mask_probs_lowres, h, w, H, W, sy, sx,
                       mask_threshold, score, threshold, part, nband);
  else
    hipLaunchKernelGGL(mf_mask_stats_kernel<false>, dim3(nband, B * Q), dim3(256), 0, stream, mask_probs_lowres, h, w, H, W, sy, sx,
                       mask_threshold, score, threshold, part, nband);
  hipLaunchKernelGGL(mf_select_kernel, dim3(B), dim3(128), 0, stream, part, nband, score, label, Q, threshold, use_mask_score, det_count,
                     det_query, det_score, det_label, det_box, det_area);
  if (fused) {
    const int64_t plane_words = (int64_t)H * (W / 32);
    int gx = (int)((plane_words / 4 + 255) / 256);
    if (gx > 16) gx = 16;
    hipLaunchKernelGGL(mf_compact_masks_kernel, dim3(gx, Q, B), dim3(256), 0, stream, allbits, plane_words, det_count, det_query, Q, mask_words);
  } else if (mask_words) {
    hipLaunchKernelGGL(mf_pack_masks_kernel, dim3(nband, Q, B), dim3(256), 0, stream, mask_probs_lowres, h, w, H, W, sy, sx, mask_threshold,
                       det_count, det_query, Q, mask_words);
  }
  return fx_launch_status();
}
