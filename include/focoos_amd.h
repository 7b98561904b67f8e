/* focoos_amd.h — C ABI of libfocoos_amd.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * focoos RT-DETR hot path.
 *
 * The reference (FocoosAI/focoos v0.25.0) is 100 % Python: it has NO native FFI for this path
 * (SURVEY.md §0.1, §8b "B6 proposed C ABI").  Each entry point below therefore names the reference
 * *Python* function(s) whose arithmetic it replaces (file:line relative to the reference root); the
 * Python adapter in focoos_amd/ binds these with ctypes and re-exposes the reference's
 * ModelManager / FAIDetr / DETRProcessor interface (INTEGRATION.md shows the binding).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (the Python host passes
 *    torch.Tensor.data_ptr()); the library never allocates, frees or retains caller memory;
 *  - activations are NHWC, bf16 (raw uint16 bits) unless a name says f32/u8/i32; "ld*" arguments are
 *    the distance in ELEMENTS between consecutive pixels/rows (>= the channel count), so a tensor may
 *    be a channel slice of a wider (concatenated) buffer;
 *  - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream); every call only
 *    enqueues work on that stream and is capturable into a hipGraph (fx_graph_*);
 *  - return value: FX_OK (0) or a negative FX_ERR_* code (fx_error_string()); nothing is swallowed.
 */
#ifndef FOCOOS_AMD_H
#define FOCOOS_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a descriptor struct's layout OR the semantics the host relies on change (version 2: fx_conv_desc grew mask / ldm /
 * reserved0 and the library applies the ReLU mask the previous bottleneck skips; version 3: fx_pw_chain_desc grew pool / ldp / img_h / img_w);
 * focoos_amd/_lib.py refuses a library of another version. */
#define FX_ABI_VERSION 8

enum { FX_OK = 0, FX_ERR_INVALID_ARGUMENT = -1, FX_ERR_LAUNCH = -2, FX_ERR_UNSUPPORTED = -3, FX_ERR_RUNTIME = -4 };
enum { FX_ACT_NONE = 0, FX_ACT_RELU = 1, FX_ACT_SILU = 2, FX_ACT_GELU = 3 };

typedef void* fx_stream_t; /* hipStream_t */

int fx_abi_version(void);
/* How the library was compiled: bit 0 = at least one translation unit was built WITH packed-fp32 VALU instructions (the
 * configuration in which two concurrent hardware queues corrupted results); the host refuses concurrent
 * batch parts / the weight-gradient side stream on such a build.  Bit 1 (ABI 7) = the library's 16-bit storage element is IEEE fp16
 * instead of bfloat16 (libfocoos_amd_fp16.so, built from the same sources with -DFX_FP16=1: every "bf16" in an entry point's name then
 * reads "the library's 16-bit element"; used for training under a loss scale, fx_adamw_step_scaled_f32). */
int fx_build_flags(void);
const char* fx_error_string(int code);
/* Device sanity: returns FX_OK iff device 0..n has a gfx950 agent; writes CU count / arch name. */
int fx_device_info(int device, int* cu_count, char* arch_name, int arch_name_len);

/* ------------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution / linear layer on the MFMA matrix cores (bf16 in, fp32 accumulate):
 *   y[b,ho,wo,n] = act( sum_{kh,kw,c} x[b, ho*stride-pad+kh, wo*stride-pad+kw, c] * w[n,kh,kw,c]
 *                        + bias[n] + residual[b,ho,wo,n] )
 * Replaces ConvNormLayer.forward = F.conv2d + eval BatchNorm (folded into w/bias by the host) +
 * activation (focoos/nn/layers/conv.py:78-98, norm.py:49-57), the residual add + ReLU of
 * BottleNeck.forward (focoos/nn/backbone/resnet.py:107-121), the avg-pool of the "d" shortcut
 * (resnet.py:89-100; `pool2`), RepVggBlock/CSPRepLayer adds (fai_detr/modelling.py:39-45,103-107)
 * and every nn.Linear on the path (H=W=1: x is [M,C] row-major).
 * w is packed [Npad][KH][KW][C] bf16 with Npad = N rounded up to 128 (zero rows); C % 32 == 0.
 */
typedef struct fx_conv_desc {
  const void* x;        /* bf16 [B,H,W,ldx] */
  const void* w;        /* bf16 [Npad, KH*KW*C] */
  const float* bias;    /* f32 [Npad] or NULL */
  const void* residual; /* bf16 [B,Ho,Wo,ldr] or NULL (added before the activation) */
  void* y;              /* bf16 (or f32 if out_f32) [B,Ho,Wo,ldy] */
  int32_t B, H, W, C, ldx;
  int32_t Ho, Wo, N, ldy, ldr;
  int32_t KH, KW, stride, pad;
  int32_t pool2;   /* 1: x is 2x2/2 average-pooled (ceil mode) on the fly; needs KH=KW=1, stride 1 */
  int32_t act;     /* FX_ACT_* */
  int32_t out_f32; /* 1: y is float32 */
  int32_t residual_after_act; /* 0: act(conv+bias+residual) (BottleNeck); 1: act(conv+bias)+residual (CSPRepLayer x_1+x_2);
                               * 2: act(conv+bias) * (residual > 0): `residual` is a saved ReLU output - the ReLU backward of the
                               *    producing layer fused into this (input-gradient) convolution of the training path */
  int64_t y_batch_stride;     /* elements between images of y; 0 = contiguous (Ho*Wo*ldy). Lets a level write its rows of
                                 the [B, sum(HW), C] decoder memory directly (modelling.py:1158-1165 flatten+concat for free) */
  const void* w_frag;         /* optional second copy of w in MFMA fragment order, bf16 [N/32][KH*KW*C/16][64][8] with
                                 Wp[nb][ks][l][i] = w[nb*32 + l%32][ks*16 + (l/32)*8 + i] (k = (kh*KW + kw)*C + c).  When present,
                                 3x3 / stride-1 / pad-1 layers with N in {64,128,256}, C % 64 == 0 run on the halo kernel
                                 (conv3x3_flat.hip: the pixels are fetched once for all nine taps).  NULL: implicit GEMM. */
  const void* mask;           /* optional bf16 [B,Ho,Wo,ldm]: the result is multiplied by (mask > 0) after residual / activation - the ReLU
                                 backward of the layer that CONSUMES y's gradient, fused into the producer of that gradient (training:
                                 a bottleneck's block-input gradient = branch2a input gradient + shortcut gradient, masked by the block
                                 input = the previous block's ReLU output; resnet.py:107-121).  Runs on the implicit-GEMM kernels only. */
  int32_t ldm, reserved0;
} fx_conv_desc;
int fx_conv2d_nhwc_bf16(const fx_conv_desc* d, fx_stream_t stream);
/* Label of the kernel fx_conv2d_nhwc_bf16 runs for this descriptor ("conv3x3_flat<256>", "pw_flat<K512>", "conv_igemm<128,128,64>",
 * "conv_igemm_dma<256,128>", ...): the same routing function decides the launch and the label, so a measurement grouped by it cannot
 * drift from what ran.  out: at least 48 bytes. */
int fx_conv2d_variant(const fx_conv_desc* d, char* out, int cap);
/* 1 iff fx_conv2d_nhwc_bf16 would run a 3x3/s1/p1 layer of C input / N output channels and image width W on the halo kernel
 * (given w_frag): N in {64,128,256}, C % 64 == 0 and the halo tile fits the 160 KiB LDS. */
int fx_conv3x3_flat_supported(int C, int N, int W);

/* ------------------------------------------------------------------------------------------------
 * Back-to-back pointwise convolutions around the residual add of the ResNet-vd bottleneck, one launch, the block output
 * consumed from LDS by the next layer (cross-layer fusion of the HBM-bound 1x1 layers):
 *   y1[m, :N1] = act1( [x1[m,:K1a] | x2[m,:K1b]] . W1^T + bias1 (+ residual[m,:N1]) )
 *   y2[m, :N2] = act2( y1[m,:N1] . W2^T + bias2 )                         (N2 = 0: first layer only)
 * Replaces BottleNeck.forward's  branch2c + short + add + ReLU  (focoos/nn/backbone/resnet.py:107-121; variant-d
 * shortcut conv :89-100 as the second K segment x2 with W1 = [W_2c | W_short], bias1 = b_2c + b_short, no residual)
 * followed by the NEXT block's branch2a (+ReLU) (resnet.py:108) - eval BatchNorm folded into W/bias by the host.
 * m indexes pixels (rows of NHWC tensors, row strides ld* in elements).  W1 / W2 are bf16 in MFMA FRAGMENT order:
 * Wp[n/32][k/16][lane][8] with Wp[nb][ks][l][i] = W[nb*32 + l%32][ks*16 + (l/32)*8 + i]  (k runs over K1a then K1b);
 * bias f32 [N1] / [N2].  N1 % 256 == 0; supported (K1a, K1b, N2) combinations: fx_pw_chain_supported(). */
typedef struct fx_pw_chain_desc {
  const void* x1;       /* bf16 [M, ldx1] */
  const void* x2;       /* bf16 [M, ldx2] or NULL (K1b = 0) */
  const void* residual; /* bf16 [M, ldr] or NULL */
  const void* w1;       /* bf16 fragment-packed [N1/32][(K1a+K1b)/16][64][8] */
  const float* bias1;   /* f32 [N1] */
  void* y1;             /* bf16 [M, ldy1]; may be NULL when `pool` and y2 are given: the block output is then not stored (round 5) */
  const void* w2;       /* bf16 fragment-packed [N2/32][N1/16][64][8] or NULL */
  const float* bias2;   /* f32 [N2] or NULL */
  void* y2;             /* bf16 [M, ldy2] or NULL */
  int32_t M, K1a, K1b, N1, N2;
  int32_t ldx1, ldx2, ldr, ldy1, ldy2;
  int32_t act1, act2;   /* FX_ACT_* */
  /* optional third output (ABI 3): pool = AvgPool2d(2, 2) of y1, bf16 [M/4, ldp] - the input of the NEXT stage's variant-d shortcut conv
   * (focoos/nn/backbone/resnet.py:46,95).  Needs the image size of the operands (even img_h, img_w; M = B * img_h * img_w): the kernel then
   * tiles the pixels by 2x2 quads.  NULL: no pooled output (img_h / img_w ignored).  Supported shapes: fx_pw_chain_pool_supported(). */
  void* pool;
  int32_t ldp, img_h, img_w, reserved0;
} fx_pw_chain_desc;
int fx_pw_chain_supported(int K1a, int K1b, int N1, int N2); /* 1 / 0 (not an error code) */
int fx_pw_chain_pool_supported(int K1a, int K1b, int N1, int N2); /* 1 / 0: the form with the pooled third output */
int fx_pw_chain_bf16(const fx_pw_chain_desc* d, fx_stream_t stream);

/* Stem: pixel normalisation (x-mean)/std (fai_detr/modelling.py:1349) fused with conv1_1 3x3/s2 +
 * folded BN + ReLU (focoos/nn/backbone/resnet.py:184-196,253).  x is HWC uint8 (in_f32=0) or HWC
 * float32 on the 0..255 scale (in_f32=1, output of fx_resize_bilinear_u8).  w: f32 [3][3][3][32]
 * (kh,kw,c,n) with BN folded (wave-uniform table read through the scalar cache), bias f32 [32];
 * y: bf16 [B,Ho,Wo,32]. */
int fx_stem_conv3x3s2(const void* x, int in_f32, const float* w, const float* bias, const float* mean,
                      const float* inv_std, void* y, int B, int H, int W, int Cout, fx_stream_t stream);

/* The first TWO stem layers in one launch from uint8 images (ABI 7; csrc/stem12.hip): y = relu(conv1_2(relu(conv1_1((x - mean) * inv_std)))) with
 * conv1_1 = fx_stem_conv3x3s2's layer (same w / bias / mean / inv_std tables) and conv1_2 = the 3x3 / s1 / p1 32 -> 32 layer given as its
 * fragment-order weights (fx_conv_desc.w_frag, [18][64][8]) and f32 bias - ResNet.conv1's conv1_1, conv1_2 (focoos/nn/backbone/resnet.py:184-196)
 * behind FAIDetr.forward's normalisation (fai_detr/modelling.py:1349).  x uint8 [B,H,W,3]; y bf16 [B,(H-1)/2+1,(W-1)/2+1,32] (pixel stride ldy).
 * The [B,H/2,W/2,32] conv1_1 activation is never written; bit-identical to fx_stem_conv3x3s2 followed by fx_conv2d_nhwc_bf16 (act = ReLU). */
int fx_stem_conv12_u8_bf16(const void* x_u8, const float* w1, const float* b1, const float* mean, const float* inv_std, const void* w2_frag,
                           const float* b2, void* y, int ldy, int B, int H, int W, fx_stream_t stream);

/* Processor.get_torch_batch resize (focoos/processor/base_processor.py:285-288):
 * F.interpolate(bilinear, align_corners=False) of one HWC uint8 image to [Ho,Wo,3] float32. */
int fx_resize_bilinear_u8(const uint8_t* x, int H, int W, float* y, int Ho, int Wo, fx_stream_t stream);

/* F.max_pool2d(k=3,s=2,p=1) (resnet.py:254) on NHWC bf16. */
int fx_maxpool3x3s2_nhwc_bf16(const void* x, int ldx, void* y, int ldy, int B, int H, int W, int C, fx_stream_t stream);
/* The last stem layer and the pool in ONE launch (ABI 7; csrc/stem_pool.hip): y = max_pool2d(relu(conv3x3_s1_p1(x) + bias), 3, 2, 1) for the
 * ResNet-vd stem's conv1_3 (32 -> 64 channels, BatchNorm folded; focoos/nn/backbone/resnet.py:184-196 `conv1`, :252-256 forward) - the
 * [B,H,W,64] activation between them is never written.  x bf16 NHWC [B,H,W,32] (pixel stride ldx), w_frag = the layer's weights in MFMA
 * fragment order (fx_conv_desc.w_frag: [2][18][64][8], k = (kh*3 + kw)*32 + c), bias f32 [64], y bf16 [B,(H-1)/2+1,(W-1)/2+1,64] (pixel
 * stride ldy).  Bit-identical to fx_conv2d_nhwc_bf16 (act = ReLU) followed by fx_maxpool3x3s2_nhwc_bf16. */
int fx_stem_conv_pool_supported(int C, int N, int H, int W);
int fx_stem_conv3x3_relu_maxpool_bf16(const void* x, int ldx, const void* w_frag, const float* bias, void* y, int ldy, int B, int H, int W,
                                      fx_stream_t stream);

/* nn.AvgPool2d(2, 2, 0, ceil_mode=True) of the "d"-variant shortcut (resnet.py:89-100) on NHWC bf16 -> [B,ceil(H/2),ceil(W/2),C].
 * (fx_conv2d_nhwc_bf16's pool2 flag fuses the same pooling into the conv's A-load; this standalone form pools once
 * when the conv has many N tiles.) */
int fx_avgpool2x2_nhwc_bf16(const void* x, int ldx, void* y, int ldy, int B, int H, int W, int C, fx_stream_t stream);

/* F.interpolate(mode="bilinear", align_corners=False) to (Ho,Wo) on NHWC bf16
 * (fai_detr/modelling.py:334,342); writing into a channel slice (ldy) makes torch.concat free. */
int fx_resize_bilinear_nhwc_bf16(const void* x, int ldx, void* y, int ldy, int B, int H, int W, int C, int Ho, int Wo,
                                 fx_stream_t stream);

/* out[r,:] = x[r,:] + y[(r % y_rows),:]  (with_pos_embed, modelling.py:918-919, transformer.py:580-581). */
int fx_add_rows_bf16(const void* x, int ldx, const void* y, int ldy_, int y_rows, void* out, int ldo, int rows, int cols,
                     fx_stream_t stream);

/* out = LayerNorm(x + residual) * gamma + beta, eps 1e-5, cols == 256 or 128 (nn.LayerNorm uses:
 * modelling.py:940,951,956,1089; transformer.py:592,600).  residual may be NULL. */
int fx_layernorm_bf16(const void* x, int ldx, const void* residual, int ldr, const float* gamma, const float* beta, void* out,
                      int ldo, int rows, int cols, fx_stream_t stream);

/* Multi-head softmax attention, head_dim 32 (nn.MultiheadAttention core, modelling.py:938,
 * transformer.py:589): q,k,v are [B, L, heads*32] bf16 views with row strides ldq/ldk/ldv (already
 * projected, bias added); out[b,l,h*32+d] bf16.  scale = 1/sqrt(32). */
int fx_mha_bf16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* out, int ldo, int B, int Lq, int Lk,
                int heads, fx_stream_t stream);

/* Multi-scale deformable attention sampling — the B4 seam: ms_deform_attn_core_pytorch
 * (focoos/nn/layers/deformable.py:10-35; bound at fai_detr/modelling.py:806, called :880).
 * value bf16 [B,S,M*D] (row stride ldv), D=32; spatial_shapes i32 [L][2]=(H,W), level_start i32 [L].
 * mode 0 ("core"): loc f32 [B,Q,M,L,P,2] in [0,1], attn f32 [B,Q,M,L,P] (already softmaxed).
 * mode 1 ("fused"): loc = raw sampling_offsets f32 [B,Q,M,L,P,2], attn = raw logits f32 [B,Q,M,L*P],
 *   ref = f32 [B,Q,4] (cx,cy,w,h); the kernel applies modelling.py:860-874 itself
 *   (softmax over L*P; loc = ref_xy + off/P * ref_wh * 0.5).
 * ld_loc / ld_attn: elements between consecutive (b,q) rows of loc / attn (>= M*L*P*2 / M*L*P), so both may be
 * column slices of one fused projection output.  out bf16 [B,Q,M*D] (row stride ldo). */
int fx_msda_bf16(const void* value, int ldv, const int32_t* spatial_shapes, const int32_t* level_start, int L, int P,
                 const float* loc, int ld_loc, const float* attn, int ld_attn, const float* ref, int mode, void* out, int ldo,
                 int B, int S, int Q, int M, fx_stream_t stream);

/* Row-wise max over the first `cols` columns of an f32 matrix (enc_outputs_class.max(-1), modelling.py:1210). */
int fx_rowmax_f32(const float* x, int ldx, float* out, int rows, int cols, fx_stream_t stream);

/* Encoder score head of the query selection (fai_detr/modelling.py:1202-1214) as one back-to-back GEMM launch:
 *   output_memory[m,:] = LayerNorm_256( W1 . (valid[m % S] ? memory[m,:] : 0) + b1 ) * gamma + beta     (bf16 out, row stride ldo)
 *   scores[m]          = max_c ( W2[c,:] . output_memory[m,:] + b2[c] )                                   (f32)
 * i.e. `valid_mask * memory` -> enc_output (Linear + LayerNorm) -> enc_score_classifier -> .max(-1); the [M, K] logits are
 * never written.  The Linear output feeds the LayerNorm from the fp32 accumulators (index-critical path).
 * memory bf16 [M, ldm] with 256 channels; valid u8 [S] or NULL; W1 / W2 fragment-packed like fx_pw_chain_desc
 * ([256/32][16][64][8] and [n2_pad/32][16][64][8]); n2_pad = classes rounded up to 128 (<= 384), padding rows of W2 zero
 * and padding entries of b2 = -3e38. */
int fx_enc_score_head_bf16(const void* memory, int ldm, const uint8_t* valid, int S, const void* w1, const float* b1,
                           const float* gamma, const float* beta, float eps, const void* w2, const float* b2, int n2_pad,
                           void* output_memory, int ldo, float* scores, int M, fx_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Row-local layer chains in one launch (csrc/row_chain.hip): a workgroup owns 32 rows of a [rows, C] activation and
 * interprets a host-written program of stages with the activations in LDS.  Used for the transformer decoder layer
 * (fai_detr/modelling.py:924-958 TransformerDecoderLayer.forward, :990-1003 query_pos_head / bbox refinement of
 * TransformerDecoder.forward): everything except the self-attention core (fx_mha_bf16) and the deformable sampling
 * (fx_msda_bf16) becomes 3 launches per layer instead of ~17.
 * LDS buffers are [32][K] bf16 at byte offsets src / dst / aux (-1 = none) of the workgroup's LDS.  Weights `w` of the GEMM
 * stages are bf16 in MFMA fragment order (see fx_pw_chain_desc).  Stage types:
 *   0 LOAD     dst <- g0[row, 0:K] (bf16 global, row stride ld)
 *   1 GEMM     act(src[32,K] . w^T + bias) -> dst (LDS, if >= 0; N in {256,512,1024}) and / or g0[row, 0:N] (row stride ld;
 *              f32 if flags&1 else bf16); N % 32 == 0, K % 128 == 0 (the weight stream moves in groups of eight 16-channel fragments)
 *   2 GEMM_LN  LayerNorm_256(src . w^T + bias + aux) * gamma + beta -> dst and optionally g0 (bf16); eps 1e-5; ld2 = LDS byte
 *              offset of a 512-byte reduction scratch
 *   3 ADD      dst <- src + aux  (K columns); if g0: also to g0[row, 0:K] (bf16, row stride ld)
 *   4 K4       dst[32,N] <- relu(ref[row,0:4] . w^T + bias), w f32 [N][4]; ref = f32 at LDS offset aux (16 B per row) if aux >= 0,
 *              else g0 (f32 [rows][4])
 *   5 BBOX     ref' = sigmoid(src[32,256] . w^T + bias + inverse_sigmoid(g0[row,0:4])), w f32 [4][256]; ref' -> g1 (f32 [rows][4])
 *              and, if aux >= 0, to LDS offset aux (for a following K4 stage)
 *   6 LN       dst and / or g0[row, 0:256] (bf16, row stride ld) <- LayerNorm_256(src) * gamma + beta (the pre-norm layers of the
 *              masked-attention decoders, fai_mf/modelling.py:453-549); src == dst allowed */
typedef struct fx_rc_stage {
  int32_t type, K, N, act;
  int32_t src, dst, aux;
  int32_t ld, ld2, flags;
  const void* w;
  const float* bias;
  const float* gamma;
  const float* beta;
  void* g0;
  void* g1;
} fx_rc_stage;
/* program_device: n_stages (<= 64) stages in DEVICE memory; rows: rows of the activation; lds_bytes: LDS the program needs. */
int fx_row_chain(const fx_rc_stage* program_device, int n_stages, int rows, int lds_bytes, fx_stream_t stream);

/* torch.topk(scores, k, dim=1) for f32 rows (modelling.py:1214; processor.py:147): for each of B rows
 * of length n writes the k largest values (descending; ties -> lower index first) and their indices. */
int fx_topk_rows_f32(const float* scores, int ld, int B, int n, int k, float* out_val, int32_t* out_idx, fx_stream_t stream);
/* Same result, two-level for long rows (n > 8192: chunk top-ks on B * ceil(n / 8192) workgroups, then the top-k of the candidates; exact incl.
 * tie order); workspace: fx_topk_rows_workspace_bytes() (0: the one-level form is used). */
size_t fx_topk_rows_workspace_bytes(int B, int n, int k);
int fx_topk_rows_ws_f32(const float* scores, int ld, int B, int n, int k, float* out_val, int32_t* out_idx, void* workspace, size_t workspace_bytes,
                        fx_stream_t stream);

/* out[b,i,:] = src[b, idx[b,i], :] (the three gathers of modelling.py:1216-1229), bf16 rows of `cols`. */
int fx_gather_rows_bf16(const void* src, int lds, int rows_per_batch, const int32_t* idx, int k, void* out, int ldo, int B, int cols,
                        fx_stream_t stream);

/* Overwrite rows listed in `rows_idx` (per batch) of a bf16 [B,R,cols] matrix with one constant row
 * (valid_mask * memory of modelling.py:1202 folded through enc_output: Linear(0)+LN = const). */
int fx_fill_rows_bf16(void* x, int ldx, int rows_per_batch, const int32_t* rows_idx, int n_idx, const void* row_bf16, int B, int cols,
                      fx_stream_t stream);

/* First layer of query_pos_head: relu(ref[r,0:4] @ W0^T + b0) -> bf16 [rows, N] (MLP(4,512,256), modelling.py:990,1084). */
int fx_linear_k4_relu(const float* ref, const float* w, const float* b, void* out, int ldo, int rows, int N, fx_stream_t stream);

/* Last layer of a bbox MLP (256->4) fused with the box update:
 *  mode 0 (decoder, modelling.py:1003): new_ref = sigmoid(h@W^T + b + inverse_sigmoid(ref)), eps 1e-5
 *  mode 1 (encoder, modelling.py:1207,1216,1221): unact = h@W^T + b + anchors[idx[r]]; new_ref = sigmoid(unact)
 * h bf16 [rows,K]; w f32 [4,K]; ref/new_ref f32 [rows,4]; anchors f32 [S,4]; idx i32 [rows] (row -> token). */
int fx_bbox_head(const void* h, int ldh, const float* w, const float* b, const float* ref, const float* anchors, const int32_t* idx,
                 int rows_per_batch, int mode, float* new_ref, float* unact_out, int rows, int K, fx_stream_t stream);

/* DETRHead.forward tail (modelling.py:392-397): probs = sigmoid(logits[:, :K]) (compact [rows,K] f32),
 * boxes_xyxy = cxcywh_to_xyxy(ref) (utils/box.py:14-17). */
int fx_detr_head_out(const float* logits, int ldl, const float* ref_cxcywh, float* probs, float* boxes_xyxy, int rows, int K,
                     fx_stream_t stream);

/* DETRProcessor.postprocess on device (fai_detr/processor.py:146-151,183-197) after fx_topk_rows_f32 over
 * the flattened [Q*K] probabilities: label = idx % K, query = idx / K, box = round(box[query]*(W,H,W,H)) as
 * int32, count[b] = #scores > threshold (a prefix, scores are sorted).  sizes i32 [B][2] = (H,W). */
int fx_detr_postprocess(const float* topk_val, const int32_t* topk_idx, const float* boxes_xyxy, const int32_t* sizes, int B, int Q,
                        int K, int top_k, float threshold, int32_t* labels, int32_t* queries, int32_t* boxes_i32, int32_t* count,
                        fx_stream_t stream);

/* ---- MaskFormer path (fai-mf-*) ------------------------------------------------------------
 * Same attention core with keys/values streamed in chunks (Lk unbounded) and the boolean attention mask of
 * MultiScaleMaskedTransformerDecoder (fai_mf/modelling.py:509-523; nn.MultiheadAttention attn_mask, shared by all heads):
 * mask_bits u32 [B*Lq][ld_mask_words], bit (key & 31) of word key/32 set = key NOT allowed; a query whose mask forbids every
 * key attends to all keys (modelling.py:509-512).  mask_bits NULL = plain attention (== fx_mha_bf16).
 * workspace (optional): fx_mha_workspace_bytes() bytes let long key sequences be sliced across workgroups
 * (partial softmax per slice + a merge kernel); without it one workgroup per (batch, head, 128 queries) walks all keys. */
int fx_mha_workspace_bytes(int B, int Lq, int Lk, int heads, int masked);
int fx_mha_masked_bf16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* out, int ldo, int B, int Lq, int Lk,
                       int heads, const uint32_t* mask_bits, int ld_mask_words, void* workspace, size_t workspace_bytes, fx_stream_t stream);

/* TransformerFPN top-down step (fai_mf/modelling.py:364): out = lateral + F.interpolate(top, size=(H,W), mode="nearest").
 * lateral/out bf16 NHWC [B,H,W,C], top bf16 NHWC [B,Hs,Ws,C]; C % 8 == 0.  Any (H, W) / (Hs, Ws): the source index follows ATen's
 * nearest_idx (float arithmetic), e.g. the ceil(H/2) levels of inputs that are not multiples of 32. */
int fx_upsample_nearest_add_nhwc_bf16(const void* lateral, int ldl, const void* top, int ldt, void* out, int ldo, int B, int H, int W,
                                      int Hs, int Ws, int C, fx_stream_t stream);
/* Adjoint of its up-sampling half (training): dtop bf16 [B,Hs,Ws,C] = sum of dy bf16 [B,H,W,C] over the pixels whose nearest source
 * (ATen's float rule: identity, exact x2, else floorf(dst * (float)in / out)) it is; gather form, deterministic.  d lateral = dy. */
int fx_upsample_nearest_bwd_nhwc_bf16(const void* dy, int lddy, void* dtop, int lddt, int B, int H, int W, int Hs, int Ws, int C,
                                      fx_stream_t stream);

/* PredictionHeads mask einsum (fai_mf/modelling.py:88): logit[b,q,p] = sum_c embed[b,q,c] * feat[b,p,c]; embed bf16
 * [B*Q, C] (row stride lde), feat bf16 [B*P, C] (row stride ldf), C == 256 (fai-mf) or 128 (bisenetformer), Q <= 128.
 *  mode 0: out f32 [B*Q][ldo] = logit;  mode 1: out = sigmoid(logit) (MaskFormerHead.forward :614);
 *  mode 2: bits u32 [B*Q][ld_words]: bit (p & 31) of word p/32 = (logit < 0), the attention mask of :104 when feat is the
 *          mask-feature map bilinearly resized to the attended level (the resize commutes with the einsum). */
int fx_query_pixel_logits_bf16(const void* embed, int lde, const void* feat, int ldf, int mode, float* out, int ldo, uint32_t* bits,
                               int ld_words, int B, int Q, int P, int C, fx_stream_t stream);

/* MaskFormerHead.forward class tail (:610-613) + the per-query max of MaskFormerProcessor.postprocess (processor.py:212):
 * probs f32 [rows][K] = softmax(logits[:, :K+1])[:, :K] (sigmoid(logits)[:, :K] when use_sigmoid), score = max_k, label = argmax_k. */
int fx_mf_class_head(const float* logits, int ldl, float* probs, float* score, int32_t* label, int rows, int K, int use_sigmoid,
                     fx_stream_t stream);

/* FAIMaskFormer.forward tail (:723): masks f32 [BQ][H][W] = bilinear(lowres f32 [BQ][h][w], align_corners=False). */
int fx_mf_upsample_probs_f32(const float* lowres, int h, int w, float* out, int H, int W, int BQ, fx_stream_t stream);
/* The same tensor in the library's 16-bit element (bf16): the engine's half-size `masks` option (the fp32 tensor is a 4.1 GB
 * write per bs = 16 800x800 step) - same taps and fp32 arithmetic, one rounding at the store. */
int fx_mf_upsample_probs_bf16(const float* lowres, int h, int w, void* out, int H, int W, int BQ, fx_stream_t stream);

/* Device side of MaskFormerProcessor.postprocess (fai_mf/processor.py:212-262 + masks_to_xyxy utils/vision.py:344-370),
 * fused with the x4 bilinear upsample so the [B,Q,H,W] tensor is never materialised: binary mask = upsampled prob >=
 * mask_threshold; keep masks with > 1 pixel; score = class score * (1e-3*sum_in_mask p)/(1e-3*area + 1e-5) when
 * use_mask_score; keep score > threshold (threshold <= 0 keeps all).  Survivors compacted in query order per image:
 * det_count i32 [B]; det_query/det_label/det_area i32 [B][Q]; det_score f32 [B][Q]; det_box i32 [B][Q][4] =
 * (x_min, y_min, x_max, y_max) inclusive; mask_words (optional) u32 [B][Q][H][ceil(W/32)], bit x&31 (bits >= W are 0) — slot j holds the
 * binary mask of detection j.  workspace: fx_mf_postprocess_workspace_bytes(B,Q,H) bytes.  Deterministic. */
int fx_mf_postprocess_workspace_bytes(int B, int Q, int H);
/* The larger workspace (16-byte aligned) that also holds one bit plane per evaluated query: for the x4 up-sample (H = 4h, W = 4w, w % 8 == 0)
 * the masks of the kept detections are then compacted from the planes the statistics pass writes anyway, instead of being
 * interpolated a second time.  Same results; fx_mf_postprocess picks the form from workspace_bytes. */
size_t fx_mf_postprocess_workspace_bytes_fused(int B, int Q, int h, int w, int H, int W);
int fx_mf_postprocess(const float* mask_probs_lowres, int h, int w, int H, int W, const float* score, const int32_t* label, int B, int Q,
                      float mask_threshold, float threshold, int use_mask_score, void* workspace, size_t workspace_bytes,
                      int32_t* det_count, int32_t* det_query, float* det_score, int32_t* det_label, int32_t* det_box, int32_t* det_area,
                      uint32_t* mask_words, fx_stream_t stream);

/* ---- mask-classification criterion (training path, forward values): SURVEY §8a row A16 ---------------------------------------
 * point_sample (focoos/nn/layers/point_rend.py:29-52): out f32 [R][P] = bilinear sample (F.grid_sample, zero padding,
 * align_corners=False) of map src[src_index ? src_index[r] : r] (f32 or u8 [.,H,W]) at coords[coord_index ? coord_index[r] : r][p] =
 * (x, y) in [0,1]^2 (coords f32 [.][P][2]). */
int fx_point_sample_f32(const void* src, int src_is_u8, int H, int W, const int32_t* src_index, const float* coords, const int32_t* coord_index,
                        float* out, int R, int P, fx_stream_t stream);

/* MaskHungarianMatcher cost blocks (fai_mf/loss.py:672-712 == bisenetformer/loss.py): cost[b][q][t] = w_mask * BCE-cost + w_class *
 * (-prob[q, label_t]) + w_dice * dice-cost over the image's P shared sample points; pred_pts f32 [B*Q][P] / tgt_pts f32 [sumT][P] =
 * fx_point_sample_f32 of the mask logits / target masks at those points; logits f32 [B,Q,ldl] (K+1 classes, softmax; sigmoid if
 * cls_sigmoid).  Layout of cost / offsets as fx_detr_match_cost_f32 (feeds fx_lsa_f32).  P <= 38400. */
int fx_mask_match_cost_f32(const float* logits, int ldl, const float* pred_pts, const float* tgt_pts, const int32_t* tgt_labels,
                           const int32_t* tgt_offsets, int B, int Q, int K, int P, int Tmax, float w_class, float w_mask, float w_dice,
                           int cls_sigmoid, float* cost, fx_stream_t stream);
/* The same cost blocks computed as tiled fp32 GEMMs over the points (16 queries x 16 targets x a slice of the points per workgroup; slices reduced
 * in a fixed order: deterministic) when given fx_mask_match_cost_workspace_bytes() bytes of 16-byte aligned scratch and Tmax <= 64; otherwise
 * identical to fx_mask_match_cost_f32.  Sums are accumulated in a different order than there (agreement ~1e-6 relative). */
size_t fx_mask_match_cost_workspace_bytes(int B, int Q, int Tmax);
int fx_mask_match_cost_ws_f32(const float* logits, int ldl, const float* pred_pts, const float* tgt_pts, const int32_t* tgt_labels,
                              const int32_t* tgt_offsets, int B, int Q, int K, int P, int Tmax, float w_class, float w_mask, float w_dice,
                              int cls_sigmoid, float* cost, void* workspace, size_t workspace_bytes, fx_stream_t stream);

/* SetCriterion.loss_labels (ce_loss branch, :411-431) + loss_masks (:463-523) of one prediction set: out3 = {w_ce * loss_ce,
 * w_mask * loss_mask, w_dice * loss_dice}.  pred_masks f32 [B,Q,h,w] (logits), tgt_masks f32 or u8 [sumT,H,W], matches as written
 * by fx_lsa_f32.  Point selection of get_uncertain_point_coords_with_randomness (point_rend.py:73-128) with the uniform draws as
 * inputs: rand_over f32 [sumT][n_over][2] (the oversampled candidates; the num_points - n_extra with the smallest |logit| are kept),
 * rand_extra f32 [sumT][n_extra][2].  num_masks = the clamped, world-averaged target count of :557-561.  workspace:
 * fx_mask_set_loss_workspace_bytes(), 8-byte aligned.  Deterministic (fixed-order float64 reductions). */
size_t fx_mask_set_loss_workspace_bytes(int B, int Q, int sum_T, int n_over);
int fx_mask_set_loss_f32(const float* logits, int ldl, const float* pred_masks, int h, int w, const void* tgt_masks, int tgt_is_u8, int H, int W,
                         const int32_t* tgt_labels, const int32_t* tgt_offsets, int sum_T, const int32_t* pred_idx, const int32_t* tgt_idx,
                         const float* rand_over, int n_over, const float* rand_extra, int n_extra, int num_points, int B, int Q, int K,
                         float eos_coef, float num_masks, float w_ce, float w_mask, float w_dice, void* workspace, size_t workspace_bytes,
                         float* out3, fx_stream_t stream);

/* ---- BiSeNetFormer path (focoos/models/bisenetformer/modelling.py, focoos/nn/backbone/stdc.py) -------
 * Depthwise 3x3 stride-2 pad-1 convolution on NHWC bf16: y[b,ho,wo,c] = bias[c] + sum_k w[k][c] * x[...] (w f32 [9][C] with the
 * eval BatchNorm scale folded in, bias f32 [C] = BN shift or NULL).  STDC CatBottleneck `avd_layer` (stdc.py:114-127); with
 * w = 1/9 and no bias it is the block's AvgPool2d(3, 2, 1) skip (:128, count_include_pad=True).  C % 8 == 0. */
int fx_dwconv3x3s2_nhwc_bf16(const void* x, int ldx, const float* w, const float* bias, void* y, int ldy, int B, int H, int W, int C,
                             fx_stream_t stream);
/* Same with an fp32 output y f32 [B,Ho,Wo,ldy] (the pre-BatchNorm tensor of the batch-statistics training path, see fx_bn_stats_bf16). */
int fx_dwconv3x3s2_nhwc_f32out(const void* x, int ldx, const float* w, const float* bias, float* y, int ldy, int B, int H, int W, int C,
                               fx_stream_t stream);

/* mean over the P pixels of each image: out f32 [B][ldo] (feat.mean(dim=(2,3)), modelling.py:161,187; adaptive_avg_pool2d :230). */
int fx_global_mean_nhwc_bf16(const void* x, int ldx, float* out, int ldo, int B, int P, int C, fx_stream_t stream);

/* 1x1 convolutions on pooled [B,C,1,1] tensors: out[b][n] = act(bias[n] + sum_c W[n][c] * in[b][c]), all f32; act = FX_ACT_NONE,
 * FX_ACT_RELU or 4 (sigmoid).  conv_avg (:188), conv_atten + bn_atten + sigmoid (:162-164), FFM conv1/relu/conv2/sigmoid (:231-234). */
int fx_pooled_linear_f32(const float* in, int ldi, const float* W, const float* bias, int act, float* out, int ldo, int B, int C, int N,
                         fx_stream_t stream);

/* y[b,p,c] = x[b,p,c] * gate[b][c] (+ x if self_add) (+ add_vec[b][c]) (+ add_map[b,p,c]); x / add_map / y bf16 NHWC, gate /
 * add_vec f32 [B][ld].  ARM output (:165) fused with ContextPath's sums (:191,196); FFM `feat * atten + feat` (:235-236). */
int fx_channel_gate_nhwc_bf16(const void* x, int ldx, const float* gate, int ldg, int self_add, const float* add_vec, int ldv,
                              const void* add_map, int ldm, void* y, int ldy, int B, int P, int C, fx_stream_t stream);

/* Device side of BisenetFormerProcessor.postprocess with predict_all_pixels (bisenetformer/processor.py:212-262 + masks_to_xyxy):
 * winner[b,y,x] = argmax_q score[b,q] * up(mask_probs_lowres[b,q])[y,x] (first maximum; `up` = the bilinear x(H/h) upsample of
 * BisenetFormer.forward :607, fused); query q's binary mask = (winner == q); keep masks with > 1 pixel; score = class score
 * [* (1e-3 * sum winning prob) / (1e-3 * area + 1e-5) if use_mask_score]; keep score > threshold (<= 0 keeps all).  Outputs as
 * fx_mf_postprocess (survivors compacted in query order); winner_out (optional) u8 [B][H][W], 16-byte aligned: the per-pixel
 * query index (the semantic map is label[winner]).  Q <= 128; workspace: fx_seg_postprocess_workspace_bytes(...), 16-byte aligned. */
size_t fx_seg_postprocess_workspace_bytes(int B, int Q, int h, int w, int H, int W);
int fx_seg_postprocess(const float* mask_probs_lowres, int h, int w, int H, int W, const float* score, const int32_t* label, int B, int Q,
                       float threshold, int use_mask_score, void* workspace, size_t workspace_bytes, int32_t* det_count, int32_t* det_query,
                       float* det_score, int32_t* det_label, int32_t* det_box, int32_t* det_area, uint32_t* mask_words, uint8_t* winner_out,
                       fx_stream_t stream);

/* ---- set criterion (training path, forward only this round): SURVEY §8a rows A14/A15 ------------------------------
 * BoxHungarianMatcher cost (fai_detr/modelling.py:714-746, focal branch; box math focoos/utils/box.py:14-64), computed
 * per image: cost[b][q][t] for t < T_b (T_b = tgt_offsets[b+1]-tgt_offsets[b]; row stride Tmax; columns >= T_b zero).
 * logits f32 [B,Q,ldl] (raw), boxes f32 [B,Q,4] cxcywh, tgt_labels i32 [sumT], tgt_boxes f32 [sumT,4] cxcywh. */
int fx_detr_match_cost_f32(const float* logits, int ldl, const float* boxes, const int32_t* tgt_labels, const float* tgt_boxes,
                           const int32_t* tgt_offsets, int B, int Q, int K, int Tmax, float w_class, float w_bbox, float w_giou, float alpha,
                           float gamma, float* cost, fx_stream_t stream);

/* scipy.optimize.linear_sum_assignment on every image's [Q, T_b] block (modelling.py:749-750; SciPy is a third-party
 * dependency of the reference, pinned scipy~=1.14.1): same shortest-augmenting-path algorithm, float64 arithmetic and
 * tie rule, so the int indices are identical to SciPy's.  Writes, at tgt_offsets[b].., the matched query indices in
 * ascending order (pred_idx) and their targets (tgt_idx).  Needs T_b <= Q <= 1024. */
int fx_lsa_f32(const float* cost, int B, int Q, int Tmax, const int32_t* tgt_offsets, int32_t* pred_idx, int32_t* tgt_idx, fx_stream_t stream);
/* The same with a device status word: bit 0 is OR-ed in when an image's assignment is infeasible (+inf costs: SciPy's "cost matrix is
 * infeasible"), bit 1 when its cost block holds NaN or -inf (SciPy's "matrix contains invalid numeric entries") - the two cases in which
 * linear_sum_assignment raises inside the reference matcher (fai_detr/modelling.py:749-750); the image's output slots stay untouched.  The word is sticky: the host reads and
 * clears it where it synchronises anyway (focoos_amd/criterion.py: raise_if_infeasible). */
int fx_lsa_status_f32(const float* cost, int B, int Q, int Tmax, const int32_t* tgt_offsets, int32_t* pred_idx, int32_t* tgt_idx, int32_t* status,
                      fx_stream_t stream);

/* SetCriterion.loss_labels_vfl + loss_boxes of one prediction set (modelling.py:464-497, 513-530, weights :576-579):
 * out3 = {w_vfl * loss_vfl, w_bbox * loss_bbox, w_giou * loss_giou}.  workspace: fx_detr_set_loss_workspace_bytes() bytes,
 * 8-byte aligned.  Deterministic (fixed-order float64 reduction). */
int fx_detr_set_loss_workspace_bytes(int B, int Q, int sum_T);
int fx_detr_set_loss_f32(const float* logits, int ldl, const float* boxes, const int32_t* tgt_labels, const float* tgt_boxes,
                         const int32_t* tgt_offsets, const int32_t* pred_idx, const int32_t* tgt_idx, int B, int Q, int K, int sum_T,
                         float num_boxes, float focal_alpha, float focal_gamma, float w_vfl, float w_bbox, float w_giou, void* workspace,
                         float* out3, fx_stream_t stream);

/* SetCriterion.loss_boxes (fai_detr/modelling.py:513-530) of one prediction set together with its gradient and the per-query targets of
 * loss_labels_vfl (:464-480), one launch (training path): loss2 = {scale_bbox * sum_pairs L1, scale_giou * sum_pairs (1 - GIoU)} with
 * scale_* = weight / num_boxes; q_cls i32 [B*Q] = matched target label or K, q_score f32 [B*Q] = IoU of the matched pair or 0;
 * pair_grad f32 [sum_T][8] = d loss2[0] / d box (4) | d loss2[1] / d box (4) of every matched pair (cxcywh; autograd's conventions
 * for max / min ties, clamp and sgn).  boxes f32 [B,Q,4] contiguous.  Deterministic.
 * fx_detr_box_loss_bwd_f32 writes all of dboxes f32 [B,Q,4] = g_bbox * d loss2[0] + g_giou * d loss2[1] (g_*: device scalars, NULL = 0). */
int fx_detr_box_loss_f32(const float* boxes, const int32_t* tgt_labels, const float* tgt_boxes, const int32_t* tgt_offsets,
                         const int32_t* pred_idx, const int32_t* tgt_idx, int B, int Q, int K, int sum_T, float scale_bbox, float scale_giou,
                         int32_t* q_cls, float* q_score, float* loss2, float* pair_grad, fx_stream_t stream);
int fx_detr_box_loss_bwd_f32(const float* pair_grad, const int32_t* tgt_offsets, const int32_t* pred_idx, int B, int Q, int sum_T,
                             const float* g_bbox, const float* g_giou, float* dboxes, fx_stream_t stream);

/* ms_deform_attn_core in fp32 with its backward - the autograd half of seam B4 (focoos/nn/layers/deformable.py:10-35;
 * fai_detr/modelling.py:806,880).  value f32 [B,S,M*32], loc f32 [B,Q,M,L,P,2], attn f32 [B,Q,M,L,P] (contiguous),
 * out / grad_out f32 [B,Q,M*32].  bwd zeroes grad_value [B,S,M*32] itself, then accumulates with fp32 atomics;
 * grad_loc / grad_attn have the shapes of loc / attn. */
int fx_msda_f32_fwd(const float* value, const int32_t* spatial_shapes, const int32_t* level_start, int L, int P, const float* loc,
                    const float* attn, float* out, int B, int S, int Q, int M, fx_stream_t stream);
int fx_msda_f32_bwd(const float* value, const int32_t* spatial_shapes, const int32_t* level_start, int L, int P, const float* loc,
                    const float* attn, const float* grad_out, float* grad_value, float* grad_loc, float* grad_attn, int B, int S, int Q, int M,
                    fx_stream_t stream);
/* The same pair with a typed, strided value operand (the training graph: the six decoder layers share one value projection whose
 * bf16 output [B,S,6*M*32] each layer reads as a column slice - no fp32 copy - and whose fp32 gradient they accumulate into column
 * slices of one buffer): value fp32 or bf16 (value_bf16), row stride ldv elements; grad_value fp32, row stride ldg, zeroed by the call
 * only when zero_grad_value != 0 (then ldg must be M*32). */
int fx_msda_train_fwd(const void* value, int value_bf16, int ldv, const int32_t* spatial_shapes, const int32_t* level_start, int L, int P,
                      const float* loc, const float* attn, float* out, int B, int S, int Q, int M, fx_stream_t stream);
int fx_msda_train_bwd(const void* value, int value_bf16, int ldv, const int32_t* spatial_shapes, const int32_t* level_start, int L, int P,
                      const float* loc, const float* attn, const float* grad_out, float* grad_value, int ldg, int zero_grad_value,
                      float* grad_loc, float* grad_attn, int B, int S, int Q, int M, fx_stream_t stream);
/* fx_msda_train_bwd without floating-point atomics: grad_loc / grad_attn by one kernel in the forward's lane layout, the value gradient
 * by binning - a workgroup per (batch, head, slab of rows of one level) files the taps of all Q*P sampling points under their pixels in
 * LDS and sums them pixel by pixel - stored once, as bf16 (the dtype the value projection's backward reads): grad_value_bf16 [B,S,ldg],
 * this layer's M*32 columns fully overwritten - no zero-fill, no accumulation across calls.  grad_out fp32 or bf16 (grad_out_bf16).
 * shapes_host = the L (H, W) pairs in host memory; spatial_shapes / level_start as above (device).  fx_msda_bwd_slab_supported: 1 iff
 * the shapes are covered (every W <= 3200, the Q*P*4 taps + Q grad_out rows fit the 160 KiB LDS); else the call returns
 * FX_ERR_UNSUPPORTED and the caller keeps fx_msda_train_bwd. */
int fx_msda_bwd_slab_supported(const int32_t* shapes_host, int L, int P, int Q, int M, int grad_out_bf16);
int fx_msda_train_bwd_slab(const void* value, int value_bf16, int ldv, const int32_t* spatial_shapes, const int32_t* level_start,
                           const int32_t* shapes_host, int L, int P, const float* loc, const float* attn, const void* grad_out,
                           int grad_out_bf16, void* grad_value_bf16, int ldg, float* grad_loc, float* grad_attn, int B, int S, int Q, int M,
                           fx_stream_t stream);

/* Sampling locations and attention weights of one deformable layer from its raw bf16 projections, and the gradients back to them
 * (MSDeformableAttention.forward, fai_detr/modelling.py:866-879, 4-d reference points, detached): aw = softmax over the L*P logits of a
 * (query, head); loc = ref_xy + off / P * ref_wh * 0.5.  off [BQ, >= M*L*P*2], logit [BQ, >= M*L*P] bf16 rows; ref f32 [BQ,4];
 * loc f32 [BQ,M,L,P,2], aw f32 [BQ,M,L,P] - the operands of fx_msda_train_fwd / _bwd.  One launch each way instead of ~7 + ~8. */
int fx_msda_prep_bf16(const void* off, int ld_off, const void* logit, int ld_logit, const float* ref, float* loc, float* aw, int BQ, int M, int L,
                      int P, fx_stream_t stream);
int fx_msda_prep_bwd_bf16(const float* grad_loc, const float* grad_attn, const float* aw, const float* ref, void* grad_off, int ld_off,
                          void* grad_logit, int ld_logit, int BQ, int M, int L, int P, fx_stream_t stream);

/* Fused multi-tensor AdamW + global-norm gradient clipping over one flat fp32 buffer (replaces the
 * ~500 single-tensor param groups of focoos/trainer/solver/build.py:39-138 and the clip of :29-36 / trainer.py:758-760).
 * The buffer is cut into chunks (chunk_start i64, chunk_len i32 <= 65536) each carrying its tensor's lr / weight_decay;
 * arithmetic of torch.optim.AdamW (decoupled decay, bias correction with `step` >= 1), clip coefficient
 * min(1, max_grad_norm / (||g|| + 1e-6)) computed on the device (max_grad_norm <= 0: no clipping).
 * workspace: fx_adamw_workspace_bytes() bytes, 8-byte aligned; total_norm_out (may be NULL) receives ||g||.
 * Gradients holding an inf / NaN: the update is skipped (params, exp_avg, exp_avg_sq untouched) and total_norm_out receives +inf -
 * what GradScaler does for the reference's default amp training (trainer/trainer.py:645,735-773). */
int fx_adamw_workspace_bytes(void);
int fx_adamw_step_f32(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t numel, const int64_t* chunk_start,
                      const int32_t* chunk_len, const float* chunk_lr, const float* chunk_wd, int nchunks, int step, float beta1, float beta2,
                      float eps, float max_grad_norm, void* workspace, float* total_norm_out, fx_stream_t stream);

/* The same step under a dynamic loss scale (ABI 7; the fp16 build's training step - torch.amp.GradScaler around optimizer.step(),
 * focoos/trainer/trainer.py:645 `GradScaler(init_scale=2**10)`, :735-773 scale(loss).backward() / unscale_ + clip / step / update): `grads` hold
 * scale * g; the update uses g = grads / scale (clip norm on the unscaled gradients); a step whose gradients contain inf / NaN is SKIPPED
 * (parameters, moments, step count untouched) and the scale multiplied by backoff_factor; after growth_interval consecutive good steps the
 * scale is multiplied by growth_factor.  All of it on the device (state lives in device memory, no host synchronisation); Adam's bias
 * correction counts the steps actually taken (state->good_steps). */
typedef struct fx_loss_scale_state {
  float scale;            /* current loss scale (initialise: 1024) */
  int32_t growth_tracker; /* consecutive good steps since the last change of the scale */
  int32_t good_steps;     /* optimizer steps taken */
  int32_t skipped_steps;  /* steps skipped on inf / NaN gradients */
} fx_loss_scale_state;
int fx_adamw_step_scaled_f32(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t numel, const int64_t* chunk_start,
                             const int32_t* chunk_len, const float* chunk_lr, const float* chunk_wd, int nchunks, float beta1, float beta2, float eps,
                             float max_grad_norm, void* workspace, float* total_norm_out, fx_loss_scale_state* state, float growth_factor,
                             float backoff_factor, int growth_interval, fx_stream_t stream);

/* ---- training path, convolution backward (autograd of F.conv2d as used by ConvNormLayer,
 * focoos/nn/layers/conv.py:78-98) -----------------------------------------------------------------------------------
 * Weight gradient: dw[n][kh][kw][c] (fp32, layout of the packed forward weights without row padding) +=
 * sum_m dz[m][n] * x[pixel(m,kh,kw)][c].  x bf16 NHWC [B,H,W,C] (pixel stride ldx), dz bf16 [B,Ho,Wo,N] (pixel stride
 * lddz) = gradient w.r.t. the convolution output (activation backward already applied).  ACCUMULATES with float atomics:
 * the caller zeroes dw.  C % 8 == 0, N % 8 == 0. */
int fx_conv2d_wgrad_nhwc_bf16(const void* x, int ldx, const void* dz, int lddz, float* dw, int B, int H, int W, int C, int Ho, int Wo, int N,
                              int KH, int KW, int stride, int pad, fx_stream_t stream);
/* Same, and dbias[n] += sum_m dz[m][n] (bias gradient of a Linear / biased conv) from the tiles the kernel stages anyway. */
int fx_conv2d_wgrad_bias_nhwc_bf16(const void* x, int ldx, const void* dz, int lddz, float* dw, float* dbias, int B, int H, int W, int C, int Ho,
                                   int Wo, int N, int KH, int KW, int stride, int pad, fx_stream_t stream);

/* Master weights fp32 [N][C][KH][KW] (reference / checkpoint layout) -> bf16 images for the MFMA kernels, optionally
 * multiplied by a per-out-channel scale (frozen BatchNorm folded): w_fwd [Npad][KH][KW][C] (fx_conv2d_nhwc_bf16 layout) and
 * w_dgrad [Cpad][KH][KW][N] = flipped + transposed filter: for a stride-1 "same" conv, dX = fx_conv2d_nhwc_bf16(dZ, w_dgrad).
 * Either output may be NULL; padding rows are not touched (zero them once).  w_fwd_frag / w_dgrad_frag (optional; need the matching
 * image, N % 32 == 0 resp. C % 32 == 0): second copies in MFMA fragment order - fx_conv_desc.w_frag of the forward / input-gradient
 * convolution, which routes eligible layers to the halo / pointwise kernels of conv3x3_flat.hip. */
int fx_pack_conv_weights_f32(const float* w, const float* scale, void* w_fwd, void* w_dgrad, void* w_fwd_frag, void* w_dgrad_frag, int N, int C,
                             int KH, int KW, fx_stream_t stream);

/* [rows][K] bf16 weight image (row stride K) -> MFMA fragment order (fx_conv_desc.w_frag); rows % 32 == 0, K % 16 == 0. */
int fx_pack_frag_bf16(const void* w_rows, void* frag, int rows, int K, fx_stream_t stream);

/* nn.Linear master weights fp32 [N][K] (+ bias [N]) -> bf16 images of the forward GEMM (w_fwd, row stride Kp) and of the
 * input-gradient GEMM (w_t = transpose, row stride Np) and the bias into its padded fp32 vector; Np >= N, Kp >= K: the padding is
 * never written (zero it once). */
int fx_pack_linear_weights_f32(const float* w, const float* bias, void* w_fwd, void* w_t, float* bias_out, int N, int K, int Np, int Kp,
                               fx_stream_t stream);

/* Multi-tensor form of the two functions above: every weight image of a model rebuilt in one launch.  `entries_dev` is a device
 * array; entry e converts master w [N][C][KH][KW] (a Linear: KH = KW = 1, C = K) times scale[n] (NULL: 1) into w_fwd (row n at
 * (n_offset + n) * ld_fwd, column (kh*KW + kw)*C + c), w_dgrad (row c at c * ld_dgrad, column ((KH-1-kh)*KW + (KW-1-kw))*n_total + n_offset + n), their
 * fragment-order copies (NULL: none; rows % 32 == 0, columns % 16 == 0) and copies bias [N] to bias_out (NULL: none).  Workgroups
 * first_block .. first_block + fx_pack_entry_blocks(N, C, KH, KW) - 1 belong to entry e (ascending, gap-free); total_blocks = their sum.
 * The image base pointers, ld_fwd and ld_dgrad must be 16-byte / 8-element aligned for the vector stores to be taken (else element stores). */
typedef struct {
  const float* w;
  const float* scale;
  const float* bias;
  void* w_fwd;
  void* w_dgrad;
  void* w_fwd_frag;
  void* w_dgrad_frag;
  float* bias_out;
  int32_t N, C, KH, KW;
  int32_t ld_fwd, ld_dgrad;
  int32_t first_block;
  int32_t n_offset, n_total; /* this entry's N output channels are channels n_offset .. n_offset+N-1 of images with n_total output channels
                                (several masters sharing one image, e.g. the six value projections of the decoder); plain layer: 0, N */
  int32_t reserved;
} fx_pack_entry;
int fx_pack_entry_blocks(int N, int C, int KH, int KW);   /* workgroups of one entry (tiles of 8 output channels x <= 2304 (channel, tap) elements); -1: unsupported filter size */
int fx_pack_weights_many_f32(const fx_pack_entry* entries_dev, int n_entries, int total_blocks, fx_stream_t stream);

/* dw_master[n][c][kh][kw] (+)= scale[n] * dw_eff[n][kh][kw][c]; dw_eff rows have C_eff >= C channels (stem: 3 of 8). */
int fx_unpack_conv_wgrad_f32(const float* dw_eff, const float* scale, float* dw_master, int N, int C, int KH, int KW, int C_eff, int accumulate,
                             fx_stream_t stream);

/* Weight (+ bias) gradient of a Linear layer whose operands are zero-padded views of a narrower layer (N = 4 / 365, K = 4 of the detection
 * heads): x bf16 [R][ldx] (Kp columns, the padding zero), dz bf16 [R][lddz] (Np columns); accumulates (fp32 atomics) only the
 * n_store x k_store corner of dW, row stride ld_dw, and dbias[0, n_store) - i.e. straight into the master gradient of the unpadded layer. */
int fx_linear_wgrad_bias_bf16(const void* x, int ldx, const void* dz, int lddz, float* dw, int ld_dw, float* dbias, int R, int Kp, int Np, int k_store,
                              int n_store, fx_stream_t stream);
/* The same weight gradient without atomics: pixel range s (0 <= s < splits = fx_conv2d_wgrad_splits(...)) STORES its partial sums at
 * partials + s * split_stride (f32 [N][KH][KW][C] each; split_stride >= N*KH*KW*C; nothing needs zeroing), and
 * fx_unpack_conv_wgrad_sum_f32 adds the slabs while it re-lays the gradient out for the master weight.  The L2 atomic units sustain
 * ~0.6 TB/s of fp32 adds, plain stores the HBM rate, so this is the path for the 3x3 / wide layers whose dW is megabytes. */
int fx_conv2d_wgrad_splits(int B, int Ho, int Wo, int C, int N, int KH, int KW);
/* Label of the kernel fx_conv2d_wgrad_partial_nhwc_bf16 runs for this layer shape ("conv_wgrad_dma<3x3>", "conv_wgrad_dma<pw>",
 * "conv_wgrad<pw>", "conv_wgrad<im2col>"): the launch's own routing predicate, for kernel censuses (ABI 7).  out: at least 32 bytes. */
int fx_conv2d_wgrad_variant(int B, int Ho, int Wo, int C, int N, int KH, int KW, int stride, int pad, char* out, int cap);
int fx_conv2d_wgrad_partial_nhwc_bf16(const void* x, int ldx, const void* dz, int lddz, float* partials, int64_t split_stride, int splits, int B,
                                      int H, int W, int C, int Ho, int Wo, int N, int KH, int KW, int stride, int pad, fx_stream_t stream);
int fx_unpack_conv_wgrad_sum_f32(const float* partials, int64_t split_stride, int splits, const float* scale, float* dw_master, int N, int C, int KH,
                                 int KW, int C_eff, int accumulate, fx_stream_t stream);

/* Backward of the fused epilogue y = relu(conv + bias [+ residual]): dz = (dy [+ dy2]) * (y > 0)  (use_relu = 0: plain sum). */
int fx_relu_bwd_bf16(const void* dy, int lddy, const void* dy2, int lddy2, const void* y, int ldy, void* dz, int lddz, int64_t rows, int cols,
                     int use_relu, fx_stream_t stream);

/* u[b,2ho,2wo,:] = dz[b,ho,wo,:], zeros elsewhere; u is [B,H,W,C]: input gradient of a stride-2 3x3 pad-1 conv =
 * fx_conv2d_nhwc_bf16(u, w_dgrad) at stride 1. */
int fx_zero_insert2_nhwc_bf16(const void* dz, int lddz, void* u, int ldu, int B, int Ho, int Wo, int H, int W, int C, fx_stream_t stream);

/* Backward of fx_avgpool2x2_nhwc_bf16 (AvgPool2d(2,2,0,ceil_mode=True)) and fx_maxpool3x3s2_nhwc_bf16 (first maximum in
 * window scan order receives the gradient, like PyTorch); dx is [B,H,W,C].  The max-pool backward needs a workspace of
 * B * Ho * Wo * C bytes (8-byte aligned): the arg-max tap of every output element. */
int fx_avgpool2x2_bwd_nhwc_bf16(const void* dp, int lddp, void* dx, int lddx, int B, int H, int W, int C, fx_stream_t stream);
int fx_maxpool3x3s2_bwd_nhwc_bf16(const void* x, int ldx, const void* dy, int lddy, void* dx, int lddx, int B, int H, int W, int C,
                                  void* workspace, fx_stream_t stream);

/* (img - mean) * inv_std as bf16 NHWC with the 3 channels padded to 8: the stem conv's input for its weight gradient. */
int fx_normalize_pad8(const void* img, int is_f32, const float* mean, const float* inv_std, void* out, int64_t pixels, fx_stream_t stream);

/* The stem convolution without its ReLU (pre-BatchNorm tensor of the batch-statistics training path). */
int fx_stem_conv3x3s2_linear(const void* x, int in_f32, const float* w, const float* bias, const float* mean, const float* inv_std, void* y,
                             int B, int H, int W, int Cout, fx_stream_t stream);

/* ---- train-mode BatchNorm2d (batch statistics; nn.BatchNorm2d / SyncBatchNorm of ConvNormLayer under model.train(),
 * focoos/nn/layers/conv.py:78-98, focoos/nn/layers/norm.py get_norm) on NHWC bf16 [rows][C] with fp32 statistics.
 *   forward : fx_bn_stats_bf16 ACCUMULATES sums[c] += sum_r z, sums[C + c] += sum_r z^2 (zero `sums` first; all-reduce it
 *             across ranks for SyncBN) -> fx_bn_finalize_f32(sums, n) writes mean, rstd, scale = gamma * rstd,
 *             shift = beta - mean * scale and updates running_mean / running_var (unbiased) / num_batches_tracked (any of
 *             the three may be NULL) -> fx_bn_apply_bf16: y = act(z * scale + shift [+ residual]).
 *   backward: da = dy * act'(z * scale + shift [+ residual]);  fx_bn_bwd_stats_bf16 ACCUMULATES sums[c] += sum da (= dbeta),
 *             sums[C + c] += sum da * xhat (= dgamma), xhat = (z - mean) * rstd; (all-reduce for SyncBN);
 *             fx_bn_bwd_apply_bf16: dz = scale * (da - sums[c] * inv_n - xhat * sums[C + c] * inv_n); da_out (optional) = da,
 *             the gradient of the residual branch; dgamma_acc / dbeta_acc (both or neither): dgamma_acc[c] += sums[C + c],
 *             dbeta_acc[c] += sums[c] - the affine gradients accumulated into the parameter gradients by the same launch (pass the
 *             LOCAL sums' buffers only when sums has not been all-reduced).
 * z_f32 != 0: z is fp32 [rows][ldz] (the conv kernel's out_f32 epilogue) instead of bf16 - y depends on z - mean, and a bf16 z keeps
 * 8 bits of z, not of z - mean; the trainable graphs keep the pre-normalisation tensor in fp32. */
int fx_bn_stats_bf16(const void* z, int ldz, int z_f32, float* sums, int64_t rows, int C, fx_stream_t stream);
int fx_bn_finalize_f32(const float* sums, float n, const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                       float* running_var, int64_t* num_batches_tracked, float* mean, float* rstd, float* scale, float* shift, int C,
                       fx_stream_t stream);
int fx_bn_apply_bf16(const void* z, int ldz, int z_f32, const float* scale, const float* shift, const void* residual, int ldr, int act, void* y, int ldy,
                     int64_t rows, int C, fx_stream_t stream);
int fx_bn_bwd_stats_bf16(const void* dy, int lddy, const void* z, int ldz, int z_f32, const void* residual, int ldr, const float* scale,
                         const float* shift, const float* mean, const float* rstd, int act, float* sums, int64_t rows, int C,
                         fx_stream_t stream);
int fx_bn_bwd_apply_bf16(const void* dy, int lddy, const void* z, int ldz, int z_f32, const void* residual, int ldr, const float* scale,
                         const float* shift, const float* mean, const float* rstd, int act, const float* sums, float inv_n, void* da_out,
                         int ldda, void* dz, int lddz, int64_t rows, int C, float* dgamma_acc, float* dbeta_acc, fx_stream_t stream);

/* ---- training path, token-space layers (A17): correctness-first fp32-math kernels ----------------------------------
 * Activation on a saved pre-activation z (FX_ACT_RELU/SILU/GELU): y = act(z); dz = dy * act'(z). */
int fx_act_fwd_bf16(const void* z, int ldz, void* y, int ldy, int64_t rows, int cols, int act, fx_stream_t stream);
int fx_act_bwd_bf16(const void* dy, int lddy, const void* z, int ldz, void* dz, int lddz, int64_t rows, int cols, int act, fx_stream_t stream);

/* Bias gradient: out[c] += sum_r x[r][c] (fp32 atomics; caller zeroes out). */
int fx_colsum_bf16(const void* x, int ldx, float* out, int64_t rows, int cols, fx_stream_t stream);

/* Backward of fx_layernorm_bf16 (no residual): dx; dgamma/dbeta (optional) accumulate with fp32 atomics.  cols == 256. */
int fx_layernorm_bwd_bf16(const void* dy, int lddy, const void* x, int ldx, const float* gamma, void* dx, int lddx, float* dgamma, float* dbeta,
                          int rows, int cols, fx_stream_t stream);

/* Backward (adjoint) of fx_resize_bilinear_nhwc_bf16 in gather form: dx bf16 [B,H,W,C] from dy bf16 [B,Ho,Wo,C]; deterministic.
 * fx_cast_f32_bf16 converts n (multiple of 8) floats to bf16. */
int fx_resize_bilinear_bwd_nhwc_bf16(const void* dy, int lddy, void* dx, int lddx, int B, int H, int W, int C, int Ho, int Wo, fx_stream_t stream);
int fx_cast_f32_bf16(const float* x, void* y, int64_t n, fx_stream_t stream);

/* Backward of fx_mha_bf16 / fx_mha_masked_bf16 on the matrix cores (head_dim 32, any Lq / Lk): dq, dk, dv from the projected q, k, v
 * and dout; `mask_bits` as in fx_mha_masked_bf16 (bit set = key not allowed; a query whose bits forbid every key attends
 * everywhere - the reference clears such mask rows, bisenetformer/modelling.py:423-426).  Replaces what autograd derives for
 * nn.MultiheadAttention's softmax(q k^T / sqrt(d)) v (focoos/nn/layers/transformer.py:83-106, 206-238, 583-601).
 * workspace: fx_mha_bwd_workspace_bytes() = 3 floats per (image, head, query): log-sum-exp, D = sum_j P_ij dP_ij, mask-in-effect.
 * `o` of the unmasked entry point is not read (kept for ABI stability). */
size_t fx_mha_bwd_workspace_bytes(int B, int Lq, int Lk, int heads);
int fx_mha_bwd_bf16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* o, int ldo, const void* dout, int lddo,
                    void* dq, int lddq, void* dk, int lddk, void* dv, int lddv, int B, int Lq, int Lk, int heads, void* workspace,
                    size_t workspace_bytes, fx_stream_t stream);
int fx_mha_masked_bwd_bf16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* dout, int lddo, void* dq, int lddq,
                           void* dk, int lddk, void* dv, int lddv, int B, int Lq, int Lk, int heads, const uint32_t* mask_bits, int ld_mask_words,
                           void* workspace, size_t workspace_bytes, fx_stream_t stream);

/* Backward of fx_gather_rows_bf16 for unique indices (top-k): dsrc[b, idx[b,j], :] = dout[b,j,:]; dsrc zeroed by the caller. */
int fx_scatter_rows_bf16(const void* dout, int ldo, const int32_t* idx, int k, void* dsrc, int lds, int rows_per_batch, int B, int cols,
                         fx_stream_t stream);

/* SetCriterion.loss_labels_vfl (fai_detr/modelling.py:464-497) and its gradient in one pass: logits bf16 [rows][ld],
 * cls i32 [rows] (matched class, K = none), score f32 [rows] (IoU of the matched pair); *loss_out += scale * sum w*BCE,
 * dlogits (optional, bf16 [rows][lddl]) = scale * w * (sigmoid(x) - t); scale = loss weight / num_boxes. */
int fx_vfl_loss_bf16(const void* logits, int ld, const int32_t* cls, const float* score, float alpha, float gamma, float scale, float* loss_out,
                     void* dlogits, int lddl, int64_t rows, int K, fx_stream_t stream);

/* Iterative box refinement of the decoder (fai_detr/modelling.py:996-1013): box = sigmoid(delta + inverse_sigmoid(ref)), eps as
 * focoos/nn/layers/functional.py's inverse_sigmoid (1e-5).  delta bf16, ref / box fp32, n elements.  Backward: d_delta bf16, d_ref fp32 (NULL:
 * ref was detached), autograd's clamp conventions. */
int fx_box_refine_f32(const void* delta, const float* ref, float* box, int64_t n, float eps, fx_stream_t stream);
int fx_box_refine_bwd_f32(const float* grad_box, const float* box, const float* ref, void* d_delta, float* d_ref, int64_t n, float eps,
                          fx_stream_t stream);

/* Fork: `side` waits for everything queued on `main` so far; join: `main` waits for `side`.  Valid inside
 * fx_graph_begin/fx_graph_end (the side stream joins the capture), which turns two launch sequences into independent
 * branches of one hipGraph - used to run the two half-batches of a step concurrently. */
int fx_stream_fork(fx_stream_t main_stream, fx_stream_t side_stream);
int fx_stream_join(fx_stream_t main_stream, fx_stream_t side_stream);

/* hipGraph capture of a launch sequence issued on `stream` (HIP graphs instead of a tracing compiler). */
int fx_graph_begin(fx_stream_t stream);
int fx_graph_end(fx_stream_t stream, void** graph_exec_out);
int fx_graph_launch(void* graph_exec, fx_stream_t stream);
int fx_graph_destroy(void* graph_exec);

/* Timing helper for bench.py: average duration in ms of `iters` graph replays measured with HIP events
 * recorded on `stream` itself. */
int fx_graph_time(void* graph_exec, fx_stream_t stream, int iters, float* ms_avg);

/* ---- training direction of the mask families (BASELINE config 5) ------------------------------------
 * Gradients of fx_mask_set_loss_f32 (SetCriterion.loss_labels / loss_masks under autograd, focoos/models/fai_mf/loss.py:411-431,
 * 463-523 == bisenetformer/loss.py): same arguments plus the forward's workspace (pair sums and CE weight sum are read from it),
 * grad3 f32 [3] ON THE DEVICE = upstream gradients of (loss_ce, loss_mask, loss_dice); writes dlogits f32 [B,Q,lddl] (all K+1
 * columns of every row) and ACCUMULATES into dmasks f32 [B,Q,h,w] (zero-initialised by the caller; only the planes of matched
 * queries are touched).  The point selection is recomputed (deterministic); sample coordinates carry no gradient.  pred_idx must not repeat within an image (a Hungarian matching never does): each matched
 * plane has one writer (the gradient is accumulated in LDS per plane band and stored, not added). */
int fx_mask_set_loss_bwd_f32(const float* logits, int ldl, const float* pred_masks, int h, int w, const void* tgt_masks, int tgt_is_u8, int H, int W,
                             const int32_t* tgt_labels, const int32_t* tgt_offsets, int sum_T, const int32_t* pred_idx, const int32_t* tgt_idx,
                             const float* rand_over, int n_over, const float* rand_extra, int n_extra, int num_points, int B, int Q, int K,
                             float eos_coef, float num_masks, float w_ce, float w_mask, float w_dice, const void* workspace, size_t workspace_bytes,
                             const float* grad3, float* dlogits, int lddl, float* dmasks, fx_stream_t stream);

/* rows[b][p][q] = bf16(planes[b][q][p]) (q < Q; zero for Q <= q < Qp, Qp % 8 == 0, Qp <= 240): the [B,Q,h*w] f32 mask-logit gradient as
 * pixel-major bf16 rows - the operand layout of the two GEMMs behind the backward of einsum("bqc,bchw->bqhw")
 * (bisenetformer/modelling.py:84). */
int fx_planes_to_rows_bf16(const float* planes, int Q, int P, void* rows, int ld_rows, int Qp, int B, fx_stream_t stream);

/* Backward of fx_dwconv3x3s2_nhwc_bf16 (CatBottleneck avd_layer / AvgPool2d(3,2,1) skip, focoos/nn/backbone/stdc.py:120-166):
 * dx bf16 [B,H,W,C] (NULL: skipped) from dy bf16 [B,Ho,Wo,C] and w f32 [9][C]; dw f32 [9][C] += sum dy * x-tap (NULL: skipped;
 * accumulated with atomics, zero-initialised by the caller). */
int fx_dwconv3x3s2_bwd_nhwc_bf16(const void* dy, int lddy, const void* x, int ldx, const float* w, void* dx, int lddx, float* dw, int B, int H, int W,
                                 int C, fx_stream_t stream);

/* out[b][c] = scale * sum_p a[b,p,c] * (b ? b[b,p,c] : 1), f32 [B,ldo]: gradient of the attention gates (feat * atten) and of
 * broadcast additions (bisenetformer/modelling.py:159-167, 186-212, 226-237).  splits > 1: pixel ranges + atomics (out zeroed by the caller). */
int fx_rowdot_nhwc_bf16(const void* a, int lda, const void* b, int ldb, float scale, float* out, int ldo, int B, int P, int C, int splits,
                        fx_stream_t stream);

/* y[b,p,c] = bf16(scale * vec[b][c]): gradient of feat.mean((2,3)) towards feat (scale = 1/P). */
int fx_bcast_vec_nhwc_bf16(const float* vec, int ldv, float scale, void* y, int ldy, int B, int P, int C, fx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FOCOOS_AMD_H */
