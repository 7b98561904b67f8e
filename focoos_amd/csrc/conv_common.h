// Shared definitions of the implicit-GEMM convolution kernels (conv_igemm.hip, conv_igemm_dma.hip).
#pragma once
#include "common.h"

#ifndef FX_K64_MIN_KTOT
#define FX_K64_MIN_KTOT 1024
#endif

struct ConvArgs {
  const bf16_t* x;
  const bf16_t* w;
  const float* bias;
  const bf16_t* res;
  void* y;
  int B, H, W, C, ldx;
  int Ho, Wo, N, ldy, ldr;
  int KH, KW, stride, pad;
  int act, out_f32, res_after;
  int M, Ktot, nNt, Nstore;
  unsigned x_bytes, w_bytes, r_bytes;  // buffer sizes for the bounds-checked buffer loads (< 4 GiB)
  int64_t y_bstride;          // 0: contiguous
  const bf16_t* mask;         // optional: result *= (mask > 0) (implicit-GEMM kernels only)
  int ldm;
  unsigned m_bytes;
};

// Byte offset of 16-byte chunk `chunk` of tile row `row` in an LDS tile with BK bf16 per row.  The XOR swizzle makes the
// 16 rows a ds_read_b128 lane group touches land on 16 distinct 16-byte slots of the 256-byte bank window.
template <int BK>
__device__ __forceinline__ int lds_off(int row, int chunk) {
  constexpr int CPR = BK / 8;  // 16-byte chunks per row
  if constexpr (BK >= 128) {
    return row * (BK * 2) + ((chunk ^ (row & 15)) << 4);  // rows are whole bank windows: rotate by the row index
  } else {
    constexpr int R = 256 / (BK * 2);  // rows per 256-byte bank window
    return row * (BK * 2) + ((chunk ^ ((row / R) & (CPR - 1))) << 4);
  }
}

// Tuning knobs (compile-time defaults, overridable through the environment for A/B runs).
int fx_tune(const char* env_name, int default_value);

// 16-byte buffer load: out-of-range offsets (>= num_records) return zeros without touching memory,
// which gives zero padding / M-tail predication for free (one v_cndmask on the offset).
__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}
#define FX_OOB 0xFFFFFFF0u


// conv_igemm_dma.hip: 8-wave 256-row tiles fed by buffer_load ... lds DMA (deep-K, large-M layers)
bool fx_conv_dma_eligible(const ConvArgs& a);
int fx_launch_conv_dma(ConvArgs& a, hipStream_t stream);

// conv3x3_flat.hip: 3x3 / stride 1 / pad 1 without im2col re-fetching (flat halo tile, loader wave, fragment-ordered weights)
extern "C" int fx_conv3x3_flat_supported(int C, int N, int W);
int fx_launch_conv3x3_flat(const ConvArgs& c, const bf16_t* w_frag, hipStream_t stream);
// conv3x3_kplane.hip: the round-3 form of the same layer class (k-plane LDS layout, immediate-offset fragment reads, direct stores)
extern "C" int fx_conv3x3_kplane_supported(int C, int N, int W);
int fx_launch_conv3x3_kplane(const ConvArgs& c, const bf16_t* w_frag, hipStream_t stream);
// conv3x3s2_kplane.hip: 3x3 / stride 2 / pad 1 as a stride-1 correlation over the four parity planes of the input (round 4)
bool fx_conv3x3s2_kplane_supported(int C, int N, int Wo, int M);
int fx_launch_conv3x3s2_kplane(const ConvArgs& c, const bf16_t* w_frag, hipStream_t stream);
// conv_pw_kplane.hip: pointwise layers with the whole reduction resident in LDS (K = 256 / 512), one workgroup per pixel tile over all N
bool fx_pw_kplane_supported(int C, int N, int mode);
int fx_launch_pw_kplane(const ConvArgs& c, const bf16_t* w_frag, hipStream_t stream);
// conv3x3_c32.hip: 3x3 / s1 with 32 input channels (the ResNet-vd stem layers conv1_2 / conv1_3): one-chunk k-plane kernel, 4 waves, two per CU
bool fx_conv3x3_c32_supported(int C, int N, int W, int mode);
int fx_launch_conv3x3_c32(const ConvArgs& c, const bf16_t* w_frag, hipStream_t stream);
// conv3x3_c64.hip: 3x3 / s1 with 64 input and 64 output channels, no residual (res2 branch2b): 2-D tiles, the whole filter LDS-resident (round 5)
bool fx_conv3x3_c64_supported(int C, int N, int mode);
int fx_launch_conv3x3_c64(const ConvArgs& c, const bf16_t* w_frag, hipStream_t stream);
int fx_c3_epilogue_mode(int act, bool has_res, int res_after);  // epilogue variant (3x3 kernel: 0-3, 5; pointwise: 0, 1, 3-6), -1: none
int fx_launch_pw_flat(const ConvArgs& c, const bf16_t* w_frag, hipStream_t stream);  // 1x1, C % 256 == 0, N % 256 == 0
