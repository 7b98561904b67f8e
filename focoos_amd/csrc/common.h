// Common device/host helpers for the focoos_amd gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/focoos_amd.h"

typedef uint16_t bf16_t;  // raw bfloat16 bits; storage type of activations / packed weights
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define FX_CHECK_ARG(cond) \
  do {                     \
    if (!(cond)) return FX_ERR_INVALID_ARGUMENT; \
  } while (0)

static inline int fx_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? FX_OK : FX_ERR_LAUNCH;
}

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// float -> bf16, round-to-nearest-even (same rounding as torch's float->bfloat16): the __bf16 cast lowers to the
// gfx950 hardware convert (v_cvt_pk_bf16_f32, two values per instruction) instead of ~5 integer VALU ops per value.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(bf16_t, h);
}

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  bf16x2_t v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}

// the two storage elements of a 32-bit word (low half = even element) as floats
__device__ __forceinline__ float bf16lo_to_f32(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16hi_to_f32(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

__device__ __forceinline__ void unpack_bf16x8(const uint4& v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}

__device__ __forceinline__ uint4 pack_bf16x8(const float* f) {
  uint4 v;
  v.x = pack_bf16x2(f[0], f[1]); v.y = pack_bf16x2(f[2], f[3]);
  v.z = pack_bf16x2(f[4], f[5]); v.w = pack_bf16x2(f[6], f[7]);
  return v;
}

__device__ __forceinline__ float fx_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }

// activation codes: FX_ACT_*
__device__ __forceinline__ float fx_act(float x, int act) {
  switch (act) {
    case FX_ACT_RELU: return fmaxf(x, 0.0f);
    case FX_ACT_SILU: return x / (1.0f + __expf(-x));
    case FX_ACT_GELU: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
    default: return x;
  }
}

// XCD-aware bijective remap of a linear workgroup id (guide §5.5 T1): block b runs on XCD b%8;
// give each XCD a contiguous chunk of the tile space so neighbouring tiles share one L2.
__device__ __forceinline__ int fx_xcd_remap(int bid, int nwg) {
  const int NX = 8;
  int xcd = bid % NX, local = bid / NX;
  int q = nwg / NX, r = nwg % NX;
  int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + local;
}
