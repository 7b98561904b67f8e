// Common device/host helpers for the focoos_amd gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/focoos_amd.h"

#define FX_CHECK_ARG(cond) \
  do {                     \
    if (!(cond)) return FX_ERR_INVALID_ARGUMENT; \
  } while (0)

static inline int fx_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? FX_OK : FX_ERR_LAUNCH;
}

// ---- the 16-bit storage element of activations / packed weights / activation gradients -----------------------------------------
// Default build: bfloat16.  -DFX_FP16=1 (the second library, libfocoos_amd_fp16.so: BASELINE configs[4] names fp16 - the reference trains
// under torch.autocast(float16) + GradScaler, trainer/trainer.py:645,735-773): IEEE half, 11 significand bits instead of 8, exponent range
// 6e-8 .. 65504 - which is why that build's training step carries a dynamic loss scale (fx_adamw_step_scaled_f32).  Every kernel takes its
// element type from the helpers below (no kernel touches the bit layout itself), the MFMA instruction from FX_MFMA_*; the names keep the
// historical "bf16" (`bf16_t` = raw 16-bit storage word, `bf16x8` = an MFMA operand fragment of eight elements).
typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#ifndef FX_FP16
#define FX_FP16 0
#endif

#if FX_FP16
typedef __attribute__((ext_vector_type(8))) _Float16 bf16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 bf16x2_t;
#define FX_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define FX_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {   // round-to-nearest-even; |f| > 65504 -> +-inf (what the loss scaler watches for)
  _Float16 h = (_Float16)f;
  return __builtin_bit_cast(bf16_t, h);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  bf16x2_t v = {(_Float16)lo, (_Float16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
// the two storage elements of a 32-bit word (low half = even element) as floats
__device__ __forceinline__ float bf16lo_to_f32(uint32_t u) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(u & 0xffffu)); }
__device__ __forceinline__ float bf16hi_to_f32(uint32_t u) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(u >> 16)); }
#else
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
#define FX_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define FX_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

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
#endif

__device__ __forceinline__ void unpack_bf16x8(const uint4& v, float* f) {
  f[0] = bf16lo_to_f32(v.x); f[1] = bf16hi_to_f32(v.x);
  f[2] = bf16lo_to_f32(v.y); f[3] = bf16hi_to_f32(v.y);
  f[4] = bf16lo_to_f32(v.z); f[5] = bf16hi_to_f32(v.z);
  f[6] = bf16lo_to_f32(v.w); f[7] = bf16hi_to_f32(v.w);
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

// Issue-priority asymmetry between the workgroups that share a CU (round 6).  Kernels whose workgroups run two (or more) per CU start
// together and, being bound by the same resource, advance in lockstep: load phase beside load phase, MFMA phase beside MFMA phase at half
// rate each, store phase beside store phase - the co-residency then overlaps nothing.  Giving the wave in the even hardware slot of every
// SIMD (HW_ID.wave_id, hwreg 4 bits 3:0) the higher priority lets one workgroup run its MFMA phase at full rate while the other fills the
// gaps and owns the pipes during the first one's load / store phases.  Scalar (wave-uniform) branch, two instructions.
__device__ __forceinline__ void fx_prio_by_hw_slot() {
  const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1u;
  if (slot) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(3);
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
