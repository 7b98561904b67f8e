// Shared pieces of the back-to-back GEMM kernels (conv_pw_chain.hip, score_head.hip): LDS-DMA tile loads with the XOR swizzle
// on the source address, fragment-ordered weight loads.
#pragma once
#include <type_traits>

#include "conv_common.h"

typedef __attribute__((address_space(3))) void pw_lds_void_t;

__device__ __forceinline__ void pw_dma16(__amdgpu_buffer_rsrc_t r, unsigned char* lds_base, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (pw_lds_void_t*)lds_base, 16, voff, 0, 0, 0);
}

// physical 16-byte chunk of logical chunk `chunk` in row `row` of an LDS tile with RL chunks per row
template <int RL>
__device__ __forceinline__ int pw_swz(int row, int chunk) {
  if constexpr (RL >= 16) {
    return chunk ^ (row & 15);
  } else {
    constexpr int R = 16 / RL;  // rows per 256-byte bank window
    return chunk ^ ((row / R) & (RL - 1));
  }
}

// DMA `nrows` rows of RL chunks (global row m0+row, element stride ld, column offset col0) into a swizzled LDS tile.
// A wave-instruction fills 64 consecutive physical chunks; the 4 waves take the instructions round-robin.
template <int RL>
__device__ __forceinline__ void pw_dma_rows(__amdgpu_buffer_rsrc_t r, unsigned char* tile, int nrows, int m0, int M, int ld, int col0,
                                            int wave, int lane) {
  const int ninstr = nrows * RL / 64;
  for (int i = wave; i < ninstr; i += 4) {
    const int q = i * 64 + lane;
    const int row = q / RL, pc = q % RL;
    const int lc = pw_swz<RL>(row, pc);
    const int m = m0 + row;
    const unsigned off = (m < M) ? (unsigned)(m * ld + col0 + lc * 8) * 2u : FX_OOB;
    pw_dma16(r, tile + i * 1024, off);
  }
}

__device__ __forceinline__ bf16x8 pw_ldg_frag(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }


// Weight-fragment loads hidden from hipcc's waitcnt bookkeeping (guide 5.7 form ii): hipcc opens every iteration of a loop that
// carries register loads with s_waitcnt vmcnt(0), which would expose one L2 round trip per filter tap.  The load is an asm
// statement on a read-write operand (the ring slot keeps ONE register across the loop: no compiler copy of a value that has
// not landed), and c3_wait<N> names the fragments an MFMA is about to read, after a counted wait: loads return in order, and
// between a slot's refill and its use exactly (KJ-1)*TN younger refills are issued.
template <int N, int I = 0, typename F>
__device__ __forceinline__ void c3_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    c3_static_for<N, I + 1>(f);
  }
}
template <int OFF = 0>
__device__ __forceinline__ void c3_ldg_async(bf16x8& dst, const bf16_t* ptr) {
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "+v"(dst) : "v"(ptr), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void c3_wait(bf16x8& f0) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(f0) : "n"(N));
}
template <int N>
__device__ __forceinline__ void c3_wait(bf16x8& f0, bf16x8& f1) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(f0), "+v"(f1) : "n"(N));
}

