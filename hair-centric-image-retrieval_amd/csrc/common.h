// common.h — shared device helpers for libhcir (gfx950 only: 64-wide waves, MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hcir.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define HCIR_WAVE 64

// hipGetLastError() is sticky per host thread: clear whatever an earlier runtime call of
// the caller (torch's event queries return hipErrorNotReady, ...) left behind, so that the
// check after our launch reports only our launch.
#define HCIR_ENTER() (void)hipGetLastError()

#define HCIR_LAUNCH_CHECK()                         \
  do {                                              \
    if (hipGetLastError() != hipSuccess) return HCIR_ERR_LAUNCH; \
  } while (0)

static inline int64_t hcir_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Row offset inside a 32x32 MFMA accumulator: register i of lane-half h holds
// row (i&3) + 8*(i>>2) + 4*h, column lane&31 (cdna_hip_programming.md §3).
__device__ __forceinline__ int acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

// (score desc, index asc) ordering used by every top-k in the library.
__device__ __forceinline__ bool better(float sa, int64_t ia, float sb, int64_t ib) {
  return (sa > sb) || (sa == sb && ia < ib);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// LDS-DMA (global_load_lds_dwordx4: 16 B per lane to wave-uniform base + lane * 16) issued OUTSIDE the compiler's
// view.  With the builtin, hipcc's waitcnt pass treats every ds_read_b64_tr_b16 (a builtin without a memory operand
// it could disambiguate) as possibly aliasing the pending transfer and puts `s_waitcnt vmcnt(0)` in front of it: a
// stage prefetched under a loop of transposed reads was waited for at the loop's first read (seen in the .s of
// gemm_tn and attn_bwd).  As inline asm the transfer is invisible to that pass; callers retire it with an explicit
// `s_waitcnt vmcnt(0)` before the barrier that publishes the stage.  VMEM operations the compiler does not know of
// only make the vmcnt(N) it computes for its own loads stricter (returns are in order), never wrong.
// Source = wave-uniform base (SGPR pair) + 32-bit per-lane byte offset; destination = wave-uniform LDS byte address
// (lds_addr() of the 1 KB the 64 lanes fill).  M0 is on the clobber list: the compiler may not keep a value of its own
// in M0 across a transfer (a builtin LDS-DMA, movrel, readlane through m0 in the same kernel).
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(size_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ void lds_dma16(const void* src_base, uint32_t src_off, uint32_t lds_wave_addr) {
  const uint32_t m = __builtin_amdgcn_readfirstlane(lds_wave_addr);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(src_off), "s"(src_base), "s"(m)
               : "memory", "m0");
}
__device__ __forceinline__ void lds_dma16_nt(const void* src_base, uint32_t src_off, uint32_t lds_wave_addr) {  // non-temporal
  const uint32_t m = __builtin_amdgcn_readfirstlane(lds_wave_addr);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(src_off), "s"(src_base), "s"(m)
               : "memory", "m0");
}
// Same, per-lane 64-bit source address (no wave-uniform base at hand).
__device__ __forceinline__ void lds_dma16_v(const void* gsrc, uint32_t lds_wave_addr) {
  const uint32_t m = __builtin_amdgcn_readfirstlane(lds_wave_addr);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(m) : "memory", "m0");
}
__device__ __forceinline__ void lds_dma16_v_nt(const void* gsrc, uint32_t lds_wave_addr) {   // non-temporal
  const uint32_t m = __builtin_amdgcn_readfirstlane(lds_wave_addr);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" ::"v"(gsrc), "s"(m)
               : "memory", "m0");
}
