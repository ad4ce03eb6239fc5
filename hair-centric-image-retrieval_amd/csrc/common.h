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
