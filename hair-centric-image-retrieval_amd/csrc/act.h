// act.h — GELU (erf form) and its derivative, shared by the GEMM epilogue (gemm.hip) and the training kernels
// (train.hip).  nn.GELU() of torchvision's MLPBlock / timm's Mlp (HP/src/main_backbone.py:554; HP/src/models_vit.py:19).
#pragma once
#include "common.h"

// gelu(x) = 0.5 x (1 + erf(x / sqrt 2)), erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the
// fp16 rounding of the output) instead of erff's ~40 instructions: the fc1 epilogue evaluates it 64K times per
// tile and is VALU-bound there.  With q = (1/2) erfc(|x|/sqrt2) = P(t) t exp(-x^2/2), t = 1/(1 + p|x|/sqrt2):
//   gelu(x) = max(x,0) - |x| q = |x| (1/2 - q) + x/2
// (no max, no sign select; exact 0 / x in the tails), arranged so that hipcc emits packed fp32 ops on pairs:
// 8.5 VALU instructions per element (3.5 v_pk_fma + 2 v_pk_mul + v_and + v_rcp + v_exp), 11.5 before.
typedef float gelu_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ gelu_f32x2 gelu_erf2(gelu_f32x2 x) {
  constexpr float kPW = 0.3275911f * 0.70710678118654752440f;      // p / sqrt2
  constexpr float kNW2 = -0.5f * 1.44269504088896340736f;          // exp(-x^2/2) = exp2(kNW2 x^2)
  const gelu_f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
  const gelu_f32x2 den = ax * kPW + 1.0f;
  const gelu_f32x2 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  gelu_f32x2 p = t * (0.5f * 1.061405429f) + (0.5f * -1.453152027f);
  p = p * t + (0.5f * 1.421413741f);
  p = p * t + (0.5f * -0.284496736f);
  p = p * t + (0.5f * 0.254829592f);
  const gelu_f32x2 s = (x * x) * kNW2;
  const gelu_f32x2 e = {__builtin_amdgcn_exp2f(s[0]), __builtin_amdgcn_exp2f(s[1])};
  const gelu_f32x2 pt = p * t;
  const gelu_f32x2 qn = 0.5f - pt * e;
  return ax * qn + 0.5f * x;
}
__device__ __forceinline__ float gelu_erf(float x) { return gelu_erf2((gelu_f32x2){x, x})[0]; }


// d/dx gelu(x) = Phi(x) + x phi(x),  Phi = 1 - q (x >= 0) or q (x < 0) with the same q = (1/2) erfc(|x| / sqrt 2)
// as above, phi(x) = exp(-x^2 / 2) / sqrt(2 pi).
__device__ __forceinline__ float gelu_erf_grad(float x) {
  constexpr float kPW = 0.3275911f * 0.70710678118654752440f;
  constexpr float kNW2 = -0.5f * 1.44269504088896340736f;
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(ax, kPW, 1.0f));
  float p = __builtin_fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
  p = __builtin_fmaf(p, t, 0.5f * 1.421413741f);
  p = __builtin_fmaf(p, t, 0.5f * -0.284496736f);
  p = __builtin_fmaf(p, t, 0.5f * 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(x * x * kNW2);
  const float q = p * t * e;                       // (1/2) erfc(|x| / sqrt 2)
  const float cdf = x >= 0.f ? 1.0f - q : q;
  return __builtin_fmaf(x * 0.39894228040143267794f, e, cdf);
}
