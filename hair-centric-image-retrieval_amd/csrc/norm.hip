// norm.hip — LayerNorm (fp32 rows -> fp16), CLS head (final LN + L2 normalise),
// patch-token mean.  One wave per row; row statistics in fp32, two-pass (mean, then
// centred variance) over registers, as ATen's layer_norm does in fp32
// (nn.LayerNorm(eps=1e-6): torchvision EncoderBlock.ln_1/ln_2, Encoder.ln via
// HP/src/main_backbone.py:554; models_vit.LayerNorm HP/src/models_vit.py:23-27).
#include "common.h"

namespace {


// 4 consecutive elements of a residual-stream row (fp32 or fp16 storage) as fp32.
template <typename XT>
__device__ __forceinline__ f32x4 load4(const XT* p) {
  if constexpr (sizeof(XT) == 4) {
    return *reinterpret_cast<const f32x4*>(p);
  } else {
    const f16x4 h = *reinterpret_cast<const f16x4*>(p);
    return (f32x4){(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
  }
}

// Loads row (d <= 2048, d % 4 == 0) as float4 per lane: element index 4*(lane + 64 j) + e.
template <int NV, typename XT>
__device__ __forceinline__ void ln_row(const XT* __restrict__ x, int d, int lane, f32x4 (&v)[NV],
                                       float& mean, float& rstd, float eps) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int k = 4 * (lane + 64 * j);
    v[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (k < d) v[j] = load4<XT>(x + k);
    s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
  }
  mean = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int k = 4 * (lane + 64 * j);
    if (k < d) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float c = v[j][e] - mean;
        q = __builtin_fmaf(c, c, q);
      }
    }
  }
  rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
}

template <int NV, typename XT>
__global__ __launch_bounds__(256) void layernorm_f16_kernel(const XT* __restrict__ x, int64_t rows,
                                                            int d, int64_t ldx,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps,
                                                            _Float16* __restrict__ y, int64_t ldy) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  f32x4 v[NV];
  float mean, rstd;
  ln_row<NV, XT>(x + row * ldx, d, lane, v, mean, rstd, eps);
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int k = 4 * (lane + 64 * j);
    if (k < d) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + k);
      const f32x4 b = *reinterpret_cast<const f32x4*>(beta + k);
      f16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (_Float16)((v[j][e] - mean) * rstd * g[e] + b[e]);
      *reinterpret_cast<f16x4*>(y + row * ldy + k) = o;
    }
  }
}

// emb[b] = l2norm?( ln?( tok[b][0] ) )
template <int NV, typename XT>
__global__ __launch_bounds__(256) void cls_head_kernel(const XT* __restrict__ tok, int64_t b, int t,
                                                       int d, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps,
                                                       int l2, float* __restrict__ e32,
                                                       _Float16* __restrict__ e16) {
  const int lane = threadIdx.x & 63;
  const int64_t bi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bi >= b) return;
  const XT* x = tok + bi * t * (int64_t)d;
  f32x4 v[NV];
  float mean = 0.f, rstd = 1.f;
  if (gamma) {
    ln_row<NV, XT>(x, d, lane, v, mean, rstd, eps);
  } else {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int k = 4 * (lane + 64 * j);
      v[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (k < d) v[j] = load4<XT>(x + k);
    }
  }
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int k = 4 * (lane + 64 * j);
    if (k < d && gamma) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + k);
      const f32x4 bb = *reinterpret_cast<const f32x4*>(beta + k);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[j][e] = (v[j][e] - mean) * rstd * g[e] + bb[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) ss = __builtin_fmaf(v[j][e], v[j][e], ss);
  }
  float inv = 1.f;
  if (l2) inv = 1.0f / fmaxf(sqrtf(wave_sum(ss)), 1e-12f);  // F.normalize eps
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int k = 4 * (lane + 64 * j);
    if (k < d) {
      f32x4 o;
      f16x4 oh;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = v[j][e] * inv;
        oh[e] = (_Float16)o[e];
      }
      if (e32) *reinterpret_cast<f32x4*>(e32 + bi * d + k) = o;
      if (e16) *reinterpret_cast<f16x4*>(e16 + bi * d + k) = oh;
    }
  }
}

// out[b] = mean over tokens 1..t-1 of ln?(tok[b][i]); one workgroup (4 waves) per image.
template <int NV, typename XT>
__global__ __launch_bounds__(256) void patch_mean_kernel(const XT* __restrict__ tok, int t, int d,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps,
                                                         float* __restrict__ out) {
  __shared__ float part[4][2048];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t bi = blockIdx.x;
  f32x4 acc[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int i = 1 + wave; i < t; i += 4) {
    const XT* x = tok + (bi * t + i) * (int64_t)d;
    f32x4 v[NV];
    float mean = 0.f, rstd = 1.f;
    if (gamma) {
      ln_row<NV, XT>(x, d, lane, v, mean, rstd, eps);
    } else {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int k = 4 * (lane + 64 * j);
        v[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (k < d) v[j] = load4<XT>(x + k);
      }
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int k = 4 * (lane + 64 * j);
      if (k < d) {
        if (gamma) {
          const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + k);
          const f32x4 bb = *reinterpret_cast<const f32x4*>(beta + k);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[j][e] += (v[j][e] - mean) * rstd * g[e] + bb[e];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[j][e] += v[j][e];
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int k = 4 * (lane + 64 * j);
    if (k < d) *reinterpret_cast<f32x4*>(&part[wave][k]) = acc[j];
  }
  __syncthreads();
  for (int k = threadIdx.x; k < d; k += 256)
    out[bi * d + k] = ((part[0][k] + part[1][k]) + (part[2][k] + part[3][k])) / (float)(t - 1);
}

#define DISPATCH_NV(d, CALL)                \
  do {                                      \
    const int nv_ = (int)hcir_cdiv(d, 256); \
    if (nv_ <= 2) { CALL(2); }              \
    else if (nv_ <= 3) { CALL(3); }         \
    else if (nv_ <= 4) { CALL(4); }         \
    else if (nv_ <= 5) { CALL(5); }         \
    else { CALL(8); }                       \
  } while (0)

}  // namespace

extern "C" {

int hcir_layernorm_f16(const void* x, int x_dtype, int64_t rows, int32_t d, int64_t ldx,
                       const float* gamma, const float* beta, float eps, void* y_f16, int64_t ldy,
                       void* stream) {
  HCIR_ENTER();
  if (!x || !gamma || !beta || !y_f16 || rows <= 0 || d <= 0 || (d & 3) || d > 2048)
    return HCIR_ERR_INVALID;
  if (ldx < d || ldy < d || (ldx & 3) || (ldy & 3)) return HCIR_ERR_INVALID;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)hcir_cdiv(rows, 4)), block(256);
  if (x_dtype != HCIR_F32 && x_dtype != HCIR_F16) return HCIR_ERR_UNSUPPORTED;
#define CALL(NV)                                                                                      \
  if (x_dtype == HCIR_F32)                                                                            \
    hipLaunchKernelGGL((layernorm_f16_kernel<NV, float>), grid, block, 0, st, (const float*)x, rows, d, \
                       ldx, gamma, beta, eps, static_cast<_Float16*>(y_f16), ldy);                    \
  else                                                                                                \
    hipLaunchKernelGGL((layernorm_f16_kernel<NV, _Float16>), grid, block, 0, st, (const _Float16*)x,    \
                       rows, d, ldx, gamma, beta, eps, static_cast<_Float16*>(y_f16), ldy)
  DISPATCH_NV(d, CALL);
#undef CALL
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}

int hcir_cls_head(const void* tok, int tok_dtype, int64_t b, int32_t t, int32_t d,
                  const float* gamma, const float* beta, float eps, int l2_normalize, float* emb_f32,
                  void* emb_f16, void* stream) {
  HCIR_ENTER();
  if (!tok || b <= 0 || t <= 0 || d <= 0 || (d & 3) || d > 2048) return HCIR_ERR_INVALID;
  if ((gamma == nullptr) != (beta == nullptr)) return HCIR_ERR_INVALID;
  if (!emb_f32 && !emb_f16) return HCIR_ERR_INVALID;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)hcir_cdiv(b, 4)), block(256);
  if (tok_dtype != HCIR_F32 && tok_dtype != HCIR_F16) return HCIR_ERR_UNSUPPORTED;
#define CALL(NV)                                                                                     \
  if (tok_dtype == HCIR_F32)                                                                         \
    hipLaunchKernelGGL((cls_head_kernel<NV, float>), grid, block, 0, st, (const float*)tok, b, t, d,   \
                       gamma, beta, eps, l2_normalize, emb_f32, static_cast<_Float16*>(emb_f16));    \
  else                                                                                               \
    hipLaunchKernelGGL((cls_head_kernel<NV, _Float16>), grid, block, 0, st, (const _Float16*)tok, b, t, \
                       d, gamma, beta, eps, l2_normalize, emb_f32, static_cast<_Float16*>(emb_f16))
  DISPATCH_NV(d, CALL);
#undef CALL
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}

int hcir_patch_mean(const void* tok, int tok_dtype, int64_t b, int32_t t, int32_t d,
                    const float* gamma, const float* beta, float eps, float* out_f32, void* stream) {
  HCIR_ENTER();
  if (!tok || !out_f32 || b <= 0 || t <= 1 || d <= 0 || (d & 3) || d > 2048) return HCIR_ERR_INVALID;
  if ((gamma == nullptr) != (beta == nullptr)) return HCIR_ERR_INVALID;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (tok_dtype != HCIR_F32 && tok_dtype != HCIR_F16) return HCIR_ERR_UNSUPPORTED;
#define CALL(NV)                                                                                      \
  if (tok_dtype == HCIR_F32)                                                                          \
    hipLaunchKernelGGL((patch_mean_kernel<NV, float>), dim3((unsigned)b), dim3(256), 0, st,            \
                       (const float*)tok, t, d, gamma, beta, eps, out_f32);                           \
  else                                                                                                \
    hipLaunchKernelGGL((patch_mean_kernel<NV, _Float16>), dim3((unsigned)b), dim3(256), 0, st,         \
                       (const _Float16*)tok, t, d, gamma, beta, eps, out_f32)
  DISPATCH_NV(d, CALL);
#undef CALL
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}

}  // extern "C"
