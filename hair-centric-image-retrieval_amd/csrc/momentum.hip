// momentum.hip — EMA update of the momentum encoder as ONE launch over all parameter tensors.
//
// Replaces lightly.models.utils.update_momentum(model, model_ema, m) as called once per training step for
// the backbone and the projection head (HP/src/pretrain_engine.py:618-619; same arithmetic in
// HP/utils/utils.py:113-115), a Python loop of ~150 tensors x 3 elementwise launches:
//     ema = ema * m + p * (1 - m)
// Bit-exact with that expression in fp32: two rounded multiplies and one rounded add (no FMA contraction),
// (1 - m) rounded to fp32 as torch does when it multiplies an fp32 tensor by a Python float.
// The caller passes a chunk table in device memory (built once per model pair): chunk c covers count[c]
// consecutive floats at dst[c] / src[c]; one workgroup per chunk, 16 B per lane per access.  HBM-bound:
// 12 B per parameter.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void ema_update_kernel(const uint64_t* __restrict__ dst,
                                                         const uint64_t* __restrict__ src,
                                                         const int64_t* __restrict__ count, float m,
                                                         float om) {
#pragma clang fp contract(off)  // a * m + b * om must stay two multiplies and an add (HIP contracts by default)
  const int64_t c = blockIdx.x;
  float* __restrict__ e = reinterpret_cast<float*>(dst[c]);
  const float* __restrict__ p = reinterpret_cast<const float*>(src[c]);
  const int64_t n = count[c];
  const bool vec = ((dst[c] | src[c]) & 15) == 0;
  const int64_t n4 = vec ? n / 4 : 0;
  for (int64_t i = threadIdx.x; i < n4; i += blockDim.x) {
    f32x4 a = reinterpret_cast<f32x4*>(e)[i];
    const f32x4 b = reinterpret_cast<const f32x4*>(p)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = a[k] * m + b[k] * om;
    reinterpret_cast<f32x4*>(e)[i] = a;
  }
  for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x)
    e[i] = e[i] * m + p[i] * om;
}

}  // namespace

extern "C" int hcir_ema_update(const uint64_t* dst_ptrs, const uint64_t* src_ptrs, const int64_t* counts,
                               int64_t n_chunks, float m, float one_minus_m, void* stream) {
  HCIR_ENTER();
  if (!dst_ptrs || !src_ptrs || !counts || n_chunks <= 0 || n_chunks > 0x7fffffff) return HCIR_ERR_INVALID;
  hipLaunchKernelGGL(ema_update_kernel, dim3((unsigned)n_chunks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     dst_ptrs, src_ptrs, counts, m, one_minus_m);
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}
