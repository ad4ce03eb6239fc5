// augment.hip — the two pieces of the HSimCLR step's DEFAULT path that were still torch / torchvision:
//
//   hcir_positive_transform   positive_transform of HP/utils/transform.py:21-24, applied to the device batch at
//                             HP/src/pretrain_engine.py:686: T.RandomRotation((-15, 15)) then
//                             T.GaussianBlur(kernel_size=3, sigma=(0.1, 0.5)).  On a batch tensor torchvision draws ONE
//                             angle and ONE sigma per call; the caller (hcir.transform.PositiveTransform) draws them
//                             the same way and passes the derived numbers.  One kernel, thread per output pixel:
//                             out(y, x) = sum_{dy,dx} k[dy] k[dx] R(reflect(y + dy), reflect(x + dx)),
//                             R = nearest-neighbour sample of the rotated image (torchvision's affine grid with the
//                             pixel-centre convention, grid_sample(nearest, zeros, align_corners = false): zero outside).
//   hcir_bn1d_fwd / _bwd      BatchNorm1d in TRAINING mode (batch statistics, running-statistic update) of lightly's
//                             SimCLRProjectionHead (HP/src/main_backbone.py:589; Linear -> BN -> ReLU -> Linear -> BN),
//                             forward and backward, so that the head's two GEMMs run on hcir_gemm_f16 / hcir_gemm_f16_tn
//                             like the rest of the step.  [B][F] row-major fp32 in; a workgroup owns 64 feature
//                             columns, its four row lanes stride the batch; fixed-order LDS reductions (deterministic).
// All HBM / latency bound: B x F is at most a few MB.
#include "common.h"

namespace {

struct PosTransformArgs {
  const float* in;
  float* out;
  int64_t planes;  // B * C
  int h, w;
  float t00, t01, t10, t11;  // rows of torchvision's inverse affine matrix, divided by (0.5 w, 0.5 h) as it does
  float k0, k1;              // Gaussian taps: k0 at +-1, k1 at 0
};

__device__ __forceinline__ int reflect1(int i, int n) {  // torch 'reflect' padding by one pixel
  return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i);
}

__device__ __forceinline__ float rotated_sample(const float* plane, const PosTransformArgs& a, int oy, int ox) {
  // _gen_affine_grid: base grid = pixel centres in [-w/2 + 0.5, w/2 - 0.5]; grid = base @ (theta^T / (0.5 w, 0.5 h))
  const float xg = (float)ox - 0.5f * (float)a.w + 0.5f, yg = (float)oy - 0.5f * (float)a.h + 0.5f;
  const float gx = xg * a.t00 + yg * a.t01, gy = xg * a.t10 + yg * a.t11;
  // grid_sampler_unnormalize (align_corners = false) + nearest = nearbyint (ties to even)
  const float fx = ((gx + 1.f) * (float)a.w - 1.f) * 0.5f, fy = ((gy + 1.f) * (float)a.h - 1.f) * 0.5f;
  const float rx = nearbyintf(fx), ry = nearbyintf(fy);
  if (!(rx >= 0.f && rx <= (float)(a.w - 1) && ry >= 0.f && ry <= (float)(a.h - 1))) return 0.f;
  return plane[(int64_t)(int)ry * a.w + (int)rx];
}

__global__ __launch_bounds__(256) void positive_transform_kernel(PosTransformArgs a) {
  const int64_t npix = (int64_t)a.h * a.w;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.planes * npix) return;
  const int64_t pl = t / npix;
  const int p = (int)(t - pl * npix);
  const int y = p / a.w, x = p - y * a.w;
  const float* plane = a.in + pl * npix;
  const float kk[3] = {a.k0, a.k1, a.k0};
  float acc = 0.f;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy) {
    const int yy = reflect1(y + dy, a.h);
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int xx = reflect1(x + dx, a.w);
      acc += (kk[dy + 1] * kk[dx + 1]) * rotated_sample(plane, a, yy, xx);
    }
  }
  a.out[t] = acc;
}

// ---- BatchNorm1d, training mode -------------------------------------------------------------------------------
constexpr int kBnCols = 64;

// y = (x - mean) * rstd * gamma + beta, optional ReLU; mean / biased variance over the batch; running statistics
// updated as torch does (momentum, UNBIASED variance).  Outputs: y32 and / or y16 (either may be null).
__global__ __launch_bounds__(256) void bn1d_fwd_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows, int f,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float eps, float momentum, int relu, float* __restrict__ running_mean,
                                                       float* __restrict__ running_var, float* __restrict__ save_mean,
                                                       float* __restrict__ save_rstd, float* __restrict__ y32, int64_t ldy32,
                                                       _Float16* __restrict__ y16, int64_t ldy16) {
  __shared__ float red[4][kBnCols];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * kBnCols + cl;
  const bool ok = c < f;
  float s = 0.f;
  if (ok)
    for (int64_t r = rl; r < rows; r += 4) s += x[r * ldx + c];
  red[rl][cl] = s;
  __syncthreads();
  const float mean = (red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]) / (float)rows;
  __syncthreads();
  float v = 0.f;
  if (ok)
    for (int64_t r = rl; r < rows; r += 4) {
      const float d = x[r * ldx + c] - mean;
      v += d * d;
    }
  red[rl][cl] = v;
  __syncthreads();
  const float m2 = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
  const float var = m2 / (float)rows;
  const float rstd = 1.0f / sqrtf(var + eps);
  if (!ok) return;
  if (rl == 0) {
    save_mean[c] = mean;
    save_rstd[c] = rstd;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (rows > 1 ? m2 / (float)(rows - 1) : var);
  }
  const float g = gamma[c] * rstd, b = beta[c];
  for (int64_t r = rl; r < rows; r += 4) {
    float y = (x[r * ldx + c] - mean) * g + b;
    if (relu) y = y > 0.f ? y : 0.f;
    if (y32) y32[r * ldy32 + c] = y;
    if (y16) y16[r * ldy16 + c] = (_Float16)y;
  }
}

// dx = gamma rstd / B * (B dy - sum(dy) - xhat sum(dy xhat)); dgamma = sum(dy xhat); dbeta = sum(dy).
// relu_out16 (optional): the fp16 ReLU output of the forward; where it is <= 0 the incoming gradient is zero.
// dy is multiplied by dy_scale (a DEVICE scalar, optional) first: the power-of-two renormalisation of the caller.
__global__ __launch_bounds__(256) void bn1d_bwd_kernel(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ x,
                                                       int64_t ldx, int64_t rows, int f, const float* __restrict__ gamma,
                                                       const float* __restrict__ save_mean, const float* __restrict__ save_rstd,
                                                       const _Float16* __restrict__ relu_out16, int64_t ldr,
                                                       const float* __restrict__ dy_scale, _Float16* __restrict__ dx16,
                                                       int64_t lddx, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[2][4][kBnCols];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * kBnCols + cl;
  const bool ok = c < f;
  const float sc = dy_scale ? *dy_scale : 1.f;
  const float mean = ok ? save_mean[c] : 0.f, rstd = ok ? save_rstd[c] : 0.f;
  float s1 = 0.f, s2 = 0.f;
  if (ok)
    for (int64_t r = rl; r < rows; r += 4) {
      float g = dy[r * lddy + c] * sc;
      if (relu_out16 && !((float)relu_out16[r * ldr + c] > 0.f)) g = 0.f;
      s1 += g;
      s2 += g * ((x[r * ldx + c] - mean) * rstd);
    }
  red[0][rl][cl] = s1;
  red[1][rl][cl] = s2;
  __syncthreads();
  const float sum_dy = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
  const float sum_dyx = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
  if (!ok) return;
  if (rl == 0) {
    dgamma[c] = sum_dyx;
    dbeta[c] = sum_dy;
  }
  const float k = gamma[c] * rstd / (float)rows;
  for (int64_t r = rl; r < rows; r += 4) {
    float g = dy[r * lddy + c] * sc;
    if (relu_out16 && !((float)relu_out16[r * ldr + c] > 0.f)) g = 0.f;
    const float xh = (x[r * ldx + c] - mean) * rstd;
    dx16[r * lddx + c] = (_Float16)(k * ((float)rows * g - sum_dy - xh * sum_dyx));
  }
}

}  // namespace

extern "C" int hcir_positive_transform(const float* images, int64_t b, int32_t c, int32_t h, int32_t w,
                                       const float* theta4, const float* taps2, float* out, void* stream) {
  HCIR_ENTER();
  if (!images || !out || !theta4 || !taps2 || b <= 0 || c <= 0 || h < 2 || w < 2 || images == out) return HCIR_ERR_INVALID;
  PosTransformArgs a{};
  a.in = images;
  a.out = out;
  a.planes = b * c;
  a.h = h;
  a.w = w;
  // torchvision: rescaled_theta = theta^T / [0.5 w, 0.5 h]  (columns of the grid: x then y)
  a.t00 = theta4[0] / (0.5f * (float)w);
  a.t01 = theta4[1] / (0.5f * (float)w);
  a.t10 = theta4[2] / (0.5f * (float)h);
  a.t11 = theta4[3] / (0.5f * (float)h);
  a.k0 = taps2[0];
  a.k1 = taps2[1];
  const int64_t n = a.planes * (int64_t)h * w;
  hipLaunchKernelGGL(positive_transform_kernel, dim3((unsigned)hcir_cdiv(n, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a);
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}

extern "C" int hcir_bn1d_fwd(const float* x, int64_t ldx, int64_t rows, int32_t f, const float* gamma, const float* beta,
                             float eps, float momentum, int relu, float* running_mean, float* running_var,
                             float* save_mean, float* save_rstd, float* y_f32, int64_t ldy32, void* y_f16, int64_t ldy16,
                             void* stream) {
  HCIR_ENTER();
  if (!x || !gamma || !beta || !save_mean || !save_rstd || (!y_f32 && !y_f16) || rows <= 0 || f <= 0) return HCIR_ERR_INVALID;
  hipLaunchKernelGGL(bn1d_fwd_kernel, dim3((unsigned)hcir_cdiv(f, kBnCols)), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                     ldx, rows, f, gamma, beta, eps, momentum, relu, running_mean, running_var, save_mean, save_rstd, y_f32,
                     ldy32, static_cast<_Float16*>(y_f16), ldy16);
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}

extern "C" int hcir_bn1d_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t rows, int32_t f,
                             const float* gamma, const float* save_mean, const float* save_rstd, const void* relu_out_f16,
                             int64_t ldr, const float* dy_scale, void* dx_f16, int64_t lddx, float* dgamma, float* dbeta,
                             void* stream) {
  HCIR_ENTER();
  if (!dy || !x || !gamma || !save_mean || !save_rstd || !dx_f16 || !dgamma || !dbeta || rows <= 0 || f <= 0)
    return HCIR_ERR_INVALID;
  hipLaunchKernelGGL(bn1d_bwd_kernel, dim3((unsigned)hcir_cdiv(f, kBnCols)), dim3(256), 0, static_cast<hipStream_t>(stream), dy,
                     lddy, x, ldx, rows, f, gamma, save_mean, save_rstd, static_cast<const _Float16*>(relu_out_f16), ldr,
                     dy_scale, static_cast<_Float16*>(dx_f16), lddx, dgamma, dbeta);
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}
