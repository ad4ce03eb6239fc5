// train.hip — the element-wise / row-wise kernels of the ViT backward and the two small losses of the HSimCLR
// step (HP/src/pretrain_engine.py:681-751).
//
//   hcir_gelu_fwd_f16 / hcir_gelu_bwd_f16   nn.GELU() of the MLP block and its derivative (act.h)
//   hcir_layernorm_bwd                       nn.LayerNorm backward: dx (added to the residual gradient), dgamma, dbeta
//   hcir_colsum_f16                          bias gradients: column sums of an fp16 matrix
//   hcir_add_f32_f16                         fp32 residual gradient -> fp16 GEMM operand (optionally + another fp32)
//   hcir_triplet_margin_fwd / _bwd           nn.TripletMarginLoss(margin, p=2, eps) (:96-97,717-721)
//   hcir_mse_fwd / _bwd                      F.mse_loss(reduction='mean') (:730)
// Column reductions (dgamma, dbeta, bias gradients) are two-stage and deterministic: per-workgroup partial rows in
// a caller-provided workspace, then a fixed-order sum.
#include "act.h"

namespace {

// ---------------------------------------------------------------- GELU
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const _Float16* __restrict__ u, int64_t n,
                                                       _Float16* __restrict__ h) {
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 8;
  for (; i < n; i += stride) {  // n % 8 == 0
    const f16x8 v = *reinterpret_cast<const f16x8*>(u + i);
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      const gelu_f32x2 y = gelu_erf2((gelu_f32x2){(float)v[e], (float)v[e + 1]});
      o[e] = (_Float16)y[0];
      o[e + 1] = (_Float16)y[1];
    }
    *reinterpret_cast<f16x8*>(h + i) = o;
  }
}

__global__ __launch_bounds__(256) void gelu_bwd_kernel(const _Float16* __restrict__ u, const _Float16* __restrict__ dh,
                                                       int64_t n, _Float16* __restrict__ du) {
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 8;
  for (; i < n; i += stride) {
    const f16x8 v = *reinterpret_cast<const f16x8*>(u + i);
    const f16x8 g = *reinterpret_cast<const f16x8*>(dh + i);
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (_Float16)((float)g[e] * gelu_erf_grad((float)v[e]));
    *reinterpret_cast<f16x8*>(du + i) = o;
  }
}

// ---------------------------------------------------------------- fp32 (+ fp32) -> fp16
__global__ __launch_bounds__(256) void add_f32_f16_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          int64_t n, _Float16* __restrict__ y) {
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  for (; i < n; i += stride) {  // n % 4 == 0
    f32x4 v = *reinterpret_cast<const f32x4*>(a + i);
    if (b) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(b + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += w[e];
    }
    f16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (_Float16)v[e];
    *reinterpret_cast<f16x4*>(y + i) = o;
  }
}

// ---------------------------------------------------------------- LayerNorm backward
// One wave per row, lanes own columns 4 (lane + 64 j) + e (d % 4 == 0, d <= 4096).  Per row:
//   xhat = (x - mean) rstd,  g = dy o gamma,
//   dx = rstd (g - mean(g) - xhat mean(g o xhat))           [biased variance, as nn.LayerNorm]
//   dres_out[row] = (dres_in ? dres_in[row] : 0) + dx       fp32
// and the workgroup's partial sums  dgamma_part[wg][c] = sum_rows dy xhat,  dbeta_part[wg][c] = sum_rows dy.
constexpr int kLnMaxJ = 16;  // d <= 4 * 64 * 16 = 4096

template <typename XT, int NJ>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const XT* __restrict__ x, int64_t ldx,
                                                            const _Float16* __restrict__ dy, int64_t lddy,
                                                            const float* __restrict__ gamma, float eps, int64_t rows,
                                                            int d, const float* __restrict__ dres_in,
                                                            float* __restrict__ dres_out, int64_t ldr,
                                                            float* __restrict__ dgamma_part,
                                                            float* __restrict__ dbeta_part,
                                                            _Float16* __restrict__ dres16, int64_t ldh,
                                                            float* __restrict__ dres_part) {
  __shared__ float red[4][2];
  extern __shared__ float colacc[];  // [2 or 3][d] per workgroup, combined across the 4 waves at the end
  // NJ = column groups of 256 per row, a compile-time bound: with a run-time bound the per-lane arrays (4 x 16 x 4
  // floats) went to scratch and the kernel ran 5x off its HBM time
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int nj = NJ;
  // optional third product (hcir_layernorm_bwd_fused): the fp16 copy of dres_out - the next dgrad / wgrad GEMM's
  // operand - and the column sums of that copy - the bias gradient of the Linear layer in front
  float gsum[NJ][4], bsum[NJ][4], rsum[NJ][4];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) gsum[j][e] = bsum[j][e] = rsum[j][e] = 0.f;
  (void)red;

  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    const XT* xr = x + row * ldx;
    const _Float16* dr = dy + row * lddy;
    float xv[NJ][4], dv[NJ][4];
    f32x4 rin[NJ];   // the residual gradient of the row, requested with x and dy (behind the three reductions its
                     // HBM latency stood exposed once per row)
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (j < nj) {
        const int c = 4 * (lane + 64 * j);
        rin[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (c < d) {
          if (dres_in) rin[j] = *reinterpret_cast<const f32x4*>(dres_in + row * ldr + c);
          if constexpr (sizeof(XT) == 2) {
            const f16x4 xq = *reinterpret_cast<const f16x4*>(xr + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) xv[j][e] = (float)xq[e];
          } else {
            const f32x4 xq = *reinterpret_cast<const f32x4*>(xr + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) xv[j][e] = xq[e];
          }
          const f16x4 dq = *reinterpret_cast<const f16x4*>(dr + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            dv[j][e] = (float)dq[e];
            s += xv[j][e];
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) xv[j][e] = dv[j][e] = 0.f;
        }
      }
    }
    const float mean = wave_sum(s) / (float)d;
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (j < nj) {
        const int c = 4 * (lane + 64 * j);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = c < d ? xv[j][e] - mean : 0.f;
          v = __builtin_fmaf(t, t, v);
        }
      }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)d + eps);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (j < nj) {
        const int c = 4 * (lane + 64 * j);
        if (c < d) {
          const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xh = (xv[j][e] - mean) * rstd;
            const float g = dv[j][e] * gm[e];
            gsum[j][e] = __builtin_fmaf(dv[j][e], xh, gsum[j][e]);
            bsum[j][e] += dv[j][e];
            xv[j][e] = xh;   // keep xhat
            dv[j][e] = g;    // keep g
            sg += g;
            sgx = __builtin_fmaf(g, xh, sgx);
          }
        }
      }
    }
    const float mg = wave_sum(sg) / (float)d, mgx = wave_sum(sgx) / (float)d;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (j < nj) {
        const int c = 4 * (lane + 64 * j);
        if (c < d) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = rstd * (dv[j][e] - mg - xv[j][e] * mgx);
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] += rin[j][e];
          *reinterpret_cast<f32x4*>(dres_out + row * ldr + c) = o;
          if (dres16) {
            f16x4 oh;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              oh[e] = (_Float16)o[e];
              rsum[j][e] += (float)oh[e];
            }
            *reinterpret_cast<f16x4*>(dres16 + row * ldh + c) = oh;
          }
        }
      }
    }
  }
  // combine the 4 waves' column sums through LDS in wave order (deterministic), one partial row per workgroup
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if (j < nj) {
          const int c = 4 * (lane + 64 * j);
          if (c < d) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              colacc[c + e] = (w ? colacc[c + e] : 0.f) + gsum[j][e];
              colacc[d + c + e] = (w ? colacc[d + c + e] : 0.f) + bsum[j][e];
              if (dres_part) colacc[2 * d + c + e] = (w ? colacc[2 * d + c + e] : 0.f) + rsum[j][e];
            }
          }
        }
      }
    }
    __syncthreads();
  }
  for (int c = threadIdx.x; c < d; c += 256) {
    dgamma_part[(int64_t)blockIdx.x * d + c] = colacc[c];
    dbeta_part[(int64_t)blockIdx.x * d + c] = colacc[d + c];
    if (dres_part) dres_part[(int64_t)blockIdx.x * d + c] = colacc[2 * d + c];
  }
}

// out[c] (+)= sum_p part[p][c]: workgroup = 16 columns x 16 part lanes (lane q sums parts q, q + 16, ... in order,
// the sixteen sums are then added in lane order: a fixed tree, deterministic); blockIdx.y selects one of up to three
// independent (part, out) pairs laid out back to back (LayerNorm: dgamma, dbeta and the bias gradient of the Linear
// in front in ONE launch); bit y of acc_mask = accumulate into out y.  History: a one-thread-per-column loop over
// ~800 partial rows took 105 us per call, 64 columns x 4 part lanes 30 us (2.7 ms of a training step over ~90 calls).
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ part, int parts, int n,
                                                              float* __restrict__ out0, float* __restrict__ out1,
                                                              float* __restrict__ out2, int acc_mask) {
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  const float* src = part + (int64_t)blockIdx.y * parts * n;
  float* out = blockIdx.y == 0 ? out0 : (blockIdx.y == 1 ? out1 : out2);
  float s = 0.f;
  if (c < n)
    for (int p = pl; p < parts; p += 16) s += src[(int64_t)p * n + c];
  red[pl][cl] = s;
  __syncthreads();
  if (pl == 0 && c < n) {
    float t = red[0][cl];
#pragma unroll
    for (int q = 1; q < 16; ++q) t += red[q][cl];
    out[c] = ((acc_mask >> blockIdx.y) & 1) ? out[c] + t : t;
  }
}

// ---------------------------------------------------------------- column sums of an fp16 matrix (bias gradients)
// workgroup = 256 threads = 32 column groups of 8 (16 B) x 8 row lanes; grid (col blocks of 256, row chunks)
__global__ __launch_bounds__(256) void colsum_kernel(const _Float16* __restrict__ x, int64_t m, int n, int64_t ldx,
                                                     int64_t rows_per, float* __restrict__ part) {
  __shared__ float red[8][256];
  const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c0 = blockIdx.x * 256 + cg * 8;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per, r1 = r0 + rows_per < m ? r0 + rows_per : m;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  if (c0 < n) {
    for (int64_t r = r0 + rl; r < r1; r += 8) {
      const f16x8 v = *reinterpret_cast<const f16x8*>(x + r * ldx + c0);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rl][cg * 8 + e] = acc[e];
  __syncthreads();
  const int c = threadIdx.x;
  if (blockIdx.x * 256 + c < n) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) s += red[w][c];
    part[(int64_t)blockIdx.y * n + blockIdx.x * 256 + c] = s;
  }
}

// ---------------------------------------------------------------- GELU backward + the bias gradient's column sums
// du = dh o gelu'(u) and part[chunk][c] = sum over the chunk's rows of the STORED fp16 du: colsum_kernel's geometry
// and summation order with the elementwise pass inside (the bias gradient of fc1 costs no pass of its own, and it
// is bit-identical to hcir_gelu_bwd_f16 followed by hcir_colsum_f16).
__global__ __launch_bounds__(256) void gelu_bwd_colsum_kernel(const _Float16* __restrict__ u,
                                                              const _Float16* __restrict__ dh, int64_t m, int n,
                                                              int64_t ld, int64_t rows_per, _Float16* __restrict__ du,
                                                              float* __restrict__ part) {
  __shared__ float red[8][256];
  const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c0 = blockIdx.x * 256 + cg * 8;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per, r1 = r0 + rows_per < m ? r0 + rows_per : m;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  if (c0 < n) {
    for (int64_t r = r0 + rl; r < r1; r += 8) {
      const f16x8 v = *reinterpret_cast<const f16x8*>(u + r * ld + c0);
      const f16x8 g = *reinterpret_cast<const f16x8*>(dh + r * ld + c0);
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        o[e] = (_Float16)((float)g[e] * gelu_erf_grad((float)v[e]));
        acc[e] += (float)o[e];
      }
      *reinterpret_cast<f16x8*>(du + r * ld + c0) = o;
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rl][cg * 8 + e] = acc[e];
  __syncthreads();
  const int c = threadIdx.x;
  if (blockIdx.x * 256 + c < n) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) s += red[w][c];
    part[(int64_t)blockIdx.y * n + blockIdx.x * 256 + c] = s;
  }
}

// ---------------------------------------------------------------- TripletMarginLoss(margin, p=2, eps), mean
// torch: d(x, y) = || x - y + eps ||_2 (pairwise_distance adds eps to the difference),
//        loss = mean_i max(d(a_i, p_i) - d(a_i, n_i) + margin, 0)
// one wave per row; row losses and both distances are kept for the backward
__global__ __launch_bounds__(256) void triplet_fwd_kernel(const float* __restrict__ a, const float* __restrict__ p,
                                                          const float* __restrict__ n, int64_t b, int d, float margin,
                                                          float eps, float* __restrict__ row_loss,
                                                          float* __restrict__ dist /*[2][b]*/) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= b) return;
  float sp = 0.f, sn = 0.f;
  for (int c = lane; c < d; c += 64) {
    const float av = a[row * d + c];
    const float dp = av - p[row * d + c] + eps, dn = av - n[row * d + c] + eps;
    sp = __builtin_fmaf(dp, dp, sp);
    sn = __builtin_fmaf(dn, dn, sn);
  }
  sp = sqrtf(wave_sum(sp));
  sn = sqrtf(wave_sum(sn));
  if (lane == 0) {
    row_loss[row] = fmaxf(sp - sn + margin, 0.f);
    dist[row] = sp;
    dist[b + row] = sn;
  }
}

// dL/da = gscale * active * ((a - p + eps)/dp - (a - n + eps)/dn), dL/dp = -gscale active (a - p + eps)/dp,
// dL/dn = +gscale active (a - n + eps)/dn, gscale = grad_out / b, active = row_loss > 0
__global__ __launch_bounds__(256) void triplet_bwd_kernel(const float* __restrict__ a, const float* __restrict__ p,
                                                          const float* __restrict__ n, int64_t b, int d, float eps,
                                                          const float* __restrict__ row_loss,
                                                          const float* __restrict__ dist, const float* grad_out,
                                                          float* __restrict__ da, float* __restrict__ dp_,
                                                          float* __restrict__ dn_) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b * d) return;
  const int64_t row = i / d;
  const float gs = row_loss[row] > 0.f ? grad_out[0] / (float)b : 0.f;
  const float up = (a[i] - p[i] + eps) / dist[row], un = (a[i] - n[i] + eps) / dist[b + row];
  da[i] = gs * (up - un);
  dp_[i] = -gs * up;
  dn_[i] = gs * un;
}

// fixed-tree mean of a vector (one workgroup): deterministic
__global__ __launch_bounds__(256) void mean_kernel(const float* __restrict__ x, int64_t n, float scale,
                                                   float* __restrict__ out) {
  __shared__ float red[256];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 256) s += x[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0] * scale;
}

// ---------------------------------------------------------------- mse_loss(x, y, 'mean'): partial sums per workgroup
__global__ __launch_bounds__(256) void mse_partial_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                          int64_t n, float* __restrict__ part) {
  __shared__ float red[256];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float dlt = x[i] - y[i];
    s = __builtin_fmaf(dlt, dlt, s);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ void mse_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, int64_t n,
                               const float* grad_out, float* __restrict__ dx, float* __restrict__ dy) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float g = 2.0f * (x[i] - y[i]) * grad_out[0] / (float)n;
  if (dx) dx[i] = g;
  if (dy) dy[i] = -g;
}

}  // namespace

extern "C" {

int hcir_gelu_fwd_f16(const void* u, int64_t n, void* h, void* stream) {
  HCIR_ENTER();
  if (!u || !h || n <= 0 || (n & 7)) return HCIR_ERR_INVALID;
  int64_t blocks = hcir_cdiv(n, 2048);
  blocks = blocks > 8192 ? 8192 : blocks;
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const _Float16*>(u), n, static_cast<_Float16*>(h));
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}

int hcir_gelu_bwd_f16(const void* u, const void* dh, int64_t n, void* du, void* stream) {
  HCIR_ENTER();
  if (!u || !dh || !du || n <= 0 || (n & 7)) return HCIR_ERR_INVALID;
  int64_t blocks = hcir_cdiv(n, 2048);
  blocks = blocks > 8192 ? 8192 : blocks;
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const _Float16*>(u), static_cast<const _Float16*>(dh), n,
                     static_cast<_Float16*>(du));
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}

int hcir_add_f32_f16(const float* a, const float* b, int64_t n, void* y, void* stream) {
  HCIR_ENTER();
  if (!a || !y || n <= 0 || (n & 3)) return HCIR_ERR_INVALID;
  int64_t blocks = hcir_cdiv(n, 1024);
  blocks = blocks > 8192 ? 8192 : blocks;
  hipLaunchKernelGGL(add_f32_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a, b,
                     n, static_cast<_Float16*>(y));
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}

int32_t hcir_layernorm_bwd_blocks(int64_t rows) {
  int64_t b = hcir_cdiv(rows, 4 * 16);  // >= 16 rows per wave; 1024 workgroups = 4 waves per SIMD in flight
  return (int32_t)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

int hcir_layernorm_bwd(const void* x, int x_dtype, int64_t rows, int32_t d, int64_t ldx, const void* dy_f16,
                       int64_t lddy, const float* gamma, float eps, const float* dres_in, float* dres_out,
                       int64_t ldr, float* dgamma, float* dbeta, int accumulate, float* workspace,
                       size_t workspace_bytes, void* stream) {
  return hcir_layernorm_bwd_fused(x, x_dtype, rows, d, ldx, dy_f16, lddy, gamma, eps, dres_in, dres_out, ldr, dgamma,
                                  dbeta, accumulate, nullptr, 0, nullptr, workspace, workspace_bytes, stream);
}

int hcir_layernorm_bwd_fused(const void* x, int x_dtype, int64_t rows, int32_t d, int64_t ldx, const void* dy_f16,
                             int64_t lddy, const float* gamma, float eps, const float* dres_in, float* dres_out,
                             int64_t ldr, float* dgamma, float* dbeta, int accumulate, void* dres_f16, int64_t ldh,
                             float* dres_colsum, float* workspace, size_t workspace_bytes, void* stream) {
  HCIR_ENTER();
  if (!x || !dy_f16 || !gamma || !dres_out || !dgamma || !dbeta || !workspace) return HCIR_ERR_INVALID;
  if (rows <= 0 || d <= 0 || (d & 3) || d > 4 * 64 * kLnMaxJ || ldx < d || lddy < d || ldr < d || (ldr & 3))
    return HCIR_ERR_INVALID;
  if (dres_colsum && !dres_f16) return HCIR_ERR_INVALID;
  if (dres_f16 && (ldh < d || (ldh & 3))) return HCIR_ERR_INVALID;
  if (x_dtype != HCIR_F32 && x_dtype != HCIR_F16) return HCIR_ERR_UNSUPPORTED;
  const int blocks = hcir_layernorm_bwd_blocks(rows);
  const int narr = dres_colsum ? 3 : 2;
  if (workspace_bytes < (size_t)blocks * d * narr * sizeof(float)) return HCIR_ERR_WORKSPACE;
  float* gpart = workspace;
  float* bpart = workspace + (size_t)blocks * d;
  float* rpart = dres_colsum ? workspace + (size_t)2 * blocks * d : nullptr;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t shm = (size_t)narr * d * sizeof(float);
#define LNB(XT, NJV)                                                                                                \
  hipLaunchKernelGGL((layernorm_bwd_kernel<XT, NJV>), dim3(blocks), dim3(256), shm, st, static_cast<const XT*>(x), ldx, \
                     static_cast<const _Float16*>(dy_f16), lddy, gamma, eps, rows, d, dres_in, dres_out, ldr, gpart,  \
                     bpart, static_cast<_Float16*>(dres_f16), ldh, rpart)
#define LNB_NJ(XT)                       \
  do {                                   \
    if (nj <= 1) LNB(XT, 1);             \
    else if (nj <= 2) LNB(XT, 2);        \
    else if (nj <= 3) LNB(XT, 3);        \
    else if (nj <= 4) LNB(XT, 4);        \
    else if (nj <= 8) LNB(XT, 8);        \
    else LNB(XT, 16);                    \
  } while (0)
  const int nj = (d + 255) / 256;
  if (x_dtype == HCIR_F32)
    LNB_NJ(float);
  else
    LNB_NJ(_Float16);
#undef LNB_NJ
#undef LNB
  HCIR_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3((unsigned)hcir_cdiv(d, 16), (unsigned)narr), dim3(256), 0, st, gpart,
                     blocks, d, dgamma, dbeta, dres_colsum, accumulate ? 3 : 0);
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}

int32_t hcir_colsum_chunks(int64_t m) {
  int64_t c = hcir_cdiv(m, 512);
  return (int32_t)(c < 1 ? 1 : (c > 256 ? 256 : c));
}

int hcir_colsum_f16(const void* x, int64_t m, int32_t n, int64_t ldx, float* out, int accumulate, float* workspace,
                    size_t workspace_bytes, void* stream) {
  HCIR_ENTER();
  if (!x || !out || !workspace || m <= 0 || n <= 0 || (n & 7) || ldx < n || (ldx & 7)) return HCIR_ERR_INVALID;
  const int chunks = hcir_colsum_chunks(m);
  if (workspace_bytes < (size_t)chunks * n * sizeof(float)) return HCIR_ERR_WORKSPACE;
  const int64_t rows_per = hcir_cdiv(m, chunks);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)hcir_cdiv(n, 256), (unsigned)chunks), dim3(256), 0, st,
                     static_cast<const _Float16*>(x), m, n, ldx, rows_per, workspace);
  HCIR_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3((unsigned)hcir_cdiv(n, 16), 1), dim3(256), 0, st, workspace, chunks, n,
                     out, out, out, accumulate ? 1 : 0);
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}

int hcir_gelu_bwd_colsum_f16(const void* u, const void* dh, int64_t m, int32_t n, int64_t ld, void* du, float* colsum,
                             int accumulate, float* workspace, size_t workspace_bytes, void* stream) {
  HCIR_ENTER();
  if (!u || !dh || !du || !colsum || !workspace || m <= 0 || n <= 0 || (n & 7) || ld < n || (ld & 7))
    return HCIR_ERR_INVALID;
  const int chunks = hcir_colsum_chunks(m);
  if (workspace_bytes < (size_t)chunks * n * sizeof(float)) return HCIR_ERR_WORKSPACE;
  const int64_t rows_per = hcir_cdiv(m, chunks);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(gelu_bwd_colsum_kernel, dim3((unsigned)hcir_cdiv(n, 256), (unsigned)chunks), dim3(256), 0, st,
                     static_cast<const _Float16*>(u), static_cast<const _Float16*>(dh), m, n, ld, rows_per,
                     static_cast<_Float16*>(du), workspace);
  HCIR_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3((unsigned)hcir_cdiv(n, 16), 1), dim3(256), 0, st, workspace, chunks, n,
                     colsum, colsum, colsum, accumulate ? 1 : 0);
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}

int hcir_triplet_margin_fwd(const float* anchor, const float* positive, const float* negative, int64_t b, int32_t d,
                            float margin, float eps, float* loss, float* row_loss, float* dist, void* stream) {
  HCIR_ENTER();
  if (!anchor || !positive || !negative || !loss || !row_loss || !dist || b <= 0 || d <= 0) return HCIR_ERR_INVALID;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(triplet_fwd_kernel, dim3((unsigned)hcir_cdiv(b, 4)), dim3(256), 0, st, anchor, positive, negative,
                     b, d, margin, eps, row_loss, dist);
  HCIR_LAUNCH_CHECK();
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, st, row_loss, b, 1.0f / (float)b, loss);
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}

int hcir_triplet_margin_bwd(const float* anchor, const float* positive, const float* negative, int64_t b, int32_t d,
                            float eps, const float* row_loss, const float* dist, const float* grad_out,
                            float* d_anchor, float* d_positive, float* d_negative, void* stream) {
  HCIR_ENTER();
  if (!anchor || !positive || !negative || !row_loss || !dist || !grad_out || !d_anchor || !d_positive || !d_negative)
    return HCIR_ERR_INVALID;
  if (b <= 0 || d <= 0) return HCIR_ERR_INVALID;
  hipLaunchKernelGGL(triplet_bwd_kernel, dim3((unsigned)hcir_cdiv(b * d, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), anchor, positive, negative, b, d, eps, row_loss, dist, grad_out,
                     d_anchor, d_positive, d_negative);
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}

int hcir_mse_fwd(const float* x, const float* y, int64_t n, float* loss, float* workspace /*[256]*/, void* stream) {
  HCIR_ENTER();
  if (!x || !y || !loss || !workspace || n <= 0) return HCIR_ERR_INVALID;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int64_t blocks = hcir_cdiv(n, 256 * 16);
  blocks = blocks < 1 ? 1 : (blocks > 256 ? 256 : blocks);
  hipLaunchKernelGGL(mse_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, y, n, workspace);
  HCIR_LAUNCH_CHECK();
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, st, workspace, blocks, 1.0f / (float)n, loss);
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}

int hcir_mse_bwd(const float* x, const float* y, int64_t n, const float* grad_out, float* dx, float* dy,
                 void* stream) {
  HCIR_ENTER();
  if (!x || !y || !grad_out || n <= 0 || (!dx && !dy)) return HCIR_ERR_INVALID;
  hipLaunchKernelGGL(mse_bwd_kernel, dim3((unsigned)hcir_cdiv(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     x, y, n, grad_out, dx, dy);
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}

}  // extern "C"
