// attn_cls.hip — attention of the CLASS-TOKEN query only, forward (with log-sum-exp) and backward, for the last
// block of the TRAINING pass: only the class token feeds the loss (cls_token = x[:, 0], HP/src/main_backbone.py:625-627;
// nn.MultiheadAttention inside torchvision's EncoderBlock, :554), so in the last block the other 196 queries — and
// the proj / MLP of their rows — are dead work in the forward AND in the backward.  The inference engine has skipped
// them since round 1 (hcir_attn_fwd with nq = 1); this is the differentiable form.
//
// One wave per (image, head); T <= 256 keys, head_dim 64.  Two lane roles:
//   lane = key   s_j = <q, k_j>, dP_j = <dO, v_j>: a lane reads its key's 128-byte row and dots it with the broadcast
//                64-vector (v_dot2_f32_f16);
//   lane = dim   o_d = sum_j p_j v_j[d], dq_d, dk_j[d] = dS_j q_d, dv_j[d] = p_j dO_d: rows are read / written as whole
//                128-byte lines, the per-key scalars come from LDS.
// Softmax in the log2 domain exactly as hcir_attn_fwd_lse writes it (lse = log2 sum 2^(s c - m) + m, c = scale log2 e).
// Latency-bound and tiny: 4 T 64 flops per (image, head) forward, 50 KB of K / V read.
#include "common.h"

namespace {

struct AttnClsArgs {
  const _Float16* qkv;  // [B][T][3][H][64]
  int64_t b;
  int t, h;
  float scale;
};

__device__ __forceinline__ float dot64(const f16x8 (&a)[8], const _Float16* row) {
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const f16x8 r = *reinterpret_cast<const f16x8*>(row + 8 * i);
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      const f16x2 x = {a[i][e], a[i][e + 1]}, y = {r[e], r[e + 1]};
      acc = __builtin_amdgcn_fdot2(x, y, acc, false);
    }
  }
  return acc;
}

__global__ __launch_bounds__(256) void attn_cls_fwd_kernel(AttnClsArgs a, _Float16* __restrict__ out, float* __restrict__ lse) {
  __shared__ float p_s[4][256];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t item = (int64_t)blockIdx.x * 4 + wave;  // (image, head)
  if (item >= a.b * a.h) return;
  const int64_t bi = item / a.h;
  const int hd = (int)(item - bi * a.h);
  const int64_t tok = (int64_t)3 * a.h * 64;  // elements per token
  const _Float16* base = a.qkv + bi * a.t * tok + (int64_t)hd * 64;
  f16x8 q[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) q[i] = *reinterpret_cast<const f16x8*>(base + 8 * i);  // token 0, q part
  const float c = a.scale * 1.4426950408889634f;
  float s[4];
  float mx = -INFINITY;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int j = lane + 64 * r;
    s[r] = j < a.t ? dot64(q, base + j * tok + (int64_t)a.h * 64) * c : -INFINITY;
    mx = fmaxf(mx, s[r]);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    s[r] = (lane + 64 * r) < a.t ? exp2f(s[r] - mx) : 0.f;
    sum += s[r];
  }
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int r = 0; r < 4; ++r) p_s[wave][lane + 64 * r] = s[r] * inv;
  if (lane == 0) lse[item] = log2f(sum) + mx;
  __builtin_amdgcn_wave_barrier();
  // o_d = sum_j p_j v_j[d], lane = d
  const _Float16* v = base + (int64_t)2 * a.h * 64 + lane;
  float o = 0.f;
  for (int j = 0; j < a.t; ++j) o += p_s[wave][j] * (float)v[j * tok];
  out[item * 64 + lane] = (_Float16)o;  // [B][H*64] compact
}

__global__ __launch_bounds__(256) void attn_cls_bwd_kernel(AttnClsArgs a, const _Float16* __restrict__ o,
                                                           const _Float16* __restrict__ d_o, const float* __restrict__ lse,
                                                           _Float16* __restrict__ d_qkv) {
  __shared__ float p_s[4][256], ds_s[4][256];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t item = (int64_t)blockIdx.x * 4 + wave;
  if (item >= a.b * a.h) return;
  const int64_t bi = item / a.h;
  const int hd = (int)(item - bi * a.h);
  const int64_t tok = (int64_t)3 * a.h * 64;
  const _Float16* base = a.qkv + bi * a.t * tok + (int64_t)hd * 64;
  _Float16* dbase = d_qkv + bi * a.t * tok + (int64_t)hd * 64;
  f16x8 q[8], g[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    q[i] = *reinterpret_cast<const f16x8*>(base + 8 * i);
    g[i] = *reinterpret_cast<const f16x8*>(d_o + item * 64 + 8 * i);
  }
  // D = sum_d dO_d O_d
  const float dd = wave_sum((float)d_o[item * 64 + lane] * (float)o[item * 64 + lane]);
  const float c = a.scale * 1.4426950408889634f;
  const float l = lse[item];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int j = lane + 64 * r;
    float p = 0.f, ds = 0.f;
    if (j < a.t) {
      p = exp2f(dot64(q, base + j * tok + (int64_t)a.h * 64) * c - l);
      const float dp = dot64(g, base + j * tok + (int64_t)2 * a.h * 64);
      ds = p * (dp - dd);
    }
    p_s[wave][j] = p;
    ds_s[wave][j] = ds;
  }
  __builtin_amdgcn_wave_barrier();
  // lane = dim: dq_d = scale sum_j dS_j k_j[d];  dk_j[d] = scale dS_j q_d;  dv_j[d] = p_j dO_d;  dq of the other
  // tokens is zero (their queries were never formed)
  const float qd = (float)base[lane], gd = (float)d_o[item * 64 + lane];
  const _Float16* k = base + (int64_t)a.h * 64 + lane;
  float dq = 0.f;
  for (int j = 0; j < a.t; ++j) {
    const float ds = ds_s[wave][j], p = p_s[wave][j];
    dq += ds * (float)k[j * tok];
    _Float16* row = dbase + j * tok + lane;
    if (j > 0) row[0] = (_Float16)0.f;
    row[(int64_t)a.h * 64] = (_Float16)(a.scale * ds * qd);
    row[(int64_t)2 * a.h * 64] = (_Float16)(p * gd);
  }
  dbase[lane] = (_Float16)(a.scale * dq);
}

}  // namespace

extern "C" int hcir_attn_cls_fwd_lse(const void* qkv, int64_t b, int32_t t, int32_t h, int32_t hd, float scale, void* out,
                                     float* lse, void* stream) {
  HCIR_ENTER();
  if (!qkv || !out || !lse || b <= 0 || t <= 0 || h <= 0) return HCIR_ERR_INVALID;
  if (hd != 64 || t > 256) return HCIR_ERR_UNSUPPORTED;
  AttnClsArgs a{static_cast<const _Float16*>(qkv), b, t, h, scale};
  hipLaunchKernelGGL(attn_cls_fwd_kernel, dim3((unsigned)hcir_cdiv(b * h, 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     a, static_cast<_Float16*>(out), lse);
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}

extern "C" int hcir_attn_cls_bwd(const void* qkv, const void* out, const void* d_out, const float* lse, int64_t b, int32_t t,
                                 int32_t h, int32_t hd, float scale, void* d_qkv, void* stream) {
  HCIR_ENTER();
  if (!qkv || !out || !d_out || !lse || !d_qkv || b <= 0 || t <= 0 || h <= 0) return HCIR_ERR_INVALID;
  if (hd != 64 || t > 256) return HCIR_ERR_UNSUPPORTED;
  AttnClsArgs a{static_cast<const _Float16*>(qkv), b, t, h, scale};
  hipLaunchKernelGGL(attn_cls_bwd_kernel, dim3((unsigned)hcir_cdiv(b * h, 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     a, static_cast<const _Float16*>(out), static_cast<const _Float16*>(d_out), lse,
                     static_cast<_Float16*>(d_qkv));
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}
