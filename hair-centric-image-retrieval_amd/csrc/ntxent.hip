// ntxent.hip — fused NT-Xent forward: normalise -> 2B x 2B cosine / T -> masked row
// log-sum-exp -> mean cross-entropy, without materialising the logits.
//
// Semantics: lightly.loss.NTXentLoss(temperature) as called at
// HP/src/pretrain_engine.py:93,725 (== experiments/DualViewHair/src/losses/ntxent_loss.py:30-57):
//   z = [normalize(z0); normalize(z1)], logits = z z^T / T, diagonal removed,
//   positive of row i is (i + B) mod 2B, loss = mean_i(logsumexp_j logits[i][j] - logits[i][pos]).
//
// Kernels
//   ntxent_prep   : one wave per row: x / max(||x||, 1e-12) (F.normalize eps) into the
//                   workspace, in the compute dtype.
//   ntxent_tiles  : sim_core.h tile engine; workgroup (x = column tile of 128, y = row block of
//                   128); a lane owns one logits ROW and folds its 16 scores per MFMA tile
//                   into an online (max, sum-of-exp2) pair; diagonal skipped, positive captured.
//   ntxent_rows   : one thread per row merges the per-column-tile partials in tile order
//                   (deterministic), writes row_lse and the row's loss term.
//   ntxent_reduce : one workgroup sums the 2B loss terms in a fixed tree -> mean loss.
#include "sim_core.h"

namespace {

constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kLn2 = 0.69314718055994530942f;

template <typename T>
__global__ __launch_bounds__(256) void ntxent_prep(const T* __restrict__ z0, const T* __restrict__ z1,
                                                   int64_t b, int d, T* __restrict__ zn) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= 2 * b) return;
  const T* p = row < b ? z0 + row * d : z1 + (row - b) * d;
  float s = 0.f;
  for (int k = lane; k < d; k += 64) {
    const float v = (float)p[k];
    s = __builtin_fmaf(v, v, s);
  }
  s = wave_sum(s);
  const float inv = 1.0f / fmaxf(sqrtf(s), 1e-12f);
  for (int k = lane; k < d; k += 64) zn[row * d + k] = (T)((float)p[k] * inv);
}

struct NtArgs {
  const void* zn;
  float* pm;   // [nct][2B] running max (log2 domain) per column tile
  float* pl;   // [nct][2B] sum of exp2
  float* pp;   // [nct][2B] positive logit (log2 domain) or -inf
  int64_t n;   // 2B
  int64_t b;
  int d;
  float scale_log2;  // inv_t * log2(e)
  // backward (BWD = true): W[i][j] = p_ij + p_ji - 2*[j == pos(i)], diagonal 0, fp16 [2B][2B]
  const float* lse2;  // [2B] row log-sum-exp in the log2 domain
  _Float16* wmat;
};

template <typename T, bool GLDS, bool BWD = false>
__global__ __launch_bounds__(256, 2) void ntxent_tiles(NtArgs a) {
  using Cfg = SimCfg<T, 2, 2, 2>;  // 128 column rows streamed x 128 logits rows resident
  constexpr int EPS = SimElem<T>::kPerStage;
  __shared__ __attribute__((aligned(16))) char lds[Cfg::LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_c = wave >> 1, wave_r = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const T* zn = static_cast<const T*>(a.zn);
  const int64_t c0 = (int64_t)blockIdx.x * 128, r0 = (int64_t)blockIdx.y * 128;
  const int nkc = (a.d + EPS - 1) / EPS;

  f32x16 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[x][y][i] = 0.f;

  if constexpr (GLDS) {
    sim_stage_glds<T, Cfg>(lds, zn, c0, a.n - 1, zn, r0, a.n - 1, a.d, 0, tid);
    for (int kc = 0; kc < nkc; ++kc) {
      const int cur = kc & 1;
      sim_glds_retire_and_sync();
      if (kc + 1 < nkc)
        sim_stage_glds<T, Cfg>(lds + (cur ^ 1) * Cfg::STAGE_BYTES, zn, c0, a.n - 1, zn, r0, a.n - 1, a.d,
                               kc + 1, tid);
      sim_stage_mfma<T, Cfg, 2>(acc, lds + cur * Cfg::STAGE_BYTES, wave_c, wave_r, lane);
    }
  } else {
    u32x4 regs[Cfg::NLOAD];
    sim_stage_load<T, Cfg>(regs, zn, c0, a.n - 1, zn, r0, a.n - 1, a.d, 0, tid);
    sim_stage_store<Cfg>(regs, lds, tid);
    __syncthreads();
    for (int kc = 0; kc < nkc; ++kc) {
      const int cur = kc & 1;
      if (kc + 1 < nkc) sim_stage_load<T, Cfg>(regs, zn, c0, a.n - 1, zn, r0, a.n - 1, a.d, kc + 1, tid);
      sim_stage_mfma<T, Cfg, 2>(acc, lds + cur * Cfg::STAGE_BYTES, wave_c, wave_r, lane);
      if (kc + 1 < nkc) sim_stage_store<Cfg>(regs, lds + (cur ^ 1) * Cfg::STAGE_BYTES, tid);
      __syncthreads();
    }
  }
  __syncthreads();  // stage buffers are re-used below for the in-workgroup merge

  const float ninf = -__builtin_huge_valf();
  if constexpr (BWD) {
    // ---- backward tile: the lane owns logits row `row`; its 16 registers per MFMA tile are 4 groups
    //      of 4 consecutive columns -> 8-byte fp16 stores into W[row][col..col+3]
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int64_t row = r0 + wave_r * 64 + rt * 32 + r;
      if (row >= a.n) continue;
      const int64_t pos = row < a.b ? row + a.b : row - a.b;
      const float lrow = a.lse2[row];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
          const int64_t col0 = c0 + wave_c * 64 + ct * 32 + 8 * grp + 4 * h;
          if (col0 >= a.n) continue;  // n is a multiple of 8: a group of 4 is inside or outside
          const f32x4 lcol = *reinterpret_cast<const f32x4*>(a.lse2 + col0);
          f16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int64_t col = col0 + e;
            const float x = acc[ct][rt][4 * grp + e] * a.scale_log2;
            float w = __builtin_amdgcn_exp2f(x - lrow) + __builtin_amdgcn_exp2f(x - lcol[e]);
            w = (col == pos) ? w - 2.0f : w;
            w = (col == row) ? 0.f : w;
            o[e] = (_Float16)w;
          }
          *reinterpret_cast<f16x4*>(a.wmat + row * a.n + col0) = o;
        }
      }
    }
    return;
  }
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const int64_t row = r0 + wave_r * 64 + rt * 32 + r;
    const int64_t pos = row < a.b ? row + a.b : row - a.b;
    float m = ninf, l = 0.f, pv = ninf;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      float x[16];
      float tm = ninf;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int64_t col = c0 + wave_c * 64 + ct * 32 + acc_row(i, h);
        const float v = acc[ct][rt][i] * a.scale_log2;
        const bool live = col < a.n && col != row;
        x[i] = live ? v : ninf;
        pv = (col == pos) ? v : pv;
        tm = fmaxf(tm, x[i]);
      }
      if (tm > ninf) {
        const float mn = fmaxf(m, tm);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) sum += __builtin_amdgcn_exp2f(x[i] - mn);
        l = l * __builtin_amdgcn_exp2f(m - mn) + sum;
        m = mn;
      }
    }
    // the four (wave_c, lane-half) partials of a row meet in LDS: [128 rows][4 src][m, l, p]
    {
      float* red = reinterpret_cast<float*>(lds);
      const int rl = wave_r * 64 + rt * 32 + r, src = wave_c * 2 + h;
      red[(rl * 4 + src) * 3 + 0] = m;
      red[(rl * 4 + src) * 3 + 1] = l;
      red[(rl * 4 + src) * 3 + 2] = pv;
    }
  }
  __syncthreads();
  if (tid < 128 && r0 + tid < a.n) {
    const float* red = reinterpret_cast<const float*>(lds) + tid * 12;
    float m = ninf, l = 0.f, pv = ninf;
#pragma unroll
    for (int s = 0; s < 4; ++s) {  // fixed order: deterministic
      const float mp = red[s * 3], lp = red[s * 3 + 1];
      pv = fmaxf(pv, red[s * 3 + 2]);
      if (lp > 0.f) {
        const float mn = fmaxf(m, mp);
        l = l * __builtin_amdgcn_exp2f(m - mn) + lp * __builtin_amdgcn_exp2f(mp - mn);
        m = mn;
      }
    }
    const int64_t o = (int64_t)blockIdx.x * a.n + r0 + tid;
    a.pm[o] = m;
    a.pl[o] = l;
    a.pp[o] = pv;
  }
}

// one thread per logits row: merge the per-column-tile partials in tile order, write
// row_lse and the row's loss term (lse - positive logit)
__global__ __launch_bounds__(256) void ntxent_rows(const float* __restrict__ pm,
                                                   const float* __restrict__ pl,
                                                   const float* __restrict__ pp, int64_t n, int nparts,
                                                   float* __restrict__ row_loss,
                                                   float* __restrict__ row_lse) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= n) return;
  float m = -__builtin_huge_valf(), l = 0.f, pv = -__builtin_huge_valf();
  for (int p = 0; p < nparts; ++p) {
    const float mp = pm[(int64_t)p * n + row], lp = pl[(int64_t)p * n + row];
    pv = fmaxf(pv, pp[(int64_t)p * n + row]);
    if (lp > 0.f) {
      const float mn = fmaxf(m, mp);
      l = l * exp2f(m - mn) + lp * exp2f(mp - mn);
      m = mn;
    }
  }
  const float lse2 = m + log2f(l);  // log2 domain
  if (row_lse) row_lse[row] = lse2 * kLn2;
  row_loss[row] = (lse2 - pv) * kLn2;
}

__global__ __launch_bounds__(1024) void ntxent_reduce(const float* __restrict__ row_loss, int64_t n,
                                                      float* __restrict__ loss) {
  __shared__ float red[1024];
  float local = 0.f;
  for (int64_t row = threadIdx.x; row < n; row += 1024) local += row_loss[row];  // fixed order
  red[threadIdx.x] = local;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) *loss = red[0] / (float)n;
}

// zt[k][i] = zn[i][k] as fp16 (the "weight" operand of dU = W . U on hcir_gemm_f16), and lse -> log2 domain
template <typename T>
__global__ void ntxent_bwd_prep(const T* __restrict__ zn, const float* __restrict__ row_lse, int64_t n, int d,
                                _Float16* __restrict__ zt, float* __restrict__ lse2) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) lse2[t] = row_lse[t] * kLog2e;
  if (t < n * d) {
    const int64_t i = t / d;
    const int k = (int)(t - i * d);
    zt[(int64_t)k * n + i] = (_Float16)(float)zn[t];
  }
}

// dx_i = coef * rn_i * (g_i - (g_i . u_i) u_i), one wave per row; g = W.U (fp32), u = normalised row
template <typename T>
__global__ __launch_bounds__(256) void ntxent_bwd_finish(const float* __restrict__ g, const T* __restrict__ zn,
                                                         const T* __restrict__ z0, const T* __restrict__ z1,
                                                         int64_t b, int d, float coef, T* __restrict__ dz0,
                                                         T* __restrict__ dz1) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= 2 * b) return;
  const T* x = row < b ? z0 + row * d : z1 + (row - b) * d;
  T* dx = row < b ? dz0 + row * d : dz1 + (row - b) * d;
  float ss = 0.f, gu = 0.f;
  for (int k = lane; k < d; k += 64) {
    const float xv = (float)x[k];
    ss = __builtin_fmaf(xv, xv, ss);
    gu = __builtin_fmaf(g[row * d + k], (float)zn[row * d + k], gu);
  }
  ss = wave_sum(ss);
  gu = wave_sum(gu);
  const float rn = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
  for (int k = lane; k < d; k += 64)
    dx[k] = (T)(coef * rn * (g[row * d + k] - gu * (float)zn[row * d + k]));
}

struct NtWorkspace {
  void* zn;
  float *pm, *pl, *pp, *row_loss;
  // backward only
  _Float16 *wmat, *zt;
  float *lse2, *g;
  size_t bytes;
};

NtWorkspace nt_carve(void* base, int64_t b, int d, int dtype, bool bwd = false) {
  NtWorkspace w;
  char* c = static_cast<char*>(base);
  size_t off = 0;
  auto take = [&](size_t n) {
    char* r = c ? c + off : nullptr;
    off += (n + 255) & ~size_t(255);
    return r;
  };
  const int64_t n = 2 * b;
  const int64_t nct = hcir_cdiv(n, 128);
  w.zn = take((size_t)n * d * (dtype == HCIR_F32 ? 4 : 2));
  w.pm = reinterpret_cast<float*>(take((size_t)nct * n * 4));
  w.pl = reinterpret_cast<float*>(take((size_t)nct * n * 4));
  w.pp = reinterpret_cast<float*>(take((size_t)nct * n * 4));
  w.row_loss = reinterpret_cast<float*>(take((size_t)n * 4));
  w.wmat = nullptr;
  w.zt = nullptr;
  w.lse2 = w.g = nullptr;
  if (bwd) {
    w.wmat = reinterpret_cast<_Float16*>(take((size_t)n * n * 2));
    w.zt = reinterpret_cast<_Float16*>(take((size_t)n * d * 2));
    w.lse2 = reinterpret_cast<float*>(take((size_t)n * 4));
    w.g = reinterpret_cast<float*>(take((size_t)n * d * 4));
  }
  w.bytes = off;
  return w;
}

template <typename T>
int nt_run(const void* z0, const void* z1, int64_t b, int d, float inv_t, float* loss, float* row_lse,
           const NtWorkspace& w, hipStream_t st) {
  const int64_t n = 2 * b;
  const int nct = (int)hcir_cdiv(n, 128);
  hipLaunchKernelGGL(ntxent_prep<T>, dim3((unsigned)hcir_cdiv(n, 4)), dim3(256), 0, st,
                     static_cast<const T*>(z0), static_cast<const T*>(z1), b, d, static_cast<T*>(w.zn));
  HCIR_LAUNCH_CHECK();
  NtArgs a{w.zn, w.pm, w.pl, w.pp, n, b, d, inv_t * kLog2e, nullptr, nullptr};
  if (d % SimElem<T>::kPerStage == 0)
    hipLaunchKernelGGL((ntxent_tiles<T, true>), dim3(nct, nct), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((ntxent_tiles<T, false>), dim3(nct, nct), dim3(256), 0, st, a);
  HCIR_LAUNCH_CHECK();
  hipLaunchKernelGGL(ntxent_rows, dim3((unsigned)hcir_cdiv(n, 256)), dim3(256), 0, st, w.pm, w.pl, w.pp, n,
                     nct, w.row_loss, row_lse);
  HCIR_LAUNCH_CHECK();
  hipLaunchKernelGGL(ntxent_reduce, dim3(1), dim3(1024), 0, st, w.row_loss, n, loss);
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}

template <typename T>
int nt_run_bwd(const void* z0, const void* z1, int64_t b, int d, float inv_t, const float* row_lse,
               float grad_out, void* dz0, void* dz1, const NtWorkspace& w, hipStream_t st) {
  const int64_t n = 2 * b;
  const int nct = (int)hcir_cdiv(n, 128);
  hipLaunchKernelGGL(ntxent_prep<T>, dim3((unsigned)hcir_cdiv(n, 4)), dim3(256), 0, st,
                     static_cast<const T*>(z0), static_cast<const T*>(z1), b, d, static_cast<T*>(w.zn));
  HCIR_LAUNCH_CHECK();
  hipLaunchKernelGGL(ntxent_bwd_prep<T>, dim3((unsigned)hcir_cdiv(n * d, 256)), dim3(256), 0, st,
                     static_cast<const T*>(w.zn), row_lse, n, d, w.zt, w.lse2);
  HCIR_LAUNCH_CHECK();
  NtArgs a{w.zn, w.pm, w.pl, w.pp, n, b, d, inv_t * kLog2e, w.lse2, w.wmat};
  if (d % SimElem<T>::kPerStage == 0)
    hipLaunchKernelGGL((ntxent_tiles<T, true, true>), dim3(nct, nct), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((ntxent_tiles<T, false, true>), dim3(nct, nct), dim3(256), 0, st, a);
  HCIR_LAUNCH_CHECK();
  // g[2B][D] = W[2B][2B] . U[2B][D]  ==  hcir_gemm_f16(A = W, "weights" = U^T [D][2B])
  const int rc = hcir_gemm_f16(w.wmat, n, w.zt, n, nullptr, nullptr, n, d, (int32_t)n, HCIR_EPI_BIAS_F32,
                               w.g, d, st);
  if (rc != HCIR_OK) return rc;
  hipLaunchKernelGGL(ntxent_bwd_finish<T>, dim3((unsigned)hcir_cdiv(n, 4)), dim3(256), 0, st, w.g,
                     static_cast<const T*>(w.zn), static_cast<const T*>(z0), static_cast<const T*>(z1), b, d,
                     grad_out * inv_t / (float)n, static_cast<T*>(dz0), static_cast<T*>(dz1));
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}

}  // namespace

extern "C" {

size_t hcir_ntxent_bwd_workspace_bytes(int64_t b, int32_t d, int dtype) {
  if (b <= 0 || d <= 0) return 0;
  return nt_carve(nullptr, b, d, dtype, true).bytes;
}

int hcir_ntxent_bwd(const void* z0, const void* z1, int64_t b, int32_t d, int dtype, float inv_t,
                    const float* row_lse, float grad_out, void* dz0, void* dz1, void* workspace,
                    size_t workspace_bytes, void* stream) {
  HCIR_ENTER();
  if (!z0 || !z1 || !row_lse || !dz0 || !dz1 || b <= 0 || d <= 0 || (d & 7) || (b & 3)) return HCIR_ERR_INVALID;
  if (dtype != HCIR_F32 && dtype != HCIR_F16 && dtype != HCIR_BF16) return HCIR_ERR_UNSUPPORTED;
  const NtWorkspace w = nt_carve(workspace, b, d, dtype, true);
  if (!workspace || workspace_bytes < w.bytes) return HCIR_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == HCIR_F32) return nt_run_bwd<float>(z0, z1, b, d, inv_t, row_lse, grad_out, dz0, dz1, w, st);
  if (dtype == HCIR_F16) return nt_run_bwd<_Float16>(z0, z1, b, d, inv_t, row_lse, grad_out, dz0, dz1, w, st);
  return nt_run_bwd<__bf16>(z0, z1, b, d, inv_t, row_lse, grad_out, dz0, dz1, w, st);
}

size_t hcir_ntxent_workspace_bytes(int64_t b, int32_t d, int dtype) {
  if (b <= 0 || d <= 0) return 0;
  return nt_carve(nullptr, b, d, dtype).bytes;
}

int hcir_ntxent_fwd(const void* z0, const void* z1, int64_t b, int32_t d, int dtype, float inv_t,
                    float* loss, float* row_lse, void* workspace, size_t workspace_bytes,
                    void* stream) {
  HCIR_ENTER();
  if (!z0 || !z1 || !loss || b <= 0 || d <= 0 || (d & 7)) return HCIR_ERR_INVALID;
  if (!(inv_t == inv_t) || inv_t == 0.f) return HCIR_ERR_INVALID;
  if (dtype != HCIR_F32 && dtype != HCIR_F16 && dtype != HCIR_BF16) return HCIR_ERR_UNSUPPORTED;
  const NtWorkspace w = nt_carve(workspace, b, d, dtype);
  if (!workspace || workspace_bytes < w.bytes) return HCIR_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == HCIR_F32) return nt_run<float>(z0, z1, b, d, inv_t, loss, row_lse, w, st);
  if (dtype == HCIR_F16) return nt_run<_Float16>(z0, z1, b, d, inv_t, loss, row_lse, w, st);
  return nt_run<__bf16>(z0, z1, b, d, inv_t, loss, row_lse, w, st);
}

}  // extern "C"
