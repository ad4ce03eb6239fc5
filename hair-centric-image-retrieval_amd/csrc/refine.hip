// refine.hip — exact fp32 re-scoring + certification of a candidate list.
//
// Used by the "exact by verification" search (hcir/gallery.py, DESIGN.md §3 "filtered search"):
//   1. hcir_sim_topk on an fp16 MIRROR of the gallery gives kc candidates per query at
//      HBM-streaming speed (scores s16, sorted);
//   2. this kernel re-scores the candidates against the fp32 gallery with the SAME fp32 fmaf
//      chain and k-order as the HCIR_F32 scan (sim_core.h "sim_topk k-order"), so values are
//      bit-identical to hcir_sim_topk(HCIR_F32), ranks them (score desc, index asc), and
//   3. certifies the query:  s16[kc-1] + err[i] < exact k-th score.  Every non-candidate row j
//      has s16(j) <= s16[kc-1] and |exact(j) - s16(j)| <= err[i] (bound supplied by the caller,
//      derived in hcir/gallery.py), hence exact(j) < exact k-th: the top-k is the exact top-k.
//      Uncertified queries are re-run by the caller through the full exact scan.
// One wave per query; lane c re-scores candidate c (kc <= 64).
#include "common.h"

namespace {

struct RefineArgs {
  const float* q;
  const float* g;
  const int64_t* cand_idx;  // [nq][kc] global indices (idx_base included), < 0 = empty
  const float* cand_val;    // [nq][kc] filter scores, sorted descending
  const float* err;         // [nq] bound on |exact - filter score| for this query, or null:
  float eg, g16max, g32max, gamma;  //   then E_i is derived here from the mirror's constants (gallery.py)
  const float* qn;          // optional inverse norms (score = (dot * gn) * qn)
  const float* gn;
  float* out_val;
  int64_t* out_idx;
  int32_t* certified;
  int64_t nq, idx_base;
  int d, kc, k;
};

__global__ __launch_bounds__(256) void topk_refine_kernel(RefineArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t qi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (qi >= a.nq) return;
  const float ninf = -__builtin_huge_valf();
  // E_i = ||q|| eg + ||q - q~|| g16max + gamma (||q|| g32max + ||q~|| g16max), q~ = fp16(q)
  float err_i = 0.f;
  if (a.err) {
    err_i = a.err[qi];
  } else {
    const float* qr0 = a.q + qi * (int64_t)a.d;
    float s_q = 0.f, s_d = 0.f, s_h = 0.f;
    for (int k = lane; k < a.d; k += 64) {
      const float x = qr0[k];
      const float xh = (float)(_Float16)x;
      s_q = __builtin_fmaf(x, x, s_q);
      s_d = __builtin_fmaf(x - xh, x - xh, s_d);
      s_h = __builtin_fmaf(xh, xh, s_h);
    }
    const float nq_ = sqrtf(wave_sum(s_q)), nd_ = sqrtf(wave_sum(s_d)), nh_ = sqrtf(wave_sum(s_h));
    err_i = (nq_ * a.eg + nd_ * a.g16max + a.gamma * (nq_ * a.g32max + nh_ * a.g16max)) * 1.01f + 1e-7f;
  }
  float sc = ninf;
  int64_t id = -1;
  if (lane < a.kc) id = a.cand_idx[qi * a.kc + lane];
  if (id >= 0) {
    const float* qr = a.q + qi * (int64_t)a.d;
    const float* gr = a.g + (id - a.idx_base) * (int64_t)a.d;
    float acc = 0.f;
    const int nchunk = (a.d + 31) / 32;
    for (int c = 0; c < nchunk; ++c) {
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const int k0 = 32 * c + 8 * cc;
        f32x4 glo = {0.f, 0.f, 0.f, 0.f}, ghi = glo, qlo = glo, qhi = glo;
        if (k0 < a.d) {  // d % 8 == 0: an 8-wide group is entirely inside or outside
          glo = *reinterpret_cast<const f32x4*>(gr + k0);
          ghi = *reinterpret_cast<const f32x4*>(gr + k0 + 4);
          qlo = *reinterpret_cast<const f32x4*>(qr + k0);
          qhi = *reinterpret_cast<const f32x4*>(qr + k0 + 4);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc = __builtin_fmaf(glo[e], qlo[e], acc);  // k = 32c + 8cc + e
          acc = __builtin_fmaf(ghi[e], qhi[e], acc);  // then k + 4
        }
      }
    }
    if (a.gn) acc = acc * a.gn[id - a.idx_base];
    if (a.qn) acc = acc * a.qn[qi];
    sc = acc;
    if (!(sc == sc)) id = -1;  // NaN never ranks
  }
  // rank among the candidates: number of candidates that order strictly before this one
  int rank = 0;
  for (int c = 0; c < a.kc; ++c) {
    const float os = __shfl(sc, c);
    const int64_t oi = __shfl(id, c);
    if (oi >= 0 && (id < 0 || better(os, oi, sc, id))) ++rank;
  }
  if (id >= 0 && rank < a.k) {
    a.out_val[qi * a.k + rank] = sc;
    a.out_idx[qi * a.k + rank] = id;
  }
  // fill slots no candidate claims (fewer than k valid candidates)
  const unsigned long long valid = __ballot(id >= 0);
  const int nvalid = __builtin_popcountll(valid);
  if (lane >= nvalid && lane < a.k) {
    a.out_val[qi * a.k + lane] = ninf;
    a.out_idx[qi * a.k + lane] = -1;
  }
  // exact k-th score: the candidate whose rank is k-1
  const unsigned long long kth_mask = __ballot(id >= 0 && rank == a.k - 1);
  float vk = ninf;
  if (kth_mask) vk = __shfl(sc, __builtin_ctzll(kth_mask));
  if (lane == 0) {
    const int64_t last = a.cand_idx[qi * a.kc + a.kc - 1];
    int ok;
    if (last < 0) {
      ok = 1;  // the filter returned every row it has: nothing outside the candidate set
    } else {
      const float t = a.cand_val[qi * a.kc + a.kc - 1];
      ok = (kth_mask != 0ull) && (t + err_i < vk);
    }
    a.certified[qi] = ok;
  }
}

}  // namespace

extern "C" int hcir_topk_refine_f32(const float* q, int64_t nq, const float* g, int64_t ng, int32_t d,
                                    const int64_t* cand_idx, const float* cand_val, int32_t kc,
                                    int32_t k, int64_t idx_base, const float* q_inv_norm,
                                    const float* g_inv_norm, const float* err_bound,
                                    const float* mirror_consts, float* out_val, int64_t* out_idx,
                                    int32_t* certified, void* stream) {
  HCIR_ENTER();
  if (!q || !g || !cand_idx || !cand_val || (!err_bound && !mirror_consts) || !out_val || !out_idx ||
      !certified)
    return HCIR_ERR_INVALID;
  if (nq <= 0 || ng <= 0 || d <= 0 || (d & 7) || kc <= 0 || kc > 64 || k <= 0 || k > kc)
    return HCIR_ERR_INVALID;
  RefineArgs a{q, g, cand_idx, cand_val, err_bound, 0.f, 0.f, 0.f, 0.f, q_inv_norm, g_inv_norm, out_val,
               out_idx, certified, nq, idx_base, d, kc, k};
  if (!err_bound) {  // host array {eg, g16max, g32max, gamma}: constants of the gallery mirror
    a.eg = mirror_consts[0];
    a.g16max = mirror_consts[1];
    a.g32max = mirror_consts[2];
    a.gamma = mirror_consts[3];
  }
  hipLaunchKernelGGL(topk_refine_kernel, dim3((unsigned)hcir_cdiv(nq, 4)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a);
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}
