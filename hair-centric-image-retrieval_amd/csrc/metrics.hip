// metrics.hip — what the reference does with the top-k list, on the device.
//
//   hcir_knn_vote          KNeighborsClassifier(n_neighbors=k).predict for EVERY k of the sweep from ONE
//                          top-max(k) neighbour list (HP/src/classification_engine.py:71,79-82): uniform-weight
//                          mode of the k neighbour labels, smallest label on ties (scipy.stats.mode as sklearn
//                          uses it).  The reference re-fits and re-scans for each of its 7 values of k.
//   hcir_confusion_matrix  accuracy_score / confusion_matrix counts (HP/src/classification_engine.py:83,85,93).
//   hcir_retrieval_metrics Recall@K and AP@K per query and their means
//                          (experiments/DualViewHair/scripts/quantitative_eval.py:194-209,228-234).
// All integer work except the AP sums, which run in fp64 in rank order like the reference's Python floats.
#include "common.h"

namespace {

constexpr int kMaxKs = 16;
struct KList {
  int n;
  int k[kMaxKs];  // ascending
};

// One wave per query.  The class histogram lives in LDS (nclass counters); neighbours are added in rank
// order, segment by segment between consecutive values of k; after each segment the wave takes the arg-max
// (count desc, label asc) of the histogram.
__global__ __launch_bounds__(64) void knn_vote_kernel(const int64_t* __restrict__ nbr, int64_t nq, int kmax,
                                                      int64_t idx_base, const int64_t* __restrict__ labels,
                                                      int64_t nlabels, int nclass, KList ks,
                                                      int64_t* __restrict__ pred, int* __restrict__ bad) {
  extern __shared__ int hist[];
  const int lane = threadIdx.x;
  const int64_t qi = blockIdx.x;
  for (int c = lane; c < nclass; c += 64) hist[c] = 0;
  __syncthreads();
  int done = 0;
  for (int s = 0; s < ks.n; ++s) {
    const int kend = ks.k[s];
    for (int j0 = done; j0 < kend; j0 += 64) {
      const int j = j0 + lane;
      if (j < kend) {
        const int64_t row = nbr[qi * kmax + j] - idx_base;
        int64_t lab = -1;
        if (row >= 0 && row < nlabels) lab = labels[row];
        if (lab >= 0 && lab < nclass)
          atomicAdd(&hist[(int)lab], 1);
        else
          *bad = 1;  // an empty slot (idx < 0) or a label outside [0, nclass): the caller raises
      }
    }
    done = kend;
    __syncthreads();
    // arg-max: key = count * 2^32 + (2^32 - 1 - label)  ->  maximum = most votes, then smallest label
    uint64_t best = 0;
    for (int c = lane; c < nclass; c += 64) {
      const uint64_t key = ((uint64_t)(uint32_t)hist[c] << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)c);
      best = key > best ? key : best;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const uint64_t o = __shfl_xor(best, off);
      best = o > best ? o : best;
    }
    if (lane == 0) pred[(int64_t)s * nq + qi] = (int64_t)(0xFFFFFFFFu - (uint32_t)best);
    __syncthreads();
  }
}

__global__ void confusion_kernel(const int64_t* __restrict__ y_true, const int64_t* __restrict__ y_pred,
                                 int64_t n, int nclass, int32_t* __restrict__ cm, int* __restrict__ bad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t t = y_true[i], p = y_pred[i];
  if (t < 0 || t >= nclass || p < 0 || p >= nclass) {
    *bad = 1;
    return;
  }
  atomicAdd(&cm[t * nclass + p], 1);
}

// One wave per query: lane l owns ranks l, l + 64, ...; a rank is a hit when its id is in the query's
// ground-truth list.  AP@K = sum over hit ranks i < K of (hits up to i) / (i + 1), divided by
// min(|gt|, K) (0 when the list is empty); the sum runs in rank order in fp64 (lane 0), as the
// reference's Python loop does.
__global__ __launch_bounds__(64) void retrieval_metrics_kernel(const int64_t* __restrict__ ret, int64_t nq,
                                                               int kmax, const int64_t* __restrict__ gt,
                                                               int gmax, KList ks, int32_t* __restrict__ hit,
                                                               double* __restrict__ ap) {
  extern __shared__ int flags[];  // [kmax] 1 = hit
  const int lane = threadIdx.x;
  const int64_t qi = blockIdx.x;
  int ngt = 0;
  for (int j = 0; j < gmax; ++j) ngt += gt[qi * gmax + j] >= 0 ? 1 : 0;
  for (int i = lane; i < kmax; i += 64) {
    const int64_t id = ret[qi * kmax + i];
    int f = 0;
    if (id >= 0)
      for (int j = 0; j < gmax; ++j) f |= (gt[qi * gmax + j] == id) ? 1 : 0;
    flags[i] = f;
  }
  __syncthreads();
  if (lane == 0) {
    int hits = 0, s = 0;
    double sum = 0.0;
    for (int i = 0; i < kmax && s < ks.n; ++i) {
      if (flags[i]) {
        ++hits;
        sum += (double)hits / (double)(i + 1);
      }
      while (s < ks.n && ks.k[s] == i + 1) {
        const int denom = ngt < ks.k[s] ? ngt : ks.k[s];
        hit[(int64_t)s * nq + qi] = hits > 0 ? 1 : 0;
        ap[(int64_t)s * nq + qi] = ngt > 0 ? sum / (double)denom : 0.0;
        ++s;
      }
    }
  }
}

// means over the queries in index order (one thread per k: deterministic, the sums are tiny)
__global__ void metrics_mean_kernel(const int32_t* __restrict__ hit, const double* __restrict__ ap, int64_t nq,
                                    int nk, double* __restrict__ recall, double* __restrict__ map) {
  const int s = threadIdx.x;
  if (s >= nk) return;
  int64_t h = 0;
  double a = 0.0;
  for (int64_t i = 0; i < nq; ++i) {
    h += hit[(int64_t)s * nq + i];
    a += ap[(int64_t)s * nq + i];
  }
  recall[s] = nq > 0 ? (double)h / (double)nq : 0.0;
  map[s] = nq > 0 ? a / (double)nq : 0.0;
}

bool make_klist(const int32_t* ks, int nk, int kmax, KList& out) {
  if (!ks || nk <= 0 || nk > kMaxKs) return false;
  out.n = nk;
  for (int i = 0; i < nk; ++i) {
    if (ks[i] < 1 || ks[i] > kmax || (i && ks[i] <= ks[i - 1])) return false;
    out.k[i] = ks[i];
  }
  return true;
}

}  // namespace

extern "C" {

int hcir_knn_vote(const int64_t* nbr_idx, int64_t nq, int32_t kmax, int64_t idx_base, const int64_t* labels,
                  int64_t nlabels, int32_t nclass, const int32_t* ks, int32_t nk, int64_t* pred, int32_t* bad,
                  void* stream) {
  HCIR_ENTER();
  if (!nbr_idx || !labels || !pred || !bad || nq <= 0 || kmax <= 0 || nlabels <= 0) return HCIR_ERR_INVALID;
  if (nclass <= 0 || nclass > 16384) return nclass <= 0 ? HCIR_ERR_INVALID : HCIR_ERR_UNSUPPORTED;
  KList kl;
  if (!make_klist(ks, nk, kmax, kl)) return HCIR_ERR_INVALID;
  hipLaunchKernelGGL(knn_vote_kernel, dim3((unsigned)nq), dim3(64), (size_t)nclass * 4,
                     static_cast<hipStream_t>(stream), nbr_idx, nq, kmax, idx_base, labels, nlabels, nclass, kl,
                     pred, bad);
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}

int hcir_confusion_matrix(const int64_t* y_true, const int64_t* y_pred, int64_t n, int32_t nclass, int32_t* cm,
                          int32_t* bad, void* stream) {
  HCIR_ENTER();
  if (!y_true || !y_pred || !cm || !bad || n <= 0 || nclass <= 0) return HCIR_ERR_INVALID;
  hipLaunchKernelGGL(confusion_kernel, dim3((unsigned)hcir_cdiv(n, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), y_true, y_pred, n, nclass, cm, bad);
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}

int hcir_retrieval_metrics(const int64_t* retrieved, int64_t nq, int32_t kmax, const int64_t* gt, int32_t gmax,
                           const int32_t* ks, int32_t nk, int32_t* hit, double* ap, double* recall_mean,
                           double* map_mean, void* stream) {
  HCIR_ENTER();
  if (!retrieved || !gt || !hit || !ap || nq <= 0 || kmax <= 0 || gmax <= 0) return HCIR_ERR_INVALID;
  if (kmax > 8192) return HCIR_ERR_UNSUPPORTED;
  KList kl;
  if (!make_klist(ks, nk, kmax, kl)) return HCIR_ERR_INVALID;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(retrieval_metrics_kernel, dim3((unsigned)nq), dim3(64), (size_t)kmax * 4, st, retrieved, nq,
                     kmax, gt, gmax, kl, hit, ap);
  HCIR_LAUNCH_CHECK();
  if (recall_mean && map_mean) {
    hipLaunchKernelGGL(metrics_mean_kernel, dim3(1), dim3(64), 0, st, hit, ap, nq, nk, recall_mean, map_mean);
    HCIR_LAUNCH_CHECK();
  }
  return HCIR_OK;
}

}  // extern "C"
