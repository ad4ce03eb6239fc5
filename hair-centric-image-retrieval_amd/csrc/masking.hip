// masking.hip — PositiveMaskingTransform on the device (HP/utils/transform.py:84-150).
//
// Reference: per image, the P x P patches whose mean over (C, P, P) exceeds `threshold` are "hair" patches;
// int(n_hair * u) of them, u ~ U(ratio range), chosen by a random permutation, are set to 0 — as a Python double
// loop over the batch and over the chosen patches with an .item() per image (one device sync and up to ~10
// slice-assignment launches per image).  Here: one workgroup per image, no host round trip.  The randomness is
// an INPUT (u[b] and one key per patch): the masked patches are the int(n_hair * u) hair patches with the
// smallest keys — the same distribution as randperm(n_hair)[:n]; tests drive the kernel and a CPU restatement
// with the same numbers and compare exactly.  HBM-bound: the image is read twice (means, copy) and written once.
#include "common.h"

namespace {

constexpr int kMaxPatches = 1024;

__global__ __launch_bounds__(256) void positive_masking_kernel(const float* __restrict__ img, float* __restrict__ out,
                                                               const float* __restrict__ u,
                                                               const float* __restrict__ keys, int c, int h, int w,
                                                               int p, float threshold, int* __restrict__ n_masked) {
  __shared__ float pmean[kMaxPatches];
  __shared__ unsigned char zero[kMaxPatches];
  __shared__ int n_hair_s;
  const int64_t b = blockIdx.x;
  const int nph = h / p, npw = w / p, np = nph * npw;
  const float* im = img + b * c * (int64_t)h * w;
  float* om = out + b * c * (int64_t)h * w;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) n_hair_s = 0;
  // ---- patch means: a wave per patch, fp32 sums of (c, p, p) in a fixed order, then a wave butterfly
  const float inv = 1.0f / (float)(c * p * p);
  for (int pi = wave; pi < np; pi += 4) {
    const int py = pi / npw, px = pi % npw;
    float s = 0.f;
    for (int e = lane; e < c * p * p; e += 64) {
      const int ch = e / (p * p), yy = (e / p) % p, xx = e % p;
      s += im[(ch * (int64_t)h + py * p + yy) * w + px * p + xx];
    }
    s = wave_sum(s);
    if (lane == 0) pmean[pi] = s * inv;
  }
  __syncthreads();
  // ---- selection: patch i is masked iff it is a hair patch and fewer than n_mask hair patches have a smaller
  //      key (ties: smaller patch index first)
  for (int i = tid; i < np; i += 256)
    if (pmean[i] > threshold) atomicAdd(&n_hair_s, 1);
  __syncthreads();
  const int n_hair = n_hair_s;
  const int n_mask = (int)((double)n_hair * (double)u[b]);   // int(len(hair_indices) * mask_ratio)
  for (int i = tid; i < np; i += 256) {
    unsigned char z = 0;
    if (n_mask > 0 && pmean[i] > threshold) {
      const float ki = keys[b * np + i];
      int rank = 0;
      for (int j = 0; j < np; ++j) {
        if (pmean[j] > threshold) {
          const float kj = keys[b * np + j];
          rank += (kj < ki || (kj == ki && j < i)) ? 1 : 0;
        }
      }
      z = rank < n_mask ? 1 : 0;
    }
    zero[i] = z;
  }
  __syncthreads();
  if (n_masked && tid == 0) n_masked[b] = n_mask > 0 ? n_mask : 0;
  // ---- copy with the chosen patches zeroed (pixels outside the patch grid, if H or W is not a multiple of
  //      the patch size, are copied)
  const int64_t total = (int64_t)c * h * w;
  for (int64_t e = tid; e < total; e += 256) {
    const int x = (int)(e % w), y = (int)((e / w) % h);
    const int py = y / p, px = x / p;
    const bool in_grid = py < nph && px < npw;
    om[e] = (in_grid && zero[py * npw + px]) ? 0.f : im[e];
  }
}

}  // namespace

extern "C" int hcir_positive_masking(const float* images, int64_t b, int32_t c, int32_t h, int32_t w,
                                     int32_t patch, float threshold, const float* u, const float* keys,
                                     float* out, int32_t* n_masked, void* stream) {
  HCIR_ENTER();
  if (!images || !out || !u || !keys || b <= 0 || c <= 0 || h <= 0 || w <= 0 || patch <= 0) return HCIR_ERR_INVALID;
  if (patch > h || patch > w) return HCIR_ERR_INVALID;
  if ((int64_t)(h / patch) * (w / patch) > kMaxPatches) return HCIR_ERR_UNSUPPORTED;
  if (b > 0x7fffffff) return HCIR_ERR_INVALID;
  hipLaunchKernelGGL(positive_masking_kernel, dim3((unsigned)b), dim3(256), 0, static_cast<hipStream_t>(stream),
                     images, out, u, keys, c, h, w, patch, threshold, n_masked);
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}
