// gemm_tn.hip — weight gradients of the ViT's Linear layers:  dW[N][K] (+)= A[M][N]^T . B[M][K]
// (A = dY, B = the layer's input; nn.Linear backward of qkv / proj / fc1 / fc2 and of the patch embedding,
// HP/src/models_vit.py:63,66,70,79; torchvision EncoderBlock via HP/src/main_backbone.py:554; the training step
// is HP/src/pretrain_engine.py:745).
//
// The contraction index M is the SLOW index of both operands, so neither can feed an MFMA from row reads.  Instead of
// transposing M x N activations through HBM, the tiles are staged as they are stored ([64 m][256 n|k] fp16, whole
// 512-B row segments by LDS-DMA) and the fragments are read TRANSPOSED: ds_read_b64_tr_b16 hands lane i of a
// 16-lane group column i of a 4-row x 16-column block, i.e. 4 consecutive m for one n - two such reads are one
// 16x16x32 operand (cdna_hip_programming.md T10).  32-B chunks of a row are XOR-swizzled with
// key(row) = (row & 3) | ((row >> 3) & 1) << 2 so that the eight rows one half-wave touches per read
// ({0..3, 8..11} + 4 s) land on eight different 32-B bank groups: conflict-free.
//
// Workgroup: 8 waves (2 along n x 4 along k), tile 256(n) x 256(k), wave tile 128 x 64 = 8 x 4 MFMA tiles (128
// accumulators), two 64 KB LDS slots.  M is split over workgroups (tiles x splits ~ 2 per CU... one resident per CU);
// every (tile, split) writes its fp32 partial tile to the workspace and a second kernel adds the splits in order:
// deterministic, no float atomics (cdna_hip_programming.md Guideline 12).
#include "common.h"

namespace {

typedef __fp16 tn_fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

struct TnArgs {
  const _Float16* a;  // [M][lda]  (dY: n along the row)
  const _Float16* b;  // [M][ldb]  (layer input: k along the row)
  float* part;        // [splits][N][K] partial tiles (or the output itself when splits == 1)
  int64_t m, lda, ldb;
  int n, k;
  int tiles_n, tiles_k, splits;
  int64_t rows_per_split;  // multiple of 64
};

__device__ __forceinline__ int tn_key(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

__global__ __launch_bounds__(512, 2) void gemm_f16_tn_kernel(TnArgs g) {
  constexpr int SLOT = 64 * 512 * 2;  // A tile 32 KB then B tile 32 KB
  __shared__ __attribute__((aligned(16))) char lds[2 * SLOT];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_n = wave >> 2, wave_k = wave & 3;
  const int tile = blockIdx.x % (g.tiles_n * g.tiles_k), split = blockIdx.x / (g.tiles_n * g.tiles_k);
  const int n0 = (tile / g.tiles_k) * 256, k0 = (tile % g.tiles_k) * 256;
  const int64_t m_begin = (int64_t)split * g.rows_per_split;
  int64_t m_end = m_begin + g.rows_per_split;
  m_end = m_end < g.m ? m_end : g.m;
  const int nsteps = m_end > m_begin ? (int)((m_end - m_begin) / 64) : 0;

  // DMA sources: piece P = tid + 512 i, i < 4: operand tile row P >> 5 (64 rows), physical 16-B slot P & 31 holding
  // logical slot ((c32 ^ key(row)) << 1) | half;  32-bit byte offsets from the step's first row of each operand
  uint32_t aoff[4], boff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int P = tid + 512 * i;
    const int row = P >> 5, pc16 = P & 31;
    const int lc16 = (((pc16 >> 1) ^ tn_key(row)) << 1) | (pc16 & 1);
    aoff[i] = (uint32_t)(((int64_t)row * g.lda + n0) * 2 + lc16 * 16);
    boff[i] = (uint32_t)(((int64_t)row * g.ldb + k0) * 2 + lc16 * 16);
  }
  const char* abase = reinterpret_cast<const char*>(g.a) + m_begin * g.lda * 2;
  const char* bbase = reinterpret_cast<const char*>(g.b) + m_begin * g.ldb * 2;
  const int64_t astep = 64 * g.lda * 2, bstep = 64 * g.ldb * 2;
  const uint32_t lds0 = lds_addr(lds);
  auto issue = [&](int step, int i) {  // i < 8 constant after unrolling: 0..3 A pieces, 4..7 B pieces
#ifdef HCIR_TN_DMA_BUILTIN   // build flag (A/B): the compiler-visible transfer, which the waitcnt pass retires with
                             // vmcnt(0) in front of the next transposed read, i.e. inside the step that issued it
    const char* sp = i < 4 ? abase + step * astep + aoff[i & 3] : bbase + step * bstep + boff[i & 3];
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)sp,
        (__attribute__((address_space(3))) void*)(lds + (step & 1) * SLOT + ((tid & ~63) + 512 * i) * 16), 16, 0, 0);
#else
    lds_dma16(i < 4 ? abase + step * astep : bbase + step * bstep, i < 4 ? aoff[i & 3] : boff[i & 3],
              lds0 + (step & 1) * SLOT + ((tid & ~63) + 512 * i) * 16);
#endif
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int x = 0; x < 8; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) acc[x][y] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (nsteps > 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) issue(0, i);
  }

  // transposed-read addresses: lane = 16 g + 4 q + p supplies row (8 g + q [+ 4]) of the 32-row substep, byte column
  // 32 * (16-col block) + 8 p inside the tile row
  const int g16 = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
  for (int step = 0; step < nsteps; ++step) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const char* at = lds + (step & 1) * SLOT;
    const char* bt = at + 64 * 512;
    const bool more = step + 1 < nsteps;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f16x8 bf[4];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const int row = 32 * ks + 8 * g16 + 4 * hf + q4;
          const int c32 = wave_k * 4 + kt;  // 16 columns = 32 B = one chunk
          const tn_fp16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
              (tn_fp16x4 __attribute__((address_space(3)))*)(bt + row * 512 + ((c32 ^ tn_key(row)) << 5) + 8 * p4));
#pragma unroll
          for (int e = 0; e < 4; ++e) bf[kt][4 * hf + e] = (_Float16)v[e];
        }
      }
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        f16x8 af[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const int row = 32 * ks + 8 * g16 + 4 * hf + q4;
            const int c32 = wave_n * 8 + 4 * half + q;
            const tn_fp16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                (tn_fp16x4 __attribute__((address_space(3)))*)(at + row * 512 + ((c32 ^ tn_key(row)) << 5) + 8 * p4));
#pragma unroll
            for (int e = 0; e < 4; ++e) af[q][4 * hf + e] = (_Float16)v[e];
          }
        }
        if (more && ks == 0) {  // the eight DMA pieces of the next stage, in the first 32-row substep
#pragma unroll
          for (int i = 0; i < 4; ++i) issue(step + 1, 4 * half + i);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int kt = 0; kt < 4; ++kt)
            acc[4 * half + q][kt] =
                __builtin_amdgcn_mfma_f32_16x16x32_f16(af[q], bf[kt], acc[4 * half + q][kt], 0, 0, 0);
      }
    }
  }

  // partial tile -> workspace slab of this split: acc[nt][kt][r] is dW[n0 + 128 wave_n + 16 nt + 4 (lane>>4) + r]
  //                                                                 [k0 + 64 wave_k + 16 kt + (lane & 15)]
  float* out = g.part + (int64_t)split * g.n * g.k;
#pragma unroll
  for (int nt = 0; nt < 8; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + wave_n * 128 + nt * 16 + g16 * 4 + r;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) out[(int64_t)n * g.k + k0 + wave_k * 64 + kt * 16 + (lane & 15)] = acc[nt][kt][r];
    }
}

// dw[n][k] (+)= sum_s part[s][n][k], s in order; 4 elements per thread
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ part, int splits, int64_t nk,
                                                        int k, float* __restrict__ dw, int64_t lddw,
                                                        int accumulate) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= nk) return;
  f32x4 s = *reinterpret_cast<const f32x4*>(part + i);
  for (int sp = 1; sp < splits; ++sp) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(part + (int64_t)sp * nk + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] += v[e];
  }
  float* o = dw + (i / k) * lddw + (i % k);
  if (accumulate) {
    const f32x4 old = *reinterpret_cast<const f32x4*>(o);
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] += old[e];
  }
  *reinterpret_cast<f32x4*>(o) = s;
}

struct TnPlan {
  int tiles_n, tiles_k, splits;
  int64_t rows_per_split;
};

TnPlan tn_plan(int64_t m, int n, int k) {
  TnPlan p;
  p.tiles_n = n / 256;
  p.tiles_k = k / 256;
  const int tiles = p.tiles_n * p.tiles_k;
  const int64_t steps = m / 64;
  // one workgroup per CU (128 KB of LDS each) and never more than 256 in all: a grid of 270 runs as one full round
  // plus a nearly empty one (27 tiles x 10 splits took twice the time of 27 x 9)
  int64_t splits = 256 / tiles;
  splits = splits > steps ? steps : splits;
  splits = splits < 1 ? 1 : splits;
  p.rows_per_split = hcir_cdiv(steps, splits) * 64;
  p.splits = (int)hcir_cdiv(m, p.rows_per_split);
  return p;
}

}  // namespace

extern "C" {

size_t hcir_gemm_f16_tn_workspace_bytes(int64_t m, int32_t n, int32_t k) {
  if (m <= 0 || n <= 0 || k <= 0 || (n & 255) || (k & 255) || (m & 63)) return 0;
  const TnPlan p = tn_plan(m, n, k);
  return (size_t)p.splits * n * k * sizeof(float);
}

int hcir_gemm_f16_tn(const void* a, int64_t lda, const void* b, int64_t ldb, int64_t m, int32_t n, int32_t k,
                     float* dw, int64_t lddw, int accumulate, void* workspace, size_t workspace_bytes,
                     void* stream) {
  HCIR_ENTER();
  if (!a || !b || !dw || !workspace || m <= 0 || n <= 0 || k <= 0) return HCIR_ERR_INVALID;
  if (lda < n || ldb < k || (lda & 7) || (ldb & 7) || lddw < k || (lddw & 3)) return HCIR_ERR_INVALID;
  if ((n & 255) || (k & 255) || (m & 63)) return HCIR_ERR_UNSUPPORTED;  // the caller pads M with zero rows
  if (64 * lda * 2 >= (int64_t(1) << 31) || 64 * ldb * 2 >= (int64_t(1) << 31)) return HCIR_ERR_UNSUPPORTED;
  const TnPlan p = tn_plan(m, n, k);
  if (workspace_bytes < (size_t)p.splits * n * k * sizeof(float)) return HCIR_ERR_WORKSPACE;
  TnArgs g{static_cast<const _Float16*>(a), static_cast<const _Float16*>(b), static_cast<float*>(workspace),
           m, lda, ldb, n, k, p.tiles_n, p.tiles_k, p.splits, p.rows_per_split};
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(gemm_f16_tn_kernel, dim3((unsigned)(p.tiles_n * p.tiles_k * p.splits)), dim3(512), 0, st, g);
  HCIR_LAUNCH_CHECK();
  const int64_t nk = (int64_t)n * k;
  hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)hcir_cdiv(nk, 1024)), dim3(256), 0, st,
                     static_cast<const float*>(workspace), p.splits, nk, k, dw, lddw, accumulate);
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}

}  // extern "C"
