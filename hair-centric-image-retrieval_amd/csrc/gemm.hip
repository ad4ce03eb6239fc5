// gemm.hip — out[M,N] = epilogue(A[M,K] . W[N,K]^T), fp16 MFMA, fp32 accumulate.
//
// Replaces nn.Linear / nn.MultiheadAttention in_proj,out_proj / MLPBlock / timm Mlp
// on the ViT path (HP/src/models_vit.py:63,66,70,79; torchvision EncoderBlock via
// HP/src/main_backbone.py:554) and Conv2d(3,768,16,16) patch embedding
// (HP/src/main_backbone.py:543; HP/src/models_vit.py:42,48).
//
// Two kernels (launch_gemm picks):
//   * gemm_f16_big_kernel — M >= 1024, N % 256 == 0, K % 64 == 0 (every encoder GEMM of the benchmark):
//     persistent 256(n) x 256(m) x 64(k) tiles, 8 waves of 128 x 64 (MFMA 16x16x32), two 64 KB LDS slots filled
//     by LDS-DMA, LDS-transposed full-line epilogues incl. the LayerNorm fold (hcir_gemm_f16_fused);
//   * gemm_f16_kernel — everything else (small M, ragged N / K): the sim_core.h tile engine with W rows on the
//     MFMA row index and activation rows on the column index (a lane owns ONE activation row and receives the
//     output features in groups of 4 registers), workgroup tile 128 x 128 x 64, 4 waves of 2x2 MFMA 32x32x16.
#include "sim_core.h"
#include "act.h"
#include <stdlib.h>

namespace {

using GemmCfg = SimCfg<_Float16, 2, 2, 2>;  // GM = 128 W rows, QB = 128 A rows

// Internal epilogue codes of hcir_gemm_f16_fused (persistent kernel only), beyond hcir_epilogue:
//   EPI_RESID_F16_STATS: BIAS_RESID_F16 that also writes per-row partial (sum, sum of squares) of the stored
//                        fp16 rows, one slice per 64 output features, for the LayerNorm that reads them next;
//   EPI_LN_BIAS_F16 / EPI_LN_BIAS_GELU_F16: the A rows are the RAW residual rows x and W carries the
//                        LayerNorm gamma (W' = gamma o W): out = act(rstd[m] (acc - mean[m] c1[n]) + bias[n])
//                        with c1[n] = sum_k W'[n][k], bias[n] = sum_k beta[k] W[n][k] + b[n]  (LayerNorm folded
//                        into the GEMM; no normalised copy of the tokens is ever written).
constexpr int EPI_RESID_F16_STATS = 7;
constexpr int EPI_LN_BIAS_F16 = 8;
constexpr int EPI_LN_BIAS_GELU_F16 = 9;
constexpr int EPI_BIAS_F16_DUAL_GELU = 10;  // out = fp16(acc + bias) AND out2 = fp16(gelu(acc + bias)) (hcir_gemm_f16_gelu_dual)

struct GemmArgs {
  const _Float16* a;
  const _Float16* w;
  const float* bias;
  const float* scale;
  void* out;
  int64_t m, lda, ldw, ldo;
  int n, k;
  const float* ln_stats;  // [m][2] (mean, rstd) of the A rows           (EPI_LN_*)
  const float* ln_c1;     // [n]                                           (EPI_LN_*)
  float* stats_part;      // [n/64][m][2] partial (sum, sumsq) of out rows (EPI_RESID_F16_STATS)
  const void* resid;      // residual rows of the *_RESID_* epilogues (same type and row pitch as out); == out: in place
  void* out2;             // second fp16 output of EPI_BIAS_F16_DUAL_GELU (row pitch ldo)
  int64_t stats_ld = 0;   // rows per slice of stats_part (0: m) - a launch over a row range of a larger matrix
};

// XCD-aware tile order: consecutive workgroup ids are dealt round-robin over the 8
// XCDs (MI355X_MICROARCH.md §Workgroup dispatch); remap so that one XCD walks a
// contiguous run of tiles and re-uses the W panel / A panel from its own L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

template <int EPI, int NTILES>
__device__ __forceinline__ void gemm_epilogue_t(const GemmArgs& g, const f32x16 (&acc)[NTILES][2],
                                                int64_t m0, int nbase, int wave_m, int lane) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int64_t m = m0 + wave_m * 64 + mt * 32 + r;
    if (m >= g.m) continue;
#pragma unroll
    for (int nt = 0; nt < NTILES; ++nt) {
#pragma unroll
      for (int grp = 0; grp < 4; ++grp) {
        const int n = nbase + nt * 32 + 8 * grp + 4 * h;
        if (n >= g.n) continue;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[nt][mt][4 * grp + e];
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
        if (g.bias) b = *reinterpret_cast<const f32x4*>(g.bias + n);
        if constexpr (EPI == HCIR_EPI_AFFINE_RELU_F16 || EPI == HCIR_EPI_AFFINE_F32) {
          const f32x4 s = *reinterpret_cast<const f32x4*>(g.scale + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(v[e], s[e], b[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += b[e];
        }
        if constexpr (EPI == HCIR_EPI_BIAS_F16 || EPI == HCIR_EPI_BIAS_GELU_F16 ||
                      EPI == HCIR_EPI_AFFINE_RELU_F16) {
          f16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float x = v[e];
            if constexpr (EPI == HCIR_EPI_BIAS_GELU_F16) x = gelu_erf(x);
            if constexpr (EPI == HCIR_EPI_AFFINE_RELU_F16) x = fmaxf(x, 0.f);
            o[e] = (_Float16)x;
          }
          *reinterpret_cast<f16x4*>(static_cast<_Float16*>(g.out) + m * g.ldo + n) = o;
        } else if constexpr (EPI == HCIR_EPI_BIAS_RESID_F16) {
          _Float16* p = static_cast<_Float16*>(g.out) + m * g.ldo + n;
          const f16x4 oh = *reinterpret_cast<const f16x4*>(static_cast<const _Float16*>(g.resid) + m * g.ldo + n);
          f16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float sc1 = g.scale ? g.scale[n + e] : 1.0f;
            o[e] = (_Float16)__builtin_fmaf(sc1, v[e], (float)oh[e]);
          }
          *reinterpret_cast<f16x4*>(p) = o;
        } else if constexpr (EPI == HCIR_EPI_BIAS_RESID_F32) {
          float* p = static_cast<float*>(g.out) + m * g.ldo + n;
          f32x4 o = *reinterpret_cast<const f32x4*>(static_cast<const float*>(g.resid) + m * g.ldo + n);
          if (g.scale) {
            const f32x4 s = *reinterpret_cast<const f32x4*>(g.scale + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = __builtin_fmaf(s[e], v[e], o[e]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += v[e];
          }
          *reinterpret_cast<f32x4*>(p) = o;
        } else {
          *reinterpret_cast<f32x4*>(static_cast<float*>(g.out) + m * g.ldo + n) = v;
        }
      }
    }
  }
}

template <int EPI>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, const f32x16 (&acc)[2][2],
                                              int64_t m0, int n0, int wave_n, int wave_m, int lane) {
  gemm_epilogue_t<EPI, 2>(g, acc, m0, n0 + wave_n * 64, wave_m, lane);
}
// Hide a value's origin from hipcc's waitcnt pass: after the explicit wait below, bias / scale
// registers are "produced by asm", not by a pending global load.  Without this the per-iteration
// control flow of the store loop makes the pass re-insert `s_waitcnt vmcnt(0)` before EVERY store
// (it can no longer prove the bias load retired), which also drains the previous store: the 16-32
// stores of a tile were fully serialised (seen in the .s; 40 % of GEMM time).
__device__ __forceinline__ void launder(f32x4& v) { asm volatile("" : "+v"(v)); }

// Accumulators of one wave tile (128 n x 64 m), either MFMA shape:
//   MF16 = false: acc32[nt 0..3][mt 0..1], f32x16: n = 32nt + 8(i>>2) + 4(lane>>5) + (i&3), m = 32mt + (lane&31)
//   MF16 = true : acc16[nt 0..7][mt 0..3], f32x4 : n = 16nt + 4(lane>>4) + i,            m = 16mt + (lane&15)
template <bool MF16>
struct WaveAcc;
template <>
struct WaveAcc<false> {
  f32x16 a[4][2];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int i = 0; i < 16; ++i) a[x][y][i] = 0.f;
  }
};
template <>
struct WaveAcc<true> {
  f32x4 a[8][4];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int x = 0; x < 8; ++x)
#pragma unroll
      for (int y = 0; y < 4; ++y)
#pragma unroll
        for (int i = 0; i < 4; ++i) a[x][y][i] = 0.f;
  }
};

// 8-lane (one 128-B output line) sum by DPP: quad xor 1, quad xor 2, then the mirror of the 8-lane half row
__device__ __forceinline__ float sum8_dpp(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  return v;
}

template <int EPI, bool FULL, bool MF16>
__device__ __forceinline__ void gemm_epilogue256_lds_impl(const GemmArgs& g, const WaveAcc<MF16>& acc,
                                                          char* region, int64_t m0w, int nbase,
                                                          int lane) {
  const int r = lane & 31, h = lane >> 5;
  constexpr bool kLn = (EPI == EPI_LN_BIAS_F16 || EPI == EPI_LN_BIAS_GELU_F16);
  constexpr bool kGelu = (EPI == HCIR_EPI_BIAS_GELU_F16 || EPI == EPI_LN_BIAS_GELU_F16);
  constexpr bool kResidH = (EPI == HCIR_EPI_BIAS_RESID_F16 || EPI == EPI_RESID_F16_STATS);
  constexpr bool kDual = (EPI == EPI_BIAS_F16_DUAL_GELU);
  static_assert(MF16 || !kDual, "the dual-output epilogue exists for the 16x16x32 accumulator layout only");
  constexpr bool kF16 = (EPI == HCIR_EPI_BIAS_F16 || kGelu || kLn ||
                         EPI == HCIR_EPI_AFFINE_RELU_F16 || kResidH || kDual);
  constexpr bool kAffine = (EPI == HCIR_EPI_AFFINE_RELU_F16 || EPI == HCIR_EPI_AFFINE_F32);
  constexpr int NPASS = kF16 ? 2 : 4;  // 64 or 32 output features (128 B) per pass
  constexpr int NB = kF16 ? 2 : 1;     // float4 of bias per lane per pass
  const int rrow = lane >> 3, rchunk = lane & 7;

  // 16x16x32 accumulators, fp16 outputs: bias / scale / activation / LayerNorm fold run on the ACCUMULATOR side, in
  // fp32, before the one rounding to fp16 - the row side only moves 16-B pieces (and adds the fp16 residual).  The
  // earlier form rounded the raw accumulator to fp16, widened it again behind the transposition, did the math
  // there and rounded a second time: five VALU instructions per output pair instead of two, and the epilogue phase
  // of a tile is VALU / LDS-issue time (in-kernel stamps: 4.6 us per qkv tile, 8.6 us per fc1 tile).
  constexpr bool kAccSide = MF16 && kF16;
  constexpr bool kScaleA = kAccSide && (EPI == HCIR_EPI_AFFINE_RELU_F16 || kResidH);
  f32x4 biasA[kAccSide ? 8 : 1], scaleA[kScaleA ? 8 : 1];
  bool has_scale = false;
  if constexpr (kAccSide) {
    has_scale = kScaleA && (EPI == HCIR_EPI_AFFINE_RELU_F16 || g.scale != nullptr);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int n = nbase + 16 * t + 4 * (lane >> 4);
      biasA[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (g.bias) biasA[t] = *reinterpret_cast<const f32x4*>(g.bias + n);
      if constexpr (kScaleA) {
        scaleA[t] = (f32x4){1.f, 1.f, 1.f, 1.f};
        if (has_scale) scaleA[t] = *reinterpret_cast<const f32x4*>(g.scale + n);
      }
    }
  }
  // per-lane bias / scale of every pass (row side: fp32 outputs and the 32x32x16 variant), loaded once, retired
  // once, then laundered
  f32x4 bias[NPASS][NB], scale[NPASS][NB];
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
    if constexpr (kAccSide) break;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int n = nbase + pass * (kF16 ? 64 : 32) + rchunk * (kF16 ? 8 : 4) + 4 * j;
      bias[pass][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      scale[pass][j] = (f32x4){1.f, 1.f, 1.f, 1.f};
      if (g.bias) bias[pass][j] = *reinterpret_cast<const f32x4*>(g.bias + n);
      if (kAffine || ((EPI == HCIR_EPI_BIAS_RESID_F32 || kResidH) && g.scale))
        scale[pass][j] = *reinterpret_cast<const f32x4*>(g.scale + n);
    }
  }
  // LayerNorm fold, out = rstd[m] (acc - mean[m] c1[n]) + bias[n].  The CENTERING runs on the accumulator side,
  // in fp32, before the fp16 image: rounding the raw accumulator first loses the result under the cancellation
  // acc - mean c1 when |mean| >> std (measured: 20x the error at mean/std = 50, 2.4x at 5; equal at 0).  The
  // scaling by rstd[row] and the bias run on the row side.
  //   accumulator layout: mean of the 4 (2) rows this lane owns there, c1 of its features there
  //   (16x16x32: n = 16 t + 4 (lane>>4) + e, t < 8;  32x32x16: n = 32 nt + 8 grp + 4 h + e)
  float ln_rs[8], ln_mean[4];
  f32x4 c1a[MF16 ? 8 : 16];
  if constexpr (kLn) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      if constexpr (kAccSide) break;
      int64_t mm = m0w + it * 8 + rrow;
      mm = mm < g.m ? mm : g.m - 1;
      ln_rs[it] = g.ln_stats[2 * mm + 1];
    }
#pragma unroll
    for (int mt = 0; mt < (MF16 ? 4 : 2); ++mt) {
      int64_t mm = m0w + (MF16 ? 16 * mt + (lane & 15) : 32 * mt + r);
      mm = mm < g.m ? mm : g.m - 1;
      ln_mean[mt] = g.ln_stats[2 * mm];
      if constexpr (kAccSide) ln_rs[mt] = g.ln_stats[2 * mm + 1];  // (the row-side rstd above is unused then)
    }
#pragma unroll
    for (int t = 0; t < (MF16 ? 8 : 16); ++t)
      c1a[t] = *reinterpret_cast<const f32x4*>(
          g.ln_c1 + nbase + (MF16 ? 16 * t + 4 * (lane >> 4) : 32 * (t >> 2) + 8 * (t & 3) + 4 * h));
  }
  // (Issuing the next tile's first stage from here, behind these constant loads and with vmcnt(8), instead of from
  // inside the tile's last k-step - so that this wait does not sit out the rest of the stage's round trip - was
  // measured: 3-5 % SLOWER on every shape, tools/ab_gemm.py; like the early-issue main loop, DESIGN.md "GEMM round 2".)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (kLn) {
#pragma unroll
    for (int it = 0; it < (kAccSide ? 4 : 8); ++it) asm volatile("" : "+v"(ln_rs[it]));
#pragma unroll
    for (int mt = 0; mt < (MF16 ? 4 : 2); ++mt) asm volatile("" : "+v"(ln_mean[mt]));
#pragma unroll
    for (int t = 0; t < (MF16 ? 8 : 16); ++t) launder(c1a[t]);
  }
  if constexpr (kAccSide) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      launder(biasA[t]);
      if constexpr (kScaleA) launder(scaleA[t]);
    }
  } else {
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass)
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        launder(bias[pass][j]);
        launder(scale[pass][j]);
      }
  }

  // (dual output: every 64-feature pass runs twice through the same wave-private image - first the pre-activation,
  // then its GELU; a wave's LDS operations execute in order, so the second write follows the first pass's reads)
  constexpr int NITER = kDual ? 2 * NPASS : NPASS;
#pragma unroll
  for (int iter = 0; iter < NITER; ++iter) {
    const int pass = kDual ? iter >> 1 : iter;
    const bool second = kDual && (iter & 1);
    // ---- accumulators -> LDS (lane = output row m, registers = features n)
    if constexpr (!MF16) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int row = mt * 32 + r;
        if constexpr (kF16) {
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int nt = 2 * pass + q;
#pragma unroll
            for (int grp = 0; grp < 4; ++grp) {
              f16x4 o;
              if constexpr (kLn) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  o[e] = (_Float16)__builtin_fmaf(-ln_mean[mt], c1a[4 * nt + grp][e], acc.a[nt][mt][4 * grp + e]);
              } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (_Float16)acc.a[nt][mt][4 * grp + e];
              }
              const int chunk = q * 4 + grp;
              *reinterpret_cast<f16x4*>(region + row * 128 + ((chunk ^ (row & 7)) << 4) + 8 * h) = o;
            }
          }
        } else {
#pragma unroll
          for (int grp = 0; grp < 4; ++grp) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = acc.a[pass][mt][4 * grp + e];
            const int chunk = 2 * grp + h;
            *reinterpret_cast<f32x4*>(region + row * 128 + ((chunk ^ (row & 7)) << 4)) = o;
          }
        }
      }
    } else {
      const int r16 = lane & 15, q16 = lane >> 4;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int row = mt * 16 + r16;
        if constexpr (kF16) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {   // 64 features per pass = 4 n-tiles of 16
            const int nt = 4 * pass + q;
            f16x4 o;
            float xs[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float x = acc.a[nt][mt][e];
              const float bb = biasA[nt][e];
              if constexpr (kLn) {
                // out = rstd[m] (acc - mean[m] c1[n]) + bias[n]: centered first (cancellation when |mean| >> std)
                x = __builtin_fmaf(ln_rs[mt], __builtin_fmaf(-ln_mean[mt], c1a[nt][e], x), bb);
              } else if constexpr (EPI == HCIR_EPI_AFFINE_RELU_F16) {
                x = fmaxf(__builtin_fmaf(x, scaleA[nt][e], bb), 0.f);
              } else if constexpr (kResidH) {
                x = scaleA[nt][e] * (x + bb);
              } else {
                x += bb;
              }
              xs[e] = x;
            }
            if (kGelu || second) {
#pragma unroll
              for (int e = 0; e < 4; e += 2) {
                const gelu_f32x2 y = gelu_erf2((gelu_f32x2){xs[e], xs[e + 1]});
                xs[e] = y[0];
                xs[e + 1] = y[1];
              }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (_Float16)xs[e];
            const int chunk = 2 * q + (q16 >> 1);
            *reinterpret_cast<f16x4*>(region + row * 128 + ((chunk ^ (row & 7)) << 4) + 8 * (q16 & 1)) = o;
          }
        } else {
#pragma unroll
          for (int q = 0; q < 2; ++q) {   // 32 features per pass = 2 n-tiles of 16
            const int nt = 2 * pass + q;
            const int chunk = 4 * q + q16;
            *reinterpret_cast<f32x4*>(region + row * 128 + ((chunk ^ (row & 7)) << 4)) = acc.a[nt][mt];
          }
        }
      }
    }
    // ---- LDS -> rows: lane = (row rrow + 8 it, 16-B chunk rchunk)
    if constexpr (kF16) {
      const int n = nbase + pass * 64 + rchunk * 8;
#pragma unroll
      for (int it0 = 0; it0 < 8; it0 += 4) {
        // fp16 residual: the four old rows of a group are requested back to back (counted waits)
        f16x8 oldh[4];
        if constexpr (kResidH) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int64_t mm = m0w + (it0 + u) * 8 + rrow;
#pragma unroll
            for (int e = 0; e < 8; ++e) oldh[u][e] = (_Float16)0.f;
            // (-DHCIR_EPI_ABL_NOOLD: timing ablation, WRONG results - the residual rows are not read: proj 237 -> 201 us,
            // fc2 709 -> 686 us at batch 880.  Requesting all sixteen pieces of the tile during its last k-step instead
            // (registers freed by keeping the bias in LDS; bit-identical) made proj 3.5 % and fc2 1.3 % SLOWER: the
            // 128 KB a tile reads here go through the same L2 -> CU path as its stages, and that path is what bounds
            // the main loop - profiles/r4_gemm_resid_prefetch.txt)
#ifndef HCIR_EPI_ABL_NOOLD
            if (FULL || mm < g.m)
              oldh[u] = *reinterpret_cast<const f16x8*>(static_cast<const _Float16*>(g.resid) + mm * g.ldo + n);
#endif
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int row = (it0 + u) * 8 + rrow;
          const f16x8 v = *reinterpret_cast<const f16x8*>(region + row * 128 + ((rchunk ^ (row & 7)) << 4));
          f16x8 o;
          if constexpr (kAccSide) {
            // the image already holds the finished fp16 values; the fp16 residual is one packed add per pair
            // (the exact sum of two fp16 numbers, rounded once)
            if constexpr (kResidH)
              o = v + oldh[u];
            else
              o = v;
          } else {
          float xs[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float x = (float)v[e];
            const float bb = bias[pass][e >> 2][e & 3];
            if constexpr (EPI == HCIR_EPI_AFFINE_RELU_F16) {
              x = fmaxf(__builtin_fmaf(x, scale[pass][e >> 2][e & 3], bb), 0.f);
            } else if constexpr (kResidH) {
              x = __builtin_fmaf(scale[pass][e >> 2][e & 3], x + bb, (float)oldh[u][e]);
            } else if constexpr (kLn) {
              x = __builtin_fmaf(ln_rs[it0 + u], x, bb);
            } else {
              x += bb;
            }
            xs[e] = x;
          }
          if constexpr (kGelu) {
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
              const gelu_f32x2 y = gelu_erf2((gelu_f32x2){xs[e], xs[e + 1]});
              xs[e] = y[0];
              xs[e + 1] = y[1];
            }
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (_Float16)xs[e];
          }
          const int64_t m = m0w + row;
          // non-temporal: the 128 KB a workgroup writes per tile would otherwise push the W panels out of the
          // XCD's L2 (W is re-fetched ~50x from the Infinity Cache per qkv launch); +1.4 % end to end
          if (FULL || m < g.m)
            __builtin_nontemporal_store(
                o, reinterpret_cast<f16x8*>(static_cast<_Float16*>(second ? g.out2 : g.out) + m * g.ldo + n));
          if constexpr (EPI == EPI_RESID_F16_STATS) {
            // (mean, sum of squared deviations from that mean) of the 64 STORED fp16 values of this row slice,
            // for the next LayerNorm.  Two-pass per slice (the values are in registers) and Chan's combination
            // in the finalize kernel: a plain (sum, sum of squares) pair loses the variance to cancellation
            // when |mean| >> std (0.2 % error in rstd at mean/std = 50).
            float xv[8], s1 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              xv[e] = (float)o[e];
              s1 += xv[e];
            }
            const float mean_s = sum8_dpp(s1) * (1.0f / 64.0f);
            float m2 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float dv = xv[e] - mean_s;
              m2 = __builtin_fmaf(dv, dv, m2);
            }
            m2 = sum8_dpp(m2);
            if (rchunk == 0 && (FULL || m < g.m)) {
              const int64_t slice = (nbase >> 6) + pass;
              *reinterpret_cast<f32x2*>(g.stats_part + (slice * (g.stats_ld ? g.stats_ld : g.m) + m) * 2) = (f32x2){mean_s, m2};
            }
          }
        }
      }
    } else {
      const int n = nbase + pass * 32 + rchunk * 4;
      const f32x4 b = bias[pass][0], sc = scale[pass][0];
#pragma unroll
      for (int it0 = 0; it0 < 8; it0 += 4) {
        // the four old-value loads of a group are issued back to back, so their waits are COUNTED
        // (vmcnt(3), ...) and the stores of the previous group stay in flight
        f32x4 oldv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int64_t mm = m0w + (it0 + u) * 8 + rrow;
          oldv[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if constexpr (EPI == HCIR_EPI_BIAS_RESID_F32) {
            if (FULL || mm < g.m)
              oldv[u] = *reinterpret_cast<const f32x4*>(static_cast<const float*>(g.resid) + mm * g.ldo + n);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int row = (it0 + u) * 8 + rrow;
          const int64_t mm = m0w + row;
          f32x4 v = *reinterpret_cast<const f32x4*>(region + row * 128 + ((rchunk ^ (row & 7)) << 4));
          if constexpr (EPI == HCIR_EPI_AFFINE_F32) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(v[e], sc[e], b[e]);
          } else if constexpr (EPI == HCIR_EPI_BIAS_RESID_F32) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(sc[e], v[e] + b[e], oldv[u][e]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += b[e];
          }
          if (FULL || mm < g.m) *reinterpret_cast<f32x4*>(static_cast<float*>(g.out) + mm * g.ldo + n) = v;
        }
      }
    }
  }
}

// Epilogue of the 256 x 256 kernel: every wave transposes its 128(n) x 64(m) accumulator tile
// through a private 8 KB piece of the LDS slot that the tile's last k-step has just released,
// 128 B of one output row at a time, so that a lane ends up with 16 contiguous bytes of ONE output
// row and 8 lanes cover a whole 128-B line (the direct layout gives every lane 8 B of a different
// row: 1.6-1.9 TB/s effective on the fp16 outputs).  Bias / GELU / residual run on the
// row-contiguous side.  [64 rows][128 B] image, 16-B chunks XOR-swizzled with row & 7.
// Full tiles (all 64 rows of the wave inside M) take a branch-free path.
template <int EPI, bool MF16>
__device__ __forceinline__ void gemm_epilogue256_lds(const GemmArgs& g, const WaveAcc<MF16>& acc,
                                                     char* region, int64_t m0w, int nbase, int lane) {
  // opaque lane id: the epilogue's ~60 loop-invariant LDS / global addresses all derive from it, so hipcc
  // cannot hoist them out of the persistent tile loop into long-lived registers
  asm volatile("" : "+v"(lane));
  if (m0w + 64 <= g.m)
    gemm_epilogue256_lds_impl<EPI, true, MF16>(g, acc, region, m0w, nbase, lane);
  else
    gemm_epilogue256_lds_impl<EPI, false, MF16>(g, acc, region, m0w, nbase, lane);
}

template <int EPI, bool GLDS>
__global__ __launch_bounds__(256, 2) void gemm_f16_kernel(GemmArgs g, int tiles_n, int tiles_m) {
  using Cfg = GemmCfg;
  __shared__ __attribute__((aligned(16))) char lds[Cfg::LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_n = wave >> 1, wave_m = wave & 1;
  // tile order: n fastest inside a run of one XCD -> the A panel (128 rows x K) stays
  // in that XCD's L2 while the W panels stream through it.
  const int t = xcd_remap(blockIdx.x, tiles_n * tiles_m);
  const int tn = t % tiles_n, tm = t / tiles_n;
  const int n0 = tn * 128;
  const int64_t m0 = (int64_t)tm * 128;
  const int nkc = (g.k + 63) / 64;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  // The tile engine addresses rows as base + row * d: lda == ldw == k is required
  // (checked on the host); "d" is the row length in elements.
  if constexpr (GLDS) {
    // K % 64 == 0: LDS-DMA straight into the double buffer, next stage issued before the MFMAs
    // of the current one; __syncthreads() waits for the DMA (vmcnt) and the barrier.
    sim_stage_glds<_Float16, Cfg>(lds, g.w, n0, g.n - 1, g.a, m0, g.m - 1, g.k, 0, tid);
    for (int kc = 0; kc < nkc; ++kc) {
      const int cur = kc & 1;
      sim_glds_retire_and_sync();  // stage kc landed for every wave; slot cur^1 is free
      if (kc + 1 < nkc)
        sim_stage_glds<_Float16, Cfg>(lds + (cur ^ 1) * Cfg::STAGE_BYTES, g.w, n0, g.n - 1, g.a, m0,
                                      g.m - 1, g.k, kc + 1, tid);
      sim_stage_mfma<_Float16, Cfg, 2>(acc, lds + cur * Cfg::STAGE_BYTES, wave_n, wave_m, lane);
    }
  } else {
    // ragged K: register-staged, zero-filled past K
    u32x4 regs[Cfg::NLOAD];
    sim_stage_load<_Float16, Cfg>(regs, g.w, n0, g.n - 1, g.a, m0, g.m - 1, g.k, 0, tid);
    sim_stage_store<Cfg>(regs, lds, tid);
    __syncthreads();
    for (int kc = 0; kc < nkc; ++kc) {
      const int cur = kc & 1;
      if (kc + 1 < nkc)
        sim_stage_load<_Float16, Cfg>(regs, g.w, n0, g.n - 1, g.a, m0, g.m - 1, g.k, kc + 1, tid);
      sim_stage_mfma<_Float16, Cfg, 2>(acc, lds + cur * Cfg::STAGE_BYTES, wave_n, wave_m, lane);
      if (kc + 1 < nkc) sim_stage_store<Cfg>(regs, lds + (cur ^ 1) * Cfg::STAGE_BYTES, tid);
      __syncthreads();
    }
  }
  gemm_epilogue<EPI>(g, acc, m0, n0, wave_n, wave_m, lane);
}

// ---------------------------------------------------------------------------
// Big-tile GEMM: 256(n) x 256(m) per workgroup, 8 waves as 2(n) x 4(m), each wave
// 128(n) x 64(m) = 4 x 2 MFMA 32x32x16 tiles (128 accumulator registers), persistent
// over tiles.  Why this geometry (measured, DESIGN.md "GEMM ablation"): with 64 x 64
// wave tiles the LDS pipe (fragment reads + DMA writes, ~170 B/clk of 256) is the limit
// and compute alone tops out at 1.24 PF; 128 x 64 wave tiles cut LDS reads per MFMA by
// 25 % and L2->LDS traffic per flop by 2x.  Rows stay 128 B (BK = 64): with 64-B row
// pieces (BK = 32) the LDS-DMA path moved 2.2-2.8x fewer bytes per second.
//
// A stage is 512 rows x 128 B = 64 KB; two slots.  Step s: wait for stage s (vmcnt(0)),
// barrier, then the eight DMA pieces of stage s+1 are issued between the fragment reads
// and the MFMAs of the FIRST two k-substeps, so that they have the rest of the step (24+
// MFMAs per wave) to land.  Stages run on across tile boundaries: the next tile's first
// stage flies under the epilogue.  Requires K % 64 == 0.
// ---------------------------------------------------------------------------
#ifdef HCIR_DIAG_GSTAMPS
// diagnostic build only (tools/diag_gemm_stamps.py): 100 MHz wall-clock stamps around the SECOND tile of every
// workgroup of the 256 x 256 kernel
__device__ unsigned long long g_gemm_stamps[256 * 8];
#define HCIR_GSTAMP(cond, i)                                                                   \
  do {                                                                                         \
    if ((cond) && threadIdx.x == 0 && blockIdx.x < 256)                                        \
      g_gemm_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime();                  \
  } while (0)
// per-wavefront stamps of the overlapped boundary (gemm_f16_ov_kernel): [0..7] start, [8..15] end, [16..23] next step
__device__ unsigned long long g_gemm_wstamps[256 * 24];
#define HCIR_WSTAMP(cond, i)                                                                              \
  do {                                                                                                    \
    if ((cond) && (threadIdx.x & 63) == 0 && blockIdx.x < 256)                                            \
      g_gemm_wstamps[blockIdx.x * 24 + 8 * (i) + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memrealtime();  \
  } while (0)
#else
#define HCIR_GSTAMP(cond, i) \
  do {                       \
  } while (0)
#define HCIR_WSTAMP(cond, i) \
  do {                       \
  } while (0)
#endif

#ifndef HCIR_GEMM_NGROUP
#define HCIR_GEMM_NGROUP 3  // tools/ab_gemm.py, batch 880: fc1 908 -> 890 us, qkv 574 -> 568 us (0 = n fastest over the whole N)
#endif
#ifndef HCIR_GEMM_NGROUP_WIDE
#define HCIR_GEMM_NGROUP_WIDE 6  // group size when it divides tiles_n (fc1: 12 n-tiles): the activation panels are
                                 // then fetched by two XCD sets instead of four (PMC: fc1 reads x5.3 -> see DESIGN);
                                 // time flat against 3 (profiles/r3_diag_ab_gemm_ngroup.txt), fabric bytes -20 %
#endif

struct G256 {
  static constexpr int NT = 512;
  static constexpr int ROWS = 512;               // 256 W rows (n) then 256 activation rows (m)
  static constexpr int STAGE_BYTES = ROWS * 128; // 64 KB
  static constexpr int NLOAD = ROWS * 8 / NT;    // 16-B pieces per thread per stage = 8
};

template <int EPI, bool MF16>
__global__ __launch_bounds__(512, 2) void gemm_f16_big_kernel(GemmArgs g, int tiles_n, int tiles_m) {
  __shared__ __attribute__((aligned(16))) char lds[2 * G256::STAGE_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_n = wave >> 2, wave_m = wave & 3;
  const int r = lane & 31, h = lane >> 5;
  const int ntiles = tiles_n * tiles_m;
  const int nkc = g.k / 64;
  const int my_tiles =
      (int)blockIdx.x < ntiles ? (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  const int nsteps = my_tiles * nkc;

  auto tile_origin = [&](int ti, int& n0, int64_t& m0) {
    const int t = xcd_remap((int)blockIdx.x + ti * (int)gridDim.x, ntiles);
#if HCIR_GEMM_NGROUP > 0
    // n-grouped order: the W panels of one group of HCIR_GEMM_NGROUP n-tiles stay in the XCD's L2 while every
    // m-tile streams past them (a W set wider than the 4 MB L2 - fc1: 12 panels, 4.7 MB - is otherwise re-fetched
    // for every row of tiles)
    const int ngrp = (HCIR_GEMM_NGROUP_WIDE > 0 && tiles_n % HCIR_GEMM_NGROUP_WIDE == 0 && tiles_n > HCIR_GEMM_NGROUP_WIDE)
                         ? HCIR_GEMM_NGROUP_WIDE : HCIR_GEMM_NGROUP;
    if (tiles_n % ngrp == 0 && tiles_n > ngrp) {
      const int per = ngrp * tiles_m;
      const int grp = t / per, rem = t - grp * per;
      n0 = (grp * ngrp + rem % ngrp) * 256;
      m0 = (int64_t)(rem / ngrp) * 256;
      return;
    }
#endif
    n0 = (t % tiles_n) * 256;
    m0 = (int64_t)(t / tiles_n) * 256;
  };

  // per-thread DMA source offsets of the tile being issued (row + swizzled chunk), in bytes from the
  // tile's W / activation origin: 32-bit VGPR offsets on wave-uniform 64-bit bases (global saddr mode)
  uint32_t soff[G256::NLOAD];
  const char* wbase = nullptr;
  const char* abase = nullptr;
  auto set_sources = [&](int t_i) {
    int n0;
    int64_t m0;
    tile_origin(t_i, n0, m0);
    wbase = reinterpret_cast<const char*>(g.w + (int64_t)n0 * g.ldw);
    abase = reinterpret_cast<const char*>(g.a + m0 * g.lda);
    // opaque thread id: the per-piece row / chunk values below are cheap to recompute once per tile; hoisted out of
    // the tile loop they stayed live through the main loop (and were SPILLED in the dual-output variant, whose
    // per-tile reload then waited - vmcnt(0) - for the stage in flight)
    int otid = tid;
    asm volatile("" : "+v"(otid));
#pragma unroll
    for (int i = 0; i < G256::NLOAD; ++i) {
      const int piece = otid + G256::NT * i;
      const int row = piece >> 3, chunk = (piece & 7) ^ ((row >> 1) & 7);
      if (row < 256) {
        const int nr = n0 + row > g.n - 1 ? g.n - 1 - n0 : row;
        soff[i] = (uint32_t)(((int64_t)nr * g.ldw + chunk * 8) * 2);
      } else {
        int64_t mr = row - 256;
        mr = m0 + mr > g.m - 1 ? g.m - 1 - m0 : mr;
        soff[i] = (uint32_t)((mr * g.lda + chunk * 8) * 2);
      }
    }
  };
  int issue_ti = 0, issue_kc = 0;  // next stage to issue
  auto issue_piece = [&](int slot, int i) {
    // pieces 0..3 are W rows, 4..7 activation rows (i is a constant after unrolling).  Default cache policy
    // on both operands: `nt` (aux 2) on the activation rows measured +-1 %, on the W rows -9..-13 %
    // (every CU of an XCD re-reads them from L2)
#ifndef HCIR_GEMM_BUILTIN_DMA   // the transfer issued outside the compiler's view (common.h lds_dma16): +1.3 % over
                                // the layer against the builtin (qkv +2.6 %), which stays behind this flag for A/B runs
#ifdef HCIR_GEMM_A_NT   // EXPERIMENT (build flag): activation rows with the non-temporal policy, so that the streamed
                        // A panels do not push the W panels out of the XCD's L2
    if (i >= 4)
      lds_dma16_nt(abase + issue_kc * 128, soff[i],
                   lds_addr(lds) + slot * G256::STAGE_BYTES + ((tid & ~63) + G256::NT * i) * 16);
    else
#endif
    lds_dma16((i < 4 ? wbase : abase) + issue_kc * 128, soff[i],
              lds_addr(lds) + slot * G256::STAGE_BYTES + ((tid & ~63) + G256::NT * i) * 16);
#else
    const char* sp = (i < 4 ? wbase : abase) + issue_kc * 128 + soff[i];
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)sp,
        (__attribute__((address_space(3))) void*)(lds + slot * G256::STAGE_BYTES +
                                                   ((tid & ~63) + G256::NT * i) * 16),
        16, 0, 0);
#endif
  };
  auto issue_advance = [&]() {
    if (++issue_kc == nkc) {
      issue_kc = 0;
      ++issue_ti;
      if (issue_ti < my_tiles) set_sources(issue_ti);
    }
  };

  WaveAcc<MF16> acc;
  acc.zero();

#ifdef HCIR_GEMM_SKEW_CLK
  // EXPERIMENT (build flag): every other workgroup of an XCD starts half a tile late, so that the epilogue
  // store bursts of the two halves of an XCD's CUs do not coincide
  if ((blockIdx.x >> 3) & 1) {
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    const uint64_t wait = (uint64_t)HCIR_GEMM_SKEW_CLK * (uint64_t)nkc;
    while (__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(8);
  }
#endif

  if (nsteps > 0) {
    set_sources(0);
#pragma unroll
    for (int i = 0; i < G256::NLOAD; ++i) issue_piece(0, i);
    issue_advance();
  }

  int kc = 0, ti = 0;
  for (int step = 0; step < nsteps; ++step) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    HCIR_GSTAMP(ti == 1 && kc == 0, 0);
    HCIR_GSTAMP(ti == 2 && kc == 0, 4);
#ifdef HCIR_DIAG_GSTAMPS
    // shader-clock counter at the same two points: (s_memtime delta) / (s_memrealtime delta) x 100 MHz = the clock
    // the chip holds inside the kernel (MI355X_MICROARCH.md, DVFS give-back item 6)
    if (kc == 0 && (ti == 1 || ti == 2) && threadIdx.x == 0 && blockIdx.x < 256)
      g_gemm_stamps[blockIdx.x * 8 + (ti == 1 ? 6 : 7)] = __builtin_amdgcn_s_memtime();
#endif

    const char* st = lds + (step & 1) * G256::STAGE_BYTES;
    const bool do_issue = step + 1 < nsteps;
    const int islot = (step + 1) & 1;
    if constexpr (!MF16) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int chunk = 2 * ks + h;
        u32x4 af[4], bf[2];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          af[nt] = *reinterpret_cast<const u32x4*>(st + sim_slot_off(wave_n * 128 + nt * 32 + r, chunk));
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          bf[mt] = *reinterpret_cast<const u32x4*>(st + sim_slot_off(256 + wave_m * 64 + mt * 32 + r, chunk));
        if (do_issue && ks < 2) {  // four of the eight DMA pieces of stage step+1 per early k-substep
#pragma unroll
          for (int i = 0; i < 4; ++i) issue_piece(islot, 4 * ks + i);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
            acc.a[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                __builtin_bit_cast(f16x8, af[nt]), __builtin_bit_cast(f16x8, bf[mt]), acc.a[nt][mt], 0, 0, 0);
      }
    } else {
      // 16x16x32: lane (r16 = lane&15, kq = lane>>4) holds row r16, k = 32*ks2 + 8*kq .. +7
      const int r16 = lane & 15, kq = lane >> 4;
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2) {
        const int chunk = 4 * ks2 + kq;
        u32x4 bf[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          bf[mt] = *reinterpret_cast<const u32x4*>(st + sim_slot_off(256 + wave_m * 64 + mt * 16 + r16, chunk));
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          u32x4 af[4];
#pragma unroll
          for (int q = 0; q < 4; ++q)
            af[q] = *reinterpret_cast<const u32x4*>(
                st + sim_slot_off(wave_n * 128 + (4 * half + q) * 16 + r16, chunk));
          if (do_issue && ks2 == 0) {  // the eight DMA pieces of stage step+1 in the first 32-k substep
#pragma unroll
            for (int i = 0; i < 4; ++i) issue_piece(islot, 4 * half + i);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
              acc.a[4 * half + q][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                  __builtin_bit_cast(f16x8, af[q]), __builtin_bit_cast(f16x8, bf[mt]), acc.a[4 * half + q][mt],
                  0, 0, 0);
        }
      }
    }
    if (do_issue) issue_advance();
    HCIR_GSTAMP(ti == 2 && kc == 0, 5);

    if (++kc == nkc) {
      int n0;
      int64_t m0;
      tile_origin(ti, n0, m0);
      HCIR_GSTAMP(ti == 1, 1);
      // all waves are done reading slot step&1 (their MFMAs have consumed it) after this barrier;
      // the slot stays free until the DMA of stage step+2 is issued behind the next step's barrier
      __builtin_amdgcn_s_barrier();
      HCIR_GSTAMP(ti == 1, 2);
      gemm_epilogue256_lds<EPI, MF16>(g, acc, lds + (step & 1) * G256::STAGE_BYTES + wave * 8192,
                                      m0 + wave_m * 64, n0 + wave_n * 128, lane);
      HCIR_GSTAMP(ti == 1, 3);
      acc.zero();
      kc = 0;
      ++ti;
      // the DMA source offsets of the tile being issued are RE-DERIVED here instead of living through the epilogue:
      // the dual-output variant spilled them, and its reload made the compiler put vmcnt(0) in front of every DMA
      // group of the main loop (MFMA busy 39 % against 57 % for the other variants)
      if (issue_ti < my_tiles) set_sources(issue_ti);
    }
  }
}

// ---------------------------------------------------------------------------
// EXPERIMENT (build flag HCIR_GEMM_OVERLAP, round 4; NOT the default - measured flat to 5 % slower):
// the 256 x 256 kernel with the epilogue of tile t run INSIDE the first k-step of tile t+1 ("overlapped boundary"),
// for the fp16-output epilogues without a residual read (qkv, fc1: 62 % of the encoder's GEMM time).
//
// Idea: at K = 768 a tile is 12 k-steps of 1.65 us plus 5-6 us (qkv) to 9-10 us (fc1 + GELU) in which the matrix
// pipes and the L2 -> LDS fill both stand still.  There is no room to park a finished tile (the accumulators are half
// of the CU's registers, the two stages 128 of its 160 KB of LDS), so the boundary is cut into EIGHT sub-passes of 4
// accumulator blocks (64 features x 16 rows): finish the blocks in fp32 on the accumulator side -> start the SAME
// blocks of the next tile (their MFMAs of k-step 0 with C = 0, fragments from the stage that has landed) -> store.
// An accumulator register is re-used by the next tile the moment its old value has been packed; the step behind the
// boundary waits with vmcnt(16) (its stage was issued BEFORE the sixteen stores; vector-memory operations retire in
// issue order); per-tile vectors (bias, c1, LayerNorm mean / rstd) are staged in 4 KB of LDS during the tile's last
// k-step so that they are read with ds_read instead of queueing behind the stores.  The k order of every
// accumulator is unchanged: results are bit-identical to gemm_f16_big_kernel / gemm_f16_mid_kernel
// (tools/cmp_gemm_variants.py, tests/test_vit_gpu.py::test_gemm_overlapped_tile_boundary).
//
// What the stamps say (tools/diag_gemm_stamps.py ... ov, profiles/r4_gemm_boundary_stamps.txt; batch 880):
//   * the boundary step takes 5.6 us (qkv) / 9.4 us (fc1) against 6.5 / 11 for "barrier + epilogue + wait + first
//     k-step" of the plain kernel, and the launch is flat: qkv 594 vs 592 us, fc1 932 vs 918 us with an LDS
//     transposition image (first form), 613 vs 580 / 922 vs 901 us with the register exchange below;
//   * per wavefront: the SIMD's first wavefront runs its boundary at full speed (2.4 us qkv, 4.6 us fc1), the
//     second one only THEN (leaves at 4.5 / 8.3 us): the two instruction streams of a SIMD serialise, so the step
//     costs the SUM of both, ~6.7 cycles per instruction, 3.5 x the essential VALU + MFMA issue time;
//   * it is not the stores' destination (every tile storing to the same 256 rows: unchanged), not HBM, not the
//     instruction count alone (1000 -> 700 per wavefront moved the first wavefront, not the step);
//   * with the finish arithmetic and the stores compiled out the tile period is 20.5 us (qkv and fc1: 1.21-1.23 PF)
//     against 24-25 / 27-28: that, not more, is what a perfect boundary would buy (-15 % on the two shapes).
// Tiles that are not followed by another tile of the workgroup, and the ragged last row of tiles, take the plain
// boundary (gemm_epilogue256_lds).  Requires K >= 128.
// ---------------------------------------------------------------------------
struct GOv {
  static constexpr int VEC_OFF = 2 * G256::STAGE_BYTES;   // bias[256] | c1[256] | (mean, rstd)[256]
  [[maybe_unused]] static constexpr int LDS_BYTES = VEC_OFF + 4096;   // 132 KB
};

// Second form, written for instruction count (the first one - LDS transposition image, per-store 64-bit address
// arithmetic - spent ~1000 instructions per wavefront on 128 x 64 outputs, this one ~650):
//   * no LDS image: two v_permlane16_swap_b32 per pair of 16-feature blocks leave every lane with 16 contiguous
//     bytes of one output row (lanes 0-15 <-> 16-31 and 32-47 <-> 48-63 trade halves: a lane then holds features
//     8 (q16 >> 1) .. +7 of block nt + (q16 & 1)), stored as 16 rows x 64 B per instruction;
//   * store addresses = wave-uniform base (SGPRs) + ONE per-lane 32-bit offset for the whole boundary;
//   * every LDS address = one of six per-lane bases + an immediate.
__device__ __forceinline__ f16x2 cvt_pk_rne(float a, float b) {
  return __builtin_convertvector((f32x2){a, b}, f16x2);
}

template <int EPI>
__device__ __forceinline__ void gemm_ov_boundary(const GemmArgs& g, WaveAcc<true>& acc, const char* st,
                                                 const char* vec, int64_t m0w, int nbase, int wave_n, int wave_m,
                                                 int lane) {
  constexpr bool kLn = (EPI == EPI_LN_BIAS_F16 || EPI == EPI_LN_BIAS_GELU_F16);
  constexpr bool kGelu = (EPI == HCIR_EPI_BIAS_GELU_F16 || EPI == EPI_LN_BIAS_GELU_F16);
  asm volatile("" : "+v"(lane));   // opaque: none of the addresses below may be hoisted into the main loop
  const int r16 = lane & 15, q16 = lane >> 4;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // per-lane bases; everything else is an immediate (row blocks are 16 rows = 2048 B apart, and a block's swizzle
  // only depends on r16: sim_slot_off(row + 16 j, c) = sim_slot_off(row, c) + 2048 j)
  const char* af0 = st + sim_slot_off(wave_n * 128 + r16, q16);
  const char* af1 = st + sim_slot_off(wave_n * 128 + r16, 4 + q16);
  const char* bf0 = st + sim_slot_off(256 + wave_m * 64 + r16, q16);
  const char* bf1 = st + sim_slot_off(256 + wave_m * 64 + r16, 4 + q16);
  const char* vb = vec + (wave_n * 128 + 4 * q16) * 4;
  const char* lnb = vec + 2048 + (wave_m * 64 + r16) * 8;
  const uint32_t voff = (uint32_t)(((int64_t)r16 * g.ldo + 16 * (q16 & 1) + 8 * (q16 >> 1)) * 2);
#ifndef HCIR_OV_ABL
#define HCIR_OV_ABL 0   // timing ablations (WRONG results): 3 = every tile stores to the first 256 rows, 4 = plain (not nt) stores
#endif
  char* const obase = reinterpret_cast<char*>(static_cast<_Float16*>(g.out) + (HCIR_OV_ABL == 3 ? (m0w & 255) : m0w) * g.ldo + nbase);
#pragma unroll
  for (int hn = 0; hn < 2; ++hn) {
    // W fragments of the next tile's k-step 0 for these four n-blocks (both 32-k halves) and the blocks' bias / c1,
    // kept over the four row blocks
    u32x4 afH[2][4];
    f32x4 bH[4], cH[kLn ? 4 : 1];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      afH[0][q] = *reinterpret_cast<const u32x4*>(af0 + 2048 * (4 * hn + q));
      afH[1][q] = *reinterpret_cast<const u32x4*>(af1 + 2048 * (4 * hn + q));
      bH[q] = *reinterpret_cast<const f32x4*>(vb + 64 * (4 * hn + q));
      if constexpr (kLn) cH[q] = *reinterpret_cast<const f32x4*>(vb + 1024 + 64 * (4 * hn + q));
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const u32x4 b0 = *reinterpret_cast<const u32x4*>(bf0 + 2048 * mt);
      const u32x4 b1 = *reinterpret_cast<const u32x4*>(bf1 + 2048 * mt);
      f32x2 ln = {0.f, 1.f};
      if constexpr (kLn) ln = *reinterpret_cast<const f32x2*>(lnb + 128 * mt);
      // ---- finish the four blocks (same arithmetic, same order as gemm_epilogue256_lds_impl)
      u32x2 o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nt = 4 * hn + q;
        float xs[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = acc.a[nt][mt][e];
          if constexpr (kLn)
            x = __builtin_fmaf(ln[1], __builtin_fmaf(-ln[0], cH[q][e], x), bH[q][e]);
          else
            x += bH[q][e];
          xs[e] = x;
        }
        if constexpr (kGelu) {
#pragma unroll
          for (int e = 0; e < 4; e += 2) {
            const gelu_f32x2 y = gelu_erf2((gelu_f32x2){xs[e], xs[e + 1]});
            xs[e] = y[0];
            xs[e + 1] = y[1];
          }
        }
        o[q][0] = __builtin_bit_cast(uint32_t, cvt_pk_rne(xs[0], xs[1]));
        o[q][1] = __builtin_bit_cast(uint32_t, cvt_pk_rne(xs[2], xs[3]));
      }
      // ---- the same blocks of the next tile: k-step 0
#pragma unroll
      for (int q = 0; q < 4; ++q)
        acc.a[4 * hn + q][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
            __builtin_bit_cast(f16x8, afH[0][q]), __builtin_bit_cast(f16x8, b0), zero4, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        acc.a[4 * hn + q][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
            __builtin_bit_cast(f16x8, afH[1][q]), __builtin_bit_cast(f16x8, b1), acc.a[4 * hn + q][mt], 0, 0, 0);
      // ---- 16 rows x 64 B per store: blocks (nt, nt + 1) side by side
      char* const orow = obase + ((int64_t)(16 * mt) * g.ldo + 64 * hn) * 2;   // wave-uniform
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const auto s0 = __builtin_amdgcn_permlane16_swap(o[2 * j][0], o[2 * j + 1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(o[2 * j][1], o[2 * j + 1][1], false, false);
        const u32x4 v = {s0[0], s1[0], s0[1], s1[1]};
#if HCIR_OV_ABL == 4
        *reinterpret_cast<u32x4*>(orow + 64 * j + voff) = v;
#else
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(orow + 64 * j + voff));
#endif
      }
      // keep hipcc from hoisting the next sub-passes' reads up here (registers); the hardware overlaps the next
      // sub-pass's arithmetic with these MFMAs anyway (in-order issue, asynchronous matrix pipe)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_f16_ov_kernel(GemmArgs g, int tiles_n, int tiles_m) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr bool kLn = (EPI == EPI_LN_BIAS_F16 || EPI == EPI_LN_BIAS_GELU_F16);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: everything derived from it stays in SGPRs
  const int wave_n = wave >> 2, wave_m = wave & 3;
  const int ntiles = tiles_n * tiles_m;
  const int nkc = g.k / 64;
  const int my_tiles =
      (int)blockIdx.x < ntiles ? (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  const int nsteps = my_tiles * nkc;

  auto tile_origin = [&](int ti, int& n0, int64_t& m0) {
    const int t = xcd_remap((int)blockIdx.x + ti * (int)gridDim.x, ntiles);
#if HCIR_GEMM_NGROUP > 0
    const int ngrp = (HCIR_GEMM_NGROUP_WIDE > 0 && tiles_n % HCIR_GEMM_NGROUP_WIDE == 0 && tiles_n > HCIR_GEMM_NGROUP_WIDE)
                         ? HCIR_GEMM_NGROUP_WIDE : HCIR_GEMM_NGROUP;
    if (tiles_n % ngrp == 0 && tiles_n > ngrp) {
      const int per = ngrp * tiles_m;
      const int grp = t / per, rem = t - grp * per;
      n0 = (grp * ngrp + rem % ngrp) * 256;
      m0 = (int64_t)(rem / ngrp) * 256;
      return;
    }
#endif
    n0 = (t % tiles_n) * 256;
    m0 = (int64_t)(t / tiles_n) * 256;
  };

  uint32_t soff[G256::NLOAD];
  const char* wbase = nullptr;
  const char* abase = nullptr;
  auto set_sources = [&](int t_i) {
    int n0;
    int64_t m0;
    tile_origin(t_i, n0, m0);
    wbase = reinterpret_cast<const char*>(g.w + (int64_t)n0 * g.ldw);
    abase = reinterpret_cast<const char*>(g.a + m0 * g.lda);
    int otid = tid;
    asm volatile("" : "+v"(otid));
#pragma unroll
    for (int i = 0; i < G256::NLOAD; ++i) {
      const int piece = otid + G256::NT * i;
      const int row = piece >> 3, chunk = (piece & 7) ^ ((row >> 1) & 7);
      if (row < 256) {
        const int nr = n0 + row > g.n - 1 ? g.n - 1 - n0 : row;
        soff[i] = (uint32_t)(((int64_t)nr * g.ldw + chunk * 8) * 2);
      } else {
        int64_t mr = row - 256;
        mr = m0 + mr > g.m - 1 ? g.m - 1 - m0 : mr;
        soff[i] = (uint32_t)((mr * g.lda + chunk * 8) * 2);
      }
    }
  };
  int issue_ti = 0, issue_kc = 0;  // next stage to issue
  auto issue_piece = [&](int slot, int i) {
    lds_dma16((i < 4 ? wbase : abase) + issue_kc * 128, soff[i],
              lds_addr(lds) + slot * G256::STAGE_BYTES + ((tid & ~63) + G256::NT * i) * 16);
  };
  auto issue_advance = [&]() {
    if (++issue_kc == nkc) {
      issue_kc = 0;
      ++issue_ti;
      if (issue_ti < my_tiles) set_sources(issue_ti);
    }
  };

  WaveAcc<true> acc;
  acc.zero();

  if (nsteps > 0) {
    set_sources(0);
#pragma unroll
    for (int i = 0; i < G256::NLOAD; ++i) issue_piece(0, i);
    issue_advance();
  }

  // tile ti ends in the overlapped boundary when the workgroup has another tile behind it and all 256 rows exist
  int n0c = 0;
  int64_t m0c = 0;
  bool ov = false;
  auto enter_tile = [&](int ti) {
    tile_origin(ti, n0c, m0c);
    ov = ti + 1 < my_tiles && m0c + 256 <= g.m;
  };
  if (my_tiles > 0) enter_tile(0);

  const int r16 = lane & 15, kq = lane >> 4;
  int kc = 0, ti = 0;
  bool stores_behind = false;   // the step follows a boundary step: its stage is older than that step's 16 stores
  for (int step = 0; step < nsteps; ++step) {
    if (stores_behind)
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stores_behind = false;
    __builtin_amdgcn_s_barrier();
    HCIR_GSTAMP(ti == 1 && kc == 1, 0);   // diagnostic build: first ordinary k-step of the workgroup's second tile
    HCIR_GSTAMP(ti == 2 && kc == 1, 5);
    HCIR_WSTAMP(ti == 2 && kc == 1, 2);
#ifdef HCIR_DIAG_GSTAMPS
    if (kc == 1 && (ti == 1 || ti == 2) && threadIdx.x == 0 && blockIdx.x < 256)
      g_gemm_stamps[blockIdx.x * 8 + (ti == 1 ? 6 : 7)] = __builtin_amdgcn_s_memtime();
#endif

    const char* st = lds + (step & 1) * G256::STAGE_BYTES;
    const bool do_issue = step + 1 < nsteps;
    const int islot = (step + 1) & 1;
    // last k-step of a tile that ends in the overlapped boundary: request its per-tile vectors (older than this step's
    // transfers, so they are back before the boundary's wait); they go to LDS at the top of the boundary step
    const bool fetch_vec = ov && kc == nkc - 1;
    float v0 = 0.f, v1 = 0.f;
    if (fetch_vec) {
      if (tid < 256) {
        if (g.bias) v0 = g.bias[n0c + tid];
      } else if constexpr (kLn) {
        v0 = g.ln_c1[n0c + tid - 256];
      }
      if constexpr (kLn) v1 = g.ln_stats[2 * m0c + tid];   // 256 full rows: in range
    }
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2) {
      const int chunk = 4 * ks2 + kq;
      u32x4 bf[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        bf[mt] = *reinterpret_cast<const u32x4*>(st + sim_slot_off(256 + wave_m * 64 + mt * 16 + r16, chunk));
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        u32x4 af[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          af[q] = *reinterpret_cast<const u32x4*>(st + sim_slot_off(wave_n * 128 + (4 * half + q) * 16 + r16, chunk));
        if (do_issue && ks2 == 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) issue_piece(islot, 4 * half + i);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
            acc.a[4 * half + q][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                __builtin_bit_cast(f16x8, af[q]), __builtin_bit_cast(f16x8, bf[mt]), acc.a[4 * half + q][mt], 0, 0, 0);
      }
    }
    if (do_issue) issue_advance();

    if (++kc == nkc) {
      if (ov) {
        // ---- boundary step = epilogue of tile ti + k-step 0 of tile ti+1 (step + 1)
        ++step;
        HCIR_GSTAMP(ti == 1, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        char* vec = lds + GOv::VEC_OFF;
        reinterpret_cast<float*>(vec)[tid] = v0;
        if constexpr (kLn) reinterpret_cast<float*>(vec)[512 + tid] = v1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        HCIR_GSTAMP(ti == 1, 2);
        const char* st1 = lds + (step & 1) * G256::STAGE_BYTES;
        if (step + 1 < nsteps) {   // always: nkc >= 2
#pragma unroll
          for (int i = 0; i < G256::NLOAD; ++i) issue_piece((step + 1) & 1, i);
          issue_advance();
        }
        HCIR_GSTAMP(ti == 1, 3);
        HCIR_WSTAMP(ti == 1, 0);
        gemm_ov_boundary<EPI>(g, acc, st1, vec,
                              m0c + wave_m * 64, n0c + wave_n * 128, wave_n, wave_m, lane);
        HCIR_GSTAMP(ti == 1, 4);
        HCIR_WSTAMP(ti == 1, 1);
        stores_behind = true;
        kc = 1;
      } else {
        __builtin_amdgcn_s_barrier();
        gemm_epilogue256_lds<EPI, true>(g, acc, lds + (step & 1) * G256::STAGE_BYTES + wave * 8192,
                                        m0c + wave_m * 64, n0c + wave_n * 128, lane);
        acc.zero();
        kc = 0;
      }
      ++ti;
      if (ti < my_tiles) enter_tile(ti);
      if (issue_ti < my_tiles) set_sources(issue_ti);
    }
  }
}

// ---------------------------------------------------------------------------
// Mid-tile GEMM: 128(n) x 192(m) per workgroup of FOUR waves (2 x 2, wave tile 64 x 96 = 4 x 6 MFMA 16x16x32
// tiles, 96 accumulator registers), 2 x 40 KB LDS slots -> TWO workgroups per CU (2 x 80 KB = the CU's 160 KB).
//
// Why: the 256 x 256 kernel runs ONE workgroup per CU, so a CU's matrix pipes idle for the whole epilogue of
// every tile (22 % of the launch at K = 768: LDS transposition, bias / GELU / LayerNorm-fold math, 128 KB of
// stores per tile, all CUs at the same moment).  Two independent workgroups per CU fall into anti-phase by
// themselves: while one transposes and stores, the other owns the matrix pipes, and the output bursts of a CU
// (and of the chip) spread over the tile time.  Price: 1.67 x the L2->LDS bytes per flop of the 256^2 tile and
// 10 instead of 12 fragment reads per 24 instead of 32 MFMAs (+11 % LDS reads per MFMA).
// Same staging (LDS-DMA, 128-B rows, XOR swizzle), same MFMA operand maps and k order as the 256^2 kernel:
// results are bit-identical to it.  Requires K % 64 == 0, N % 128 == 0.
// ---------------------------------------------------------------------------
struct GMid {
  static constexpr int NT = 256;
  static constexpr int TN = 128, TM = 192;
  static constexpr int ROWS = TN + TM;             // 128 W rows (n) then 192 activation rows (m)
  static constexpr int STAGE_BYTES = ROWS * 128;   // 40 KB
  static constexpr int NLOAD = ROWS * 8 / NT;      // 10 pieces of 16 B per thread per stage (4 W + 6 activation)
  static constexpr int NW = TN * 8 / NT;           // 4
};

// Epilogue of a 16x16x32 accumulator block acc[NTW n-tiles][.. MTH m-tiles from mt0]: (16 NTW) features x (16 MTH)
// rows, transposed through `region` (16 MTH rows x 128 B, private to the wave) one 128-B output line per row and
// pass; the arithmetic of gemm_epilogue256_lds_impl (same order of operations -> same bits).
template <int EPI, bool FULL, int NTW, int MTT, int MTH>
__device__ __forceinline__ void gemm_epilogue16_lds_impl(const GemmArgs& g, const f32x4 (&acc)[NTW][MTT], int mt0,
                                                         char* region, int64_t m0h, int nbase, int lane) {
  constexpr bool kLn = (EPI == EPI_LN_BIAS_F16 || EPI == EPI_LN_BIAS_GELU_F16);
  constexpr bool kGelu = (EPI == HCIR_EPI_BIAS_GELU_F16 || EPI == EPI_LN_BIAS_GELU_F16);
  constexpr bool kResidH = (EPI == HCIR_EPI_BIAS_RESID_F16 || EPI == EPI_RESID_F16_STATS);
  constexpr bool kF16 = (EPI == HCIR_EPI_BIAS_F16 || kGelu || kLn || EPI == HCIR_EPI_AFFINE_RELU_F16 || kResidH);
  constexpr bool kAffine = (EPI == HCIR_EPI_AFFINE_RELU_F16 || EPI == HCIR_EPI_AFFINE_F32);
  constexpr int FPP = kF16 ? 64 : 32;      // features per pass = one 128-B line per row
  constexpr int NPASS = NTW * 16 / FPP;
  constexpr int NB = kF16 ? 2 : 1;
  constexpr int NIT = MTH * 2;             // groups of 8 rows
  constexpr int UB = (NIT % 4 == 0) ? 4 : 3;
  static_assert(NIT % UB == 0 && (NTW * 16) % FPP == 0, "geometry");
  const int rrow = lane >> 3, rchunk = lane & 7;
  const int r16 = lane & 15, q16 = lane >> 4;

  f32x4 bias[NPASS][NB], scale[NPASS][NB];
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
    if constexpr (kF16) break;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int n = nbase + pass * FPP + rchunk * (kF16 ? 8 : 4) + 4 * j;
      bias[pass][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      scale[pass][j] = (f32x4){1.f, 1.f, 1.f, 1.f};
      if (g.bias) bias[pass][j] = *reinterpret_cast<const f32x4*>(g.bias + n);
      if (kAffine || ((EPI == HCIR_EPI_BIAS_RESID_F32 || kResidH) && g.scale))
        scale[pass][j] = *reinterpret_cast<const f32x4*>(g.scale + n);
    }
  }
  // fp16 outputs: bias / scale / activation / LayerNorm fold run on the ACCUMULATOR side, in fp32, before the ONE
  // rounding to fp16, exactly as in gemm_epilogue256_lds_impl (the earlier form rounded the raw accumulator first:
  // a second rounding, one fp16 ulp away from the 256 x 256 kernel's result)
  constexpr bool kScaleA = kF16 && (EPI == HCIR_EPI_AFFINE_RELU_F16 || kResidH);
  f32x4 biasA[kF16 ? NTW : 1], scaleA[kScaleA ? NTW : 1];
  if constexpr (kF16) {
    const bool has_scale = kScaleA && (EPI == HCIR_EPI_AFFINE_RELU_F16 || g.scale != nullptr);
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int n = nbase + 16 * t + 4 * q16;
      biasA[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (g.bias) biasA[t] = *reinterpret_cast<const f32x4*>(g.bias + n);
      if constexpr (kScaleA) {
        scaleA[t] = (f32x4){1.f, 1.f, 1.f, 1.f};
        if (has_scale) scaleA[t] = *reinterpret_cast<const f32x4*>(g.scale + n);
      }
    }
  }
  float ln_rs[MTH], ln_mean[MTH];
  f32x4 c1a[NTW];
  if constexpr (kLn) {
#pragma unroll
    for (int mt = 0; mt < MTH; ++mt) {
      int64_t mm = m0h + 16 * mt + r16;
      mm = mm < g.m ? mm : g.m - 1;
      ln_mean[mt] = g.ln_stats[2 * mm];
      ln_rs[mt] = g.ln_stats[2 * mm + 1];
    }
#pragma unroll
    for (int t = 0; t < NTW; ++t) c1a[t] = *reinterpret_cast<const f32x4*>(g.ln_c1 + nbase + 16 * t + 4 * q16);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (kLn) {
#pragma unroll
    for (int mt = 0; mt < MTH; ++mt) {
      asm volatile("" : "+v"(ln_rs[mt]));
      asm volatile("" : "+v"(ln_mean[mt]));
    }
#pragma unroll
    for (int t = 0; t < NTW; ++t) launder(c1a[t]);
  }
  if constexpr (kF16) {
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      launder(biasA[t]);
      if constexpr (kScaleA) launder(scaleA[t]);
    }
  }
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
    if constexpr (kF16) break;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      launder(bias[pass][j]);
      launder(scale[pass][j]);
    }
  }

#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
    // ---- accumulators -> LDS (lane = output row m, registers = features n)
#pragma unroll
    for (int mt = 0; mt < MTH; ++mt) {
      const int row = mt * 16 + r16;
      if constexpr (kF16) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int nt = 4 * pass + q;
          f16x4 o;
          float xs[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float x = acc[nt][mt0 + mt][e];
            const float bb = biasA[nt][e];
            if constexpr (kLn) {
              // out = rstd[m] (acc - mean[m] c1[n]) + bias[n]: centered first (cancellation when |mean| >> std)
              x = __builtin_fmaf(ln_rs[mt], __builtin_fmaf(-ln_mean[mt], c1a[nt][e], x), bb);
            } else if constexpr (EPI == HCIR_EPI_AFFINE_RELU_F16) {
              x = fmaxf(__builtin_fmaf(x, scaleA[nt][e], bb), 0.f);
            } else if constexpr (kResidH) {
              x = scaleA[nt][e] * (x + bb);
            } else {
              x += bb;
            }
            xs[e] = x;
          }
          if constexpr (kGelu) {
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
              const gelu_f32x2 y = gelu_erf2((gelu_f32x2){xs[e], xs[e + 1]});
              xs[e] = y[0];
              xs[e + 1] = y[1];
            }
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (_Float16)xs[e];
          const int chunk = 2 * q + (q16 >> 1);
          *reinterpret_cast<f16x4*>(region + row * 128 + ((chunk ^ (row & 7)) << 4) + 8 * (q16 & 1)) = o;
        }
      } else {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int nt = 2 * pass + q;
          const int chunk = 4 * q + q16;
          *reinterpret_cast<f32x4*>(region + row * 128 + ((chunk ^ (row & 7)) << 4)) = acc[nt][mt0 + mt];
        }
      }
    }
    // ---- LDS -> rows: lane = (row rrow + 8 it, 16-B chunk rchunk)
    if constexpr (kF16) {
      const int n = nbase + pass * 64 + rchunk * 8;
#pragma unroll
      for (int it0 = 0; it0 < NIT; it0 += UB) {
        f16x8 oldh[UB];
        if constexpr (kResidH) {
#pragma unroll
          for (int u = 0; u < UB; ++u) {
            const int64_t mm = m0h + (it0 + u) * 8 + rrow;
#pragma unroll
            for (int e = 0; e < 8; ++e) oldh[u][e] = (_Float16)0.f;
            if (FULL || mm < g.m)
              oldh[u] = *reinterpret_cast<const f16x8*>(static_cast<const _Float16*>(g.resid) + mm * g.ldo + n);
          }
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const int row = (it0 + u) * 8 + rrow;
          const f16x8 v = *reinterpret_cast<const f16x8*>(region + row * 128 + ((rchunk ^ (row & 7)) << 4));
          // the image already holds the finished fp16 values; the fp16 residual is one packed add per pair (the
          // exact sum of two fp16 numbers, rounded once)
          f16x8 o;
          if constexpr (kResidH)
            o = v + oldh[u];
          else
            o = v;
          const int64_t m = m0h + row;
          if (FULL || m < g.m)
            __builtin_nontemporal_store(o, reinterpret_cast<f16x8*>(static_cast<_Float16*>(g.out) + m * g.ldo + n));
          if constexpr (EPI == EPI_RESID_F16_STATS) {
            float xv[8], s1 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              xv[e] = (float)o[e];
              s1 += xv[e];
            }
            const float mean_s = sum8_dpp(s1) * (1.0f / 64.0f);
            float m2 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float dv = xv[e] - mean_s;
              m2 = __builtin_fmaf(dv, dv, m2);
            }
            m2 = sum8_dpp(m2);
            if (rchunk == 0 && (FULL || m < g.m)) {
              const int64_t slice = (nbase >> 6) + pass;
              *reinterpret_cast<f32x2*>(g.stats_part + (slice * (g.stats_ld ? g.stats_ld : g.m) + m) * 2) = (f32x2){mean_s, m2};
            }
          }
        }
      }
    } else {
      const int n = nbase + pass * 32 + rchunk * 4;
      const f32x4 b = bias[pass][0], sc = scale[pass][0];
#pragma unroll
      for (int it0 = 0; it0 < NIT; it0 += UB) {
        f32x4 oldv[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const int64_t mm = m0h + (it0 + u) * 8 + rrow;
          oldv[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if constexpr (EPI == HCIR_EPI_BIAS_RESID_F32) {
            if (FULL || mm < g.m)
              oldv[u] = *reinterpret_cast<const f32x4*>(static_cast<const float*>(g.resid) + mm * g.ldo + n);
          }
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const int row = (it0 + u) * 8 + rrow;
          const int64_t mm = m0h + row;
          f32x4 v = *reinterpret_cast<const f32x4*>(region + row * 128 + ((rchunk ^ (row & 7)) << 4));
          if constexpr (EPI == HCIR_EPI_AFFINE_F32) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(v[e], sc[e], b[e]);
          } else if constexpr (EPI == HCIR_EPI_BIAS_RESID_F32) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(sc[e], v[e] + b[e], oldv[u][e]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += b[e];
          }
          if (FULL || mm < g.m) *reinterpret_cast<f32x4*>(static_cast<float*>(g.out) + mm * g.ldo + n) = v;
        }
      }
    }
  }
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_f16_mid_kernel(GemmArgs g, int tiles_n, int tiles_m) {
  __shared__ __attribute__((aligned(16))) char lds[2 * GMid::STAGE_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_n = wave >> 1, wave_m = wave & 1;
  const int ntiles = tiles_n * tiles_m;
  const int nkc = g.k / 64;
  const int my_tiles =
      (int)blockIdx.x < ntiles ? (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  const int nsteps = my_tiles * nkc;

  auto tile_origin = [&](int ti, int& n0, int64_t& m0) {
    const int t = xcd_remap((int)blockIdx.x + ti * (int)gridDim.x, ntiles);
    n0 = (t % tiles_n) * GMid::TN;
    m0 = (int64_t)(t / tiles_n) * GMid::TM;
  };

  uint32_t soff[GMid::NLOAD];
  const char* wbase = nullptr;
  const char* abase = nullptr;
  auto set_sources = [&](int t_i) {
    int n0;
    int64_t m0;
    tile_origin(t_i, n0, m0);
    wbase = reinterpret_cast<const char*>(g.w + (int64_t)n0 * g.ldw);
    abase = reinterpret_cast<const char*>(g.a + m0 * g.lda);
#pragma unroll
    for (int i = 0; i < GMid::NLOAD; ++i) {
      const int piece = tid + GMid::NT * i;
      const int row = piece >> 3, chunk = (piece & 7) ^ ((row >> 1) & 7);
      if (row < GMid::TN) {
        const int nr = n0 + row > g.n - 1 ? g.n - 1 - n0 : row;
        soff[i] = (uint32_t)(((int64_t)nr * g.ldw + chunk * 8) * 2);
      } else {
        int64_t mr = row - GMid::TN;
        mr = m0 + mr > g.m - 1 ? g.m - 1 - m0 : mr;
        soff[i] = (uint32_t)((mr * g.lda + chunk * 8) * 2);
      }
    }
  };
  int issue_ti = 0, issue_kc = 0;
  auto issue_piece = [&](int slot, int i) {  // i constant after unrolling: pieces 0..3 W rows, 4..9 activation rows
#ifndef HCIR_GEMM_BUILTIN_DMA
    lds_dma16((i < GMid::NW ? wbase : abase) + issue_kc * 128, soff[i],
              lds_addr(lds) + slot * GMid::STAGE_BYTES + ((tid & ~63) + GMid::NT * i) * 16);
#else
    const char* sp = (i < GMid::NW ? wbase : abase) + issue_kc * 128 + soff[i];
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)sp,
        (__attribute__((address_space(3))) void*)(lds + slot * GMid::STAGE_BYTES + ((tid & ~63) + GMid::NT * i) * 16),
        16, 0, 0);
#endif
  };
  auto issue_advance = [&]() {
    if (++issue_kc == nkc) {
      issue_kc = 0;
      ++issue_ti;
      if (issue_ti < my_tiles) set_sources(issue_ti);
    }
  };

  f32x4 acc[4][6];
  auto zero_acc = [&]() {
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int y = 0; y < 6; ++y) acc[x][y] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  zero_acc();

  if (nsteps > 0) {
    set_sources(0);
#pragma unroll
    for (int i = 0; i < GMid::NLOAD; ++i) issue_piece(0, i);
    issue_advance();
  }

  const int r16 = lane & 15, kq = lane >> 4;
  int kc = 0, ti = 0;
  for (int step = 0; step < nsteps; ++step) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const char* st = lds + (step & 1) * GMid::STAGE_BYTES;
    const bool do_issue = step + 1 < nsteps;
    const int islot = (step + 1) & 1;
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2) {
      const int chunk = 4 * ks2 + kq;
      u32x4 bf[6], af[4];
#pragma unroll
      for (int mt = 0; mt < 6; ++mt)
        bf[mt] = *reinterpret_cast<const u32x4*>(st + sim_slot_off(GMid::TN + wave_m * 96 + mt * 16 + r16, chunk));
#pragma unroll
      for (int q = 0; q < 4; ++q)
        af[q] = *reinterpret_cast<const u32x4*>(st + sim_slot_off(wave_n * 64 + q * 16 + r16, chunk));
#pragma unroll
      for (int hq = 0; hq < 2; ++hq) {
        if (do_issue && ks2 == 0) {  // the ten DMA pieces of stage step+1 in the first 32-k substep
#pragma unroll
          for (int i = 0; i < 5; ++i) issue_piece(islot, 5 * hq + i);
        }
#pragma unroll
        for (int q = 2 * hq; q < 2 * hq + 2; ++q)
#pragma unroll
          for (int mt = 0; mt < 6; ++mt)
            acc[q][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af[q]),
                                                               __builtin_bit_cast(f16x8, bf[mt]), acc[q][mt], 0, 0, 0);
      }
    }
    if (do_issue) issue_advance();

    if (++kc == nkc) {
      int n0;
      int64_t m0;
      tile_origin(ti, n0, m0);
      // every wave is done reading slot step&1; it stays free until the DMA of stage step+2 goes out behind the
      // next step's barrier.  Per wave a private 6 KB piece of it: 48 rows x 128 B, two row halves per tile.
      __builtin_amdgcn_s_barrier();
      char* region = lds + (step & 1) * GMid::STAGE_BYTES + wave * 6144;
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));  // opaque: the epilogue's addresses are not hoisted out of the tile loop
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int64_t m0h = m0 + wave_m * 96 + hh * 48;
        if (m0h + 48 <= g.m)
          gemm_epilogue16_lds_impl<EPI, true, 4, 6, 3>(g, acc, 3 * hh, region, m0h, n0 + wave_n * 64, lane_o);
        else
          gemm_epilogue16_lds_impl<EPI, false, 4, 6, 3>(g, acc, 3 * hh, region, m0h, n0 + wave_n * 64, lane_o);
      }
      zero_acc();
      kc = 0;
      ++ti;
    }
  }
}

inline bool gemm_takes_mid(int64_t m, int n, int k) { return k % 64 == 0 && m >= 1024 && n % 128 == 0; }

template <int EPI>
void launch_gemm_mid(const GemmArgs& g, hipStream_t st) {
  const int tn = g.n / GMid::TN, tm = (int)hcir_cdiv(g.m, GMid::TM);
  const int grid = tn * tm < 512 ? tn * tm : 512;  // persistent: two workgroups per CU
  hipLaunchKernelGGL((gemm_f16_mid_kernel<EPI>), dim3(grid), dim3(256), 0, st, g, tn, tm);
}

// ---------------------------------------------------------------------------
// Patch embedding: Conv2d(C, D, P, P) as a GEMM whose A operand is gathered from
// the fp32 NCHW image on the fly (no im2col buffer), P == 16.
//   k = c*256 + ky*16 + kx ; a 64-wide k chunk = 4 image rows of one channel;
//   a 16-B LDS slot = 8 consecutive pixels of one row (32 B of fp32 source).
// ---------------------------------------------------------------------------
struct PatchArgs {
  int p, kreal, kpad;  // patch size, C*P*P, row length of the (zero-padded) weight matrix
  const float* img;
  const _Float16* w;
  const float* bias;
  const float* pos;
  void* tok;
  int64_t b;
  int c, h, w_px, gh, gw, d;
  float pos_mult;
};

template <typename TokT, bool P16>
__global__ __launch_bounds__(256) void patch_embed_kernel(PatchArgs p, int tiles_n, int tiles_m) {
  using Cfg = GemmCfg;
  __shared__ __attribute__((aligned(16))) char lds[Cfg::LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_n = wave >> 1, wave_m = wave & 1;
  const int t = xcd_remap(blockIdx.x, tiles_n * tiles_m);
  const int tn = t % tiles_n, tm = t / tiles_n;
  const int n0 = tn * 128;
  const int64_t m0 = (int64_t)tm * 128;
  const int np = p.gh * p.gw;
  const int64_t mtot = p.b * np;
  const int kdim = p.kpad;
  const int nkc = kdim / 64;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  u32x4 regs[Cfg::NLOAD];
  auto load = [&](int kc) {
#pragma unroll
    for (int i = 0; i < Cfg::NLOAD; ++i) {
      const int slot = tid + 256 * i;
      const int row = slot >> 3, chunk = slot & 7;
      const int k0 = kc * 64 + chunk * 8;
      if (row < Cfg::GM) {
        int nr = n0 + row;
        nr = nr > p.d - 1 ? p.d - 1 : nr;
        regs[i] = *reinterpret_cast<const u32x4*>(p.w + (int64_t)nr * kdim + k0);
      } else {
        int64_t m = m0 + (row - Cfg::GM);
        m = m > mtot - 1 ? mtot - 1 : m;
        const int64_t bi = m / np;
        const int pi = (int)(m % np);
        const int py = pi / p.gw, px = pi % p.gw;
        f16x8 v;
        if constexpr (P16) {
          const int c = k0 >> 8, ky = (k0 >> 4) & 15, kx = k0 & 15;
          const float* src = p.img + ((bi * p.c + c) * p.h + (py * 16 + ky)) * (int64_t)p.w_px + px * 16 + kx;
          const f32x4 lo = *reinterpret_cast<const f32x4*>(src);
          const f32x4 hi = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = (_Float16)lo[e];
            v[4 + e] = (_Float16)hi[e];
          }
        } else {
          // any patch size: k = (c * P + ky) * P + kx, element by element, zero past C*P*P
          const int pp = p.p * p.p;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int k = k0 + e;
            float x = 0.f;
            if (k < p.kreal) {
              const int c = k / pp, rem = k - c * pp;
              const int ky = rem / p.p, kx = rem - ky * p.p;
              x = p.img[((bi * p.c + c) * p.h + (py * p.p + ky)) * (int64_t)p.w_px + px * p.p + kx];
            }
            v[e] = (_Float16)x;
          }
        }
        regs[i] = __builtin_bit_cast(u32x4, v);
      }
    }
  };
  load(0);
  sim_stage_store<Cfg>(regs, lds, tid);
  __syncthreads();
  for (int kc = 0; kc < nkc; ++kc) {
    const int cur = kc & 1;
    if (kc + 1 < nkc) load(kc + 1);
    sim_stage_mfma<_Float16, Cfg, 2>(acc, lds + cur * Cfg::STAGE_BYTES, wave_n, wave_m, lane);
    if (kc + 1 < nkc) sim_stage_store<Cfg>(regs, lds + (cur ^ 1) * Cfg::STAGE_BYTES, tid);
    __syncthreads();
  }
  // epilogue: tok[b][1 + p][n] = acc + bias[n] + pos_mult * pos[1 + p][n]
  const int r = lane & 31, h = lane >> 5;
  if constexpr (sizeof(TokT) == 2) {
    // fp16 token rows: the wave transposes its 64 x 64 tile through 8 KB of the (now idle) stage buffers and
    // stores whole 128-B lines; the direct layout below writes 8 B of 32 different rows per instruction
    // (the same fix took 13 % off the attention kernel)
    char* ot = lds + wave * 8192;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int row = mt * 32 + r;
      int64_t m = m0 + wave_m * 64 + row;
      m = m < mtot ? m : mtot - 1;
      const float* prow = p.pos + (int64_t)(1 + (int)(m % np)) * p.d;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
          int n = n0 + wave_n * 64 + nt * 32 + 8 * grp + 4 * h;
          n = n < p.d ? n : p.d - 4;  // (d is a multiple of 8: a clamped read, the column is not stored)
          const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n);
          const f32x4 ps = *reinterpret_cast<const f32x4*>(prow + n);
          f16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (_Float16)(acc[nt][mt][4 * grp + e] + b[e] + p.pos_mult * ps[e]);
          *reinterpret_cast<f16x4*>(ot + row * 128 + (((4 * nt + grp) ^ (row & 7)) << 4) + 8 * h) = o;
        }
      }
    }
    const int rr = lane >> 3, cc = lane & 7;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = it * 8 + rr;
      const int64_t m = m0 + wave_m * 64 + row;
      const int n = n0 + wave_n * 64 + cc * 8;
      const u32x4 v = *reinterpret_cast<const u32x4*>(ot + row * 128 + ((cc ^ (row & 7)) << 4));
      if (m < mtot && n < p.d) {
        const int64_t bi = m / np;
        const int pi = (int)(m % np);
        *reinterpret_cast<u32x4*>(static_cast<_Float16*>(p.tok) + (bi * (np + 1) + 1 + pi) * (int64_t)p.d + n) = v;
      }
    }
    return;
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int64_t m = m0 + wave_m * 64 + mt * 32 + r;
    if (m >= mtot) continue;
    const int64_t bi = m / np;
    const int pi = (int)(m % np);
    TokT* orow = static_cast<TokT*>(p.tok) + (bi * (np + 1) + 1 + pi) * (int64_t)p.d;
    const float* prow = p.pos + (int64_t)(1 + pi) * p.d;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
      for (int grp = 0; grp < 4; ++grp) {
        const int n = n0 + wave_n * 64 + nt * 32 + 8 * grp + 4 * h;
        if (n >= p.d) continue;
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n);
        const f32x4 ps = *reinterpret_cast<const f32x4*>(prow + n);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[nt][mt][4 * grp + e] + b[e] + p.pos_mult * ps[e];
        if constexpr (sizeof(TokT) == 4) {
          *reinterpret_cast<f32x4*>(orow + n) = v;
        } else {
          f16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (_Float16)v[e];
          *reinterpret_cast<f16x4*>(orow + n) = o;
        }
      }
    }
  }
}

template <typename TokT>
__global__ void cls_row_kernel(const float* __restrict__ cls, const float* __restrict__ pos,
                               float pos_mult, int64_t b, int t, int d, TokT* __restrict__ tok) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b * d) return;
  const int64_t bi = i / d;
  const int n = (int)(i % d);
  tok[bi * t * (int64_t)d + n] = (TokT)(cls[n] + pos_mult * pos[n]);
}

inline bool gemm_takes_big(int64_t m, int n, int k) { return k % 64 == 0 && m >= 1024 && n % 256 == 0; }

template <int EPI>
void launch_gemm_mid(const GemmArgs& g, hipStream_t st);

// The 256 x 256 persistent kernel over tn x tm tiles (with -DHCIR_GEMM_OVERLAP: the overlapped-boundary experiment
// for the epilogues it covers, when some workgroup gets a second tile).
template <int EPI>
void launch_big_tiles(const GemmArgs& g, int tn, int tm, hipStream_t st) {
  const int grid = tn * tm < 256 ? tn * tm : 256;  // persistent: one workgroup per CU
#ifdef HCIR_GEMM_MFMA32
  hipLaunchKernelGGL((gemm_f16_big_kernel<EPI, false>), dim3(grid), dim3(512), 0, st, g, tn, tm);
#else
#ifdef HCIR_GEMM_OVERLAP   // EXPERIMENT (build flag; measured flat to -5 %, see the kernel's header): off by default
  if constexpr (EPI == HCIR_EPI_BIAS_F16 || EPI == HCIR_EPI_BIAS_GELU_F16 || EPI == EPI_LN_BIAS_F16 ||
                EPI == EPI_LN_BIAS_GELU_F16) {
    if (g.k >= 128 && tn * tm > grid) {
      // 148 KB of dynamic LDS: one attribute call per instantiation; a refusal falls back to the plain kernel
      static const bool attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16_ov_kernel<EPI>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   GOv::LDS_BYTES) == hipSuccess;
      if (attr) {
        hipLaunchKernelGGL((gemm_f16_ov_kernel<EPI>), dim3(grid), dim3(512), GOv::LDS_BYTES, st, g, tn, tm);
        return;
      }
    }
  }
#endif
  hipLaunchKernelGGL((gemm_f16_big_kernel<EPI, true>), dim3(grid), dim3(512), 0, st, g, tn, tm);
#endif
}

template <int EPI>
void launch_gemm_big(const GemmArgs& g, hipStream_t st) {
#ifdef HCIR_GEMM_MID
  if (g.n % 128 == 0) {   // EXPERIMENT (build flag): every persistent-kernel shape on the 2-workgroups-per-CU kernel
    launch_gemm_mid<EPI>(g, st);
    return;
  }
#endif
  const int tn = (int)hcir_cdiv(g.n, 256), tm = (int)hcir_cdiv(g.m, 256);
#if !defined(HCIR_GEMM_NO_TAIL_SPLIT) && !defined(HCIR_GEMM_MFMA32)
  // Tail split: when the last round of 256 x 256 tiles would be less than half full (64 images: fc1 = 600 tiles = 2.34
  // rounds; ViT-L/14 at 128 images: 516 / 1548 / 2064 tiles = 2.02 / 6.05 / 8.06 rounds - a whole round for a sliver),
  // the m-tiles that fill whole rounds run here and the remaining ROWS go to the 128 x 192 kernel as a second launch
  // (pointers advanced to the row range; the per-row statistics keep the full matrix's slice pitch).
  if constexpr (EPI != EPI_BIAS_F16_DUAL_GELU) {
    const int ntiles = tn * tm;
    const int tm1 = (ntiles / 256) * 256 / tn;          // m-tiles of the whole rounds
    const int tail_tiles = ntiles - tm1 * tn;
    const int64_t m1 = (int64_t)tm1 * 256;
    if (ntiles > 256 && tm1 > 0 && tm1 < tm && tail_tiles < 128 && g.n % GMid::TN == 0 &&
        (g.n / GMid::TN) * hcir_cdiv(g.m - m1, GMid::TM) <= 512) {
      GemmArgs g1 = g, g2 = g;
      g1.m = m1;
      g1.stats_ld = g.stats_ld ? g.stats_ld : g.m;
      g2.m = g.m - m1;
      g2.stats_ld = g1.stats_ld;
      g2.a = g.a + m1 * g.lda;
      const bool f32out = EPI == HCIR_EPI_BIAS_RESID_F32 || EPI == HCIR_EPI_BIAS_F32 || EPI == HCIR_EPI_AFFINE_F32;
      const int64_t obytes = m1 * g.ldo * (f32out ? 4 : 2);
      g2.out = static_cast<char*>(g.out) + obytes;
      if (g.resid) g2.resid = static_cast<const char*>(g.resid) + obytes;
      if (g.ln_stats) g2.ln_stats = g.ln_stats + 2 * m1;
      if (g.stats_part) g2.stats_part = g.stats_part + 2 * m1;
      launch_big_tiles<EPI>(g1, tn, tm1, st);
      launch_gemm_mid<EPI>(g2, st);
      return;
    }
  }
#endif
#if !defined(HCIR_GEMM_NO_MID_SMALL) && !defined(HCIR_GEMM_MFMA32)
  // Small M: a launch whose 256 x 256 tiles fill less than 0.7 of ONE round of the 256 CUs (64 images: proj / fc2 give
  // 150 tiles) runs on the 128 x 192 kernel at two workgroups per CU instead - 2.7 x the tiles, every CU busy; its
  // results are bit-identical (same k order).  12 % slower per flop on full rounds, which is why only these take it.
  if constexpr (EPI != EPI_BIAS_F16_DUAL_GELU) {
    if (tn * tm * 10 < 256 * 7 && g.n % GMid::TN == 0 && (g.n / GMid::TN) * hcir_cdiv(g.m, GMid::TM) <= 512) {
      launch_gemm_mid<EPI>(g, st);
      return;
    }
  }
#endif
  // MFMA shape: a BUILD flag (make CXXFLAGS+=-DHCIR_GEMM_MFMA32 builds the 32x32x16 variant for A/B runs through
  // HCIR_LIB_PATH); the library reads no environment variables
  launch_big_tiles<EPI>(g, tn, tm, st);
}

// (mean, M2) of the 64-feature slices of hcir_gemm_f16_fused -> (mean, rstd) per row by Chan's parallel-variance
// combination (equal slice sizes), slices visited in index order: deterministic, no cancellation
__global__ void ln_stats_finalize_kernel(const float* __restrict__ part, int parts, int64_t m, float slice_n,
                                         float eps, float* __restrict__ stats) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= m) return;
  float ms = 0.f;
  for (int p = 0; p < parts; ++p) ms += part[((int64_t)p * m + row) * 2];
  const float mean = ms / (float)parts;
  float m2 = 0.f;
  for (int p = 0; p < parts; ++p) {
    const f32x2 v = *reinterpret_cast<const f32x2*>(part + ((int64_t)p * m + row) * 2);
    const float dm = v[0] - mean;
    m2 += v[1] + slice_n * dm * dm;
  }
  const float var = m2 / (slice_n * (float)parts);
  *reinterpret_cast<f32x2*>(stats + 2 * row) = (f32x2){mean, 1.0f / sqrtf(var + eps)};
}

template <int EPI>
void launch_gemm(const GemmArgs& g, hipStream_t st) {
  if (gemm_takes_big(g.m, g.n, g.k)) {
    launch_gemm_big<EPI>(g, st);
    return;
  }
  const int tiles_n = (int)hcir_cdiv(g.n, 128), tiles_m = (int)hcir_cdiv(g.m, 128);
  if (g.k % 64 == 0)
    hipLaunchKernelGGL((gemm_f16_kernel<EPI, true>), dim3(tiles_n * tiles_m), dim3(256), 0, st, g,
                       tiles_n, tiles_m);
  else
    hipLaunchKernelGGL((gemm_f16_kernel<EPI, false>), dim3(tiles_n * tiles_m), dim3(256), 0, st, g,
                       tiles_n, tiles_m);
}

}  // namespace

extern "C" {

int hcir_gemm_f16(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias,
                  const float* scale, int64_t m, int32_t n, int32_t k, int epilogue, void* out,
                  int64_t ldo, void* stream) {
  return hcir_gemm_f16_resid(a, lda, w, ldw, bias, scale, m, n, k, epilogue, nullptr, out, ldo, stream);
}

int hcir_gemm_f16_resid(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias,
                        const float* scale, int64_t m, int32_t n, int32_t k, int epilogue, const void* resid,
                        void* out, int64_t ldo, void* stream) {
  HCIR_ENTER();
  if (resid && epilogue != HCIR_EPI_BIAS_RESID_F16 && epilogue != HCIR_EPI_BIAS_RESID_F32) return HCIR_ERR_INVALID;
  if (!a || !w || !out || m <= 0 || n <= 0 || k <= 0) return HCIR_ERR_INVALID;
  if ((k & 7) || (n & 7) || lda < k || ldw < k || (lda & 7) || (ldw & 7) || ldo < n || (ldo & 3))
    return HCIR_ERR_INVALID;
  // row pitches other than k: only the 256x256 persistent kernel carries separate pitches
  if ((lda != k || ldw != k) && !(k % 64 == 0 && m >= 1024 && n % 256 == 0)) return HCIR_ERR_UNSUPPORTED;
  if ((epilogue == HCIR_EPI_AFFINE_RELU_F16 || epilogue == HCIR_EPI_AFFINE_F32) && !scale)
    return HCIR_ERR_INVALID;
  if (hcir_cdiv(m, 128) * hcir_cdiv(n, 128) > 0x7fffffff) return HCIR_ERR_INVALID;
  GemmArgs g{static_cast<const _Float16*>(a), static_cast<const _Float16*>(w), bias, scale, out, m,
             lda, ldw, ldo, n, k, nullptr, nullptr, nullptr, resid ? resid : out, nullptr};
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (epilogue) {
    case HCIR_EPI_BIAS_F16: launch_gemm<HCIR_EPI_BIAS_F16>(g, st); break;
    case HCIR_EPI_BIAS_GELU_F16: launch_gemm<HCIR_EPI_BIAS_GELU_F16>(g, st); break;
    case HCIR_EPI_BIAS_RESID_F32: launch_gemm<HCIR_EPI_BIAS_RESID_F32>(g, st); break;
    case HCIR_EPI_BIAS_F32: launch_gemm<HCIR_EPI_BIAS_F32>(g, st); break;
    case HCIR_EPI_AFFINE_RELU_F16: launch_gemm<HCIR_EPI_AFFINE_RELU_F16>(g, st); break;
    case HCIR_EPI_AFFINE_F32: launch_gemm<HCIR_EPI_AFFINE_F32>(g, st); break;
    case HCIR_EPI_BIAS_RESID_F16: launch_gemm<HCIR_EPI_BIAS_RESID_F16>(g, st); break;
    default: return HCIR_ERR_UNSUPPORTED;
  }
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}

int hcir_gemm_f16_gelu_dual(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias, int64_t m,
                            int32_t n, int32_t k, void* out_pre, void* out_act, int64_t ldo, void* stream) {
  HCIR_ENTER();
  if (!a || !w || !out_pre || !out_act || m <= 0 || n <= 0 || k <= 0) return HCIR_ERR_INVALID;
  if (lda < k || ldw < k || (lda & 7) || (ldw & 7) || ldo < n || (ldo & 7)) return HCIR_ERR_INVALID;
  if (!gemm_takes_big(m, n, k)) return HCIR_ERR_UNSUPPORTED;
#if defined(HCIR_GEMM_MFMA32) || defined(HCIR_GEMM_MID)
  return HCIR_ERR_UNSUPPORTED;
#else
  GemmArgs g{static_cast<const _Float16*>(a), static_cast<const _Float16*>(w), bias, nullptr, out_pre, m,
             lda, ldw, ldo, n, k, nullptr, nullptr, nullptr, out_pre, out_act};
  launch_gemm_big<EPI_BIAS_F16_DUAL_GELU>(g, static_cast<hipStream_t>(stream));
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
#endif
}

#ifdef HCIR_DIAG_GSTAMPS
int hcir_debug_gemm_stamps(unsigned long long* host_dst) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_gemm_stamps), sizeof(unsigned long long) * 256 * 8);
}
int hcir_debug_gemm_wstamps(unsigned long long* host_dst) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_gemm_wstamps), sizeof(unsigned long long) * 256 * 24);
}
#endif

int hcir_gemm_fused_supported(int64_t m, int32_t n, int32_t k) { return gemm_takes_big(m, n, k) ? 1 : 0; }

int32_t hcir_gemm_stats_slices(int32_t n) { return n / 64; }

int hcir_gemm_f16_fused(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias,
                        const float* scale, int64_t m, int32_t n, int32_t k, int epilogue, void* out,
                        int64_t ldo, const float* ln_stats, const float* ln_c1, float* stats_part,
                        void* stream) {
  HCIR_ENTER();
  if (!a || !w || !out || m <= 0 || n <= 0 || k <= 0) return HCIR_ERR_INVALID;
  if ((k & 7) || (n & 7) || lda < k || ldw < k || (lda & 7) || (ldw & 7) || ldo < n || (ldo & 7))
    return HCIR_ERR_INVALID;
  const bool ln = ln_stats || ln_c1;
  if (ln && (!ln_stats || !ln_c1 || !bias)) return HCIR_ERR_INVALID;
  if (ln && stats_part) return HCIR_ERR_INVALID;
  if (!ln && !stats_part) return HCIR_ERR_INVALID;  // nothing fused: use hcir_gemm_f16
  if (ln && epilogue != HCIR_EPI_BIAS_F16 && epilogue != HCIR_EPI_BIAS_GELU_F16) return HCIR_ERR_UNSUPPORTED;
  if (stats_part && epilogue != HCIR_EPI_BIAS_RESID_F16) return HCIR_ERR_UNSUPPORTED;
  if (!gemm_takes_big(m, n, k)) return HCIR_ERR_UNSUPPORTED;
  GemmArgs g{static_cast<const _Float16*>(a), static_cast<const _Float16*>(w), bias, scale, out, m,
             lda, ldw, ldo, n, k, ln_stats, ln_c1, stats_part, out, nullptr};
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (stats_part) launch_gemm_big<EPI_RESID_F16_STATS>(g, st);
  else if (epilogue == HCIR_EPI_BIAS_F16) launch_gemm_big<EPI_LN_BIAS_F16>(g, st);
  else launch_gemm_big<EPI_LN_BIAS_GELU_F16>(g, st);
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}

int hcir_ln_stats_finalize(const float* stats_part, int32_t slices, int64_t m, int32_t n_features, float eps,
                           float* ln_stats, void* stream) {
  HCIR_ENTER();
  if (!stats_part || !ln_stats || slices <= 0 || m <= 0 || n_features <= 0) return HCIR_ERR_INVALID;
  if ((int64_t)slices * 64 != n_features) return HCIR_ERR_INVALID;  // slices are 64 features wide
  hipLaunchKernelGGL(ln_stats_finalize_kernel, dim3((unsigned)hcir_cdiv(m, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), stats_part, slices, m, 64.0f, eps, ln_stats);
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}

int hcir_patch_embed(const float* img, int64_t b, int32_t c, int32_t h, int32_t w_px, int32_t p,
                     const void* w_f16, int64_t ldw, const float* bias, const float* cls,
                     const float* pos, float pos_mult, int32_t d, void* tok, int tok_dtype,
                     void* stream) {
  HCIR_ENTER();
  if (!img || !w_f16 || !bias || !cls || !pos || !tok || b <= 0) return HCIR_ERR_INVALID;
  if (p <= 0 || p > 32 || c <= 0 || h % p || w_px % p || (d & 7)) return HCIR_ERR_INVALID;
  const int kreal = c * p * p;
  if (ldw < kreal || (ldw & 63)) return HCIR_ERR_INVALID;  // weight rows zero-padded to a multiple of 64
  PatchArgs a{p, kreal, (int)ldw, img, static_cast<const _Float16*>(w_f16), bias, pos, tok, b, c, h, w_px,
              h / p, w_px / p, d, pos_mult};
  const int np = a.gh * a.gw;
  const int tiles_n = (int)hcir_cdiv(d, 128), tiles_m = (int)hcir_cdiv(b * np, 128);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (tok_dtype != HCIR_F32 && tok_dtype != HCIR_F16) return HCIR_ERR_UNSUPPORTED;
  const dim3 cgrid((unsigned)hcir_cdiv(b * d, 256));
  if (tok_dtype == HCIR_F32) {
    if (p == 16)
      hipLaunchKernelGGL((patch_embed_kernel<float, true>), dim3(tiles_n * tiles_m), dim3(256), 0, st, a, tiles_n, tiles_m);
    else
      hipLaunchKernelGGL((patch_embed_kernel<float, false>), dim3(tiles_n * tiles_m), dim3(256), 0, st, a, tiles_n, tiles_m);
    HCIR_LAUNCH_CHECK();
    hipLaunchKernelGGL(cls_row_kernel<float>, cgrid, dim3(256), 0, st, cls, pos, pos_mult, b, np + 1, d,
                       static_cast<float*>(tok));
  } else {
    if (p == 16)
      hipLaunchKernelGGL((patch_embed_kernel<_Float16, true>), dim3(tiles_n * tiles_m), dim3(256), 0, st, a, tiles_n, tiles_m);
    else
      hipLaunchKernelGGL((patch_embed_kernel<_Float16, false>), dim3(tiles_n * tiles_m), dim3(256), 0, st, a, tiles_n, tiles_m);
    HCIR_LAUNCH_CHECK();
    hipLaunchKernelGGL(cls_row_kernel<_Float16>, cgrid, dim3(256), 0, st, cls, pos, pos_mult, b, np + 1, d,
                       static_cast<_Float16*>(tok));
  }
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}

}  // extern "C"
