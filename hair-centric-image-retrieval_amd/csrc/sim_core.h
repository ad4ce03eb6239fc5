// sim_core.h — streaming "rows x rows" similarity tiles on MFMA (gfx950).
//
// One workgroup (4 waves) multiplies a tile of GM "stream" rows (gallery rows /
// NT-Xent columns) against QB "resident" rows (queries / NT-Xent rows):
//   S^T[g][q] = sum_k G[g][k] * Q[q][k]
// computed as MFMA 32x32 tiles with the STREAM row on the MFMA row index and the
// RESIDENT row on the MFMA column index, so that one lane owns one resident row
// (column lane&31) and receives 16 stream-row scores per tile in its registers:
// the per-query epilogue (running top-k, online log-sum-exp) is lane-local.
//
// Staging: K is cut in 128-byte chunks per row (64 fp16 / 32 fp32 elements).
// Each stage holds GM + QB rows x 128 B in LDS, XOR-swizzled per 16-B slot with
// ((row >> 1) & 7) so that the ds_read_b128 fragment reads of 16 lanes on 16
// different rows hit 16 different 16-B slots of the 256-B bank row
// (cdna_hip_programming.md §2, T2).  Global loads are 16 B per lane, 8 lanes per
// 128-B line (full lines), register-staged one chunk ahead of the MFMAs (T14).
//
// HCIR_F32 k-order ("sim_topk k-order", mirrored by oracle/knn_oracle.c):
// within each 32-element chunk c the products enter ONE fp32 fmaf chain in the
// order  for cc in 0..3: for e in 0..3: k = 32c + 8cc + e, then k + 4
// because lane-half h of v_mfma_f32_32x32x2_f32 supplies k index h and the
// instruction is a k-ordered fmaf chain (cdna_hip_programming.md §3).
#pragma once
#include "common.h"

template <typename T>
struct SimElem;
template <>
struct SimElem<float> {
  static constexpr int kPerChunk = 4;   // elements per 16-B slot
  static constexpr int kPerStage = 32;  // elements per 128-B row chunk
};
template <>
struct SimElem<_Float16> {
  static constexpr int kPerChunk = 8;
  static constexpr int kPerStage = 64;
};
template <>
struct SimElem<__bf16> {
  static constexpr int kPerChunk = 8;
  static constexpr int kPerStage = 64;
};

// WGG x WQ waves; each wave owns 2 stream tiles x QT resident tiles of 32x32.
template <typename T, int WGG, int WQ, int QT>
struct SimCfg {
  static_assert(WGG * WQ == 4, "4 waves per workgroup");
  static constexpr int NT = 256;             // threads per workgroup
  static constexpr int GM = 64 * WGG;        // stream rows per WG tile
  static constexpr int QB = 32 * QT * WQ;    // resident rows per WG
  static constexpr int ROWS = GM + QB;       // rows staged per chunk
  static constexpr int NLOAD = ROWS * 8 / 256;  // 16-B slots per thread per stage
  static constexpr int STAGE_BYTES = ROWS * 128;
  static constexpr int LDS_BYTES = 2 * STAGE_BYTES;
  static_assert((ROWS * 8) % 256 == 0, "slot count must divide evenly");
};

__device__ __forceinline__ int sim_slot_off(int row, int chunk) {
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// Issue the global loads of one stage (chunk index kc) into registers.
// Stream rows come from g (rows g_row0 .. clamped to g_last), resident rows from
// q (rows q_row0 .. clamped to q_last).  Slots past d read as zero.
template <typename T, typename Cfg>
__device__ __forceinline__ void sim_stage_load(u32x4 (&regs)[Cfg::NLOAD], const T* __restrict__ g,
                                               int64_t g_row0, int64_t g_last,
                                               const T* __restrict__ q, int64_t q_row0,
                                               int64_t q_last, int d, int kc, int tid) {
  constexpr int EPC = SimElem<T>::kPerChunk;
  constexpr int EPS = SimElem<T>::kPerStage;
#pragma unroll
  for (int i = 0; i < Cfg::NLOAD; ++i) {
    const int slot = tid + 256 * i;
    const int row = slot >> 3, chunk = slot & 7;
    const int k0 = kc * EPS + chunk * EPC;
    const T* src;
    if (row < Cfg::GM) {
      int64_t gr = g_row0 + row;
      gr = gr > g_last ? g_last : gr;
      src = g + gr * (int64_t)d + k0;
    } else {
      int64_t qr = q_row0 + (row - Cfg::GM);
      qr = qr > q_last ? q_last : qr;
      src = q + qr * (int64_t)d + k0;
    }
    u32x4 v = {0u, 0u, 0u, 0u};
    if (k0 < d) v = *reinterpret_cast<const u32x4*>(src);
    regs[i] = v;
  }
}

template <typename Cfg>
__device__ __forceinline__ void sim_stage_store(const u32x4 (&regs)[Cfg::NLOAD], char* stage,
                                                int tid) {
#pragma unroll
  for (int i = 0; i < Cfg::NLOAD; ++i) {
    const int slot = tid + 256 * i;
    const int row = slot >> 3, chunk = slot & 7;
    *reinterpret_cast<u32x4*>(stage + sim_slot_off(row, chunk)) = regs[i];
  }
}

// LDS-DMA staging (global_load_lds_dwordx4): one wave instruction writes 64 consecutive
// 16-B slots = 8 rows x 128 B of the stage image; the LDS destination is wave-uniform
// base + lane*16, so the XOR swizzle is applied to the per-lane SOURCE address (the lane
// that fills physical slot p of row r fetches logical chunk p ^ ((r>>1)&7)); the 8 lanes of a
// row still cover one whole 128-B line (cdna_hip_programming.md §5 caveat, rule 21).
// No VGPR round trip and no ds_write: the register-staged path is bound by the
// ds_write_b128 VGPR->LDS transfer (~79 B/clk/CU).  Requires d % (elements per stage) == 0.
template <typename T, typename Cfg, int AUX_STREAM = 0>
__device__ __forceinline__ void sim_stage_glds(char* stage, const T* __restrict__ g, int64_t g_row0,
                                               int64_t g_last, const T* __restrict__ q,
                                               int64_t q_row0, int64_t q_last, int d, int kc,
                                               int tid) {
  constexpr int EPC = SimElem<T>::kPerChunk;
  constexpr int EPS = SimElem<T>::kPerStage;
  const int wave_base = tid & ~63;
#pragma unroll
  for (int i = 0; i < Cfg::NLOAD; ++i) {
    const int slot = tid + Cfg::NT * i;
    const int row = slot >> 3, pslot = slot & 7;
    const int chunk = pslot ^ ((row >> 1) & 7);
    const int k0 = kc * EPS + chunk * EPC;
    const T* src;
    if (row < Cfg::GM) {
      int64_t gr = g_row0 + row;
      gr = gr > g_last ? g_last : gr;
      src = g + gr * (int64_t)d + k0;
    } else {
      int64_t qr = q_row0 + (row - Cfg::GM);
      qr = qr > q_last ? q_last : qr;
      src = q + qr * (int64_t)d + k0;
    }
#ifndef HCIR_SIM_BUILTIN_DMA   // the transfer outside the compiler's view (common.h); the callers retire it
                          // with sim_glds_retire_and_sync().  (No cache-policy bit on this path.)
    if (AUX_STREAM != 0 && ((Cfg::NT * i) / 8 + 63 / 8 < Cfg::GM || row < Cfg::GM))
      lds_dma16_v_nt(src, lds_addr(stage) + (wave_base + Cfg::NT * i) * 16);
    else
      lds_dma16_v(src, lds_addr(stage) + (wave_base + Cfg::NT * i) * 16);
    continue;
#endif
    // rows of one instruction are all stream rows or all resident rows (GM is a multiple of 8)
    if ((Cfg::NT * i) / 8 + 63 / 8 < Cfg::GM || row < Cfg::GM)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)src,
          (__attribute__((address_space(3))) void*)(stage + (wave_base + Cfg::NT * i) * 16), 16, 0,
          AUX_STREAM);
    else
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)src,
          (__attribute__((address_space(3))) void*)(stage + (wave_base + Cfg::NT * i) * 16), 16, 0, 0);
  }
}

// Retire every LDS-DMA this wave has issued, then meet the workgroup.  The explicit wait is
// REQUIRED: hipcc (ROCm 7.2) tracks an LDS-DMA issued under a condition in a loop only up to the
// back-edge and emits `s_waitcnt lgkmcnt(0); s_barrier` without the vmcnt wait there (seen in
// sim_topk_scan: stale fragments, run-to-run different top-k).  asm is invisible to that pass.
__device__ __forceinline__ void sim_glds_retire_and_sync() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// MFMAs of one stage.  acc[gt][qt]: stream tile gt (0,1) x resident tile qt.
template <typename T, typename Cfg, int QT>
__device__ __forceinline__ void sim_stage_mfma(f32x16 (&acc)[2][QT], const char* stage,
                                               int wave_g, int wave_q, int lane) {
  const int r = lane & 31, h = lane >> 5;
  int grow[2], qrow[QT];
#pragma unroll
  for (int gt = 0; gt < 2; ++gt) grow[gt] = wave_g * 64 + gt * 32 + r;
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) qrow[qt] = Cfg::GM + wave_q * (32 * QT) + qt * 32 + r;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int chunk = 2 * s + h;
    u32x4 a[2], b[QT];
#pragma unroll
    for (int gt = 0; gt < 2; ++gt)
      a[gt] = *reinterpret_cast<const u32x4*>(stage + sim_slot_off(grow[gt], chunk));
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
      b[qt] = *reinterpret_cast<const u32x4*>(stage + sim_slot_off(qrow[qt], chunk));
#pragma unroll
    for (int gt = 0; gt < 2; ++gt) {
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        if constexpr (sizeof(T) == 4) {
          const f32x4 af = __builtin_bit_cast(f32x4, a[gt]);
          const f32x4 bf = __builtin_bit_cast(f32x4, b[qt]);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[gt][qt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e], acc[gt][qt], 0, 0, 0);
        } else if constexpr (__is_same(T, _Float16)) {
          acc[gt][qt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
              __builtin_bit_cast(f16x8, a[gt]), __builtin_bit_cast(f16x8, b[qt]), acc[gt][qt], 0,
              0, 0);
        } else {
          acc[gt][qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
              __builtin_bit_cast(bf16x8, a[gt]), __builtin_bit_cast(bf16x8, b[qt]), acc[gt][qt], 0,
              0, 0);
        }
      }
    }
  }
}
