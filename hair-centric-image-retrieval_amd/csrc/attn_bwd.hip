// attn_bwd.hip — backward of the fused multi-head self-attention (hcir_attn_fwd) for ViT token counts.
//
// Replaces autograd through Attention.forward (HP/src/models_vit.py:69-78) / nn.MultiheadAttention inside
// torchvision's EncoderBlock (HP/src/main_backbone.py:554) in the training step (HP/src/pretrain_engine.py:745):
//   P = softmax(scale q k^T)   dV = P^T dO   dP = dO V^T   dS = P o (dP - D),  D_q = sum_d dO[q][d] O[q][d]
//   dQ = scale dS K            dK = scale dS^T Q
// One workgroup per (batch, head) of 8 / KT waves; wave w owns keys [32 KT w, 32 KT (w + 1)) (T <= 256) and keeps dK^T
// and dV^T of its keys in registers for the whole sweep over the query tiles.  KT = 1 (eight waves of one 32-key tile,
// two per SIMD) is what runs: with four waves of two tiles (the first version: ~350 registers, one wave per SIMD)
// every LDS round trip and exp2 chain of the per-tile sequence stood exposed and the fourth wave owned 5 real keys of
// 64 at T = 197 - 1.56 ms per launch at batch 1024 against the forward's 0.31 (cdna_hip_programming.md Appendix B, "Attention
// backward": the KEY sits on the MFMA lane):
//   S  = Q . K^T  and  dP = dO . V^T   MFMA 32x32x16, query on the row, key on the column: the accumulators
//                                      ARE the B operands (contraction over their row index = the query) of
//   dV^T += dO^T . P,  dK^T += Q^T . dS   whose A operands dO^T / Q^T come from the row-major LDS images by
//                                      ds_read_b64_tr_b16, in the permuted k order of an accumulator operand
//                                      (cdna_hip_programming.md §3);
//   dQ^T = K^T . dS^T                  contracts over the key = the lane index: dS crosses LDS once ([q][key] image,
//                                      private to the wave), the per-wave partial dQ tiles (its keys) meet in four
//                                      padded fp32 LDS slabs (KT = 1: wave 2j writes slab j, wave 2j+1 adds to it),
//                                      summed in slab order, and leave as whole fp16 rows: deterministic.
// P is recomputed from the forward's per-row log2-sum-exp (hcir_attn_fwd_lse); D from dO and O at kernel start.
// LDS: Q, dO, K images (3 x 32 KB), dS staging 16 KB, dQ slabs 4 x 8.3 KB, row constants 2 KB: 1 workgroup per CU.
// The images carry the row-read swizzle only; transposed reads see some bank conflicts.
#include <type_traits>

#include "common.h"

namespace {

typedef __fp16 ab_fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

struct AttnBwdArgs {
  const _Float16* qkv;   // [B][T][3][H][64]
  const _Float16* out;   // [B][T][H*64]   forward output O
  const _Float16* dout;  // [B][T][H*64]
  const float* lse;      // [B][H][T]      log2-sum-exp of the forward (scaled scores, log2 domain)
  _Float16* dqkv;        // [B][T][3][H][64]
  int t, h;
  float scale, scale_log2e;
};

constexpr int kTP = 256;  // keys per (b, head): 4 waves x 64

__device__ __forceinline__ int img_off(int row, int c16) { return row * 128 + ((c16 ^ ((row >> 1) & 7)) << 4); }
// byte address of element column e0 (a multiple of 4) of `row`: a transposed read's 8-byte piece
__device__ __forceinline__ int img_off_e(int row, int e0) { return img_off(row, e0 >> 3) + (e0 & 7) * 2; }

template <int KT>
__global__ __launch_bounds__(64 * (8 / KT), 1) void attn_bwd_kernel(AttnBwdArgs a) {
  constexpr int NW = 8 / KT;        // waves
  constexpr int NT = 64 * NW;       // threads
  constexpr int KW = 32 * KT;       // keys per wave
  constexpr int DSROW = 64 * KT;    // bytes per row of a wave's dS image (KW fp16)
  constexpr int DSMASK = 4 * KT - 1;
  auto ds_off = [](int row, int c16) { return row * DSROW + ((c16 ^ ((row >> 1) & DSMASK)) << 4); };
  __shared__ __attribute__((aligned(16))) char lds[3 * kTP * 128 + 4 * 4096 + 4 * 32 * 65 * 4 + 2 * kTP * 4];
  char* qs = lds;
  char* dos = lds + kTP * 128;
  char* ks = lds + 2 * kTP * 128;
  char* dss = lds + 3 * kTP * 128;                                  // [NW waves][32 q][DSROW B]: 16 KB
  float* dqt = reinterpret_cast<float*>(lds + 3 * kTP * 128 + 4 * 4096);  // [4 waves][32][65]
  float* dsum = dqt + 4 * 32 * 65;                                  // D[q]
  float* lrow = dsum + kTP;                                         // lse[q]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int grp = lane >> 4, li = lane & 15;
  const int head = blockIdx.x % a.h;
  const int64_t b = blockIdx.x / a.h;
  const int64_t qkv_stride = (int64_t)3 * a.h * 64, o_stride = (int64_t)a.h * 64;
  const _Float16* qg = a.qkv + b * a.t * qkv_stride + head * 64;
  const _Float16* kg = qg + (int64_t)a.h * 64;
  const _Float16* vg = qg + (int64_t)2 * a.h * 64;
  const _Float16* og = a.out + b * a.t * o_stride + head * 64;
  const _Float16* dog = a.dout + b * a.t * o_stride + head * 64;
  _Float16* dqg = a.dqkv + b * a.t * qkv_stride + head * 64;
  _Float16* dkg = dqg + (int64_t)a.h * 64;
  _Float16* dvg = dqg + (int64_t)2 * a.h * 64;

  // ---- stage Q, dO, K images (rows past T: Q, K clamped to the last real row - finite, masked below; dO zero)
  for (int slot = tid; slot < kTP * 8; slot += NT) {
    const int row = slot >> 3, c = slot & 7;
    const int src = row < a.t ? row : a.t - 1;
    const u32x4 qv = *reinterpret_cast<const u32x4*>(qg + src * qkv_stride + c * 8);
    const u32x4 kv = *reinterpret_cast<const u32x4*>(kg + src * qkv_stride + c * 8);
    u32x4 dv = {0u, 0u, 0u, 0u};
    if (row < a.t) dv = *reinterpret_cast<const u32x4*>(dog + src * o_stride + c * 8);
    *reinterpret_cast<u32x4*>(qs + img_off(row, c)) = qv;
    *reinterpret_cast<u32x4*>(ks + img_off(row, c)) = kv;
    *reinterpret_cast<u32x4*>(dos + img_off(row, c)) = dv;
  }
  // ---- row constants: D[q] = <dO[q], O[q]>, lse[q] (+inf past T: P = 0 there)
  if (tid < kTP) {
    const int q = tid;
    float dsv = 0.f, lv = __builtin_huge_valf();
    if (q < a.t) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const f16x8 ov = *reinterpret_cast<const f16x8*>(og + q * o_stride + c * 8);
        const f16x8 dv = *reinterpret_cast<const f16x8*>(dog + q * o_stride + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) dsv = __builtin_fmaf((float)ov[e], (float)dv[e], dsv);
      }
      lv = a.lse[(b * a.h + head) * (int64_t)a.t + q];
    }
    dsum[q] = dsv;
    lrow[q] = lv;
  }
  // ---- this wave's K and V fragments (B operands: lane (key = 64 w + 32 kt + r, half h) holds [key][16 s + 8 h ..])
  f16x8 kf[KT][4], vf[KT][4];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    int key = KW * wave + 32 * kt + r;
    key = key < a.t ? key : a.t - 1;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      kf[kt][s] = *reinterpret_cast<const f16x8*>(kg + key * qkv_stride + 16 * s + 8 * h);
      vf[kt][s] = *reinterpret_cast<const f16x8*>(vg + key * qkv_stride + 16 * s + 8 * h);
    }
  }
  f32x16 dkt[2][KT], dvt[2][KT];  // [dim tile][key tile]: lane = key column, registers = dims
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < KT; ++y)
#pragma unroll
      for (int i = 0; i < 16; ++i) dkt[x][y][i] = dvt[x][y][i] = 0.f;
  __syncthreads();

  char* myds = dss + wave * (32 * DSROW);
  const int nqt = (a.t + 31) >> 5;
  for (int qt = 0; qt < nqt; ++qt) {
    const int q0 = qt * 32;
    // ---- S = Q K^T, dP = dO V^T (A operands: row reads of the Q / dO images)
    f16x8 qf[4], dof[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qf[s] = *reinterpret_cast<const f16x8*>(qs + img_off(q0 + r, 2 * s + h));
      dof[s] = *reinterpret_cast<const f16x8*>(dos + img_off(q0 + r, 2 * s + h));
    }
    f32x16 sc[KT], dp[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
      for (int i = 0; i < 16; ++i) sc[kt][i] = dp[kt][i] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qf[s], kf[kt][s], sc[kt], 0, 0, 0);
        dp[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dof[s], vf[kt][s], dp[kt], 0, 0, 0);
      }
    }
    // ---- P and dS in the accumulator layout (row = query acc_row(i, h), column = key r)
    float lq[16], dq_[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      lq[i] = lrow[q0 + acc_row(i, h)];
      dq_[i] = dsum[q0 + acc_row(i, h)];
    }
    f16x8 pf[KT][2], dsf[KT][2];  // [key tile][16-query k-step]
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const bool live = KW * wave + 32 * kt + r < a.t;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[kt][i], a.scale_log2e, -lq[i]));
        p = live ? p : 0.f;
        const float dsv = p * (dp[kt][i] - dq_[i]);
        pf[kt][i >> 3][i & 7] = (_Float16)p;
        dsf[kt][i >> 3][i & 7] = (_Float16)dsv;
        // dS to the wave's [q][key] image for the dQ product (2-byte stores: 32 lanes = 64 contiguous bytes)
        const int qrow = acc_row(i, h), kcol = 32 * kt + r;
        *reinterpret_cast<_Float16*>(myds + ds_off(qrow, kcol >> 3) + (kcol & 7) * 2) = (_Float16)dsv;
      }
    }
    asm volatile("" ::: "memory");  // the 2-byte dS stores above are read back below through another pointer type
    // ---- dV^T += dO^T P,  dK^T += Q^T dS: A operands by transposed reads, k order of an accumulator operand:
    //      element j of half h is query 16 s + 8 (j >> 2) + 4 h + (j & 3)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const int c0 = 32 * dt + 16 * (grp & 1) + 4 * (li & 3);
        const int qb = q0 + 16 * s + 4 * (grp >> 1) + (li >> 2);
        f16x8 dot, qtf;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int row = qb + 8 * half;
          const ab_fp16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
              (ab_fp16x4 __attribute__((address_space(3)))*)(dos + img_off_e(row, c0)));
          const ab_fp16x4 v2 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
              (ab_fp16x4 __attribute__((address_space(3)))*)(qs + img_off_e(row, c0)));
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            dot[4 * half + e] = (_Float16)v1[e];
            qtf[4 * half + e] = (_Float16)v2[e];
          }
        }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          dvt[dt][kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dot, pf[kt][s], dvt[dt][kt], 0, 0, 0);
          dkt[dt][kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qtf, dsf[kt][s], dkt[dt][kt], 0, 0, 0);
        }
      }
    }
    // ---- dQ^T[dim][q] = sum over this wave's keys of K^T[dim][key] dS^T[key][q]
    f32x16 dqa[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int i = 0; i < 16; ++i) dqa[dt][i] = 0.f;
#pragma unroll
    for (int s = 0; s < 2 * KT; ++s) {
      // B operand: lane (q = r, half h) holds dS[q][16 s + 8 h .. + 7] of the wave's image (natural k order)
      const f16x8 dsb = *reinterpret_cast<const f16x8*>(myds + ds_off(r, 2 * s + h));
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const int c0 = 32 * dt + 16 * (grp & 1) + 4 * (li & 3);
        const int kb = KW * wave + 16 * s + 8 * (grp >> 1) + (li >> 2);
        f16x8 ktf;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const ab_fp16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
              (ab_fp16x4 __attribute__((address_space(3)))*)(ks + img_off_e(kb + 4 * half, c0)));
#pragma unroll
          for (int e = 0; e < 4; ++e) ktf[4 * half + e] = (_Float16)v[e];
        }
        dqa[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ktf, dsb, dqa[dt], 0, 0, 0);
      }
    }
    // the waves' partial tiles (their keys) go to four LDS slabs (row pitch 65 floats: the 32 queries of a register
    // spread over the banks) and are summed in slab order: deterministic, no LDS atomics (an atomic version changed
    // low bits from run to run, which fp16 roundings downstream amplified to 1e-4).  KT = 1: wave 2j stores slab j,
    // then wave 2j+1 adds its tile to it.
    {
      float* slab = dqt + (wave / (NW / 4)) * (32 * 65);
      if (wave % (NW / 4) == 0) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int i = 0; i < 16; ++i) slab[r * 65 + 32 * dt + acc_row(i, h)] = dqa[dt][i];
      }
      __syncthreads();
      if constexpr (NW > 4) {
        if (wave % (NW / 4) == 1) {
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) slab[r * 65 + 32 * dt + acc_row(i, h)] += dqa[dt][i];
        }
        __syncthreads();
      }
    }
    if (tid < 256) {
      const int q = tid >> 3, c = tid & 7;
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float sacc = dqt[q * 65 + c * 8 + e];
#pragma unroll
        for (int w = 1; w < 4; ++w) sacc += dqt[w * (32 * 65) + q * 65 + c * 8 + e];
        o[e] = (_Float16)(sacc * a.scale);
      }
      if (q0 + q < a.t) *reinterpret_cast<f16x8*>(dqg + (q0 + q) * qkv_stride + c * 8) = o;
    }
    __syncthreads();
  }

  // ---- dK = scale dK^T, dV = dV^T: lane = key, registers = dims in groups of 4
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const int key = KW * wave + 32 * kt + r;
    if (key < a.t) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          f16x4 ok, ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            ok[e] = (_Float16)(dkt[dt][kt][4 * g4 + e] * a.scale);
            ov[e] = (_Float16)dvt[dt][kt][4 * g4 + e];
          }
          const int dim = 32 * dt + 8 * g4 + 4 * h;
          *reinterpret_cast<f16x4*>(dkg + key * qkv_stride + dim) = ok;
          *reinterpret_cast<f16x4*>(dvg + key * qkv_stride + dim) = ov;
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// T <= 224 (ViT-B/16: 197): persistent two-pass kernel.  The single-pass kernel above pays three workgroup barriers,
// a dS round trip and a four-slab dQ reduction per query tile, and its one workgroup per CU loads, computes and
// stores in sequence (202 KB of HBM traffic per (b, head): 10 us at the CU's share of 5 TB/s, never overlapped).
// Here a workgroup of ceil(T / 32) waves walks over (b, head) items and splits each into
//   pass 1  wave w owns KEYS 32w..: S = Q K^T, dP = dO V^T (query on the row, key on the lane), dV^T += dO^T P,
//           dK^T += Q^T dS from the accumulators - needs the Q / dO images only;
//   pass 2  wave w owns QUERIES 32w..: S^T = K Q^T, dP^T = V dO^T (key on the row, query on the lane) recomputed,
//           dQ^T += K^T dS^T with dS^T taken from the accumulators (contraction over their row index: no LDS round
//           trip, no cross-wave reduction) - needs the K / V images only.
// 40 % more MFMA work (28 instead of 20 per tile pair; the matrix pipe was 19 % busy) for NO barrier inside a pass,
// no LDS writes, no atomics.  Two LDS regions filled by LDS-DMA: R0 = {Q, dO, O} of the item, R1 = {K, V}; R1 is in
// flight under pass 1, the NEXT item's R0 (and its K / V register fragments, its lse) under pass 2, so the loads of
// an item hide under the compute of the previous one.  Three barriers per item.  D = rowsum(dO o O) comes from the
// LDS images.  The chunk swizzle (bit-reversed (row >> 1) & 7) is conflict-free for the row reads AND for the
// transposed reads (a half wave reads 4 consecutive rows x 64 B: rows r, r + 2 land in different 64-B windows).
constexpr int kRows2 = 224;
#ifdef HCIR_ATTN_BWD_STAMPS   // diagnostic build (tools/diag_attn_bwd.py): wave 0 of workgroup 0 stamps the phase
                              // boundaries of its fourth item with (s_memtime, s_memrealtime)
__device__ unsigned long long g_attn_bwd_stamps[16];
#define AB2_STAMP(i)                                                         \
  if (blockIdx.x == 0 && tid == 0 && nitem == 3) {                           \
    g_attn_bwd_stamps[2 * (i)] = __builtin_amdgcn_s_memtime();               \
    g_attn_bwd_stamps[2 * (i) + 1] = __builtin_amdgcn_s_memrealtime();       \
  }
#else
#define AB2_STAMP(i)
#endif
constexpr int kImg2 = kRows2 * 128;
constexpr int kNW2 = kRows2 / 32;

__device__ __forceinline__ int swz2(int row) {
  return (((row >> 1) & 1) << 2) | (((row >> 2) & 1) << 1) | ((row >> 3) & 1);
}
__device__ __forceinline__ int img2_off(int row, int c16) { return row * 128 + ((c16 ^ swz2(row)) << 4); }
__device__ __forceinline__ int img2_off_e(int row, int e0) { return img2_off(row, e0 >> 3) + (e0 & 7) * 2; }

#ifndef HCIR_AB2_STAGGER
#define HCIR_AB2_STAGGER 0   // s_sleep units (64 cycles) the second wave of each SIMD starts a pass late (A/B flag)
#endif

__global__ __launch_bounds__(64 * kNW2, 1) void attn_bwd2_kernel(AttnBwdArgs a, int items) {
  __shared__ __attribute__((aligned(16))) char lds[5 * kImg2 + 2 * 256 * 4];
  char* const qs = lds;                 // R0
  char* const dos = lds + kImg2;
  char* const os = lds + 2 * kImg2;
  char* const ks = lds + 3 * kImg2;     // R1
  char* const vs = lds + 4 * kImg2;
  float* const dsum = reinterpret_cast<float*>(lds + 5 * kImg2);
  float* const lrow = dsum + 256;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int grp = lane >> 4, li = lane & 15;
  const int qkv_stride = 3 * a.h * 64, o_stride = a.h * 64;   // elements between tokens
  const int myrow = 32 * wave + r;                            // this lane's key (pass 1) / query (pass 2)
  const int myrow_c = myrow < a.t ? myrow : a.t - 1;
  const bool live = myrow < a.t;
  const uint32_t lds0 = lds_addr(lds);
  const float inv_c = 1.0f / a.scale_log2e;

  // One image = 4 nw wave instructions of 8 rows; piece u (0..3) of wave w is instruction w + u nw.  Rows past T
  // fetch the last real row: finite data; P is exactly 0 there through lse = +inf (queries) or the key mask.
  // Memory instructions are NOT issued in bursts: a burst of 12 transfers + 8 loads + 8 stores per wave stood 7-10 k
  // cycles at the head of a pass (the issuing wave stalls while the CU's memory pipe is backlogged - the kernel
  // moves 202 KB per item, ~20 k cycles at the CU's share of HBM); piece u goes out at the top of tile min(u, nw - 1)
  // of the pass it hides under, the row stores in the tiles behind.
  auto dma_piece = [&](const char* img, const _Float16* g, int stride, int u) {
    const int ii = wave + u * nw;
    const int row = ii * 8 + (lane >> 3), pc = lane & 7;
    const int src = row < a.t ? row : a.t - 1;
    lds_dma16(g, (uint32_t)(src * stride * 2 + ((pc ^ swz2(row)) << 4)), lds0 + (uint32_t)(img - lds) + ii * 1024);
  };
#if defined(HCIR_AB2_ABL) && HCIR_AB2_ABL == 5   // timing ablation (wrong results): one tile per pass, all memory traffic
  const int nwl = 1;
#else
  const int nwl = nw;
#endif
  auto piece_tile = [&](int u) { return u < nwl - 1 ? u : nwl - 1; };
  auto qkv_base = [&](int item) { return ((int64_t)(item / a.h) * a.t) * qkv_stride + (item % a.h) * 64; };
  auto o_base = [&](int item) { return ((int64_t)(item / a.h) * a.t) * o_stride + (item % a.h) * 64; };
  auto issue_r0 = [&](int item, int u) {
    const int64_t ob = o_base(item);
    dma_piece(qs, a.qkv + qkv_base(item), qkv_stride, u);
    dma_piece(dos, a.dout + ob, o_stride, u);
    dma_piece(os, a.out + ob, o_stride, u);
  };

  f16x8 kf[4], vf[4];   // K / V rows of this wave's keys: B operands of pass 1
  float lv = 0.f;
  auto load_frags = [&](int item) {
    const char* kg = reinterpret_cast<const char*>(a.qkv + qkv_base(item) + (int64_t)a.h * 64);
    const char* vg = kg + (int64_t)a.h * 64 * 2;
    const uint32_t off = (uint32_t)(myrow_c * qkv_stride * 2 + 16 * h);   // this lane's row, dims 8 h ..
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      kf[s] = *reinterpret_cast<const f16x8*>(kg + (off + 32 * s));
      vf[s] = *reinterpret_cast<const f16x8*>(vg + (off + 32 * s));
    }
    lv = __builtin_huge_valf();
    if ((tid >> 1) < a.t) lv = (a.lse + (int64_t)item * a.t)[tid >> 1];   // row tid / 2: see the D phase
  };

  // Rows leave in 16-B pieces: a lane pair (r, r + 32) holds one row of the transposed accumulators in alternating
  // 8-B pieces (dims 8 g + 4 h ..), so one v_permlane32_swap per dword hands each lane two adjacent pieces (chunk c
  // of half 0: dims 16 c .. + 7, of half 1: dims 16 c + 8 .. + 15).  As 8-B stores every instruction touched 32
  // lines twice over.
  auto pack_rows = [&](const f32x16 (&acc)[2], float mul, u32x4 (&o)[4]) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f16x4 x, y;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          x[e] = (_Float16)(acc[dt][8 * j + e] * mul);
          y[e] = (_Float16)(acc[dt][8 * j + 4 + e] * mul);
        }
        u32x2 xu = __builtin_bit_cast(u32x2, x), yu = __builtin_bit_cast(u32x2, y);
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const auto sw = __builtin_amdgcn_permlane32_swap(xu[d], yu[d], false, false);
          xu[d] = sw[0];
          yu[d] = sw[1];
        }
        o[2 * dt + j] = (u32x4){xu[0], xu[1], yu[0], yu[1]};
      }
  };
  const uint32_t row_off = (uint32_t)(myrow * qkv_stride * 2 + 16 * h);
  auto store_chunk = [&](_Float16* g, int c, const u32x4& v) {
    if (live) *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(g) + (row_off + 32 * c)) = v;
  };

  u32x4 dqo[4];   // the previous item's dQ rows, packed: stored behind the next barrier
  int prev = -1;
  int item = blockIdx.x;
  if (item < items) {
#pragma unroll
    for (int u = 0; u < 4; ++u) issue_r0(item, u);
    load_frags(item);
  }
  [[maybe_unused]] int nitem = 0;
  while (item < items) {
    const int64_t qb = qkv_base(item);
    const _Float16* kgl = a.qkv + qb + (int64_t)a.h * 64;
    const _Float16* vgl = kgl + (int64_t)a.h * 64;
    AB2_STAMP(0)
    // R0 of this item has landed (issued under the first tiles of the previous pass 2).  The eight dK / dV row stores
    // of that pass were issued BEHIND the transfers (vmcnt retires in order): they may stay in flight - waiting for
    // their acknowledgements cost ~4 k cycles per item.
    if (prev < 0)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    // the fragments / lse prefetched with it have landed too: take them out of the compiler's pending set, or it
    // re-waits (vmcnt(0)) at their first use, behind the K / V transfer issued below
    asm volatile("" : "+v"(lv));
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      asm volatile("" : "+v"(kf[s]));
      asm volatile("" : "+v"(vf[s]));
    }
    __syncthreads();                                   // ... and every wave has left pass 2 of the previous item
    AB2_STAMP(1)
    if (prev >= 0) {
      _Float16* dqg = a.dqkv + qkv_base(prev);
#pragma unroll
      for (int c = 0; c < 4; ++c) store_chunk(dqg, c, dqo[c]);
    }
    {  // D[q] = <dO[q], O[q]>: two threads per row (blockDim = 2 x rows), four 16-B chunks each
      const int drow = tid >> 1, dc = (tid & 1) * 4;
      float dsv = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f16x8 ov = *reinterpret_cast<const f16x8*>(os + img2_off(drow, dc + c));
        const f16x8 dv = *reinterpret_cast<const f16x8*>(dos + img2_off(drow, dc + c));
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const f16x2 x = {ov[e], ov[e + 1]}, y = {dv[e], dv[e + 1]};
          dsv = __builtin_amdgcn_fdot2(x, y, dsv, false);
        }
      }
      dsv += __shfl_xor(dsv, 1);
      if ((tid & 1) == 0) {   // stored as the accumulator start values of pass 1: -D and -lse / (scale log2e)
        dsum[drow] = -dsv;
        lrow[drow] = -lv * inv_c;   // lse = +inf past T: -inf, so P = exp2(-inf) = 0 there
      }
    }
    __syncthreads();
    AB2_STAMP(2)
#if HCIR_AB2_STAGGER > 0
    if (wave >= 4) __builtin_amdgcn_s_sleep(HCIR_AB2_STAGGER);
#endif

    // ---- pass 1: this wave's keys against every query tile; the K / V images (R1) go out under its first tiles
    f32x16 dkt[2], dvt[2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int i = 0; i < 16; ++i) dkt[x][i] = dvt[x][i] = 0.f;
    for (int qt = 0; qt < nwl; ++qt) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (piece_tile(u) == qt) {
          dma_piece(ks, kgl, qkv_stride, u);
          dma_piece(vs, vgl, qkv_stride, u);
        }
      const int q0 = qt * 32;
      // Issue order is pinned with sched_barriers: left to itself hipcc fetched the row constants and the transposed
      // A operands just in time, into the same eight registers, and every pair of dV / dK MFMAs waited out an LDS
      // round trip (pass 1 without that MFMA group: 7.4 k cycles instead of 13.7 k, 2 x the group's pipe time).
      // ---- all row reads of the tile: Q / dO fragments, then lse / D of the 16 accumulator rows
      f16x8 qf[4], dof[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        qf[s] = *reinterpret_cast<const f16x8*>(qs + img2_off(q0 + r, 2 * s + h));
        dof[s] = *reinterpret_cast<const f16x8*>(dos + img2_off(q0 + r, 2 * s + h));
      }
      // the accumulators START from the row constants: sc = -lse / (scale log2e), dp = -D (both stored that way by
      // the D phase), so that the MFMAs leave S - lse / c and dP - D: no per-element subtraction, no constants live
      // beside the accumulators (rows q0 + 8 g + 4 h .. + 3 = accumulator registers 4 g .. 4 g + 3)
      f32x16 sc, dp;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(lrow + q0 + 8 * g + 4 * h);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(dsum + q0 + 8 * g + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          sc[4 * g + e] = l4[e];
          dp[4 * g + e] = d4[e];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(qf[s], kf[s], sc, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_f16(dof[s], vf[s], dp, 0, 0, 0);
      }
      // ---- the transposed A operands of dV^T += dO^T P, dK^T += Q^T dS, in flight under the exp chain below
      //      (element j of half h is query 16 s + 8 (j >> 2) + 4 h + (j & 3))
      f16x8 dotf[2][2], qtff[2][2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const int c0 = 32 * dt + 16 * (grp & 1) + 4 * (li & 3);
          const int qrow = q0 + 16 * s + 4 * (grp >> 1) + (li >> 2);
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            const ab_fp16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                (ab_fp16x4 __attribute__((address_space(3)))*)(dos + img2_off_e(qrow + 8 * half, c0)));
            const ab_fp16x4 v2 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                (ab_fp16x4 __attribute__((address_space(3)))*)(qs + img2_off_e(qrow + 8 * half, c0)));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              dotf[s][dt][4 * half + e] = (_Float16)v1[e];
              qtff[s][dt][4 * half + e] = (_Float16)v2[e];
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      f16x8 pf[2], dsf[2];   // [16-query k-step], accumulator-operand k order
      // (keys past T - this lane's column - need no mask here: they only reach dK^T / dV^T columns that are never stored)
      // (-DHCIR_AB2_HALVES: the exp chain of the second 16-query k-step issued in the shadow of the first k-step's four
      // MFMAs - measured 2659 against 2662 us per launch, i.e. nothing: the two waves of a SIMD already cover each
      // other's phases; the plain order is what ships)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float p = __builtin_amdgcn_exp2f(sc[i] * a.scale_log2e);
        pf[0][i] = (_Float16)p;
        dsf[0][i] = (_Float16)(p * dp[i]);
      }
#ifndef HCIR_AB2_HALVES
#pragma unroll
      for (int i = 8; i < 16; ++i) {
        const float p = __builtin_amdgcn_exp2f(sc[i] * a.scale_log2e);
        pf[1][i & 7] = (_Float16)p;
        dsf[1][i & 7] = (_Float16)(p * dp[i]);
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        dvt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dotf[0][dt], pf[0], dvt[dt], 0, 0, 0);
        dkt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qtff[0][dt], dsf[0], dkt[dt], 0, 0, 0);
      }
#ifdef HCIR_AB2_HALVES
#pragma unroll
      for (int i = 8; i < 16; ++i) {
        const float p = __builtin_amdgcn_exp2f(sc[i] * a.scale_log2e);
        pf[1][i & 7] = (_Float16)p;
        dsf[1][i & 7] = (_Float16)(p * dp[i]);
      }
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {   // one MFMA, then a quarter of the chain (8 x (mul, exp, mul) + 8 cvt = 32 VALU)
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        dvt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dotf[1][dt], pf[1], dvt[dt], 0, 0, 0);
        dkt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qtff[1][dt], dsf[1], dkt[dt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    AB2_STAMP(3)
    // Q / dO rows of this wave's QUERIES (B operands of pass 2) and their row constants, while R0 is still this item's
    f16x8 qb2[4], dob2[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qb2[s] = *reinterpret_cast<const f16x8*>(qs + img2_off(myrow, 2 * s + h));
      dob2[s] = *reinterpret_cast<const f16x8*>(dos + img2_off(myrow, 2 * s + h));
    }
    const float lq2 = -lrow[myrow] * a.scale_log2e, dq2 = -dsum[myrow];   // back to lse and D (pass 2 subtracts them)
    // dK = scale dK^T, dV = dV^T: lane = key, registers = dims; packed now, stored under pass 2
    u32x4 dko[4], dvo[4];
    pack_rows(dkt, a.scale, dko);
    pack_rows(dvt, 1.f, dvo);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // R1 has landed (issued under the first tiles of pass 1)
    __syncthreads();                                   // ... and every wave is done with R0
    AB2_STAMP(4)
#if HCIR_AB2_STAGGER > 0
    if (wave >= 4) __builtin_amdgcn_s_sleep(HCIR_AB2_STAGGER);
#endif
    const int next = item + gridDim.x;
    const bool has_next = next < items;
    // the next item's register fragments: here, not inside the loop below - there the compiler guards the rewrite of
    // these registers with a vmcnt wait that also retires the transfers issued just before it
    if (has_next) load_frags(next);

    // ---- pass 2: this wave's queries against every key tile; the next item's R0, its fragments and this item's
    //      dK / dV rows go out under it
    f32x16 dqa[2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int i = 0; i < 16; ++i) dqa[x][i] = 0.f;
    auto key_tile = [&](int kt, auto masked) {
      if (has_next) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (piece_tile(u) == kt) issue_r0(next, u);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (piece_tile(c + 3) == kt) {
          store_chunk(a.dqkv + qb + (int64_t)a.h * 64, c, dko[c]);
          store_chunk(a.dqkv + qb + (int64_t)2 * a.h * 64, c, dvo[c]);
        }
      const int k0 = kt * 32;
      __builtin_amdgcn_sched_barrier(0);
      f16x8 ka[4], va[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        ka[s] = *reinterpret_cast<const f16x8*>(ks + img2_off(k0 + r, 2 * s + h));
        va[s] = *reinterpret_cast<const f16x8*>(vs + img2_off(k0 + r, 2 * s + h));
      }
      __builtin_amdgcn_sched_barrier(0);
      f32x16 st, dpt;
#pragma unroll
      for (int i = 0; i < 16; ++i) st[i] = dpt[i] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka[s], qb2[s], st, 0, 0, 0);
        dpt = __builtin_amdgcn_mfma_f32_32x32x16_f16(va[s], dob2[s], dpt, 0, 0, 0);
      }
      // K^T fragments of dQ^T += K^T dS^T (accumulator operand's k order), in flight under the exp chain
      f16x8 ktf[2][2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const int c0 = 32 * dt + 16 * (grp & 1) + 4 * (li & 3);
          const int krow = k0 + 16 * s + 4 * (grp >> 1) + (li >> 2);
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            const ab_fp16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                (ab_fp16x4 __attribute__((address_space(3)))*)(ks + img2_off_e(krow + 8 * half, c0)));
#pragma unroll
            for (int e = 0; e < 4; ++e) ktf[s][dt][4 * half + e] = (_Float16)v[e];
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      f16x8 dsb[2];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[i], a.scale_log2e, -lq2));
        if (decltype(masked)::value) p = k0 + acc_row(i, h) < a.t ? p : 0.f;   // keys past T: the last tile only
        dsb[i >> 3][i & 7] = (_Float16)(p * (dpt[i] - dq2));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          dqa[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ktf[s][dt], dsb[s], dqa[dt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    };
    for (int kt = 0; kt + 1 < nwl; ++kt) key_tile(kt, std::false_type{});
    key_tile(nwl - 1, std::true_type{});
    pack_rows(dqa, a.scale, dqo);   // dQ = scale dQ^T: lane = query
    AB2_STAMP(6)
    prev = item;
    item = next;
    ++nitem;
  }
  if (prev >= 0) {
    _Float16* dqg = a.dqkv + qkv_base(prev);
#pragma unroll
    for (int c = 0; c < 4; ++c) store_chunk(dqg, c, dqo[c]);
  }
}

}  // namespace

#ifdef HCIR_ATTN_BWD_STAMPS
extern "C" int hcir_diag_attn_bwd_stamps(unsigned long long* host16) {
  return hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_attn_bwd_stamps), sizeof(g_attn_bwd_stamps)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int hcir_attn_bwd(const void* qkv, const void* out, const void* d_out, const float* lse, int64_t b,
                             int32_t t, int32_t h, int32_t hd, float scale, void* d_qkv, void* stream) {
  HCIR_ENTER();
  if (!qkv || !out || !d_out || !lse || !d_qkv || b <= 0 || t <= 0 || h <= 0) return HCIR_ERR_INVALID;
  if (hd != 64 || t > kTP) return HCIR_ERR_UNSUPPORTED;
  if (b * h > 0x7fffffff) return HCIR_ERR_INVALID;
  AttnBwdArgs a{static_cast<const _Float16*>(qkv), static_cast<const _Float16*>(out),
                static_cast<const _Float16*>(d_out), lse, static_cast<_Float16*>(d_qkv), t, h, scale,
                scale * 1.44269504088896340736f};
#ifndef HCIR_ATTN_BWD_V1   // build flag: the single-pass kernel at every T, for A/B runs
  if (t <= kRows2) {
    const int items = (int)(b * h), nw = (t + 31) / 32;
    hipLaunchKernelGGL(attn_bwd2_kernel, dim3((unsigned)(items < 256 ? items : 256)), dim3(64 * nw), 0,
                       static_cast<hipStream_t>(stream), a, items);
    HCIR_LAUNCH_CHECK();
    return HCIR_OK;
  }
#endif
#ifdef HCIR_ATTN_BWD_KT2   // build flag: the first version (four waves of two key tiles), for A/B runs
  hipLaunchKernelGGL(attn_bwd_kernel<2>, dim3((unsigned)(b * h)), dim3(256), 0, static_cast<hipStream_t>(stream), a);
#else
  hipLaunchKernelGGL(attn_bwd_kernel<1>, dim3((unsigned)(b * h)), dim3(512), 0, static_cast<hipStream_t>(stream), a);
#endif
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}
