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
constexpr int kImg2 = kRows2 * 128;
constexpr int kNW2 = kRows2 / 32;

__device__ __forceinline__ int swz2(int row) {
  return (((row >> 1) & 1) << 2) | (((row >> 2) & 1) << 1) | ((row >> 3) & 1);
}
__device__ __forceinline__ int img2_off(int row, int c16) { return row * 128 + ((c16 ^ swz2(row)) << 4); }
__device__ __forceinline__ int img2_off_e(int row, int e0) { return img2_off(row, e0 >> 3) + (e0 & 7) * 2; }

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
  const int64_t qkv_stride = (int64_t)3 * a.h * 64, o_stride = (int64_t)a.h * 64;
  const int nrows = 32 * nw;
  const int myrow = 32 * wave + r;                          // this lane's key (pass 1) / query (pass 2)
  const int myrow_c = myrow < a.t ? myrow : a.t - 1;
  const bool live = myrow < a.t;

  // one image = 4 nw wave instructions of 8 rows; wave w issues instructions w, w + nw, ...  (rows past T: the last
  // real row - finite data; P is exactly 0 there through lse = +inf (queries) or the key mask).  The per-lane source
  // byte offsets are the same for every item: row * pitch + swizzled chunk, for the two row pitches in use.
  uint32_t offq[4], offo[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int row = (wave + u * nw) * 8 + (lane >> 3), pc = lane & 7;
    const int src = row < a.t ? row : a.t - 1;
    offq[u] = (uint32_t)(src * (int)qkv_stride * 2 + ((pc ^ swz2(row)) << 4));
    offo[u] = (uint32_t)(src * (int)o_stride * 2 + ((pc ^ swz2(row)) << 4));
  }
  const uint32_t lds0 = lds_addr(lds);
  auto dma_image = [&](const char* img, const _Float16* g, const uint32_t (&off)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) lds_dma16(g, off[u], lds0 + (uint32_t)(img - lds) + (wave + u * nw) * 1024);
  };
  auto qkv_base = [&](int item) { return ((int64_t)(item / a.h) * a.t) * qkv_stride + (item % a.h) * 64; };
  auto o_base = [&](int item) { return ((int64_t)(item / a.h) * a.t) * o_stride + (item % a.h) * 64; };
  const uint32_t myq_off = (uint32_t)(myrow_c * (int)qkv_stride * 2 + 16 * h);   // bytes: this lane's row, dims 8 h ..

  f16x8 kf[4], vf[4];   // K / V rows of this wave's keys: B operands of pass 1
  float lv = 0.f;
  auto prefetch = [&](int item) {
    const _Float16* qg = a.qkv + qkv_base(item);
    const int64_t ob = o_base(item);
    dma_image(qs, qg, offq);
    dma_image(dos, a.dout + ob, offo);
    dma_image(os, a.out + ob, offo);
    const char* kg = reinterpret_cast<const char*>(qg + (int64_t)a.h * 64);
    const char* vg = reinterpret_cast<const char*>(qg + (int64_t)2 * a.h * 64);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      kf[s] = *reinterpret_cast<const f16x8*>(kg + (myq_off + 32 * s));
      vf[s] = *reinterpret_cast<const f16x8*>(vg + (myq_off + 32 * s));
    }
    lv = __builtin_huge_valf();
    if (tid < a.t) lv = (a.lse + (int64_t)item * a.t)[tid];
  };

  f32x16 dqa[2];
  int prev = -1;
  // dQ = scale dQ^T: lane = query, registers = dims in groups of 4
  auto store_rows = [&](_Float16* g, const f32x16 (&acc)[2], float mul) {
    char* gp = reinterpret_cast<char*>(g);
    const uint32_t off = (uint32_t)(myrow * (int)qkv_stride * 2 + 8 * h);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (_Float16)(acc[dt][4 * g4 + e] * mul);
        *reinterpret_cast<f16x4*>(gp + (off + 64 * dt + 16 * g4)) = o;
      }
  };
  auto store_dq = [&](int item) {
    if (live) store_rows(a.dqkv + qkv_base(item), dqa, a.scale);
  };

  int item = blockIdx.x;
  if (item < items) prefetch(item);
  while (item < items) {
    const int64_t qb = qkv_base(item);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // R0 of this item has landed (issued a pass ago)
    // the fragments / lse prefetched with it have landed too: take them out of the compiler's pending set, or it
    // re-waits (vmcnt(0)) at their first use, behind the K / V transfer issued below
    asm volatile("" : "+v"(lv));
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      asm volatile("" : "+v"(kf[s]));
      asm volatile("" : "+v"(vf[s]));
    }
    __syncthreads();                                   // ... and every wave has left pass 2 of the previous item
    if (prev >= 0) store_dq(prev);
    dma_image(ks, a.qkv + qb + (int64_t)a.h * 64, offq);
    dma_image(vs, a.qkv + qb + (int64_t)2 * a.h * 64, offq);
    if (tid < nrows) {
      float dsv = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const f16x8 ov = *reinterpret_cast<const f16x8*>(os + img2_off(tid, c));
        const f16x8 dv = *reinterpret_cast<const f16x8*>(dos + img2_off(tid, c));
#pragma unroll
        for (int e = 0; e < 8; ++e) dsv = __builtin_fmaf((float)ov[e], (float)dv[e], dsv);
      }
      dsum[tid] = dsv;
      lrow[tid] = lv;
    }
    __syncthreads();

    // ---- pass 1: this wave's keys against every query tile
    f32x16 dkt[2], dvt[2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int i = 0; i < 16; ++i) dkt[x][i] = dvt[x][i] = 0.f;
    for (int qt = 0; qt < nw; ++qt) {
      const int q0 = qt * 32;
      f16x8 qf[4], dof[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        qf[s] = *reinterpret_cast<const f16x8*>(qs + img2_off(q0 + r, 2 * s + h));
        dof[s] = *reinterpret_cast<const f16x8*>(dos + img2_off(q0 + r, 2 * s + h));
      }
      f32x16 sc, dp;
#pragma unroll
      for (int i = 0; i < 16; ++i) sc[i] = dp[i] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(qf[s], kf[s], sc, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_f16(dof[s], vf[s], dp, 0, 0, 0);
      }
      f16x8 pf[2], dsf[2];   // [16-query k-step], accumulator-operand k order
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float lq = lrow[q0 + acc_row(i, h)], dq = dsum[q0 + acc_row(i, h)];
        float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[i], a.scale_log2e, -lq));
        p = live ? p : 0.f;
        pf[i >> 3][i & 7] = (_Float16)p;
        dsf[i >> 3][i & 7] = (_Float16)(p * (dp[i] - dq));
      }
      // dV^T += dO^T P, dK^T += Q^T dS: A operands by transposed reads; element j of half h is query
      // 16 s + 8 (j >> 2) + 4 h + (j & 3)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const int c0 = 32 * dt + 16 * (grp & 1) + 4 * (li & 3);
          const int qrow = q0 + 16 * s + 4 * (grp >> 1) + (li >> 2);
          f16x8 dot, qtf;
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            const ab_fp16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                (ab_fp16x4 __attribute__((address_space(3)))*)(dos + img2_off_e(qrow + 8 * half, c0)));
            const ab_fp16x4 v2 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                (ab_fp16x4 __attribute__((address_space(3)))*)(qs + img2_off_e(qrow + 8 * half, c0)));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              dot[4 * half + e] = (_Float16)v1[e];
              qtf[4 * half + e] = (_Float16)v2[e];
            }
          }
          dvt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dot, pf[s], dvt[dt], 0, 0, 0);
          dkt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qtf, dsf[s], dkt[dt], 0, 0, 0);
        }
      }
    }
    // Q / dO rows of this wave's QUERIES (B operands of pass 2) and their row constants, while R0 is still this item's
    f16x8 qb2[4], dob2[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qb2[s] = *reinterpret_cast<const f16x8*>(qs + img2_off(myrow, 2 * s + h));
      dob2[s] = *reinterpret_cast<const f16x8*>(dos + img2_off(myrow, 2 * s + h));
    }
    const float lq2 = lrow[myrow], dq2 = dsum[myrow];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // R1 has landed (issued before pass 1)
    __syncthreads();                                   // ... and every wave is done with R0
    // dK = scale dK^T, dV = dV^T: lane = key, registers = dims in groups of 4
    if (live) {
      store_rows(a.dqkv + qb + (int64_t)a.h * 64, dkt, a.scale);
      store_rows(a.dqkv + qb + (int64_t)2 * a.h * 64, dvt, 1.f);
    }
    const int next = item + gridDim.x;
    if (next < items) prefetch(next);   // R0, kf / vf, lse of the next item: in flight under pass 2

    // ---- pass 2: this wave's queries against every key tile
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int i = 0; i < 16; ++i) dqa[x][i] = 0.f;
    for (int kt = 0; kt < nw; ++kt) {
      const int k0 = kt * 32;
      f16x8 ka[4], va[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        ka[s] = *reinterpret_cast<const f16x8*>(ks + img2_off(k0 + r, 2 * s + h));
        va[s] = *reinterpret_cast<const f16x8*>(vs + img2_off(k0 + r, 2 * s + h));
      }
      f32x16 st, dpt;
#pragma unroll
      for (int i = 0; i < 16; ++i) st[i] = dpt[i] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka[s], qb2[s], st, 0, 0, 0);
        dpt = __builtin_amdgcn_mfma_f32_32x32x16_f16(va[s], dob2[s], dpt, 0, 0, 0);
      }
      f16x8 dsb[2];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[i], a.scale_log2e, -lq2));
        p = k0 + acc_row(i, h) < a.t ? p : 0.f;
        dsb[i >> 3][i & 7] = (_Float16)(p * (dpt[i] - dq2));
      }
      // dQ^T += K^T dS^T: A operand by transposed reads of the K image in the accumulator operand's k order
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const int c0 = 32 * dt + 16 * (grp & 1) + 4 * (li & 3);
          const int krow = k0 + 16 * s + 4 * (grp >> 1) + (li >> 2);
          f16x8 ktf;
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            const ab_fp16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                (ab_fp16x4 __attribute__((address_space(3)))*)(ks + img2_off_e(krow + 8 * half, c0)));
#pragma unroll
            for (int e = 0; e < 4; ++e) ktf[4 * half + e] = (_Float16)v[e];
          }
          dqa[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ktf, dsb[s], dqa[dt], 0, 0, 0);
        }
      }
    }
    prev = item;
    item = next;
  }
  if (prev >= 0) store_dq(prev);
}

}  // namespace

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
