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
#ifdef HCIR_ATTN_BWD_KT2   // build flag: the first version (four waves of two key tiles), for A/B runs
  hipLaunchKernelGGL(attn_bwd_kernel<2>, dim3((unsigned)(b * h)), dim3(256), 0, static_cast<hipStream_t>(stream), a);
#else
  hipLaunchKernelGGL(attn_bwd_kernel<1>, dim3((unsigned)(b * h)), dim3(512), 0, static_cast<hipStream_t>(stream), a);
#endif
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}
