// attn.hip — fused multi-head self-attention forward for ViT token counts (T <= 288).
//
// Replaces Attention.forward (HP/src/models_vit.py:69-78) and the
// nn.MultiheadAttention inside torchvision's EncoderBlock (HP/src/main_backbone.py:554):
//   out = softmax((q * scale) k^T) v   per (batch, head), head_dim 64.
//
// One workgroup per (b, head); wave w owns query rows 32w .. 32w+31; K and V of the
// head (T x 64 fp16 each) are staged ONCE into LDS and shared by all waves.
//   S^T = K . Q^T   : MFMA 32x32x16 f16, keys on the MFMA row, queries on the column
//                     -> a lane owns ONE query and holds its scores in registers:
//                     row max / exp2 / row sum need one cross-half shuffle only.
//   O^T = V^T . P   : the fp32 score accumulators, converted pairwise to fp16, ARE the
//                     B operand of the second MFMA (cdna_hip_programming.md §3, "An
//                     accumulator tile as the next MFMA's operand"); the A operand V^T is
//                     read from the row-major V image with ds_read_b64_tr_b16 (T10).
// LDS images: K rows are 128 B, 16-B slots XOR-swizzled with (key>>1)&7 (ds_read_b128
// conflict-free); V rows are 128 B with the two 64-B halves swapped when key bit 1 is
// set (the 4-row x 32-col transposed reads of a half-wave then cover all 64 banks).
#include "common.h"

namespace {

typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));

struct AttnArgs {
  const _Float16* qkv;  // [B][T][3][H][64]
  _Float16* out;        // [B][T][H*64]
  int t, h;
  float scale_log2e;
};

template <int NKT>
__global__ __launch_bounds__(64 * NKT) void attn_fwd_kernel(AttnArgs a) {
  constexpr int TP = 32 * NKT;
  __shared__ __attribute__((aligned(16))) char lds[2 * TP * 128];
  char* ks = lds;
  char* vs = lds + TP * 128;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nthreads = blockDim.x;
  const int r = lane & 31, h = lane >> 5;
  const int head = blockIdx.x % a.h;
  const int64_t b = blockIdx.x / a.h;
  const int64_t row_stride = (int64_t)3 * a.h * 64;  // elements between tokens
  const _Float16* base = a.qkv + b * a.t * row_stride + head * 64;
  const _Float16* qg = base;
  const _Float16* kg = base + (int64_t)a.h * 64;
  const _Float16* vg = base + (int64_t)2 * a.h * 64;

  // ---- stage K and V (zero-filled past T) ----
  for (int slot = tid; slot < TP * 8; slot += nthreads) {
    const int key = slot >> 3, c = slot & 7;
    u32x4 kv = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
    if (key < a.t) {
      kv = *reinterpret_cast<const u32x4*>(kg + key * row_stride + c * 8);
      vv = *reinterpret_cast<const u32x4*>(vg + key * row_stride + c * 8);
    }
    *reinterpret_cast<u32x4*>(ks + key * 128 + ((c ^ ((key >> 1) & 7)) << 4)) = kv;
    *reinterpret_cast<u32x4*>(vs + key * 128 + ((c ^ (((key >> 1) & 1) << 2)) << 4)) = vv;
  }

  // ---- Q fragments straight from global: lane (q = r, half h) holds Q[q][16s + 8h .. +7]
  const int q0 = wave * 32;
  int qrow = q0 + r;
  qrow = qrow < a.t ? qrow : a.t - 1;
  f16x8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
    qf[s] = *reinterpret_cast<const f16x8*>(qg + qrow * row_stride + 16 * s + 8 * h);

  __syncthreads();

  // ---- S^T = K . Q^T ----
  f32x16 sc[NKT];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
    for (int i = 0; i < 16; ++i) sc[kt][i] = 0.f;
    const int key = kt * 32 + r;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int c = 2 * s + h;
      const f16x8 kf =
          *reinterpret_cast<const f16x8*>(ks + key * 128 + ((c ^ ((key >> 1) & 7)) << 4));
      sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], sc[kt], 0, 0, 0);
    }
  }

  // ---- softmax over keys (registers + one cross-half exchange) ----
  float mx = -__builtin_huge_valf();
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int key = kt * 32 + acc_row(i, h);
      const float s = key < a.t ? sc[kt][i] : -__builtin_huge_valf();
      sc[kt][i] = s;
      mx = fmaxf(mx, s);
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float p = __builtin_amdgcn_exp2f((sc[kt][i] - mx) * a.scale_log2e);
      sc[kt][i] = p;
      sum += p;
    }
  }
  sum += __shfl_xor(sum, 32);
  const float inv = 1.0f / sum;

  // ---- O^T = V^T . P ----
  f32x16 oacc[2];
#pragma unroll
  for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[t2][i] = 0.f;

  const int grp = lane >> 4, li = lane & 15;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f16x8 pf;
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[j] = (_Float16)sc[kt][8 * s + j];
#pragma unroll
      for (int hdt = 0; hdt < 2; ++hdt) {
        // transposed read: this lane supplies row (kb + li>>2), columns c0 + 4*(li&3) .. +3
        const int c0 = 32 * hdt + 16 * (grp & 1) + 4 * (li & 3);
        const int kb = 32 * kt + 16 * s + 4 * (grp >> 1) + (li >> 2);
        f16x8 vf;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int key = kb + 8 * half;
          const int col = c0 ^ (((key >> 1) & 1) << 5);
          const fp16x4_t v4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
              (fp16x4_t __attribute__((address_space(3)))*)(vs + key * 128 + col * 2));
#pragma unroll
          for (int e = 0; e < 4; ++e) vf[4 * half + e] = (_Float16)v4[e];
        }
        oacc[hdt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, oacc[hdt], 0, 0, 0);
      }
    }
  }

  // ---- store: lane owns query row q0 + r, registers hold head dims ----
  const int q = q0 + r;
  if (q < a.t) {
    _Float16* orow = a.out + (b * a.t + q) * ((int64_t)a.h * 64) + head * 64;
#pragma unroll
    for (int hdt = 0; hdt < 2; ++hdt) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (_Float16)(oacc[hdt][4 * g4 + e] * inv);
        *reinterpret_cast<f16x4*>(orow + 32 * hdt + 8 * g4 + 4 * h) = o;
      }
    }
  }
}

}  // namespace

extern "C" int hcir_attn_fwd(const void* qkv, int64_t b, int32_t t, int32_t h, int32_t hd,
                             float scale, void* out, void* stream) {
  HCIR_ENTER();
  if (!qkv || !out || b <= 0 || t <= 0 || h <= 0) return HCIR_ERR_INVALID;
  if (hd != 64 || t > 288) return HCIR_ERR_UNSUPPORTED;
  if (b * h > 0x7fffffff) return HCIR_ERR_INVALID;
  AttnArgs a{static_cast<const _Float16*>(qkv), static_cast<_Float16*>(out), t, h,
             scale * 1.44269504088896340736f};
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nqt = (t + 31) / 32;
  const dim3 grid((unsigned)(b * h));
#define LAUNCH(N) hipLaunchKernelGGL(attn_fwd_kernel<N>, grid, dim3(64 * N), 0, st, a)
  // the kernel is built for NKT key tiles AND NKT waves (one 32-row query tile each)
  switch (nqt) {
    case 1: LAUNCH(1); break;
    case 2: LAUNCH(2); break;
    case 3: LAUNCH(3); break;
    case 4: LAUNCH(4); break;
    case 5: LAUNCH(5); break;
    case 6: LAUNCH(6); break;
    case 7: LAUNCH(7); break;
    case 8: LAUNCH(8); break;
    default: LAUNCH(9); break;
  }
#undef LAUNCH
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}
