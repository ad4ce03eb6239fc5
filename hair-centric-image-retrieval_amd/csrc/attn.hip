// attn.hip — fused multi-head self-attention forward for ViT token counts (T <= 288).
//
// Replaces Attention.forward (HP/src/models_vit.py:69-78) and the
// nn.MultiheadAttention inside torchvision's EncoderBlock (HP/src/main_backbone.py:554):
//   out = softmax((q * scale) k^T) v   per (batch, head), head_dim 64.
//
// One 4-wave workgroup per (b, head); wave w owns query tiles w, w+4, ... (32 rows each);
// K and V of the head (T x 64 fp16 each) are staged ONCE into LDS and shared by all waves.
// Four waves (not one per query tile) keep two workgroups resident per CU, so one workgroup's
// K/V staging latency hides under the other's MFMAs.
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
  _Float16* out;        // [B][NQ][H*64]
  int t, h;
  int nq;               // query rows computed per (b, head): T, or fewer (CLS-only last layer: 1)
  float scale_log2e;
  float* lse;           // optional [B][H][T]: log2 of the softmax denominator incl. the row maximum (training)
};

template <int NKT>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnArgs a) {
  constexpr int TP = 32 * NKT;
  // K and V images, then a 4 KB output-transposition tile per wave
  constexpr bool kTrans = NKT <= 7;  // with 8-9 key tiles the extra 16 KB would cost the second workgroup per CU
  __shared__ __attribute__((aligned(16))) char lds[2 * TP * 128 + (kTrans ? 4 * 4096 : 0)];
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

  // ---- stage K and V by LDS-DMA: every piece of the head in flight at once, no VGPR round trip.
  // The LDS destination of a wave instruction is linear (base + lane*16), so both swizzles are
  // applied to the per-lane SOURCE chunk.  Keys past T are clamped to the last real key: finite
  // data whose scores are masked to -inf (K) / whose probabilities are exactly 0 (V).
  for (int slot0 = (tid & ~63); slot0 < TP * 8; slot0 += nthreads) {
    const int slot = slot0 + lane;
    const int key = slot >> 3, pc = slot & 7;
    const int src_key = key < a.t ? key : a.t - 1;
    const int kc = pc ^ ((key >> 1) & 7);
    const int vc = pc ^ (((key >> 1) & 1) << 2);
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)(kg + src_key * row_stride + kc * 8),
        (__attribute__((address_space(3))) void*)(ks + slot0 * 16), 16, 0, 0);
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)(vg + src_key * row_stride + vc * 8),
        (__attribute__((address_space(3))) void*)(vs + slot0 * 16), 16, 0, 0);
  }
  // Q fragments come straight from global: lane (q = r, half h) holds Q[q][16s + 8h .. +7].
  // The first tile's fragments are requested together with the K/V DMA, the next tile's at the
  // top of the current tile, so their latency is never exposed.
  auto load_q = [&](int qt, f16x8 (&dst)[4]) {
    int qrow = qt * 32 + r;
    qrow = qrow < a.t ? qrow : a.t - 1;
#pragma unroll
    for (int s = 0; s < 4; ++s)
      dst[s] = *reinterpret_cast<const f16x8*>(qg + qrow * row_stride + 16 * s + 8 * h);
  };
  f16x8 qf[4], qn[4];
  load_q(wave, qf);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // explicit: see sim_glds_retire_and_sync()
  __syncthreads();
  const int nqt = (a.nq + 31) >> 5;
  const float ninf = -__builtin_huge_valf();
  const int grp = lane >> 4, li = lane & 15;

  for (int qt = wave; qt < nqt; qt += 4) {
    // opaque per-iteration offset: stops hipcc from hoisting the ~80 loop-invariant LDS addresses
    // out of the loop (that cost 256 VGPRs + scratch spills); they are re-derived from `lane` here
    int opaque = 0;
    asm volatile("" : "+v"(opaque));
    const char* ksl = ks + opaque;
    const char* vsl = vs + opaque;
    const int q0 = qt * 32;
    if (qt + 4 < nqt) load_q(qt + 4, qn);

    // ---- S^T = K . Q^T ----
#ifdef HCIR_ATTN_ABL   // timing ablation (wrong results): one key tile of compute, all memory traffic
    constexpr int NKC = 1;
#else
    constexpr int NKC = NKT;
#endif
    f32x16 sc[NKT];
#pragma unroll
    for (int kt = 0; kt < NKC; ++kt) {
#pragma unroll
      for (int i = 0; i < 16; ++i) sc[kt][i] = 0.f;
      const int key = kt * 32 + r;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int c = 2 * s + h;
        const f16x8 kf =
            *reinterpret_cast<const f16x8*>(ksl + key * 128 + ((c ^ ((key >> 1) & 7)) << 4));
        sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], sc[kt], 0, 0, 0);
      }
    }

    // ---- softmax over keys (registers + one cross-half exchange); only the LAST key tile can
    //      hold keys >= T (TP - T < 32), so only it is masked
    float mx = ninf;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int key = (NKC - 1) * 32 + acc_row(i, h);
      sc[NKC - 1][i] = key < a.t ? sc[NKC - 1][i] : ninf;
    }
#pragma unroll
    for (int kt = 0; kt < NKC; ++kt)
#pragma unroll
      for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sc[kt][i]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mxs = mx * a.scale_log2e;
    float sum = 0.f;
    // ---- P = exp2(S scale log2e - max), O^T = V^T . P, software-pipelined over the key tiles: the exponentials of
    //      key tile kt + 1 stand between the transposed V reads and the MFMAs of key tile kt IN ONE BASIC BLOCK (the
    //      lse store, a branch, moved behind the loop), so the 16 x (fma, v_exp_f32, add) chains issue in the shadow
    //      of the matrix pipe.  As one block of 112 v_exp_f32 in front of the PV MFMAs (what hipcc emitted for the
    //      plain loop nest) the wave's matrix pipe idled for ~1800 cycles per query tile.
#define HCIR_EXP_TILE(KT)                                                                            \
  _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                                    \
    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[KT][i], a.scale_log2e, -mxs));           \
    sc[KT][i] = p;                                                                                    \
    sum += p;                                                                                         \
  }
    f32x16 oacc[2];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int i = 0; i < 16; ++i) oacc[t2][i] = 0.f;
    HCIR_EXP_TILE(0)
#pragma unroll
    for (int kt = 0; kt < NKC; ++kt) {
      f16x8 pf[2], vf[2][2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[s][j] = (_Float16)sc[kt][8 * s + j];
#pragma unroll
        for (int hdt = 0; hdt < 2; ++hdt) {
          // transposed read: this lane supplies row (kb + li>>2), columns c0 + 4*(li&3) .. +3
          const int c0 = 32 * hdt + 16 * (grp & 1) + 4 * (li & 3);
          const int kb = 32 * kt + 16 * s + 4 * (grp >> 1) + (li >> 2);
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            const int key = kb + 8 * half;
            const int col = c0 ^ (((key >> 1) & 1) << 5);
            const fp16x4_t v4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                (fp16x4_t __attribute__((address_space(3)))*)(vsl + key * 128 + col * 2));
#pragma unroll
            for (int e = 0; e < 4; ++e) vf[s][hdt][4 * half + e] = (_Float16)v4[e];
          }
        }
      }
      if (kt + 1 < NKC) {
        HCIR_EXP_TILE(kt + 1)
      }
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int hdt = 0; hdt < 2; ++hdt)
          oacc[hdt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[s][hdt], pf[s], oacc[hdt], 0, 0, 0);
#ifdef HCIR_ATTN_SGB   // build flag (A/B): force one MFMA per 12 VALU of the neighbouring exponentials
      if (kt + 1 < NKC) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
        }
      }
#endif
    }
#undef HCIR_EXP_TILE
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    if (a.lse && h == 0 && q0 + r < a.nq)   // p = exp2(s * scale_log2e - lse): what hcir_attn_bwd recomputes P from
      a.lse[(b * a.h + head) * (int64_t)a.t + q0 + r] = mxs + __builtin_amdgcn_logf(sum);

    // ---- store: the lane owns query row q0 + r and 4-dim pieces of it; a direct store would write 16 B of 32
    //      different rows per instruction (partial lines).  The wave transposes its 32 x 64 tile through 4 KB of
    //      LDS (16-B chunks XOR-swizzled with row & 7) and stores whole 128-B rows: 8 lanes x 16 B.
    if constexpr (kTrans) {
      char* ot = lds + 2 * TP * 128 + wave * 4096 + opaque;
#pragma unroll
      for (int hdt = 0; hdt < 2; ++hdt) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          f16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (_Float16)(oacc[hdt][4 * g4 + e] * inv);
          const int chunk = 4 * hdt + g4;  // 16-B chunk = 8 head dims; h picks its 8-B half
          *reinterpret_cast<f16x4*>(ot + r * 128 + ((chunk ^ (r & 7)) << 4) + 8 * h) = o;
        }
      }
      const int rr = lane >> 3, cc = lane & 7;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + rr;
        const u32x4 v = *reinterpret_cast<const u32x4*>(ot + row * 128 + ((cc ^ (row & 7)) << 4));
        const int q = q0 + row;
        if (q < a.nq)
          *reinterpret_cast<u32x4*>(a.out + (b * a.nq + q) * ((int64_t)a.h * 64) + head * 64 + cc * 8) = v;
      }
    }
    if constexpr (!kTrans) {
      const int q = q0 + r;
      if (q < a.nq) {
        _Float16* orow = a.out + (b * a.nq + q) * ((int64_t)a.h * 64) + head * 64;
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
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = qn[s];
  }
}

// --------------------------------------------------------------------------
// T in (192, 224] (ViT-B/16: 197 tokens), all query rows: PERSISTENT workgroup of seven waves, wave w owns query tile
// w of every (b, head) item the workgroup walks over; two K / V image pairs in LDS, the next item's pair filled by
// LDS-DMA (lds_dma16: invisible to hipcc, see common.h) and the next item's Q fragments requested while the current
// item computes; one barrier per item.  The 4-wave kernel above loads, computes and stores in sequence and relies on
// its second co-resident workgroup for overlap: with one key tile of compute it takes 197 us per launch at 880 images
// (all bytes moved), with all seven 286 us - 89 us of compute standing outside the memory time.  (Round 2's persistent
// attempt gained 2.5 %: its pending transfers were retired by the compiler in front of every transposed V read.)
// Same S / softmax / PV code, same LDS images, same results as attn_fwd_kernel<7>.
// --------------------------------------------------------------------------
constexpr int kF2NKT = 7, kF2Rows = 32 * kF2NKT, kF2Img = kF2Rows * 128;

__global__ __launch_bounds__(64 * kF2NKT, 1) void attn_fwd2_kernel(AttnArgs a, int items) {
  constexpr int NKT = kF2NKT;
  __shared__ __attribute__((aligned(16))) char lds[4 * kF2Img + kF2NKT * 4096];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int grp = lane >> 4, li = lane & 15;
  const int row_stride = 3 * a.h * 64;   // elements between tokens
  const uint32_t lds0 = lds_addr(lds);
  const float ninf = -__builtin_huge_valf();
  const int q0 = wave * 32;
  const bool active = q0 < a.nq;                 // this wave's query tile exists
  auto item_base = [&](int item) { return ((int64_t)(item / a.h) * a.t) * row_stride + (item % a.h) * 64; };

  // K / V images of `item` into buffer `buf`: 2 x 28 wave instructions of 8 rows, four of each per wave
  auto dma_kv = [&](int item, int buf) {
    const _Float16* kg = a.qkv + item_base(item) + (int64_t)a.h * 64;
    const _Float16* vg = kg + (int64_t)a.h * 64;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ii = wave + u * NKT;
      const int key = ii * 8 + (lane >> 3), pc = lane & 7;
      const int src = key < a.t ? key : a.t - 1;   // past T: the last real key (finite; masked / P = 0)
      const uint32_t rowoff = (uint32_t)(src * row_stride * 2);
      lds_dma16(kg, rowoff + ((pc ^ ((key >> 1) & 7)) << 4), lds0 + buf * 2 * kF2Img + ii * 1024);
      lds_dma16(vg, rowoff + ((pc ^ (((key >> 1) & 1) << 2)) << 4), lds0 + buf * 2 * kF2Img + kF2Img + ii * 1024);
    }
  };
  f16x8 qf[4], qn[4];
  auto load_q = [&](int item, f16x8 (&dst)[4]) {
    int qrow = q0 + r;
    qrow = qrow < a.t ? qrow : a.t - 1;
    const char* qg = reinterpret_cast<const char*>(a.qkv + item_base(item));
    const uint32_t off = (uint32_t)(qrow * row_stride * 2 + 16 * h);
#pragma unroll
    for (int s = 0; s < 4; ++s) dst[s] = *reinterpret_cast<const f16x8*>(qg + (off + 32 * s));
  };

  int item = blockIdx.x, buf = 0;
  if (item < items) {
    dma_kv(item, 0);
    load_q(item, qf);
  }
  while (item < items) {
    // this item's images and Q fragments have landed (issued an item ago).  (A counted wait that leaves the previous
    // item's row stores in flight is of no use here: the stores sit under per-lane predicates, so the compiler's own
    // wait for the Q registers below counts none of them and drains everything anyway.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(qf[s]));
    __syncthreads();   // ... and every wave is done with the other buffer (the previous item's)
    const int next = item + gridDim.x;
    if (next < items) {
      dma_kv(next, buf ^ 1);
      load_q(next, qn);
    }
    if (active) {
      int opaque = 0;
      asm volatile("" : "+v"(opaque));
      const char* ksl = lds + buf * 2 * kF2Img + opaque;
      const char* vsl = ksl + kF2Img;
      const int head = item % a.h;
      const int64_t b = item / a.h;
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
          const f16x8 kf = *reinterpret_cast<const f16x8*>(ksl + key * 128 + ((c ^ ((key >> 1) & 7)) << 4));
          sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], sc[kt], 0, 0, 0);
        }
      }
      // ---- softmax over keys; only the last key tile can hold keys >= T
      float mx = ninf;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = (NKT - 1) * 32 + acc_row(i, h);
        sc[NKT - 1][i] = key < a.t ? sc[NKT - 1][i] : ninf;
      }
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sc[kt][i]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float mxs = mx * a.scale_log2e;
      float sum = 0.f;
#define HCIR_EXP_TILE2(KT)                                                                           \
  _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                                    \
    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[KT][i], a.scale_log2e, -mxs));           \
    sc[KT][i] = p;                                                                                    \
    sum += p;                                                                                         \
  }
      f32x16 oacc[2];
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[t2][i] = 0.f;
      HCIR_EXP_TILE2(0)
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        f16x8 pf[2], vf[2][2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
          for (int j = 0; j < 8; ++j) pf[s][j] = (_Float16)sc[kt][8 * s + j];
#pragma unroll
          for (int hdt = 0; hdt < 2; ++hdt) {
            const int c0 = 32 * hdt + 16 * (grp & 1) + 4 * (li & 3);
            const int kb = 32 * kt + 16 * s + 4 * (grp >> 1) + (li >> 2);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
              const int key = kb + 8 * half;
              const int col = c0 ^ (((key >> 1) & 1) << 5);
              const fp16x4_t v4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                  (fp16x4_t __attribute__((address_space(3)))*)(vsl + key * 128 + col * 2));
#pragma unroll
              for (int e = 0; e < 4; ++e) vf[s][hdt][4 * half + e] = (_Float16)v4[e];
            }
          }
        }
        if (kt + 1 < NKT) {
          HCIR_EXP_TILE2(kt + 1)
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int hdt = 0; hdt < 2; ++hdt)
            oacc[hdt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[s][hdt], pf[s], oacc[hdt], 0, 0, 0);
      }
#undef HCIR_EXP_TILE2
      sum += __shfl_xor(sum, 32);
      const float inv = 1.0f / sum;
      // ---- whole 128-B rows through the wave's 4 KB transposition tile, then the lse
      char* ot = lds + 4 * kF2Img + wave * 4096 + opaque;
#pragma unroll
      for (int hdt = 0; hdt < 2; ++hdt) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          f16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (_Float16)(oacc[hdt][4 * g4 + e] * inv);
          const int chunk = 4 * hdt + g4;
          *reinterpret_cast<f16x4*>(ot + r * 128 + ((chunk ^ (r & 7)) << 4) + 8 * h) = o;
        }
      }
      const int rr = lane >> 3, cc = lane & 7;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + rr;
        const u32x4 v = *reinterpret_cast<const u32x4*>(ot + row * 128 + ((cc ^ (row & 7)) << 4));
        const int q = q0 + row;
        if (q < a.nq)
          *reinterpret_cast<u32x4*>(a.out + (b * a.nq + q) * ((int64_t)a.h * 64) + head * 64 + cc * 8) = v;
      }
      if (a.lse && h == 0 && q0 + r < a.nq)
        a.lse[(b * a.h + head) * (int64_t)a.t + q0 + r] = mxs + __builtin_amdgcn_logf(sum);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = qn[s];
    item = next;
    buf ^= 1;
  }
}

// --------------------------------------------------------------------------
// Any head_dim that is a multiple of 16 up to 128 (vit_huge_patch14: 1280 / 16 = 80, HP/src/models_vit.py:266-270).
// Same algorithm and MFMA operand maps as attn_fwd_kernel; correctness-first staging: K and V go through registers
// into plain (unswizzled) LDS images, K rows HD * 2 bytes, V rows padded with zeros to a multiple of 32 dims so that
// the O^T = V^T . P tiles of 32 dims need no edge case (the padded output rows are computed and not stored).
// The tuned head_dim-64 kernel above is what every BASELINE config runs.
// --------------------------------------------------------------------------
template <int HD, int NKT>
__global__ __launch_bounds__(256, 1) void attn_fwd_generic_kernel(AttnArgs a) {
  constexpr int TP = 32 * NKT;
  constexpr int HDP = (HD + 31) / 32 * 32;     // V / output dims padded to whole 32-row MFMA tiles
  constexpr int KPB = HD * 2, VPB = HDP * 2;   // row pitches in bytes
  constexpr int NS = HD / 16, NDT = HDP / 32;
  static_assert(HD % 16 == 0 && HD <= 128, "head_dim");
  __shared__ __attribute__((aligned(16))) char lds[TP * (KPB + VPB)];
  char* ks = lds;
  char* vs = lds + TP * KPB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int head = blockIdx.x % a.h;
  const int64_t b = blockIdx.x / a.h;
  const int64_t row_stride = (int64_t)3 * a.h * HD;
  const _Float16* base = a.qkv + b * a.t * row_stride + head * HD;
  const _Float16* qg = base;
  const _Float16* kg = base + (int64_t)a.h * HD;
  const _Float16* vg = base + (int64_t)2 * a.h * HD;
  for (int slot = tid; slot < TP * (HDP / 8); slot += 256) {
    const int key = slot / (HDP / 8), c = slot % (HDP / 8);
    const int src_key = key < a.t ? key : a.t - 1;
    u32x4 kv = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
    if (c * 8 < HD) {
      kv = *reinterpret_cast<const u32x4*>(kg + src_key * row_stride + c * 8);
      vv = *reinterpret_cast<const u32x4*>(vg + src_key * row_stride + c * 8);
      *reinterpret_cast<u32x4*>(ks + key * KPB + c * 16) = kv;
    }
    *reinterpret_cast<u32x4*>(vs + key * VPB + c * 16) = vv;
  }
  __syncthreads();
  const int nqt = (a.nq + 31) >> 5;
  const float ninf = -__builtin_huge_valf();
  const int grp = lane >> 4, li = lane & 15;
  for (int qt = wave; qt < nqt; qt += 4) {
    const int q0 = qt * 32;
    int qrow = q0 + r;
    qrow = qrow < a.t ? qrow : a.t - 1;
    f16x8 qf[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) qf[s] = *reinterpret_cast<const f16x8*>(qg + qrow * row_stride + 16 * s + 8 * h);
    f32x16 sc[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int i = 0; i < 16; ++i) sc[kt][i] = 0.f;
      const int key = kt * 32 + r;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const f16x8 kf = *reinterpret_cast<const f16x8*>(ks + key * KPB + (2 * s + h) * 16);
        sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], sc[kt], 0, 0, 0);
      }
    }
    float mx = ninf;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int key = (NKT - 1) * 32 + acc_row(i, h);
      sc[NKT - 1][i] = key < a.t ? sc[NKT - 1][i] : ninf;
    }
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sc[kt][i]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mxs = mx * a.scale_log2e;
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[kt][i], a.scale_log2e, -mxs));
        sc[kt][i] = p;
        sum += p;
      }
    }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    f32x16 oacc[NDT];
#pragma unroll
    for (int t2 = 0; t2 < NDT; ++t2)
#pragma unroll
      for (int i = 0; i < 16; ++i) oacc[t2][i] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        f16x8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (_Float16)sc[kt][8 * s + j];
#pragma unroll
        for (int hdt = 0; hdt < NDT; ++hdt) {
          const int c0 = 32 * hdt + 16 * (grp & 1) + 4 * (li & 3);
          const int kb = 32 * kt + 16 * s + 4 * (grp >> 1) + (li >> 2);
          f16x8 vf;
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            const int key = kb + 8 * half;
            const fp16x4_t v4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                (fp16x4_t __attribute__((address_space(3)))*)(vs + key * VPB + c0 * 2));
#pragma unroll
            for (int e = 0; e < 4; ++e) vf[4 * half + e] = (_Float16)v4[e];
          }
          oacc[hdt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, oacc[hdt], 0, 0, 0);
        }
      }
    }
    const int q = q0 + r;
    if (q < a.nq) {
      _Float16* orow = a.out + (b * a.nq + q) * ((int64_t)a.h * HD) + head * HD;
#pragma unroll
      for (int hdt = 0; hdt < NDT; ++hdt) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int dim = 32 * hdt + 8 * g4 + 4 * h;
          if (dim < HD) {
            f16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (_Float16)(oacc[hdt][4 * g4 + e] * inv);
            *reinterpret_cast<f16x4*>(orow + dim) = o;
          }
        }
      }
    }
  }
}

template <int HD>
static int attn_generic_launch(const AttnArgs& a, int nqt, dim3 grid, hipStream_t st) {
#define LAUNCHG(N) hipLaunchKernelGGL((attn_fwd_generic_kernel<HD, N>), grid, dim3(256), 0, st, a)
  switch (nqt) {
    case 1: LAUNCHG(1); break;
    case 2: LAUNCHG(2); break;
    case 3: LAUNCHG(3); break;
    case 4: LAUNCHG(4); break;
    case 5: LAUNCHG(5); break;
    case 6: LAUNCHG(6); break;
    case 7: LAUNCHG(7); break;
    case 8: LAUNCHG(8); break;
    default: LAUNCHG(9); break;
  }
#undef LAUNCHG
  return 0;
}

}  // namespace

static int attn_fwd_launch(const void* qkv, int64_t b, int32_t t, int32_t h, int32_t hd, float scale, int32_t nq,
                           void* out, float* lse, void* stream) {
  HCIR_ENTER();
  if (!qkv || !out || b <= 0 || t <= 0 || h <= 0 || nq <= 0 || nq > t) return HCIR_ERR_INVALID;
  if (t > 288) return HCIR_ERR_UNSUPPORTED;
  if (hd != 64 && (lse || (hd != 32 && hd != 48 && hd != 80 && hd != 96 && hd != 128))) return HCIR_ERR_UNSUPPORTED;
  if (b * h > 0x7fffffff) return HCIR_ERR_INVALID;
  AttnArgs a{static_cast<const _Float16*>(qkv), static_cast<_Float16*>(out), t, h, nq,
             scale * 1.44269504088896340736f, lse};
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nqt = (t + 31) / 32;
  const dim3 grid((unsigned)(b * h));
  if (hd != 64) {   // generic (unswizzled, register-staged) kernel
    if (hd == 32) attn_generic_launch<32>(a, nqt, grid, st);
    else if (hd == 48) attn_generic_launch<48>(a, nqt, grid, st);
    else if (hd == 80) attn_generic_launch<80>(a, nqt, grid, st);
    else if (hd == 96) attn_generic_launch<96>(a, nqt, grid, st);
    else attn_generic_launch<128>(a, nqt, grid, st);
    HCIR_LAUNCH_CHECK();
    return HCIR_OK;
  }
#ifndef HCIR_ATTN_FWD_V1   // build flag: the 4-wave kernel at every T (A/B runs)
  if (nqt == kF2NKT && nq == t && b * h >= 512) {
    const int items = (int)(b * h);
    hipLaunchKernelGGL(attn_fwd2_kernel, dim3(256), dim3(64 * kF2NKT), 0, st, a, items);
    HCIR_LAUNCH_CHECK();
    return HCIR_OK;
  }
#endif
#define LAUNCH(N) hipLaunchKernelGGL(attn_fwd_kernel<N>, grid, dim3(256), 0, st, a)
  // the kernel is built for NKT key tiles; 4 waves walk the query tiles
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

extern "C" int hcir_attn_fwd(const void* qkv, int64_t b, int32_t t, int32_t h, int32_t hd, float scale, int32_t nq,
                             void* out, void* stream) {
  return attn_fwd_launch(qkv, b, t, h, hd, scale, nq, out, nullptr, stream);
}

// training forward: all T query rows, and the per-row log2-sum-exp for hcir_attn_bwd
extern "C" int hcir_attn_fwd_lse(const void* qkv, int64_t b, int32_t t, int32_t h, int32_t hd, float scale,
                                 void* out, float* lse, void* stream) {
  if (!lse) return HCIR_ERR_INVALID;
  return attn_fwd_launch(qkv, b, t, h, hd, scale, t, out, lse, stream);
}
