// sim_topk.hip — brute-force query x gallery cosine / inner-product top-k.
//
// Replaces sklearn KNeighborsClassifier(metric="cosine").kneighbors
// (HP/src/classification_engine.py:80-82), cosine_similarity + argsort
// (src/models/hair_encoder.py:193-194) and torch.mm + torch.sort
// (HP/src/neg_sampling.py:37,45-51).  See include/hcir.h for the contract.
//
// Structure (DESIGN.md "sim_topk"):
//   scan   : every workgroup streams gallery tiles (sim_core.h) against one block
//            of <=128 queries; each lane owns one query column and keeps a sorted
//            top-KP list in registers; lists of a workgroup are merged through LDS
//            and written as one partial list per (workgroup, query).
//   merge  : one wave per query merges the partial lists (k rounds of wave arg-max
//            over per-lane cached list heads).
//   prefix : for big galleries the first S rows are scanned and merged first; the
//            k-th score of that prefix is a per-query floor for the main scan, so
//            that register-list insertions become rare there (a gallery row with
//            score <= floor can never enter the final top-k: every prefix row has
//            a smaller index and wins the tie).
//   k > 64 : repeated passes, each taking the next <=64 ranks below a per-query
//            ceiling (val, idx) left by the previous pass.
//   > 128 queries (k <= 16, fp16/bf16): the rows behind the prefix go through
//            sim_scan_big_kernel - the GEMM's 256 x 256 tile, no lists: the rare rows above
//            the floor are appended to per-query candidate buffers; a device flag gates the
//            list-keeping scan as fallback if a buffer overflows.
#include "sim_core.h"
#include <math.h>
#include <stdlib.h>

#ifndef HCIR_SCAN_AUX
#define HCIR_SCAN_AUX 2  // cache policy of the once-read gallery stream: 2 = nt (non-temporal), +5 % GB/s
#endif

#ifdef HCIR_DIAG_STAMPS
// diagnostic build only (tools/build_variant.sh stamps "-DHCIR_DIAG_STAMPS"): wall-clock stamps (100 MHz) of the
// floorless (prefix) launches, 8 per workgroup, read back through hcir_debug_stamps
__device__ unsigned long long g_stamps[1024 * 8];
#define HCIR_STAMP(i)                                                                              \
  do {                                                                                             \
    if (!a.floor_val && threadIdx.x == 0 && blockIdx.x < 512)                                      \
      g_stamps[(blockIdx.y * 512 + blockIdx.x) * 8 + (i)] = __builtin_amdgcn_s_memrealtime();     \
  } while (0)
#else
#define HCIR_STAMP(i) \
  do {                \
  } while (0)
#endif

namespace {

constexpr float kNegInf = -__builtin_huge_valf();

// Sorted (score desc, arrival order) register list.  Callers feed increasing row
// indices, so equal scores keep the smaller index first.  Precondition: s > v[KP-1].
template <int KP>
__device__ __forceinline__ void topk_insert(float (&v)[KP], int (&id)[KP], float s, int i) {
#pragma unroll
  for (int j = KP - 1; j > 0; --j) {
    const bool up = s > v[j - 1];
    const bool here = s > v[j];
    v[j] = up ? v[j - 1] : (here ? s : v[j]);
    id[j] = up ? id[j - 1] : (here ? i : id[j]);
  }
  const bool top = s > v[0];
  v[0] = top ? s : v[0];
  id[0] = top ? i : id[0];
}
template <int KP>
__device__ __forceinline__ float topk_kth(const float (&v)[KP], int k) {
  // select chain kept opaque: hipcc otherwise rewrites it as v[k-1] through scratch
  float t = v[KP - 1];
#pragma unroll
  for (int j = 0; j < KP - 1; ++j) {
    t = (j == k - 1) ? v[j] : t;
    asm volatile("" : "+v"(t));
  }
  return t;
}

// (score, row) as ONE 64-bit key: [orderable(score) : ~row]; "better" (score desc, row asc) is the unsigned
// maximum, an empty slot is key 0.  Used by the in-workgroup and the final merges.
__device__ __forceinline__ uint64_t merge_key(float v, int id) {
  if (id < 0) return 0ull;
  uint32_t u = __builtin_bit_cast(uint32_t, v);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((uint64_t)u << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)id);
}
__device__ __forceinline__ float merge_key_val(uint64_t k) {
  uint32_t u = (uint32_t)(k >> 32);
  u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
  return __builtin_bit_cast(float, u);
}
__device__ __forceinline__ int merge_key_idx(uint64_t k) { return (int)(0xFFFFFFFFu - (uint32_t)k); }

// Batch insertion (scan epilogue): when many of a lane's 16 scores of one 32 x 32 MFMA tile are candidates - every
// one of them while the lists fill from a workgroup's first tile - the slot-by-slot insertion costs ~115 VALU
// instructions per slot for the whole wave.  The batch path sorts the 16 keys with a bitonic network (80
// compare-exchanges), takes max(list[j], batch[KP-1-j]) - the KP best of the union, as a bitonic sequence - and
// finishes with one bitonic merge: ~130 (KP = 16) compare-exchanges of 5 instructions, whatever the number of
// candidates.  Keys are merge_key's (score, row) words: all distinct, so the network needs no stability.
__device__ __forceinline__ void key_ce_desc(uint64_t& a, uint64_t& b) {  // a <- better, b <- worse
  const bool sw = b > a;
  const uint64_t t = sw ? b : a;
  b = sw ? a : b;
  a = t;
}
template <int N>
__device__ __forceinline__ void key_bitonic_merge_desc(uint64_t (&x)[N]) {  // bitonic -> sorted, best first
#pragma unroll
  for (int st = N / 2; st >= 1; st >>= 1)
#pragma unroll
    for (int i = 0; i < N; ++i)
      if ((i & st) == 0) key_ce_desc(x[i], x[i + st]);
}
__device__ __forceinline__ void key_bitonic_sort16_desc(uint64_t (&x)[16]) {
#pragma unroll
  for (int k = 2; k <= 16; k <<= 1)
#pragma unroll
    for (int j = k >> 1; j >= 1; j >>= 1)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int l = i ^ j;
        if (l > i) {
          if ((i & k) == 0)
            key_ce_desc(x[i], x[l]);
          else
            key_ce_desc(x[l], x[i]);
        }
      }
}
constexpr int kBatchMin = 8;  // candidates slots (wave union) from which the batch path is taken

struct ScanArgs {
  const void* q;
  const void* g;
  const float* qn;         // [nq] or null
  const float* gn;         // [ng] or null
  const float* floor_val;  // [nq] or null: only scores > floor are candidates
  const float* ceil_val;   // [nq] or null: only (score, idx) strictly after
  const int* ceil_idx;     //   (ceil_val, ceil_idx) in (desc, asc) order are candidates
  float* part_val;         // [nparts][nq][KP]
  int* part_idx;
  int64_t nq, row_begin, row_end;  // gallery rows [row_begin, row_end)
  int d, k;
  int shared_stream;  // > 1 query block reads every gallery tile: keep the stream in L2 (no `nt`)
  const int* gate;    // optional device flag: the launch does nothing unless *gate != 0 (fallback of the big scan)
  // candidate-append mode (template CAND): no lists; every score above the floor goes to the query's buffer
  float* cand_val;    // [cap][nq]
  int* cand_idx;      // [cap][nq]
  int* cand_cnt;      // [nq]
  int* overflow;      // [1]
  int cap;
  int cand_query_major;  // 1: buffers are [nq][cap] (what the select kernels read, coalesced); 0: [cap][nq]
  int floor_groups;   // floor = min over this many [nq] arrays behind floor_val (group floors of the prefix)
  int floor_inclusive;  // 1: candidates are scores >= floor (the scan covers the rows the floor came from)
};

// BATCH: the launch fills its lists from nothing (no floor: prefix scans, single scans of small galleries, the
// k > 64 passes) - the epilogue gets the batch-insertion path.  A separate instantiation, and only of the
// one-query-tile geometry: the path's ~100 extra live registers spill in the two-query-tile kernels at two
// workgroups per CU (measured: the 64-query main scan 260 -> 780 us) and at one workgroup per CU they lose what the
// batch gains (64-query call 333 -> 340 us); running a 64-query prefix as two 32-query blocks of this kernel was a
// wash (prefix 52 -> 49 us).  The streaming launches behind a floor keep the lean kernel.
// (KP = 16, one query tile x two wave columns, two row groups: 24 KB stages -> THREE workgroups per CU; the 64-query
// geometry of the streaming launches, see make_plan)
template <typename T, int KP, int QT, int WQ, int WGG, bool GLDS, bool CAND = false, bool BATCH = false>
__global__ __launch_bounds__(256, ((KP == 16 && QT == 1 && WQ == 2 && WGG == 2 && GLDS && !BATCH)
                                       ? 3
                                       : ((KP <= 32 && GLDS) ? 2 : 1))) void sim_topk_scan(ScanArgs a) {
  using Cfg = SimCfg<T, WGG, WQ, QT>;
  constexpr int EPS = SimElem<T>::kPerStage;
  constexpr int NSRC = 2 * WGG;  // lists per query inside a workgroup
  // queries merged per LDS round (power of two, the lists must fit the staging LDS)
  constexpr int QR_FIT = Cfg::LDS_BYTES / (NSRC * KP * 8);
  constexpr int QR = QR_FIT >= Cfg::QB ? Cfg::QB : (QR_FIT >= 64 ? 64 : (QR_FIT >= 32 ? 32 : 16));
  static_assert(QR * NSRC * KP * 8 <= Cfg::LDS_BYTES, "merge round must fit LDS");
  // LDS-DMA stage ring.  KP = 64 runs one workgroup per CU (128 list registers per lane): a 4-slot ring keeps
  // three stages (108 KB) in flight per CU instead of one; the 2-workgroup configurations keep two slots each.
  // (A 4-slot ring for the one-tile-per-workgroup prefix launches of the other configurations was measured:
  // no gain - their ~80 us is not the stage-latency chain.)
  constexpr int NST = (GLDS && KP == 64) ? 4 : 2;
  static_assert(NST * Cfg::STAGE_BYTES <= 160 * 1024, "stage ring must fit the CU's LDS");
  __shared__ __attribute__((aligned(16))) char lds[NST * Cfg::STAGE_BYTES];

  if (a.gate && *a.gate == 0) return;  // uniform: every wave of the grid reads the same flag
  HCIR_STAMP(0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_g = wave / WQ, wave_q = wave % WQ;
  const int r = lane & 31, h = lane >> 5;
  const T* __restrict__ g = static_cast<const T*>(a.g);
  const T* __restrict__ q = static_cast<const T*>(a.q);
  const int64_t q_row0 = (int64_t)blockIdx.y * Cfg::QB;
  const int64_t q_last = a.nq - 1;

  // per-lane query state (CAND keeps no lists: one dummy entry)
  constexpr int KL = CAND ? 1 : KP;
  float lv[QT][KL];
  int li[QT][KL];
  float thr[QT], qnv[QT], cval[QT];
  int cidx[QT];
  int64_t myq[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
#pragma unroll
    for (int j = 0; j < KL; ++j) {
      lv[qt][j] = kNegInf;
      li[qt][j] = -1;
    }
    myq[qt] = q_row0 + wave_q * (32 * QT) + qt * 32 + r;
    const bool live = myq[qt] < a.nq;
    const int64_t qc = live ? myq[qt] : q_last;
    qnv[qt] = a.qn ? a.qn[qc] : 1.0f;
    thr[qt] = a.floor_val ? a.floor_val[qc] : kNegInf;
    if constexpr (CAND) {
      // group floors of the prefix: every group holds >= k/groups rows at or above its floor, so the minimum
      // is a valid floor for k; padding queries never match
      for (int gidx = 1; gidx < a.floor_groups; ++gidx) thr[qt] = fminf(thr[qt], a.floor_val[gidx * a.nq + qc]);
      if (!live) thr[qt] = __builtin_huge_valf();
    }
    cval[qt] = a.ceil_val ? a.ceil_val[qc] : __builtin_huge_valf();
    cidx[qt] = a.ceil_val ? a.ceil_idx[qc] : -1;
  }
  float floorv[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) floorv[qt] = thr[qt];

  const int nkc = (a.d + EPS - 1) / EPS;
  const int64_t nrows = a.row_end - a.row_begin;
  const int64_t ntiles = (nrows + Cfg::GM - 1) / Cfg::GM;
  const int64_t my_tiles = blockIdx.x < ntiles ? (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  const int64_t nsteps = my_tiles * nkc;
  const int64_t g_last = a.row_end - 1;

  f32x16 acc[2][QT];
#pragma unroll
  for (int gt = 0; gt < 2; ++gt)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[gt][qt][i] = 0.f;

  // Staging.  GLDS (d a multiple of the 128-B stage): LDS-DMA straight into the double buffer,
  // stage step+1 issued right after the barrier that retires stage step.  Otherwise register
  // staging one stage ahead with zero fill past d.
  u32x4 regs[GLDS ? 1 : Cfg::NLOAD];
  auto tile_row0 = [&](int64_t ti) { return a.row_begin + ((int64_t)blockIdx.x + ti * gridDim.x) * Cfg::GM; };
  // DMA sources as 32-bit per-lane byte offsets from a wave-uniform base (tile's first gallery row / block's first
  // query row, + 128 B per k chunk): computed ONCE - they are the same for every full tile - so a stage issues
  // NLOAD loads and nothing else.  (Re-deriving row, clamp and a 64-bit product per piece and stage cost ~300
  // VALU instructions per stage: with one stage in flight per workgroup that was on the critical path of both the
  // one-tile prefix workgroups and the streaming main scan.)  Only the gallery's last tile can be partial: its
  // rows are clamped to the last row when the issue cursor reaches it (it is the last tile of its workgroup).
  constexpr int NPG = Cfg::GM / 32, NPQ = Cfg::QB / 32;  // pieces of a stage: gallery rows, then query rows
  static_assert(NPG + NPQ == Cfg::NLOAD, "pieces");
  constexpr int EPC_ = SimElem<T>::kPerChunk;
  uint32_t goff[NPG], qoff[NPQ];
  auto set_goff = [&](int64_t rows_valid) {  // rows_valid >= GM: full tile
#pragma unroll
    for (int i = 0; i < NPG; ++i) {
      const int slot = tid + Cfg::NT * i;
      int row = slot >> 3;
      const int chunk = (slot & 7) ^ ((row >> 1) & 7);
      row = row < rows_valid ? row : (int)rows_valid - 1;
      goff[i] = (uint32_t)(((int64_t)row * a.d + chunk * EPC_) * (int64_t)sizeof(T));
    }
  };
  if constexpr (GLDS) {
    set_goff(Cfg::GM);
#pragma unroll
    for (int i = 0; i < NPQ; ++i) {
      const int slot = tid + Cfg::NT * (NPG + i);
      const int row = (slot >> 3) - Cfg::GM;
      const int chunk = (slot & 7) ^ (((slot >> 3) >> 1) & 7);
      int64_t qr = q_row0 + row;
      qr = qr > q_last ? q_last : qr;
      qoff[i] = (uint32_t)(((qr - q_row0) * a.d + chunk * EPC_) * (int64_t)sizeof(T));
    }
  }
  // stage s of this workgroup = (tile s / nkc, chunk s % nkc); the issue cursor runs NST-1 stages ahead
  int64_t issue_tile = 0;
  int issue_kc = 0;
  auto issue_stage = [&](int slot) {
    const int64_t r0 = tile_row0(issue_tile);
    if (issue_kc == 0 && r0 + Cfg::GM > a.row_end) set_goff(a.row_end - r0);
    const char* gb = reinterpret_cast<const char*>(g) + (r0 * a.d) * (int64_t)sizeof(T) + issue_kc * 128;
    const char* qb = reinterpret_cast<const char*>(q) + (q_row0 * a.d) * (int64_t)sizeof(T) + issue_kc * 128;
    char* dst = lds + slot * Cfg::STAGE_BYTES + (tid & ~63) * 16;
#ifndef HCIR_SIM_BUILTIN_DMA
#pragma unroll
    for (int i = 0; i < NPG; ++i) {
      if (a.shared_stream || HCIR_SCAN_AUX == 0)
        lds_dma16_v(gb + goff[i], lds_addr(dst) + Cfg::NT * i * 16);
      else
        lds_dma16_v_nt(gb + goff[i], lds_addr(dst) + Cfg::NT * i * 16);
    }
#pragma unroll
    for (int i = 0; i < NPQ; ++i) lds_dma16_v(qb + qoff[i], lds_addr(dst) + Cfg::NT * (NPG + i) * 16);
#else
#pragma unroll
    for (int i = 0; i < NPG; ++i) {
      if (a.shared_stream)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb + goff[i]),
                                         (__attribute__((address_space(3))) void*)(dst + Cfg::NT * i * 16), 16, 0, 0);
      else
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb + goff[i]),
                                         (__attribute__((address_space(3))) void*)(dst + Cfg::NT * i * 16), 16, 0,
                                         HCIR_SCAN_AUX);
    }
#pragma unroll
    for (int i = 0; i < NPQ; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qb + qoff[i]),
                                       (__attribute__((address_space(3))) void*)(dst + Cfg::NT * (NPG + i) * 16), 16,
                                       0, 0);
#endif
    if (++issue_kc == nkc) {
      issue_kc = 0;
      ++issue_tile;
    }
  };
  if (nsteps > 0) {
    if constexpr (GLDS) {
#pragma unroll
      for (int s0 = 0; s0 < NST - 1; ++s0)
        if (s0 < nsteps) issue_stage(s0);
    } else {
      sim_stage_load<T, Cfg>(regs, g, tile_row0(0), g_last, q, q_row0, q_last, a.d, 0, tid);
      sim_stage_store<Cfg>(regs, lds, tid);
    }
  }
  if constexpr (!GLDS) __syncthreads();
  HCIR_STAMP(1);

  int64_t tile_i = 0;  // index among my tiles
  int kc = 0;
  int cur = 0;         // ring slot of stage `step`
  for (int64_t step = 0; step < nsteps; ++step) {
    // next stage (possibly first chunk of my next tile)
    int nkc_next = kc + 1;
    int64_t ntile_i = tile_i;
    if (nkc_next == nkc) {
      nkc_next = 0;
      ntile_i = tile_i + 1;
    }
    const bool has_next = step + 1 < nsteps;
    if constexpr (GLDS) {
      // Stage `step` landed for every wave: the NST-2 younger stages may stay in flight (vmcnt retires in
      // order; the only other vector-memory operations of the loop are the optional gallery-norm loads of the
      // tile epilogue, whose consumers already waited for everything older).  Near the tail fewer stages
      // are outstanding than the count allows: wait for all.
      if (NST > 2 && !a.gn && step + NST - 1 <= nsteps) {
        if constexpr (NST == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * Cfg::NLOAD) : "memory");
        __syncthreads();
      } else {
        sim_glds_retire_and_sync();
      }
      // the slot read in step-1 is free after this barrier: issue stage step+NST-1 into it
      if (step == 0) HCIR_STAMP(2);
      if (step + NST - 1 < nsteps) issue_stage((cur + NST - 1) % NST);
      sim_stage_mfma<T, Cfg, QT>(acc, lds + cur * Cfg::STAGE_BYTES, wave_g, wave_q, lane);
      if (step == nsteps - 1) HCIR_STAMP(3);
    } else {
      if (has_next)
        sim_stage_load<T, Cfg>(regs, g, tile_row0(ntile_i), g_last, q, q_row0, q_last, a.d, nkc_next, tid);
      sim_stage_mfma<T, Cfg, QT>(acc, lds + cur * Cfg::STAGE_BYTES, wave_g, wave_q, lane);
      if (has_next) sim_stage_store<Cfg>(regs, lds + (cur ^ 1) * Cfg::STAGE_BYTES, tid);
    }

    if (kc == nkc - 1) {
      // ---- per-tile epilogue: scale the 16 scores of each 32x32 tile, mask the ones
      //      above the lane's threshold, then pop candidates in row order into the list
      const int64_t row0 = tile_row0(tile_i);
#pragma unroll
      for (int gt = 0; gt < 2; ++gt) {
        const int64_t rbase = row0 + wave_g * 64 + gt * 32 + 4 * h;
        float gnv[16];
        if (a.gn) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int64_t row = rbase + (i & 3) + 8 * (i >> 2);
            gnv[i] = a.gn[row < a.row_end ? row : g_last];
          }
        }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          unsigned mask = 0;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int64_t row = rbase + (i & 3) + 8 * (i >> 2);
            float s = acc[gt][qt][i];
            if (a.gn) s = s * gnv[i];
            if (a.qn) s = s * qnv[qt];
            acc[gt][qt][i] = s;
            bool cand = (row < a.row_end) && (s > thr[qt]);
            if constexpr (CAND) cand = cand || ((row < a.row_end) && a.floor_inclusive && s == thr[qt]);
            if (a.ceil_val)
              cand = cand && ((s < cval[qt]) || (s == cval[qt] && (int)row > cidx[qt]));
            mask |= cand ? (1u << i) : 0u;
          }
          // common case behind the prefix floor: no lane has a candidate -> one ballot, no work.
          // Otherwise a wave-uniform slot loop: acc[..][i] with a scalar i stays in registers
          // (movrel), a per-lane index would be lowered through scratch.
          unsigned any = 0u;
          if (__ballot(mask != 0u) != 0ull) {
            any = mask;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) any |= __shfl_xor(any, off);
            any = __builtin_amdgcn_readfirstlane(any);
          }
          if constexpr (!CAND && BATCH) {
            if (__builtin_popcount(any) >= kBatchMin) {  // wave-uniform
              uint64_t bk[16];
#pragma unroll
              for (int i = 0; i < 16; ++i) {
                const bool c = ((mask >> i) & 1u) && acc[gt][qt][i] > thr[qt];
                bk[i] = c ? merge_key(acc[gt][qt][i], (int)(rbase + (i & 3) + 8 * (i >> 2))) : 0ull;
              }
              key_bitonic_sort16_desc(bk);
              uint64_t lk[KP];
#pragma unroll
              for (int j = 0; j < KP; ++j) lk[j] = merge_key(lv[qt][j], li[qt][j]);
#pragma unroll
              for (int j = 0; j < 16; ++j) {
                const uint64_t o = bk[15 - j];
                lk[KP - 16 + j] = o > lk[KP - 16 + j] ? o : lk[KP - 16 + j];
              }
              key_bitonic_merge_desc<KP>(lk);
#pragma unroll
              for (int j = 0; j < KP; ++j) {
                lv[qt][j] = lk[j] ? merge_key_val(lk[j]) : kNegInf;
                li[qt][j] = lk[j] ? merge_key_idx(lk[j]) : -1;
              }
              thr[qt] = fmaxf(floorv[qt], topk_kth<KP>(lv[qt], a.k));
              any = 0u;
            }
          }
          while (any != 0u) {
            const int i = __builtin_ctz(any);
            any &= any - 1u;
            const float s = acc[gt][qt][i];
            if constexpr (CAND) {
              if ((mask >> i) & 1u) {  // rare behind the floor: one global atomic per candidate
                const int pos = atomicAdd(a.cand_cnt + myq[qt], 1);
                if (pos < a.cap) {
                  const int64_t at = a.cand_query_major ? myq[qt] * a.cap + pos : (int64_t)pos * a.nq + myq[qt];
                  a.cand_val[at] = s;
                  a.cand_idx[at] = (int)(rbase + (i & 3) + 8 * (i >> 2));
                } else {
                  *a.overflow = 1;
                }
              }
            } else {
              if (((mask >> i) & 1u) && s > thr[qt]) {
                topk_insert<KP>(lv[qt], li[qt], s, (int)(rbase + (i & 3) + 8 * (i >> 2)));
                thr[qt] = fmaxf(floorv[qt], topk_kth<KP>(lv[qt], a.k));
              }
            }
          }
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[gt][qt][i] = 0.f;
        }
      }
    }
    if constexpr (!GLDS) __syncthreads();
    kc = nkc_next;
    tile_i = ntile_i;
    cur = (cur + 1 == NST) ? 0 : cur + 1;
  }
  HCIR_STAMP(4);
  if constexpr (!CAND) {  // (CAND: nothing to merge, the candidates are already in the per-query buffers)
  __syncthreads();  // every wave is done with the stage buffers before they are re-used below

  // ---- merge the NSRC lists of each query through LDS, one thread per query,
  //      QR queries per round
  float* mv = reinterpret_cast<float*>(lds);
  int* mi = reinterpret_cast<int*>(lds + QR * NSRC * KP * 4);
  for (int q0r = 0; q0r < Cfg::QB; q0r += QR) {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const int ql = wave_q * (32 * QT) + qt * 32 + r - q0r;
      const int src = wave_g * 2 + h;
      if (ql >= 0 && ql < QR) {
#pragma unroll
        for (int j = 0; j < KP; ++j) {
          mv[(ql * NSRC + src) * KP + j] = lv[qt][j];
          mi[(ql * NSRC + src) * KP + j] = li[qt][j];
        }
      }
    }
    __syncthreads();
    if (tid < QR && q_row0 + q0r + tid < a.nq) {
      const float* lv = mv + tid * NSRC * KP;
      const int* li = mi + tid * NSRC * KP;
      // list heads live in registers as 64-bit keys; only the winner's next entry is re-read from LDS
      // (re-reading all NSRC heads per output through dependent LDS loads took 21 us per workgroup)
      uint64_t hk[NSRC];
      int hp[NSRC];
#pragma unroll
      for (int s = 0; s < NSRC; ++s) {
        hp[s] = 0;
        hk[s] = merge_key(lv[s * KP], li[s * KP]);
      }
      // the merged list is collected in registers and stored once, as 16-B pieces: with a store inside the
      // loop hipcc put `s_waitcnt vmcnt(0)` into every iteration (16 store round trips = 21 us per workgroup)
      float* ov = a.part_val + ((int64_t)blockIdx.x * a.nq + q_row0 + q0r + tid) * KP;
      int* oi = a.part_idx + ((int64_t)blockIdx.x * a.nq + q_row0 + q0r + tid) * KP;
      for (int o0 = 0; o0 < KP; o0 += 16) {  // 16 outputs at a time in registers
        float ovr[16];
        int oir[16];
#pragma unroll
        for (int o = 0; o < 16; ++o) {
          uint64_t best = hk[0];
          int bs = 0;
#pragma unroll
          for (int s = 1; s < NSRC; ++s) {
            const bool take = hk[s] > best;
            best = take ? hk[s] : best;
            bs = take ? s : bs;
          }
          ovr[o] = best ? merge_key_val(best) : kNegInf;
          oir[o] = best ? merge_key_idx(best) : -1;
          // advance the winner's list: ONE LDS read pair at a per-lane address, then select-chain updates
          // (a branch per source serialised eight LDS round trips per output across the divergent lanes)
          int np = 0;
#pragma unroll
          for (int s = 0; s < NSRC; ++s) np = (s == bs) ? hp[s] + 1 : np;
          const bool more = best != 0ull && np < KP;
          const int at = bs * KP + (more ? np : 0);
          const uint64_t nk = more ? merge_key(lv[at], li[at]) : 0ull;
#pragma unroll
          for (int s = 0; s < NSRC; ++s) {
            const bool upd = best != 0ull && s == bs;
            hp[s] = upd ? np : hp[s];
            hk[s] = upd ? nk : hk[s];
          }
        }
#pragma unroll
        for (int o = 0; o < 16; o += 4) {
          *reinterpret_cast<f32x4*>(ov + o0 + o) = (f32x4){ovr[o], ovr[o + 1], ovr[o + 2], ovr[o + 3]};
          *reinterpret_cast<u32x4*>(oi + o0 + o) =
              (u32x4){(uint32_t)oir[o], (uint32_t)oir[o + 1], (uint32_t)oir[o + 2], (uint32_t)oir[o + 3]};
        }
      }
    }
    __syncthreads();
  }
  }  // !CAND
  HCIR_STAMP(5);
}

// --------------------------------------------------------------------------
// Big-tile scan of the rows behind the prefix, for MANY queries (nq > 128, k <= 16, fp16 / bf16, d % 64 == 0).
//
// With >128 queries the scan is MFMA-bound and the 128 x 128 tile of sim_topk_scan (64 x 64 per wave) tops
// out near 0.45-0.55 PFLOP/s; the 256 x 256 tile of the ViT GEMM (gemm.hip: 8 waves, 128 x 64 per wave, two
// 64 KB LDS slots filled by LDS-DMA) runs the same product at about twice that.  Its 128 accumulators leave no
// room for per-lane top-k lists, and it does not need them: behind the prefix floor a candidate is RARE
// (a row enters only if it beats the k-th score of the first rows: ~15 k rows per query in a gallery in
// random order), so the kernel keeps no lists at all - a lane that sees score > floor appends (score, row) to
// the query's candidate buffer through an atomic counter.  The final merge takes the prefix list plus the
// <= cap candidates as one-element lists.  If any query overflows its buffer (a gallery whose later rows
// systematically beat its first ones) a device flag is raised and the list-keeping scan runs instead, gated
// on that flag: exact either way, no host round trip.
//
// MFMA operands, k-order and accumulation are those of sim_topk_scan (32x32x16, gallery row on the MFMA row,
// query on the column, chunks 2ks+h of a 64-wide stage): the scores are bit-identical to that kernel's.
// --------------------------------------------------------------------------
struct BigScanArgs {
  const void* q;
  const void* g;
  const float* floor_val;  // [nq] k-th score of the prefix
  float* cand_val;         // [cap][nq]
  int* cand_idx;           // [cap][nq]
  int* cand_cnt;           // [nq], zeroed by the caller
  int* overflow;           // [1],  zeroed by the caller
  int64_t nq, row_begin, row_end;
  int d, cap;
};

template <typename T>
__global__ __launch_bounds__(512, 2) void sim_scan_big_kernel(BigScanArgs a) {
  constexpr int STAGE = 512 * 128;  // 256 gallery rows then 256 query rows, 128 B (64 elements) each
  __shared__ __attribute__((aligned(16))) char lds[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_g = wave >> 2, wave_q = wave & 3;
  const int r = lane & 31, h = lane >> 5;
  const int64_t q_row0 = (int64_t)blockIdx.y * 256;
  const int nkc = a.d / 64;
  const int64_t nrows = a.row_end - a.row_begin;
  const int64_t ntiles = (nrows + 255) / 256;
  const int64_t my_tiles =
      (int64_t)blockIdx.x < ntiles ? (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  const int64_t nsteps = my_tiles * nkc;
  auto tile_row0 = [&](int64_t ti) { return a.row_begin + ((int64_t)blockIdx.x + ti * gridDim.x) * 256; };

  float flo[2];
  int64_t myq[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    myq[qt] = q_row0 + wave_q * 64 + qt * 32 + r;
    flo[qt] = myq[qt] < a.nq ? a.floor_val[myq[qt]] : __builtin_huge_valf();  // padding queries never match
  }

  // DMA sources: piece = tid + 512 i -> LDS row piece>>3, physical 16-B slot piece&7 holding logical chunk
  // slot ^ ((row>>1)&7).  Pieces 0..3 are gallery rows (offsets from the tile's first row, clamped to the last
  // gallery row), 4..7 query rows (the same for every tile, clamped to the last query).
  const char* qbase = static_cast<const char*>(a.q) + q_row0 * a.d * (int64_t)sizeof(T);
  const char* gbase = nullptr;
  uint32_t goff[4], qoff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = tid + 512 * (4 + i);
    const int row = (piece >> 3) - 256, chunk = (piece & 7) ^ ((((piece >> 3)) >> 1) & 7);
    int64_t qr = row;
    qr = q_row0 + qr > a.nq - 1 ? a.nq - 1 - q_row0 : qr;
    qoff[i] = (uint32_t)((qr * a.d + chunk * 8) * (int64_t)sizeof(T));
  }
  int64_t issue_tile = 0;
  int issue_kc = 0;
  auto set_gallery_sources = [&](int64_t ti) {
    const int64_t r0 = tile_row0(ti);
    gbase = static_cast<const char*>(a.g) + r0 * a.d * (int64_t)sizeof(T);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int piece = tid + 512 * i;
      const int row = piece >> 3, chunk = (piece & 7) ^ ((row >> 1) & 7);
      int64_t gr = row;
      gr = r0 + gr > a.row_end - 1 ? a.row_end - 1 - r0 : gr;
      goff[i] = (uint32_t)((gr * a.d + chunk * 8) * (int64_t)sizeof(T));
    }
  };
  auto issue_piece = [&](int slot, int i) {  // i is a constant after unrolling
#ifndef HCIR_SIM_BUILTIN_DMA   // the transfer outside the compiler's view (common.h lds_dma16): big scan +3.8 %, streaming
                               // scans +0.7..1.4 % against the builtin (kept behind this flag for A/B runs)
    lds_dma16((i < 4 ? gbase : qbase) + issue_kc * 128, i < 4 ? goff[i & 3] : qoff[i & 3],
              lds_addr(lds) + slot * STAGE + ((tid & ~63) + 512 * i) * 16);
#else
    const char* sp = (i < 4 ? gbase + goff[i & 3] : qbase + qoff[i & 3]) + issue_kc * 128;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sp,
                                     (__attribute__((address_space(3))) void*)(lds + slot * STAGE +
                                                                                ((tid & ~63) + 512 * i) * 16),
                                     16, 0, 0);
#endif
  };
  auto issue_advance = [&]() {
    if (++issue_kc == nkc) {
      issue_kc = 0;
      ++issue_tile;
      if (issue_tile < my_tiles) set_gallery_sources(issue_tile);
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int gt = 0; gt < 4; ++gt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[gt][qt][i] = 0.f;

  if (nsteps > 0) {
    set_gallery_sources(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) issue_piece(0, i);
    issue_advance();
  }

  int kc = 0;
  int64_t ti = 0;
  for (int64_t step = 0; step < nsteps; ++step) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // explicit: see sim_glds_retire_and_sync()
    __builtin_amdgcn_s_barrier();
    const char* st = lds + (step & 1) * STAGE;
    const bool do_issue = step + 1 < nsteps;
    const int islot = (int)((step + 1) & 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int chunk = 2 * ks + h;
      u32x4 af[4], bf[2];
#pragma unroll
      for (int gt = 0; gt < 4; ++gt)
        af[gt] = *reinterpret_cast<const u32x4*>(st + sim_slot_off(wave_g * 128 + gt * 32 + r, chunk));
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
        bf[qt] = *reinterpret_cast<const u32x4*>(st + sim_slot_off(256 + wave_q * 64 + qt * 32 + r, chunk));
      if (do_issue && ks < 2) {  // four of the eight DMA pieces of stage step+1 per early k-substep
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_piece(islot, 4 * ks + i);
      }
#pragma unroll
      for (int gt = 0; gt < 4; ++gt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          if constexpr (__is_same(T, _Float16))
            acc[gt][qt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[gt]),
                                                                 __builtin_bit_cast(f16x8, bf[qt]), acc[gt][qt], 0, 0, 0);
          else
            acc[gt][qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[gt]),
                                                                  __builtin_bit_cast(bf16x8, bf[qt]), acc[gt][qt], 0, 0, 0);
        }
    }
    if (do_issue) issue_advance();

    if (++kc == nkc) {
      // ---- tile epilogue: compare against the floor; candidates are rare -> one ballot per 32 x 32 tile
      const int64_t row0 = tile_row0(ti) + wave_g * 128;
#pragma unroll
      for (int gt = 0; gt < 4; ++gt) {
        const int64_t rbase = row0 + gt * 32 + 4 * h;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          unsigned mask = 0;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int64_t row = rbase + (i & 3) + 8 * (i >> 2);
            const bool cand = (row < a.row_end) && (acc[gt][qt][i] > flo[qt]);
            mask |= cand ? (1u << i) : 0u;
          }
          if (__ballot(mask != 0u) != 0ull) {
            unsigned any = mask;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) any |= __shfl_xor(any, off);
            any = __builtin_amdgcn_readfirstlane(any);
            while (any != 0u) {  // wave-uniform slot loop: acc[..][i] with a scalar i stays in registers
              const int i = __builtin_ctz(any);
              any &= any - 1u;
              const float sc = acc[gt][qt][i];
              if ((mask >> i) & 1u) {
                const int pos = atomicAdd(a.cand_cnt + myq[qt], 1);
                if (pos < a.cap) {
                  a.cand_val[(int64_t)pos * a.nq + myq[qt]] = sc;
                  a.cand_idx[(int64_t)pos * a.nq + myq[qt]] = (int)(rbase + (i & 3) + 8 * (i >> 2));
                } else {
                  *a.overflow = 1;
                }
              }
            }
          }
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[gt][qt][i] = 0.f;
        }
      }
      kc = 0;
      ++ti;
    }
  }
}

// --------------------------------------------------------------------------
// merge: one wave per query; lists [nlists][nq][kin] sorted (desc, idx asc),
// empty slots have idx < 0.  Lane l owns lists l, l+64, ... (<= LPL of them) and
// caches each list's head; k_out rounds of wave arg-max.
// --------------------------------------------------------------------------
constexpr int kMergeLPL = 9;  // lists per lane -> 576 lists per pass

template <typename IdxT>
struct MergeArgs {
  const float* vals;
  const IdxT* idx;
  float* out_val;
  int64_t* out_idx;      // final output (idx_base added), or
  int* out_idx32;        // intermediate output
  float* kth_val;        // optional [nq]: value of rank k_out-1 (floor for the main scan)
  int* kth_idx;          // optional [nq]
  int64_t nq, idx_base;
  int nlists, kin, kout;
  // optional extra list per query (the prefix result), [nq][kin_extra]
  const float* extra_val;
  const int* extra_idx;
  int kin_extra;
  const int* nlists_q;   // optional [nq]: lists of this query = min(nlists_q[q], nlists) (candidate buffers)
  const int* gate;       // optional device flag: run only if (*gate != 0) == (gate_want != 0)
  int gate_want;
  int* zero_cnt;         // optional [nq + 1]: cleared here (candidate counters + overflow flag of the big scan)
  int ngroups;           // > 1: blockIdx.y = group g merges lists g, g + ngroups, ... into out[g][nq][kout],
                         //      kth_val[g][nq] (group floors of the prefix); 0 / 1: one group
};

template <typename IdxT, int LPL = kMergeLPL>
__global__ __launch_bounds__(256) void topk_merge_kernel(MergeArgs<IdxT> a) {
  const int lane = threadIdx.x & 63;
  const int64_t qi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (qi >= a.nq) return;
  if (a.gate && ((*a.gate != 0) != (a.gate_want != 0))) return;
  if (a.zero_cnt && lane == 0) {
    a.zero_cnt[qi] = 0;
    if (qi == 0) a.zero_cnt[a.nq] = 0;
  }
  const bool has_extra = a.extra_val != nullptr;
  int nl = a.nlists;
  if (a.nlists_q) nl = a.nlists_q[qi] < nl ? a.nlists_q[qi] : nl;
  const int total = a.nlists + (has_extra ? 1 : 0);

  float cv[LPL];
  int64_t ci[LPL];
  int hd[LPL];
  auto fetch = [&](int list, int pos, float& v, int64_t& id) {
    v = kNegInf;
    id = -1;
    if (list < a.nlists) {
      if (pos < a.kin && list < nl) {
        const int64_t o = ((int64_t)list * a.nq + qi) * a.kin + pos;
        v = a.vals[o];
        id = (int64_t)a.idx[o];
      }
    } else if (list < total) {
      if (pos < a.kin_extra) {
        v = a.extra_val[qi * a.kin_extra + pos];
        id = a.extra_idx[qi * a.kin_extra + pos];
      }
    }
    if (id < 0) v = kNegInf;
  };
#pragma unroll
  for (int j = 0; j < LPL; ++j) {
    hd[j] = 0;
    fetch(lane + 64 * j, 0, cv[j], ci[j]);
  }
  float keep_v = kNegInf;
  int64_t keep_i = -1;
  for (int o = 0; o < a.kout; ++o) {
    float bv = cv[0];
    int64_t bi = ci[0];
    int bj = 0;
#pragma unroll
    for (int j = 1; j < LPL; ++j) {
      const bool take = (ci[j] >= 0) && (bi < 0 || better(cv[j], ci[j], bv, bi));
      bv = take ? cv[j] : bv;
      bi = take ? ci[j] : bi;
      bj = take ? j : bj;
    }
    // wave arg-max under (score desc, idx asc); empty = idx < 0
    float wv = bv;
    int64_t wi = bi;
    int wl = lane;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(wv, off);
      const int64_t oi = __shfl_xor(wi, off);
      const int ol = __shfl_xor(wl, off);
      const bool take = (oi >= 0) && (wi < 0 || better(ov, oi, wv, wi));
      wv = take ? ov : wv;
      wi = take ? oi : wi;
      wl = take ? ol : wl;
    }
    // winner o kept by lane o & 63, stored once per 64 rounds (see topk_merge32_kernel: no store inside the loop)
    if (lane == (o & 63)) {
      keep_v = wv;
      keep_i = wi;
    }
    if ((o & 63) == 63 || o == a.kout - 1) {
      const int oo = (o & ~63) + lane;
      if (oo <= o) {
        a.out_val[qi * a.kout + oo] = keep_i < 0 ? kNegInf : keep_v;
        if (a.out_idx) a.out_idx[qi * a.kout + oo] = keep_i < 0 ? -1 : keep_i + a.idx_base;
        if (a.out_idx32) a.out_idx32[qi * a.kout + oo] = (int)keep_i;
        if (oo == a.kout - 1) {
          if (a.kth_val) a.kth_val[qi] = keep_i < 0 ? kNegInf : keep_v;
          if (a.kth_idx) a.kth_idx[qi] = (int)keep_i;
        }
      }
    }
    if (lane == wl && wi >= 0) {
#pragma unroll
      for (int j = 0; j < LPL; ++j) {
        if (j == bj) {
          hd[j] += 1;
          fetch(lane + 64 * j, hd[j], cv[j], ci[j]);
        }
      }
    }
  }
}

// --------------------------------------------------------------------------
// merge32: the same merge for 32-bit row indices (every intra-GPU merge), latency-tuned.
//   * a (score, row) pair is ONE 64-bit key: [orderable(score) : ~row]; "better" (score desc, row asc) is the
//     unsigned maximum, empty slots are key 0;
//   * the first FOUR entries of each list are fetched up front as two 16-B loads: the k rounds then run from
//     registers (the old kernel re-fetched the winner's next entry from memory in every round, ~1.5 us each);
//     a list whose four are used up is refilled in place (rare);
//   * the wave arg-max is 4 DPP butterfly steps inside each row of 16 lanes + 4 v_readlane + scalar max,
//     instead of 18 ds_bpermute per round.
// --------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ uint64_t dpp_max_u64(uint64_t v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, 0xF, 0xF, true);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, 0xF, 0xF, true);
  const uint64_t o = ((uint64_t)hi << 32) | lo;
  return o > v ? o : v;
}
// maximum over the 64 lanes, returned wave-uniform
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
  v = dpp_max_u64<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_max_u64<0x4E>(v);   // quad_perm [2,3,0,1]
  v = dpp_max_u64<0x141>(v);  // row_half_mirror
  v = dpp_max_u64<0x140>(v);  // row_mirror: every lane of a 16-lane row now holds the row maximum
  uint64_t m = 0;
#pragma unroll
  for (int row = 0; row < 4; ++row) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 16 * row);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 16 * row);
    const uint64_t o = ((uint64_t)hi << 32) | lo;
    m = o > m ? o : m;
  }
  return m;
}

template <int LPL>
__global__ __launch_bounds__(256) void topk_merge32_kernel(MergeArgs<int> a) {
  const int lane = threadIdx.x & 63;
  const int64_t qi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (qi >= a.nq) return;
  if (a.gate && ((*a.gate != 0) != (a.gate_want != 0))) return;
  const int ng_ = a.ngroups > 1 ? a.ngroups : 1, grp = (int)blockIdx.y;
  if (a.zero_cnt && lane == 0 && grp == 0) {
    a.zero_cnt[qi] = 0;
    if (qi == 0) a.zero_cnt[a.nq] = 0;
  }
  const bool has_extra = a.extra_val != nullptr;
  int nl = (a.nlists - grp + ng_ - 1) / ng_;   // lists of this group
  if (a.nlists_q) nl = a.nlists_q[qi] < nl ? a.nlists_q[qi] : nl;
  const int extra_list = nl;  // list index of the optional extra list
  const int64_t gout = (int64_t)grp * a.nq;   // output row offset of the group

  // entries [pos, pos + 4) of a list as keys (0 where the list has ended)
  auto fetch4 = [&](int list, int pos, uint64_t (&k4)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) k4[e] = 0ull;
    const float* pv = nullptr;
    const int* pi = nullptr;
    int len = 0;
    if (list < nl) {
      pv = a.vals + ((int64_t)(list * ng_ + grp) * a.nq + qi) * a.kin;
      pi = a.idx + ((int64_t)(list * ng_ + grp) * a.nq + qi) * a.kin;
      len = a.kin;
    } else if (has_extra && list == extra_list) {
      pv = a.extra_val + qi * a.kin_extra;
      pi = a.extra_idx + qi * a.kin_extra;
      len = a.kin_extra;
    }
    if (pos + 4 <= len && (len & 3) == 0) {  // whole 16-B groups (lists of 16 / 32 / 64 entries)
      const f32x4 v = *reinterpret_cast<const f32x4*>(pv + pos);
      const u32x4 id = *reinterpret_cast<const u32x4*>(pi + pos);
#pragma unroll
      for (int e = 0; e < 4; ++e) k4[e] = merge_key(v[e], (int)id[e]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (pos + e < len) k4[e] = merge_key(pv[pos + e], pi[pos + e]);
    }
  };

  uint64_t kq[LPL][4];
  int used[LPL];
#pragma unroll
  for (int j = 0; j < LPL; ++j) {
    used[j] = 0;
    fetch4(lane + 64 * j, 0, kq[j]);
  }
  // Winner o is KEPT by lane o and the whole result leaves after the loop, 64 outputs per store instruction.
  // (With lane 0 storing inside the loop hipcc put `s_waitcnt vmcnt(0)` in front of every round - the refill
  // load below shares the counter - so each of the k rounds waited for a store round trip: ~1.1 us per round,
  // 18-23 us per merge of 16 ranks and 55 us per merge of 50; k <= 64 per pass, so one register holds it.)
  uint64_t mine = 0ull;
  for (int o = 0; o < a.kout; ++o) {
    uint64_t best = kq[0][0];
    int bj = 0;
#pragma unroll
    for (int j = 1; j < LPL; ++j) {
      const bool take = kq[j][0] > best;
      best = take ? kq[j][0] : best;
      bj = take ? j : bj;
    }
    const uint64_t wm = wave_max_u64(best);
    mine = (lane == (o & 63)) ? wm : mine;
    if ((o & 63) == 63 || o == a.kout - 1) {   // a full wave of results (kout > 64 only from hcir_topk_merge callers)
      const int oo = (o & ~63) + lane;
      if (oo <= o) {
        const bool none = mine == 0ull;
        const float wv = none ? kNegInf : merge_key_val(mine);
        const int wi = none ? -1 : merge_key_idx(mine);
        a.out_val[(gout + qi) * a.kout + oo] = wv;
        if (a.out_idx) a.out_idx[(gout + qi) * a.kout + oo] = none ? -1 : (int64_t)wi + a.idx_base;
        if (a.out_idx32) a.out_idx32[(gout + qi) * a.kout + oo] = wi;
        if (oo == a.kout - 1) {
          if (a.kth_val) a.kth_val[gout + qi] = wv;
          if (a.kth_idx) a.kth_idx[gout + qi] = wi;
        }
      }
    }
    if (wm != 0ull && best == wm) {  // row indices are unique: exactly one lane and one list hold the winner
#pragma unroll
      for (int j = 0; j < LPL; ++j) {
        if (j == bj) {
          kq[j][0] = kq[j][1];
          kq[j][1] = kq[j][2];
          kq[j][2] = kq[j][3];
          kq[j][3] = 0ull;
          used[j] += 1;
          if ((used[j] & 3) == 0) fetch4(lane + 64 * j, used[j], kq[j]);  // its four are used up: refill
        }
      }
    }
  }
}

// --------------------------------------------------------------------------
// merge32q: merge32 with FOUR waves per query (kout <= 16).  The one-wave merge is a serial chain: every one of
// the k rounds walks all LPL cached list heads of a lane (9 - 13 compare-select steps) before the wave maximum -
// 17 - 26 us per launch for 512 - 768 lists, whatever the number of queries (<= 128 of them keep 16 - 32 CUs busy).
// Here wave w of the workgroup takes lists w, w + 4, ... (a quarter of the heads per lane), the four 16-entry
// results meet in LDS as 64 keys, and wave 0 ranks them by counting (lane i compares its key with the other 63):
// rank r < kout stores output r.  Keys are all distinct (row indices are unique), so ranks are a permutation.
// --------------------------------------------------------------------------
template <int LPL>
__global__ __launch_bounds__(256) void topk_merge32q_kernel(MergeArgs<int> a) {
  __shared__ uint64_t part[4][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t qi = blockIdx.x;
  if (a.gate && ((*a.gate != 0) != (a.gate_want != 0))) return;
  if (a.zero_cnt && threadIdx.x == 0) {
    a.zero_cnt[qi] = 0;
    if (qi == 0) a.zero_cnt[a.nq] = 0;
  }
  const bool has_extra = a.extra_val != nullptr;
  int nl = a.nlists;
  if (a.nlists_q) nl = a.nlists_q[qi] < nl ? a.nlists_q[qi] : nl;
  const int extra_list = nl;

  auto fetch4 = [&](int list, int pos, uint64_t (&k4)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) k4[e] = 0ull;
    const float* pv = nullptr;
    const int* pi = nullptr;
    int len = 0;
    if (list < nl) {
      pv = a.vals + ((int64_t)list * a.nq + qi) * a.kin;
      pi = a.idx + ((int64_t)list * a.nq + qi) * a.kin;
      len = a.kin;
    } else if (has_extra && list == extra_list) {
      pv = a.extra_val + qi * a.kin_extra;
      pi = a.extra_idx + qi * a.kin_extra;
      len = a.kin_extra;
    }
    if (pos + 4 <= len && (len & 3) == 0) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(pv + pos);
      const u32x4 id = *reinterpret_cast<const u32x4*>(pi + pos);
#pragma unroll
      for (int e = 0; e < 4; ++e) k4[e] = merge_key(v[e], (int)id[e]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (pos + e < len) k4[e] = merge_key(pv[pos + e], pi[pos + e]);
    }
  };
  // list of (wave, lane, j): wave + 4 (lane + 64 j)
  uint64_t kq[LPL][4];
  int used[LPL];
#pragma unroll
  for (int j = 0; j < LPL; ++j) {
    used[j] = 0;
    fetch4(wave + 4 * (lane + 64 * j), 0, kq[j]);
  }
  uint64_t mine = 0ull;
  for (int o = 0; o < a.kout; ++o) {
    uint64_t best = kq[0][0];
    int bj = 0;
#pragma unroll
    for (int j = 1; j < LPL; ++j) {
      const bool take = kq[j][0] > best;
      best = take ? kq[j][0] : best;
      bj = take ? j : bj;
    }
    const uint64_t wm = wave_max_u64(best);
    mine = (lane == o) ? wm : mine;
    if (wm != 0ull && best == wm) {
#pragma unroll
      for (int j = 0; j < LPL; ++j) {
        if (j == bj) {
          kq[j][0] = kq[j][1];
          kq[j][1] = kq[j][2];
          kq[j][2] = kq[j][3];
          kq[j][3] = 0ull;
          used[j] += 1;
          if ((used[j] & 3) == 0) fetch4(wave + 4 * (lane + 64 * j), used[j], kq[j]);
        }
      }
    }
  }
  if (lane < 16) part[wave][lane] = lane < a.kout ? mine : 0ull;
  __syncthreads();
  if (wave != 0) return;
  const uint64_t key = part[lane >> 4][lane & 15];
  int rank = 0;
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, i);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), i);
    const uint64_t o = ((uint64_t)hi << 32) | lo;
    rank += o > key ? 1 : 0;
  }
  const int nnz = __builtin_popcountll(__ballot(key != 0ull));
  if (key != 0ull && rank < a.kout) {
    const float wv = merge_key_val(key);
    const int wi = merge_key_idx(key);
    a.out_val[qi * a.kout + rank] = wv;
    if (a.out_idx) a.out_idx[qi * a.kout + rank] = (int64_t)wi + a.idx_base;
    if (a.out_idx32) a.out_idx32[qi * a.kout + rank] = wi;
    if (rank == a.kout - 1) {
      if (a.kth_val) a.kth_val[qi] = wv;
      if (a.kth_idx) a.kth_idx[qi] = wi;
    }
  }
  if (lane >= nnz && lane < a.kout) {   // fewer than kout entries in all lists together
    a.out_val[qi * a.kout + lane] = kNegInf;
    if (a.out_idx) a.out_idx[qi * a.kout + lane] = -1;
    if (a.out_idx32) a.out_idx32[qi * a.kout + lane] = -1;
    if (lane == a.kout - 1) {
      if (a.kth_val) a.kth_val[qi] = kNegInf;
      if (a.kth_idx) a.kth_idx[qi] = -1;
    }
  }
}

// --------------------------------------------------------------------------
// select: top-k of an UNSORTED candidate buffer (what the candidate-append scans leave: [cap][nq], cnt[nq]) plus an
// optional sorted list per query (the prefix result).  One wave per query, every entry a 64-bit (score, ~row) key
// in a register (lane l owns candidates l, l + 64, ...; its last slot takes entry l of the sorted list).
//   1. the k-th largest SCORE by a bitwise search (32 steps of "how many scores are >= t": one compare-and-count
//      per slot and a wave sum) - no key ever moves;
//   2. ties at that score (rare) are cut by the same search on the row half of the key: exactly k keys selected;
//   3. the selected keys are compacted to one per lane through LDS (ballot / mbcnt positions) and ranked by
//      counting, lane i comparing with every other lane's key (64 readlane steps); lane i stores at its rank.
// (k rounds of wave maximum over all slots - the first version - cost 14.6 us at 15 slots and 72-78 us at 47.)
// cap = 64 (LPL - 1).
// --------------------------------------------------------------------------
struct SelectArgs {
  const float* cand_val;   // [nq][cap] (query-major: a query's candidates are contiguous)
  const int* cand_idx;
  const int* cand_cnt;
  int cap;
  const float* pre_val;  // optional [nq][kpre], sorted
  const int* pre_idx;
  int kpre;
  float* out_val;
  int64_t* out_idx;
  int64_t nq, idx_base;
  int k;
  const int* gate;       // run only if (*gate != 0) == (gate_want != 0)
  int gate_want;
};

// wave-uniform sum of an int over the 64 lanes: four DPP steps inside each row of 16 lanes, then four v_readlane.
// (A __shfl_xor butterfly is six dependent ds_bpermute round trips: the 32-step searches below spent 11 us in them.)
__device__ __forceinline__ int wave_sum_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);  // row_half_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);  // row_mirror: every lane holds its row's sum
  return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
         __builtin_amdgcn_readlane(v, 48);
}

// The k largest of the wave's keys kq[LPL] (0 = empty slot), one per lane (unsorted) plus its rank among them:
// steps 1-3 of the select comment above.  `compact` is the wave's 64-entry LDS scratch.  Returns the number of keys
// found (min(k, live keys)); lanes >= that hold key 0.
template <int LPL>
__device__ __forceinline__ int select_keys(const uint64_t (&kq)[LPL], int kwant, int lane, uint64_t* compact,
                                           uint64_t& mine, int& rank) {
  int live = 0;
#pragma unroll
  for (int j = 0; j < LPL; ++j) live += kq[j] != 0ull ? 1 : 0;
  live = wave_sum_i32(live);
  const int k = kwant < live ? kwant : live;
  // 1. T = k-th largest score half
  uint32_t T = 0u;
  if (k > 0) {
#pragma unroll 1
    for (int bit = 31; bit >= 0; --bit) {
      const uint32_t t = T | (1u << bit);
      int c = 0;
#pragma unroll
      for (int j = 0; j < LPL; ++j) c += (uint32_t)(kq[j] >> 32) >= t ? 1 : 0;
      if (wave_sum_i32(c) >= k) T = t;
    }
  }
  // 2. keys above T are in; `need` of the keys AT T are in, largest row half (= smallest row) first
  int cgt = 0, ceq = 0;
#pragma unroll
  for (int j = 0; j < LPL; ++j) {
    const uint32_t u = (uint32_t)(kq[j] >> 32);
    cgt += u > T ? 1 : 0;
    ceq += (u == T && kq[j] != 0ull) ? 1 : 0;
  }
  cgt = wave_sum_i32(cgt);
  ceq = wave_sum_i32(ceq);
  const int need = k - cgt;
  uint32_t Lo = 0u;  // smallest accepted row half among the ties
  if (need < ceq) {
#pragma unroll 1
    for (int bit = 31; bit >= 0; --bit) {
      const uint32_t t = Lo | (1u << bit);
      int c = 0;
#pragma unroll
      for (int j = 0; j < LPL; ++j) c += ((uint32_t)(kq[j] >> 32) == T && (uint32_t)kq[j] >= t && kq[j] != 0ull) ? 1 : 0;
      if (wave_sum_i32(c) >= need) Lo = t;
    }
  }
  // 3. compaction: position = keys selected in earlier slots + selected lanes below this one in this slot
  int base = 0;
#pragma unroll
  for (int j = 0; j < LPL; ++j) {
    const uint32_t u = (uint32_t)(kq[j] >> 32);
    const bool sel = k > 0 && kq[j] != 0ull && (u > T || (u == T && (uint32_t)kq[j] >= Lo));
    const uint64_t m = __ballot(sel);
    if (sel) compact[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = kq[j];
    base += __builtin_popcountll(m);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same wave wrote and reads: LDS is in order, this pins the compiler
  mine = lane < k ? compact[lane] : 0ull;
  rank = 0;
#pragma unroll 8
  for (int j = 0; j < 64; ++j) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mine, j);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mine >> 32), j);
    rank += ((((uint64_t)hi << 32) | lo) > mine) ? 1 : 0;
  }
  return k;
}

template <int LPL>
__global__ __launch_bounds__(256) void topk_select_kernel(SelectArgs a) {
  __shared__ uint64_t compact[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t qi = (int64_t)blockIdx.x * 4 + wave;
  if (qi >= a.nq) return;
  if (a.gate && ((*a.gate != 0) != (a.gate_want != 0))) return;
  int cnt = a.cand_cnt[qi];
  cnt = cnt < a.cap ? cnt : a.cap;
  uint64_t kq[LPL];
#pragma unroll
  for (int j = 0; j < LPL - 1; ++j) {
    const int e = lane + 64 * j;
    kq[j] = 0ull;
    if (e < cnt) kq[j] = merge_key(a.cand_val[qi * a.cap + e], a.cand_idx[qi * a.cap + e]);
  }
  kq[LPL - 1] = 0ull;
  if (a.pre_val && lane < a.kpre) kq[LPL - 1] = merge_key(a.pre_val[qi * a.kpre + lane], a.pre_idx[qi * a.kpre + lane]);
  uint64_t mine;
  int rank;
  const int k = select_keys<LPL>(kq, a.k, lane, compact[wave], mine, rank);
  if (lane < k) {
    a.out_val[qi * a.k + rank] = merge_key_val(mine);
    a.out_idx[qi * a.k + rank] = (int64_t)merge_key_idx(mine) + a.idx_base;
  } else if (lane < a.k) {   // fewer than k rows in all
    a.out_val[qi * a.k + lane] = kNegInf;
    a.out_idx[qi * a.k + lane] = -1;
  }
}

// select with FOUR waves per query (big candidate buffers: 16 < k <= 64): wave w selects the k best of candidates
// w, w + 4, ... (wave 0 also holds the sorted extra list), the four results meet in LDS and wave 0 selects and ranks
// the k best of those <= 4k keys.  LPLQ = slots per lane of a wave = ceil(cap / 256) + 1.
template <int LPLQ>
__global__ __launch_bounds__(256) void topk_select4_kernel(SelectArgs a) {
  __shared__ uint64_t compact[4][64];
  __shared__ uint64_t stage[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t qi = blockIdx.x;
  if (a.gate && ((*a.gate != 0) != (a.gate_want != 0))) return;
  int cnt = a.cand_cnt[qi];
  cnt = cnt < a.cap ? cnt : a.cap;
  uint64_t kq[LPLQ];
#pragma unroll
  for (int j = 0; j < LPLQ - 1; ++j) {
    const int e = 4 * (lane + 64 * j) + wave;
    kq[j] = 0ull;
    if (e < cnt) kq[j] = merge_key(a.cand_val[qi * a.cap + e], a.cand_idx[qi * a.cap + e]);
  }
  kq[LPLQ - 1] = 0ull;
  if (wave == 0 && a.pre_val && lane < a.kpre)
    kq[LPLQ - 1] = merge_key(a.pre_val[qi * a.kpre + lane], a.pre_idx[qi * a.kpre + lane]);
  uint64_t mine;
  int rank;
  select_keys<LPLQ>(kq, a.k, lane, compact[wave], mine, rank);
  stage[wave][lane] = mine;   // 0 beyond the wave's count
  __syncthreads();
  if (wave != 0) return;
  uint64_t k2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) k2[j] = stage[j][lane];
  const int k = select_keys<4>(k2, a.k, lane, compact[0], mine, rank);
  if (lane < k) {
    a.out_val[qi * a.k + rank] = merge_key_val(mine);
    a.out_idx[qi * a.k + rank] = (int64_t)merge_key_idx(mine) + a.idx_base;
  } else if (lane < a.k) {   // fewer than k rows in all
    a.out_val[qi * a.k + lane] = kNegInf;
    a.out_idx[qi * a.k + lane] = -1;
  }
}

// --------------------------------------------------------------------------
// merge of SORTED partial lists (the prefix lists: nlists x [nq][16]) by selection instead of k rounds of wave
// maximum: (a) the kout lists with the largest HEAD keys are found with select_keys on the heads (one slot per
// list); an entry of any other list is below its own head, hence below kout entries of those lists, and cannot be
// in the top kout: (b) only those kout x 16 <= 256 entries are loaded (4 slots per lane) and (c) selected and ranked.
// Exact under the (score desc, row asc) order (keys are unique).  ~8 us where the rounds took 17-19 us.
// Same groups / outputs as topk_merge32_kernel (out_idx32, kth_val, zero_cnt); kin == 16, kout <= 16.
// --------------------------------------------------------------------------
template <int HLPL>
__global__ __launch_bounds__(256) void topk_merge_sel_kernel(MergeArgs<int> a) {
  __shared__ uint64_t compact[4][64];
  __shared__ int sel_list[4][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t qi = (int64_t)blockIdx.x * 4 + wave;
  if (qi >= a.nq) return;
  const int ng_ = a.ngroups > 1 ? a.ngroups : 1, grp = (int)blockIdx.y;
  if (a.zero_cnt && lane == 0 && grp == 0) {
    a.zero_cnt[qi] = 0;
    if (qi == 0) a.zero_cnt[a.nq] = 0;
  }
  const int nl = (a.nlists - grp + ng_ - 1) / ng_;
  const int64_t gout = (int64_t)grp * a.nq;
  // (a) heads: slot j of lane l = list l + 64 j of this group, as TRUE keys (a tie between heads must resolve by
  //     row index exactly as in the final order); the kout-th largest head key is then the admission bar
  uint64_t hk[HLPL];
#pragma unroll
  for (int j = 0; j < HLPL; ++j) {
    const int list = lane + 64 * j;
    hk[j] = 0ull;
    if (list < nl) {
      const int64_t o = ((int64_t)(list * ng_ + grp) * a.nq + qi) * a.kin;
      hk[j] = merge_key(a.vals[o], a.idx[o]);
    }
  }
  uint64_t mine;
  int rank;
  const int nsel = select_keys<HLPL>(hk, a.kout, lane, compact[wave], mine, rank);
  if (lane < nsel && rank == nsel - 1) compact[wave][0] = mine;   // the bar (keys are unique: one lane writes)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const uint64_t bar = nsel > 0 ? compact[wave][0] : ~0ull;
  int nlist = 0;
#pragma unroll
  for (int j = 0; j < HLPL; ++j) {
    const bool sel = hk[j] != 0ull && hk[j] >= bar;
    const uint64_t m = __ballot(sel);
    if (sel) sel_list[wave][nlist + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = lane + 64 * j;
    nlist += __builtin_popcountll(m);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // (b) the entries of the selected lists: entry e = lane + 64 j -> list e / 16, position e % 16
  uint64_t kq[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = lane + 64 * j;
    kq[j] = 0ull;
    if ((e >> 4) < nlist) {
      const int list = sel_list[wave][e >> 4];
      const int64_t o = ((int64_t)(list * ng_ + grp) * a.nq + qi) * a.kin + (e & 15);
      kq[j] = merge_key(a.vals[o], a.idx[o]);
    }
  }
  // (c)
  const int k = select_keys<4>(kq, a.kout, lane, compact[wave], mine, rank);
  if (lane < a.kout) {
    const bool have = lane < k;
    const int slot = have ? rank : lane;
    const float wv = have ? merge_key_val(mine) : kNegInf;
    const int wi = have ? merge_key_idx(mine) : -1;
    a.out_val[(gout + qi) * a.kout + slot] = wv;
    if (a.out_idx) a.out_idx[(gout + qi) * a.kout + slot] = have ? (int64_t)wi + a.idx_base : -1;
    if (a.out_idx32) a.out_idx32[(gout + qi) * a.kout + slot] = wi;
    if (slot == a.kout - 1) {
      if (a.kth_val) a.kth_val[gout + qi] = wv;
      if (a.kth_idx) a.kth_idx[gout + qi] = wi;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void row_invnorm_kernel(const T* __restrict__ x, int64_t n, int d,
                                                          int64_t ldx, float eps,
                                                          float* __restrict__ out) {
  // one wave per row; lane l sums elements 4*(l + 64 j) + e in order, then an
  // xor butterfly 32,16,...,1 (mirrored by oracle/knn_oracle.c: hcir_oracle_row_invnorm).
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const T* p = x + row * ldx;
  float s = 0.f;
  for (int k = 4 * lane; k < d; k += 256) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v = (float)p[k + e];
      s = __builtin_fmaf(v, v, s);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) out[row] = 1.0f / fmaxf(sqrtf(s), eps);
}

__global__ __launch_bounds__(256) void l2_normalize_kernel(const float* __restrict__ x, int64_t n,
                                                           int d, float eps, float* __restrict__ y32,
                                                           _Float16* __restrict__ y16) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const float* p = x + row * (int64_t)d;
  float s = 0.f;
  for (int k = 4 * lane; k < d; k += 256) {
#pragma unroll
    for (int e = 0; e < 4; ++e) s = __builtin_fmaf(p[k + e], p[k + e], s);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  const float nrm = fmaxf(sqrtf(s), eps);
  for (int k = 4 * lane; k < d; k += 256) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v = p[k + e] / nrm;
      if (y32) y32[row * (int64_t)d + k + e] = v;
      if (y16) y16[row * (int64_t)d + k + e] = (_Float16)v;
    }
  }
}

template <typename T>
__global__ void convert_kernel(const float* __restrict__ x, int64_t n, T* __restrict__ y) {
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  for (; i + 3 < n; i += stride) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) y[i + e] = (T)v[e];
  }
  if (i < n && i + 3 >= n)
    for (int64_t j = i; j < n; ++j) y[j] = (T)x[j];
}

__global__ void pass_copy_kernel(const float* __restrict__ pv, const int* __restrict__ pi, int64_t nq,
                                 int kk, int k, int done, int64_t idx_base,
                                 float* __restrict__ out_val, int64_t* __restrict__ out_idx) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nq * kk) return;
  const int64_t qi = t / kk;
  const int j = (int)(t % kk);
  out_val[qi * k + done + j] = pv[t];
  out_idx[qi * k + done + j] = pi[t] < 0 ? -1 : (int64_t)pi[t] + idx_base;
}

// ----- host-side planning ---------------------------------------------------
struct Plan {
  int kp;          // list capacity per pass (16 or 64)
  int qb;          // queries per workgroup (32, 64, 128)
  int gm;          // gallery rows per workgroup tile
  int64_t prefix;  // rows of the prefix scan (== ng: single scan)
  int grid_main;   // workgroups (x) of the widest scan
  int npass;       // passes for k > 64
};

constexpr int kMaxGridX = 512;  // 2 workgroups per CU; also <= 576 lists per merge pass
// 33..64 queries with k <= 16 run as <one query tile x two wave columns> workgroups on 128-row tiles: 24 KB stages,
// THREE workgroups per CU for the streaming launches (768 workgroups; the floorless launches get the batch-insertion
// kernel of the one-query-tile geometry, two per CU) instead of two <two query tiles> workgroups with 40 KB stages.
// Same-box A/B, 1 M x 768 fp16, 64 queries (tools/ab_sim.py): 318 -> 307 us (the prefix launch 52 -> 38 us; the main
// scan unchanged at 5.7 TB/s).  -DHCIR_SCAN_Q64_TWO_TILES builds the earlier geometry.
#ifndef HCIR_SCAN_Q64_TWO_TILES
constexpr bool kQ64ThreePerCU = true;
constexpr int kMaxGridQ64 = 768;
#else
constexpr bool kQ64ThreePerCU = false;
constexpr int kMaxGridQ64 = kMaxGridX;
#endif
constexpr int kMaxParts = kMaxGridQ64 > kMaxGridX ? kMaxGridQ64 : kMaxGridX;

// Workgroups along the gallery for `qblocks` query blocks.  Every query block scans every gallery tile; the
// grid is sized so that ALL (tile run, query block) workgroups are resident at once (512 slots) and the blocks
// of one tile run sit on the same XCD (linear id = x + grid_x * y, grid_x a multiple of 8): they stream the
// same tiles in step and all but the first read them from that XCD's L2.  (With 512 workgroups per query
// block the blocks ran one after the other and the gallery came from HBM once per block.)
inline int scan_grid_x(int64_t tiles, int64_t qblocks, int max_wg = kMaxGridX) {
  int64_t cap = max_wg / (qblocks < 1 ? 1 : qblocks);
  cap = cap < 8 ? 8 : (cap & ~int64_t(7));
  return (int)(tiles < cap ? tiles : cap);
}

// Prefix launches of 16-entry lists with 33..kPrefixQb32MaxQ queries run as 32-query blocks (the batch-insertion
// kernel; the prefix is latency- and insertion-bound, not bandwidth-bound, and the blocks of a tile share its L2 lines).
Plan make_plan(int64_t nq, int64_t ng, int k) {
  Plan p;
  p.kp = k <= 16 ? 16 : (k <= 32 ? 32 : 64);
  p.npass = (k + 63) / 64;
  if (p.kp == 64) {
    p.qb = 32;
  } else {
    p.qb = nq <= 32 ? 32 : (nq <= 64 ? 64 : 128);
  }
  p.gm = p.qb == 128 ? 128 : 256;
  if (kQ64ThreePerCU && p.kp == 16 && p.qb == 64) p.gm = 128;   // <QT 1, WQ 2, WGG 2>: 128-row tiles, 3 workgroups per CU
  const int64_t tiles = hcir_cdiv(ng, p.gm);
  p.grid_main = (int)(tiles < kMaxGridX ? tiles : kMaxGridX);
  // prefix: ~1/16 of the gallery, at least 64 rows per list-k, in whole tiles
  p.prefix = ng;
  if (p.npass == 1 && ng >= 32768) {
    int64_t s = ng / 16;
    // 64-entry lists: a workgroup's fixed cost (filling the lists from its first tile, then the in-workgroup
    // merge: ~85 us) dwarfs its streaming time, so the prefix is ONE round of at most 64 workgroups x 1 tile;
    // 16 K rows still put the floor within ~k ln(N/16K) insertions per query of the final k-th score
    const int64_t lo = 8192, hi = p.kp == 64 ? 16384 : 131072;
    s = s < lo ? lo : (s > hi ? hi : s);
    s = hcir_cdiv(s, p.gm) * p.gm;
    if (s < ng) p.prefix = s;
  }
  return p;
}

// candidates per query the big-tile scan can hold: one-element lists of the final merge (16 per lane), next
// to the prefix list
constexpr int kCandLPL = 16;
constexpr int kCandCap = 64 * kCandLPL - 64;  // 960
// candidate-append scans of <= 128 queries with 16 < k <= 64: the floor is the minimum of ceil(k/16) group floors
// (weaker than one k-th score), so more rows clear it
constexpr int kSelLPLBig = 48;
constexpr int kCandCapBig = 64 * kSelLPLBig - 64;  // 3008
constexpr int kMaxFloorGroups = 4;

// the merge of up to `nlists` (+ the extra) sorted 32-bit-index lists: four waves per query for kout <= 16 with one
// group, the one-wave kernel otherwise
inline void launch_merge32(const MergeArgs<int>& m, hipStream_t st) {
  const int total = m.nlists + (m.extra_val ? 1 : 0);
  const int merge_grid = (int)hcir_cdiv(m.nq, 4);
  if (m.kout <= 16 && m.ngroups <= 1 && total <= 256 * 4) {
    const int lpl = (total + 255) / 256;
    if (lpl <= 1)
      hipLaunchKernelGGL(topk_merge32q_kernel<1>, dim3((unsigned)m.nq), dim3(256), 0, st, m);
    else if (lpl == 2)
      hipLaunchKernelGGL(topk_merge32q_kernel<2>, dim3((unsigned)m.nq), dim3(256), 0, st, m);
    else if (lpl == 3)
      hipLaunchKernelGGL(topk_merge32q_kernel<3>, dim3((unsigned)m.nq), dim3(256), 0, st, m);
    else
      hipLaunchKernelGGL(topk_merge32q_kernel<4>, dim3((unsigned)m.nq), dim3(256), 0, st, m);
    return;
  }
  if (total > 64 * kMergeLPL)   // (callers keep their list counts within 64 * kCandLPL)
    hipLaunchKernelGGL(topk_merge32_kernel<kCandLPL>, dim3(merge_grid), dim3(256), 0, st, m);
  else
    hipLaunchKernelGGL(topk_merge32_kernel<kMergeLPL>, dim3(merge_grid), dim3(256), 0, st, m);
}


struct Workspace {
  float* part_val;
  int* part_idx;
  float* pre_val;  // [nq][kp] prefix / previous-pass result
  int* pre_idx;
  float* floor_val;  // [nq]
  float* ceil_val;   // [nq]
  int* ceil_idx;     // [nq]
  float* cand_val;   // [kCandCap][nq]  big-tile scan: candidates behind the prefix floor
  int* cand_idx;     // [kCandCap][nq]
  int* cand_cnt;     // [nq] + 1 overflow flag behind it
  size_t bytes;
};

Workspace carve(void* base, int64_t nq, int k, const Plan& p) {
  Workspace w;
  char* c = static_cast<char*>(base);
  size_t off = 0;
  auto take = [&](size_t n) {
    char* r = c ? c + off : nullptr;
    off += (n + 255) & ~size_t(255);
    return r;
  };
  const size_t part = (size_t)kMaxParts * nq * p.kp;
  w.part_val = reinterpret_cast<float*>(take(part * 4));
  w.part_idx = reinterpret_cast<int*>(take(part * 4));
  w.pre_val = reinterpret_cast<float*>(take((size_t)nq * p.kp * 4));
  w.pre_idx = reinterpret_cast<int*>(take((size_t)nq * p.kp * 4));
  w.floor_val = reinterpret_cast<float*>(take((size_t)nq * 4 * kMaxFloorGroups));
  w.ceil_val = reinterpret_cast<float*>(take((size_t)nq * 4));
  w.ceil_idx = reinterpret_cast<int*>(take((size_t)nq * 4));
  const size_t ccap = (k > 16 && k <= 64 && nq <= 128) ? kCandCapBig : kCandCap;
  w.cand_val = reinterpret_cast<float*>(take(ccap * nq * 4));
  w.cand_idx = reinterpret_cast<int*>(take(ccap * nq * 4));
  w.cand_cnt = reinterpret_cast<int*>(take(((size_t)nq + 1) * 4));
  w.bytes = off;
  return w;
}

template <typename T, int KP, int QT, int WQ, int WGG>
void launch_scan_cfg(const ScanArgs& a, int grid_x, int grid_y, hipStream_t st) {
  if (QT == 1 && a.d % SimElem<T>::kPerStage == 0 && !a.floor_val)
    hipLaunchKernelGGL((sim_topk_scan<T, KP, QT, WQ, WGG, true, false, (QT == 1)>), dim3(grid_x, grid_y), dim3(256), 0,
                       st, a);
  else if (a.d % SimElem<T>::kPerStage == 0)
    hipLaunchKernelGGL((sim_topk_scan<T, KP, QT, WQ, WGG, true>), dim3(grid_x, grid_y), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((sim_topk_scan<T, KP, QT, WQ, WGG, false>), dim3(grid_x, grid_y), dim3(256), 0, st, a);
}

template <typename T>
void launch_scan(const Plan& p, const ScanArgs& a, int grid_x, hipStream_t st) {
  const int grid_y = (int)hcir_cdiv(a.nq, p.qb);
  if (p.kp == 64) {
    launch_scan_cfg<T, 64, 1, 1, 4>(a, grid_x, grid_y, st);
  } else if (p.kp == 32) {
    if (p.qb == 32)
      launch_scan_cfg<T, 32, 1, 1, 4>(a, grid_x, grid_y, st);
    else if (p.qb == 64)
      launch_scan_cfg<T, 32, 2, 1, 4>(a, grid_x, grid_y, st);
    else
      launch_scan_cfg<T, 32, 2, 2, 2>(a, grid_x, grid_y, st);
  } else if (p.qb == 32) {
    launch_scan_cfg<T, 16, 1, 1, 4>(a, grid_x, grid_y, st);
  } else if (p.qb == 64) {
    if (kQ64ThreePerCU)
      launch_scan_cfg<T, 16, 1, 2, 2>(a, grid_x, grid_y, st);
    else
      launch_scan_cfg<T, 16, 2, 1, 4>(a, grid_x, grid_y, st);
  } else {
    launch_scan_cfg<T, 16, 2, 2, 2>(a, grid_x, grid_y, st);
  }
}

// candidate-append scan (no lists): geometry by query count only
template <typename T, int QT, int WQ, int WGG>
void launch_cand_cfg(const ScanArgs& a, int grid_x, int grid_y, hipStream_t st) {
  if (a.d % SimElem<T>::kPerStage == 0)
    hipLaunchKernelGGL((sim_topk_scan<T, 16, QT, WQ, WGG, true, true>), dim3(grid_x, grid_y), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((sim_topk_scan<T, 16, QT, WQ, WGG, false, true>), dim3(grid_x, grid_y), dim3(256), 0, st, a);
}
template <typename T>
void launch_cand(int qb, const ScanArgs& a, int grid_x, hipStream_t st) {
  const int grid_y = (int)hcir_cdiv(a.nq, qb);
  if (qb == 32)
    launch_cand_cfg<T, 1, 1, 4>(a, grid_x, grid_y, st);
  else if (qb == 64)
    launch_cand_cfg<T, 1, 2, 2>(a, grid_x, grid_y, st);   // 128-row tiles, three workgroups per CU (C5 at 64 queries: 719 -> 706 us)
  else
    launch_cand_cfg<T, 2, 2, 2>(a, grid_x, grid_y, st);
}
void launch_cand_dtype(int dtype, int qb, const ScanArgs& a, int grid_x, hipStream_t st) {
  if (dtype == HCIR_F32)
    launch_cand<float>(qb, a, grid_x, st);
  else if (dtype == HCIR_F16)
    launch_cand<_Float16>(qb, a, grid_x, st);
  else
    launch_cand<__bf16>(qb, a, grid_x, st);
}

// -DHCIR_SCAN_NO_BIG (build flag) keeps the list-keeping scan for every query count (A/B libraries through
// HCIR_LIB_PATH); the library reads no environment variables
inline bool big_scan_enabled() {
#ifdef HCIR_SCAN_NO_BIG
  return false;
#else
  return true;
#endif
}

void launch_scan_dtype(int dtype, const Plan& p, const ScanArgs& a, int grid_x, hipStream_t st) {
  if (dtype == HCIR_F32)
    launch_scan<float>(p, a, grid_x, st);
  else if (dtype == HCIR_F16)
    launch_scan<_Float16>(p, a, grid_x, st);
  else
    launch_scan<__bf16>(p, a, grid_x, st);
}

}  // namespace

extern "C" {

#ifdef HCIR_DIAG_STAMPS
int hcir_debug_stamps(unsigned long long* host_dst) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 1024 * 8);
}
#endif

size_t hcir_sim_topk_workspace_bytes(int64_t nq, int64_t ng, int32_t d, int32_t k, int dtype) {
  (void)d;
  (void)dtype;
  if (nq <= 0 || ng <= 0 || k <= 0) return 0;
  const Plan p = make_plan(nq, ng, k);
  return carve(nullptr, nq, k, p).bytes;
}

int hcir_sim_topk(const void* q, int64_t nq, const void* g, int64_t ng, int32_t d, int32_t k,
                  int dtype, const float* q_inv_norm, const float* g_inv_norm, int64_t idx_base,
                  float* out_val, int64_t* out_idx, void* workspace, size_t workspace_bytes,
                  void* stream) {
  HCIR_ENTER();
  if (!q || !g || !out_val || !out_idx) return HCIR_ERR_INVALID;
  if (nq <= 0 || ng <= 0 || d <= 0 || (d & 7) || k <= 0 || k > HCIR_TOPK_MAX) return HCIR_ERR_INVALID;
  if (k > ng || ng >= (int64_t(1) << 31)) return HCIR_ERR_INVALID;
  if (dtype != HCIR_F32 && dtype != HCIR_F16 && dtype != HCIR_BF16) return HCIR_ERR_UNSUPPORTED;
  const Plan p = make_plan(nq, ng, k);
  Workspace w = carve(workspace, nq, k, p);
  if (!workspace || workspace_bytes < w.bytes) return HCIR_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int merge_grid = (int)hcir_cdiv(nq, 4);

  const int64_t qblocks = hcir_cdiv(nq, p.qb);
  ScanArgs a{};
  a.q = q;
  a.g = g;
  a.qn = q_inv_norm;
  a.gn = g_inv_norm;
  a.part_val = w.part_val;
  a.part_idx = w.part_idx;
  a.nq = nq;
  a.d = d;
  a.shared_stream = qblocks > 1 ? 1 : 0;

  // ---- <= 128 queries, k <= 64, big gallery: candidate-append flow (DESIGN.md "sim_topk: candidate flow").
  //   A  rows [0, S): the 16-entry list kernel (its fixed cost is 53 us; the 64-entry one costs 131 us), workgroup
  //      x in group x % G, G = 1 (k <= 16) or ceil(k / 16); one merge launch gives every group's floor - the
  //      k-th (G = 1) or 16th best score of the group - and, for G = 1, the prefix top-k list;
  //   B  a scan WITHOUT lists appends every row at / above the floor (min over the groups: >= k rows sit at or
  //      above it) to the query's candidate buffer: G = 1 over rows [S, N) with `> floor` (a tie loses to the k
  //      prefix rows with smaller indices), G > 1 over ALL rows with `>= floor` (the group lists do not hold the
  //      prefix's top-k);
  //   C  select: top-k of the candidates (+ the prefix list), one wave per query.
  // A gallery whose later rows systematically beat the prefix overflows a buffer: the device flag then gates the
  // list-keeping flow below (all of it for G > 1, its second phase for G = 1), whose launches are empty otherwise.
  const int* fallback_gate = nullptr;
  bool prefix_done = false;
  int64_t cand_prefix = 0;
  // Same-box A/B against the list-keeping flow (tools/ab_sim.py, 1 M x 768 fp16, k = 16): 1 query 313 -> 302 us,
  // 32 queries 327 -> 318 us, but 64 queries 333 -> 352 us and 128 queries 443 -> 453 us (two query tiles per wave:
  // the list kernel's main scan is as fast there, and the flow adds two gated launches): k <= 16 takes the
  // candidate flow up to 32 queries only.  k > 16 (1.25 M x 1024, top-50): 1001 -> 639 us at 32 queries,
  // 1662 -> 742 us at 64: always.
  if (p.npass == 1 && p.prefix < ng && nq <= 128 && (k > 16 || nq <= 32)) {
    const int G = k <= 16 ? 1 : (k + 15) / 16;
    int cap = k <= 16 ? kCandCap : kCandCapBig;
    Plan pc = p;
    pc.kp = 16;
    pc.qb = nq <= 32 ? 32 : (nq <= 64 ? 64 : 128);
    pc.gm = pc.qb == 128 ? 128 : 256;
    int64_t S = p.prefix;
    if (G == 1) {
      // Without lists behind the floor a longer prefix costs little (one round of <= 512 one-tile workgroups
      // takes the same ~55 us as half a round) and pays twice: fewer rows for the main scan and r = 7 instead of
      // 15, i.e. <= 38 * 7 candidates per query, a 7-slot select instead of a 15-slot one.
      // (measured, 1 M x 768 fp16: 32 queries 0.314 -> 0.306 ms; at 64 queries the 2-tile-deep list prefix costs
      // what the shorter main scan and select save, 0.330 -> 0.335 ms: N/16 kept there)
      S = pc.qb == 32 ? ng / 8 : ng / 16;
      S = S < 8192 ? 8192 : (S > 131072 ? 131072 : S);
      S = hcir_cdiv(S, pc.gm) * pc.gm;
      if (S < ng && 38 * ((ng - S) / S + 1) <= 64 * 7) cap = 64 * 7;  // small select (one buffer size for scan AND select)
    }
    if (G > 1) {
      // Rows at or above the floor, r + 1 = N / S.  The tail probability at a group's 16th best of n rows is
      // ~ Gamma(16) / n (mean 16, sd 4), the floor is the weakest of the G groups: the count is ~ X G (r + 1) with
      // X the largest of G Gamma(16) draws.  P(X > 38) ~ 1e-5 per draw: sized for 38 G (r + 1) <= cap (an
      // earlier 16 G + 5 sigma sizing overflowed a few percent of the QUERIES, and one overflow reruns the call).
      const float per = 38.0f * G;
      int64_t rp1 = (int64_t)((float)cap / per);
      rp1 = rp1 < 2 ? 2 : rp1;
      S = hcir_cdiv(ng, rp1);
      S = S < 8192 ? 8192 : S;
      S = hcir_cdiv(S, pc.gm) * pc.gm;
    }
    if (S < ng && hcir_cdiv(S, pc.gm) >= 4 * G) {
      const int64_t qbl = hcir_cdiv(nq, pc.qb);
      ScanArgs c = a;
      c.shared_stream = qbl > 1 ? 1 : 0;
      c.k = G == 1 ? k : 16;
      c.row_begin = 0;
      c.row_end = S;
      // (the 16-entry list kernel of 33..64 queries works on 128-row tiles)
      const int gm_a = (kQ64ThreePerCU && pc.qb == 64) ? 128 : pc.gm;
      const int grid_a = scan_grid_x(hcir_cdiv(S, gm_a), qbl);
      launch_scan_dtype(dtype, pc, c, grid_a, st);
      HCIR_LAUNCH_CHECK();
      MergeArgs<int> m{};
      m.vals = w.part_val;
      m.idx = w.part_idx;
      m.nq = nq;
      m.nlists = grid_a;
      m.kin = 16;
      m.kout = c.k;
      m.out_val = w.pre_val;
      m.out_idx32 = w.pre_idx;
      m.kth_val = w.floor_val;
      m.zero_cnt = w.cand_cnt;
      m.ngroups = G;
      if (G == 1)
        launch_merge32(m, st);   // four waves per query (the selection merge below: 21.6 us at 32 queries)
      else
        hipLaunchKernelGGL(topk_merge_sel_kernel<kMergeLPL>, dim3(merge_grid, G), dim3(256), 0, st, m);
      HCIR_LAUNCH_CHECK();
      int* overflow = w.cand_cnt + nq;
      c.row_begin = G == 1 ? S : 0;
      c.row_end = ng;
      c.floor_val = w.floor_val;
      c.floor_groups = G;
      c.floor_inclusive = G > 1 ? 1 : 0;
      c.cand_val = w.cand_val;
      c.cand_idx = w.cand_idx;
      c.cand_cnt = w.cand_cnt;
      c.overflow = overflow;
      c.cap = cap;
      c.cand_query_major = 1;
      const bool cg3 = pc.qb == 64;   // launch_cand's 64-query geometry: 128-row tiles, up to 768 workgroups
      const int grid_b = scan_grid_x(hcir_cdiv(c.row_end - c.row_begin, cg3 ? 128 : pc.gm), qbl, cg3 ? 768 : kMaxGridX);
      launch_cand_dtype(dtype, pc.qb, c, grid_b, st);
      HCIR_LAUNCH_CHECK();
      SelectArgs sa{};
      sa.cand_val = w.cand_val;
      sa.cand_idx = w.cand_idx;
      sa.cand_cnt = w.cand_cnt;
      sa.cap = cap;
      if (G == 1) {
        sa.pre_val = w.pre_val;
        sa.pre_idx = w.pre_idx;
        sa.kpre = k;
      }
      sa.out_val = out_val;
      sa.out_idx = out_idx;
      sa.nq = nq;
      sa.idx_base = idx_base;
      sa.k = k;
      sa.gate = overflow;
      sa.gate_want = 0;
      // expected candidates <= 38 (N - S) / S per query (k <= 16): a small select when the prefix is long
      if (cap == 64 * 7)
        hipLaunchKernelGGL(topk_select_kernel<8>, dim3(merge_grid), dim3(256), 0, st, sa);
      else if (k <= 16)
        hipLaunchKernelGGL(topk_select_kernel<kCandLPL>, dim3(merge_grid), dim3(256), 0, st, sa);
      else   // 3008 candidates: four waves per query, 12 + 1 slots per lane
        hipLaunchKernelGGL(topk_select4_kernel<(kCandCapBig + 255) / 256 + 1>, dim3((unsigned)nq), dim3(256), 0, st, sa);
      HCIR_LAUNCH_CHECK();
      fallback_gate = overflow;
      prefix_done = G == 1;   // the list flow's phase A is exactly what ran above (same kernel, rows [0, S))
      if (G == 1) cand_prefix = S;
    }
  }

  Plan pp = p;  // (the prefix may shrink below)
  if (cand_prefix) pp.prefix = cand_prefix;
  // Group-floor candidate flow (k > 16): its fallback (a candidate buffer overflowed, ~1e-5 per call) is the ONE-phase
  // list scan over all rows + one merge - two gated launches that exit at once in the common case, where the
  // two-phase flow cost four (~4.5 us each even when empty: 3 % of a 64-query C5 call).
  if (fallback_gate && !prefix_done) pp.prefix = ng;
  // Many queries (MFMA-bound): the 256 x 256 tile scan collects the rare rows above the prefix floor.  The
  // prefix itself runs on the list-keeping kernel at half that rate, so it is only as long as the candidate
  // buffers require.  With r = rows behind the prefix / prefix rows, the number of rows that beat the
  // prefix's k-th score is negative-binomial: mean k r, variance k r (1 + r) ~ (r sqrt k)^2; r is chosen so that
  // mean + 5 sigma fits the buffer (overflow ~1e-6 per query for a gallery in random order; it is handled).
  const bool big = big_scan_enabled() && p.npass == 1 && p.prefix < ng && dtype != HCIR_F32 && nq > 128 && k <= 16 &&
                   d % 64 == 0 && !q_inv_norm && !g_inv_norm && ng - p.prefix >= 4096;
  if (big) {
    // (measured flat in r = 8..26 at 220 queries: a shorter prefix is paid back by a longer candidate merge;
    // r is capped at 15 - the 1/16 prefix of the list-keeping path - which leaves 11 sigma of headroom)
    int64_t rr = (int64_t)((float)kCandCap / ((float)k + 5.0f * sqrtf((float)k)));
    rr = rr > 15 ? 15 : rr;
    int64_t s = hcir_cdiv(ng, (rr < 1 ? 1 : rr) + 1);
    s = s < 8192 ? 8192 : s;
    s = hcir_cdiv(s, p.gm) * p.gm;
    if (s < pp.prefix) pp.prefix = s;
  }
  if (p.npass == 1) {
    a.k = k;
    const bool two_phase = pp.prefix < ng;
    // phase A: rows [0, prefix)
    a.row_begin = 0;
    a.row_end = pp.prefix;
    const int64_t tiles_a = hcir_cdiv(pp.prefix, p.gm);
    const int grid_a = scan_grid_x(tiles_a, qblocks);
    a.gate = fallback_gate;
    if (!prefix_done) {
      launch_scan_dtype(dtype, p, a, grid_a, st);
      HCIR_LAUNCH_CHECK();
    }
    MergeArgs<int> m{};
    m.gate = fallback_gate;
    m.gate_want = 1;
    m.vals = w.part_val;
    m.idx = w.part_idx;
    m.nq = nq;
    m.nlists = grid_a;
    m.kin = p.kp;
    m.kout = k;
    if (!two_phase) {
      m.out_val = out_val;
      m.out_idx = out_idx;
      m.idx_base = idx_base;
      launch_merge32(m, st);
      HCIR_LAUNCH_CHECK();
      return HCIR_OK;
    }
    m.out_val = w.pre_val;
    m.out_idx32 = w.pre_idx;
    m.kth_val = w.floor_val;
    if (big) m.zero_cnt = w.cand_cnt;  // the big scan's counters and overflow flag start at zero
    if (!prefix_done) {
      launch_merge32(m, st);
      HCIR_LAUNCH_CHECK();
    }
    // phase B: rows [prefix, ng) with the prefix k-th score as floor
    a.row_begin = pp.prefix;
    a.row_end = ng;
    a.floor_val = w.floor_val;
    // Many queries (MFMA-bound): the 256 x 256 tile scan collects the rare rows above the floor; the
    // list-keeping scan below then only runs (device-side gate) if a candidate buffer overflowed.
    int* overflow = w.cand_cnt + nq;
    if (big) {
      BigScanArgs b{};
      b.q = q;
      b.g = g;
      b.floor_val = w.floor_val;
      b.cand_val = w.cand_val;
      b.cand_idx = w.cand_idx;
      b.cand_cnt = w.cand_cnt;
      b.overflow = overflow;
      b.nq = nq;
      b.row_begin = pp.prefix;
      b.row_end = ng;
      b.d = d;
      b.cap = kCandCap;
      const int64_t qb256 = hcir_cdiv(nq, 256);
      const int64_t tiles256 = hcir_cdiv(ng - pp.prefix, 256);
      int64_t gx = 256 / qb256;  // one 128 KB workgroup per CU, every query block of a tile run resident
      gx = gx < 8 ? 8 : (gx & ~int64_t(7));
      gx = tiles256 < gx ? tiles256 : gx;
      if (dtype == HCIR_F16)
        hipLaunchKernelGGL(sim_scan_big_kernel<_Float16>, dim3((unsigned)gx, (unsigned)qb256), dim3(512), 0, st, b);
      else
        hipLaunchKernelGGL(sim_scan_big_kernel<__bf16>, dim3((unsigned)gx, (unsigned)qb256), dim3(512), 0, st, b);
      HCIR_LAUNCH_CHECK();
      MergeArgs<int> mc{};
      mc.vals = w.cand_val;
      mc.idx = w.cand_idx;
      mc.nq = nq;
      mc.nlists = kCandCap;
      mc.nlists_q = w.cand_cnt;
      mc.kin = 1;
      mc.kout = k;
      mc.extra_val = w.pre_val;
      mc.extra_idx = w.pre_idx;
      mc.kin_extra = k;
      mc.out_val = out_val;
      mc.out_idx = out_idx;
      mc.idx_base = idx_base;
      mc.gate = overflow;
      mc.gate_want = 0;
      launch_merge32(mc, st);
      HCIR_LAUNCH_CHECK();
      a.gate = overflow;  // the launches below: fallback only
    }
    const int64_t tiles_b = hcir_cdiv(ng - pp.prefix, p.gm);
    const bool q64g3 = kQ64ThreePerCU && p.kp == 16 && p.qb == 64;
    const int grid_b = scan_grid_x(tiles_b, qblocks, q64g3 ? kMaxGridQ64 : kMaxGridX);
    launch_scan_dtype(dtype, p, a, grid_b, st);
    HCIR_LAUNCH_CHECK();
    MergeArgs<int> m2{};
    m2.vals = w.part_val;
    m2.idx = w.part_idx;
    m2.nq = nq;
    m2.nlists = grid_b;
    m2.kin = p.kp;
    m2.kout = k;
    m2.extra_val = w.pre_val;
    m2.extra_idx = w.pre_idx;
    m2.kin_extra = k;
    m2.out_val = out_val;
    m2.out_idx = out_idx;
    m2.idx_base = idx_base;
    if (big) {
      m2.gate = overflow;
      m2.gate_want = 1;
    } else if (fallback_gate) {
      m2.gate = fallback_gate;
      m2.gate_want = 1;
    }
    launch_merge32(m2, st);
    HCIR_LAUNCH_CHECK();
    return HCIR_OK;
  }

  // k > 64: successive passes of <= 64 ranks below a moving ceiling.
  a.row_begin = 0;
  a.row_end = ng;
  const int grid_x = scan_grid_x(hcir_cdiv(ng, p.gm), qblocks);
  int done = 0;
  for (int pass = 0; pass < p.npass; ++pass) {
    const int kk = (k - done) < 64 ? (k - done) : 64;
    a.k = kk;
    a.ceil_val = pass ? w.ceil_val : nullptr;
    a.ceil_idx = pass ? w.ceil_idx : nullptr;
    launch_scan_dtype(dtype, p, a, grid_x, st);
    HCIR_LAUNCH_CHECK();
    MergeArgs<int> m{};
    m.vals = w.part_val;
    m.idx = w.part_idx;
    m.nq = nq;
    m.nlists = grid_x;
    m.kin = p.kp;
    m.kout = kk;
    m.out_val = w.pre_val;
    m.out_idx32 = w.pre_idx;
    m.kth_val = w.ceil_val;
    m.kth_idx = w.ceil_idx;
    hipLaunchKernelGGL(topk_merge32_kernel<kMergeLPL>, dim3(merge_grid), dim3(256), 0, st, m);
    HCIR_LAUNCH_CHECK();
    // scatter this pass's [nq][kk] block into out[:, done:done+kk]
    hipLaunchKernelGGL(pass_copy_kernel, dim3((unsigned)hcir_cdiv(nq * kk, 256)), dim3(256), 0, st,
                       w.pre_val, w.pre_idx, nq, kk, k, done, idx_base, out_val, out_idx);
    HCIR_LAUNCH_CHECK();
    done += kk;
  }
  return HCIR_OK;
}

int hcir_topk_merge(const float* vals, const int64_t* idx, int32_t nlists, int64_t nq, int32_t k_in,
                    int32_t k_out, float* out_val, int64_t* out_idx, void* stream) {
  HCIR_ENTER();
  if (!vals || !idx || !out_val || !out_idx) return HCIR_ERR_INVALID;
  if (nlists <= 0 || nq <= 0 || k_in <= 0 || k_out <= 0) return HCIR_ERR_INVALID;
  if (nlists > 64 * kMergeLPL) return HCIR_ERR_UNSUPPORTED;
  MergeArgs<int64_t> m{};
  m.vals = vals;
  m.idx = idx;
  m.nq = nq;
  m.nlists = nlists;
  m.kin = k_in;
  m.kout = k_out;
  m.out_val = out_val;
  m.out_idx = out_idx;
  m.idx_base = 0;
  hipLaunchKernelGGL(topk_merge_kernel<int64_t>, dim3((int)hcir_cdiv(nq, 4)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), m);
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}

int hcir_row_invnorm(const void* x, int64_t n, int32_t d, int64_t ldx, int dtype, float eps,
                     float* out, void* stream) {
  HCIR_ENTER();
  if (!x || !out || n <= 0 || d <= 0 || (d & 3) || ldx < d) return HCIR_ERR_INVALID;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)hcir_cdiv(n, 4)), block(256);
  if (dtype == HCIR_F32)
    hipLaunchKernelGGL(row_invnorm_kernel<float>, grid, block, 0, st, (const float*)x, n, d, ldx, eps, out);
  else if (dtype == HCIR_F16)
    hipLaunchKernelGGL(row_invnorm_kernel<_Float16>, grid, block, 0, st, (const _Float16*)x, n, d, ldx, eps, out);
  else if (dtype == HCIR_BF16)
    hipLaunchKernelGGL(row_invnorm_kernel<__bf16>, grid, block, 0, st, (const __bf16*)x, n, d, ldx, eps, out);
  else
    return HCIR_ERR_UNSUPPORTED;
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}

int hcir_l2_normalize(const float* x, int64_t n, int32_t d, float eps, float* y_f32, void* y_f16,
                      void* stream) {
  HCIR_ENTER();
  if (!x || n <= 0 || d <= 0 || (d & 3)) return HCIR_ERR_INVALID;
  hipLaunchKernelGGL(l2_normalize_kernel, dim3((unsigned)hcir_cdiv(n, 4)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, n, d, eps, y_f32, (_Float16*)y_f16);
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}

int hcir_convert_f32(const float* x, int64_t n, int dtype, void* y, void* stream) {
  HCIR_ENTER();
  if (!x || !y || n <= 0) return HCIR_ERR_INVALID;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int64_t blocks = hcir_cdiv(n, 1024);
  if (blocks > 4096) blocks = 4096;
  if (dtype == HCIR_F16)
    hipLaunchKernelGGL(convert_kernel<_Float16>, dim3((unsigned)blocks), dim3(256), 0, st, x, n, (_Float16*)y);
  else if (dtype == HCIR_BF16)
    hipLaunchKernelGGL(convert_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, st, x, n, (__bf16*)y);
  else
    return HCIR_ERR_UNSUPPORTED;
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}

}  // extern "C"
