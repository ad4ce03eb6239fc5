// png.hip — PNG decode of CenterCrop windows on the device (include/hcir.h "PNG decode on the device").
//
// Stands where the reference decodes on the host: HP/utils/dataloader.py:28-31 (read_file + decode_image(RGB)) and
// src/models/hair_encoder.py:108,169 (PIL), for the format every hair-region crop it lists is stored in
// (HairPretraining/data/data_train.csv: *_hair.png; assets/hair_region_only/*.png).
//
// Two kernels; per image the first runs three 64-lane wavefronts (lookup, decoder, copier), the second one (a batch is
// hundreds of images: a few waves per SIMD):
//
//  png_inflate_kernel   zlib/deflate (RFC 1950/1951) up to the last scanline the window needs.
//    * The compressed words sit in two VGPRs (lane i = word base+i, base+64+i), fetched 256 B at a time; the
//      decoder takes its 64-bit look at bit position bp with v_readlane: no memory latency on the serial chain
//      except the code-table lookup itself.
//    * Code tables are built per block IN LDS by the wave: symbols are ranked inside their length class with
//      ballots (no serial pass over the 286 + 30 lengths), then every table entry finds its code by the canonical
//      first-code / count test.  10-bit literal/length root, 9-bit distance root; the rare longer codes take the
//      canonical range search.
//    * Symbols are decoded in batches of up to 64: symbol k is parked in lane k (v_writelane).  The batch is
//      then resolved by the whole wave: a wave prefix sum of the output lengths gives every symbol its position,
//      literals are written to the ring by their lanes at once, each match is one lane-parallel copy.
//    * The 32 KB history is a ring in LDS.  Finished 1 KB units are written to the image's scanline buffer in
//      HBM as 16 B per lane.
//  png_unfilter_kernel  filters None/Sub/Up/Average/Paeth of rows 0..last, lane = row, step t handles pixel t - lane
//    of every row of a 64-row band: left is the lane's previous output, up and upper-left are the previous lane's
//    last two outputs (DPP wave shift), so the whole recurrence runs in registers.  Writes the window as RGB8.
#include <atomic>
#include <thread>
#include <vector>

#include "common.h"
#include "png_stage.h"

namespace {

constexpr int kRing = 32768;  // deflate's maximum distance: the ring never needs to be larger (the reads of a copy
                              // come before its writes; literals that could alias a far source are written in order)
constexpr int kLitRoot = 10, kDistRoot = 8, kClRoot = 7;
constexpr uint32_t kQueue = 128;  // symbol queue entries (a ring) between the decoding and the copying wavefront
constexpr uint32_t kBatchMin = 32;  // the copier takes a batch as soon as this many symbols wait (at most 64 at a time):
                                   // the decoder can therefore always count on kQueue - kBatchMin free slots
constexpr uint32_t K_INVALID = 0, K_LIT = 1, K_LEN = 2, K_EOB = 3, K_DIST = 4, K_LONG = 5, K_CL = 6;
enum { T_CL = 0, T_LIT = 1, T_DIST = 2 };
// A decoded symbol, packed: bits 0-6 the walk's step = stream bits it takes (code + extra bits; a match: length AND
// distance codes), or 64 for an entry the walk must stop on; bits 7-8 kind; literal: bits 9-16 the byte; match:
// bits 9-16 length - 3, bits 17-31 distance - 1.
constexpr uint32_t S_LIT = 0, S_MATCH = 1, S_EOB = 2, S_SLOW = 3;  // S_SLOW: left to the serial decoder
constexpr uint32_t kStop = 64;
__device__ __forceinline__ uint32_t pack_lit(uint32_t step, uint32_t byte) { return step | (S_LIT << 7) | (byte << 9); }
// Two literals behind one another as ONE entry (bit 17: a second byte follows in bits 18-25; step = both codes): the
// walk is the serial part of the decoder, and 65-81 % of a hair crop's symbols are literals of 4.6-5.1 bits - a pair
// per step saves 22-31 % of the steps (root-10 lookups: both codes <= 10 bits).  -DHCIR_PNG_NO_PAIRS: A/B switch.
#ifdef HCIR_PNG_NO_PAIRS
constexpr bool kPairLits = false;
#else
constexpr bool kPairLits = true;
#endif
__device__ __forceinline__ uint32_t pack_lit2(uint32_t step, uint32_t b1, uint32_t b2) {
  return step | (S_LIT << 7) | (b1 << 9) | (1u << 17) | (b2 << 18);
}
__device__ __forceinline__ uint32_t pack_match(uint32_t step, uint32_t len, uint32_t dist) {
  return step | (S_MATCH << 7) | ((len - 3) << 9) | ((dist - 1) << 17);
}
constexpr int kSpan = 4;                                           // 64-bit windows looked up per pass (256 positions)

struct PngBatch {
  const uint8_t* blob;
  int64_t b;
  int32_t win_h, win_w;
  uint8_t* out;
  int32_t* status;
  uint8_t* raw;
  uint64_t raw_stride;
  uint64_t* diag;  // -DHCIR_PNG_STAMPS: 16 counters per image (tools/diag_png.py)
};

#ifdef HCIR_PNG_STAMPS
#define STAMP_DECL uint64_t st_t0 = __builtin_readcyclecounter(), st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_n[4] = {0, 0, 0, 0}
#define STAMP(i)                                         \
  do {                                                   \
    const uint64_t st_now = __builtin_readcyclecounter(); \
    st_acc[i] += st_now - st_t0;                         \
    st_t0 = st_now;                                      \
  } while (0)
#define COUNT(i, v) st_n[i] += (v)
#else
#define STAMP_DECL
#define STAMP(i)
#define COUNT(i, v)
#endif

// One wave per workgroup: LDS operations of a wave complete in order, so lanes see each other's LDS writes without
// a barrier instruction; what is needed is that the compiler keeps the order (no s_waitcnt vmcnt(0) as __syncthreads
// would put in front of its s_barrier: the ring's write-out stores to HBM need not have landed).
#define WAVE_SYNC()                                           \
  do {                                                        \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    \
    __builtin_amdgcn_wave_barrier();                          \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    \
  } while (0)

struct Canon {  // canonical code ranges by length
  uint32_t first[16], count[16], offs[16];
};

struct Smem {
  uint8_t ring[kRing];
  uint32_t lit_tab[1 << kLitRoot];
  uint32_t dist_tab[1 << kDistRoot];  // also the code-length code's table while a block header is read
  Canon lit_cn, dist_cn;
  uint16_t lit_sorted[288], dist_sorted[32];
  uint8_t lens[320 + 12];
  uint8_t cl_lens[20];
  uint32_t symq[kQueue + 1];      // decoded symbols on their way to the copying wavefront (a ring) + a dump slot
  uint32_t ent_buf[kSpan][64];    // the looked-up symbols of one pass, from the lookup wavefront to the decoding one
  uint32_t lk_wbase, lk_req, lk_quit;  // written by the decoding wavefront: bit position and number of the pass wanted
  uint32_t lk_ready;              // written by the lookup wavefront: number of the pass that is in ent_buf
  uint32_t q_tail, q_eos;         // written by the decoding wavefront: entries published; 1 = no more will come
  uint32_t q_head, q_stop;        // written by the copying wavefront: entries consumed; 1 = stop decoding
  int32_t errs[2];
  uint8_t dump[64];               // where the lanes beyond a copy's length write
  uint32_t u_bp, u_last, u_type;  // wave-uniform state handed between block_header() and the kernel
  int32_t u_err;
};
typedef __attribute__((address_space(3))) Smem Smem3;

__device__ __forceinline__ uint32_t U(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ uint32_t entry(uint32_t len, uint32_t kind, uint32_t extra, uint32_t val) {
  return len | (kind << 4) | (extra << 8) | (val << 16);
}

// the table entry of symbol `sym` with code length `len` (RFC 1951 3.2.5 base values, in closed form)
template <int TYPE>
__device__ __forceinline__ uint32_t symbol_entry(uint32_t sym, uint32_t len) {
  if (TYPE == T_CL) return entry(len, K_CL, 0, sym);
  if (TYPE == T_LIT) {
    if (sym < 256) return entry(len, K_LIT, 0, sym);
    if (sym == 256) return entry(len, K_EOB, 0, 0);
    const uint32_t s = sym - 257;
    if (s > 28) return entry(len, K_INVALID, 0, 0);
    if (s < 8) return entry(len, K_LEN, 0, 3 + s);
    if (s == 28) return entry(len, K_LEN, 0, 258);
    const uint32_t eb = (s - 4) >> 2;
    return entry(len, K_LEN, eb, 3 + ((4 + (s & 3)) << eb));
  }
  if (sym > 29) return entry(len, K_INVALID, 0, 0);
  if (sym < 4) return entry(len, K_DIST, 0, 1 + sym);
  const uint32_t eb = (sym - 2) >> 1;
  return entry(len, K_DIST, eb, 1 + ((2 + (sym & 1)) << eb));
}

// Build one decode table from code lengths lens[0..n) (LDS).  Returns 0, or 1 when the set of lengths is not
// acceptable to zlib's inflate_table: over-subscribed, or incomplete with anything but a single 1-bit code
// (a code-length code must be complete).  ROOT-bit table `tab`, symbols sorted by (length, value) in `sorted`.
template <int ROOT, int TYPE, int MAXN>
__device__ __forceinline__ int build_table(const __attribute__((address_space(3))) uint8_t* lens, int n,
                                           __attribute__((address_space(3))) uint32_t* tab,
                                           __attribute__((address_space(3))) uint16_t* sorted,
                                           __attribute__((address_space(3))) Canon* cn, int lane) {
  constexpr int kChunks = (MAXN + 63) / 64;
  uint32_t run[16];
#pragma unroll
  for (int l = 0; l < 16; ++l) run[l] = 0;
  uint32_t myrank[kChunks], mylen[kChunks];
#pragma unroll
  for (int c = 0; c < kChunks; ++c) {  // rank every symbol inside its length class: ballots, no serial pass
    const int s = c * 64 + lane;
    const uint32_t l = s < n ? lens[s] : 0;
    mylen[c] = l;
    uint32_t r = 0;
#pragma unroll
    for (int L = 1; L < 16; ++L) {
      const uint64_t m = __ballot(l == (uint32_t)L);
      if (l == (uint32_t)L) r = run[L] + (uint32_t)__popcll(m & ((1ull << lane) - 1));
      run[L] += (uint32_t)__popcll(m);
    }
    myrank[c] = r;
  }
  // first code / offset of every length (uniform), validity as zlib's inflate_table
  uint32_t first[16], offs[16];
  int left = 1, maxlen = 0;
  uint32_t code = 0, o = 0;
  first[0] = offs[0] = 0;
#pragma unroll
  for (int L = 1; L < 16; ++L) {
    code = (code + run[L - 1]) << 1;  // run[0] is never counted (stays 0)
    first[L] = code;
    offs[L] = o;
    o += run[L];
    left = (left << 1) - (int)run[L];
    if (run[L]) maxlen = L;
    if (left < 0) return 1;
  }
  if (left > 0 && (TYPE == T_CL || maxlen > 1)) return 1;
  if (lane < 16) {
    uint32_t f = 0, c = 0, of = 0;
#pragma unroll
    for (int L = 1; L < 16; ++L)
      if (lane == L) f = first[L], c = run[L], of = offs[L];
    cn->first[lane] = f;
    cn->count[lane] = c;
    cn->offs[lane] = of;
  }
  WAVE_SYNC();
#pragma unroll
  for (int c = 0; c < kChunks; ++c) {
    const int s = c * 64 + lane;
    if (mylen[c]) sorted[cn->offs[mylen[c]] + myrank[c]] = (uint16_t)s;
  }
  WAVE_SYNC();
  // every ROOT-bit index finds the code it starts with: the first L bits (stream order = most significant code
  // bit first) form the L-bit number c; c is a code of length L iff first[L] <= c < first[L] + count[L]
  for (int e = lane; e < (1 << ROOT); e += 64) {
    uint32_t v = entry(0, maxlen > ROOT ? K_LONG : K_INVALID, 0, 0);
    const uint32_t rev = __brev((uint32_t)e);
#pragma unroll
    for (int L = (ROOT < 15 ? ROOT : 15); L >= 1; --L) {
      const uint32_t c = rev >> (32 - L), idx = c - first[L];
      if (c >= first[L] && idx < run[L]) v = symbol_entry<TYPE>(sorted[offs[L] + idx], (uint32_t)L);
    }
    tab[e] = v;
  }
  WAVE_SYNC();
  return 0;
}

// A code longer than the table's root: canonical range test, lane L tests length L (ROOT < L < 16).  Uniform result.
template <int ROOT, int TYPE>
__device__ __forceinline__ uint32_t long_code(uint32_t bits, const __attribute__((address_space(3))) uint16_t* sorted,
                                              const __attribute__((address_space(3))) Canon* cn, int lane) {
  const int L = (lane & 15) > ROOT ? (lane & 15) : ROOT + 1;
  const uint32_t f = cn->first[L], cnt = cn->count[L], of = cn->offs[L];
  const uint32_t c = __brev(bits) >> (32 - L), idx = c - f;
  const uint64_t hit = __ballot(lane < 16 && (lane & 15) > ROOT && c >= f && idx < cnt);
  if (!hit) return entry(0, K_INVALID, 0, 0);
  const int hl = __builtin_ctzll(hit);
  const uint32_t si = (uint32_t)__builtin_amdgcn_readlane((int)(of + idx), hl);
  return U(symbol_entry<TYPE>(sorted[si], (uint32_t)hl));
}

struct Words {  // the compressed stream, 128 words at a time in two VGPRs (lane i: words base+i, base+64+i)
  const uint32_t* w;
  uint32_t nwords, base;
  uint32_t v0, v1;
  int lane;
  __device__ __forceinline__ uint32_t fetch(uint32_t first) const {
    const uint32_t i = first + (uint32_t)lane;
    return i < nwords ? w[i] : 0u;
  }
  __device__ __forceinline__ void seek(uint32_t word) {
    base = word & ~63u;
    v0 = fetch(base);
    v1 = fetch(base + 64);
  }
  // make words [first, first + 11) readable; returns first - base (< 64)
  __device__ __forceinline__ uint32_t cover(uint32_t first) {
    uint32_t k = first - base;
    if (k >= 64) {
      if (k < 128) {
        v0 = v1;
        base += 64;
        v1 = fetch(base + 64);  // needed 53+ words from now: its latency is not waited for here
      } else {
        seek(first);
      }
      k = first - base;
    }
    return k;
  }
  __device__ __forceinline__ uint32_t word(uint32_t k) const {  // k uniform, < 128
    if (k < 64) return (uint32_t)__builtin_amdgcn_readlane((int)v0, (int)k);
    return (uint32_t)__builtin_amdgcn_readlane((int)v1, (int)(k - 64));
  }
  // the 64 stream bits from bit position bp on (uniform)
  __device__ __forceinline__ uint64_t peek(uint32_t bp) {
    const uint32_t k = cover(bp >> 5);
    const uint32_t w0 = word(k), w1 = word(k + 1), w2 = word(k + 2), sh = bp & 31;
    const uint64_t lo = ((uint64_t)w1 << 32) | w0;
    return sh ? (lo >> sh) | ((uint64_t)w2 << (64 - sh)) : lo;
  }
};

// ---- one block header: type, and for Huffman blocks the code tables (RFC 1951 3.2.3 - 3.2.7).  Kept out of line:
// its uniform arrays live in scalar registers and would otherwise crowd the decode loop's.  State goes through LDS.
__device__ __noinline__ void block_header(const uint32_t* words, uint32_t nwords, uint32_t total_bits, Smem3* sm,
                                          int lane) {
  nwords = U(nwords);
  total_bits = U(total_bits);
  Words in;
  in.w = reinterpret_cast<const uint32_t*>(
      ((uint64_t)U((uint32_t)((uint64_t)words >> 32)) << 32) | U((uint32_t)(uint64_t)words));
  in.nwords = nwords;
  in.lane = lane;
  uint32_t bp = U(sm->u_bp);
  in.seek(bp >> 5);
  int err = 0;
  uint32_t last = 0, type = 0;
  do {
    if (bp + 3 > total_bits) {
      err = 1;
      break;
    }
    uint64_t win = in.peek(bp);
    last = (uint32_t)win & 1;
    type = (uint32_t)(win >> 1) & 3;
    bp += 3;
    if (type == 3) {
      err = 1;
      break;
    }
    if (type == 0) break;  // stored: the kernel copies the bytes
    if (type == 1) {
      for (int s = lane; s < 288; s += 64) sm->lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
      WAVE_SYNC();
      build_table<kLitRoot, T_LIT, 288>(sm->lens, 288, sm->lit_tab, sm->lit_sorted, &sm->lit_cn, lane);
      if (lane < 32) sm->lens[lane] = 5;  // 30 and 31 are part of the fixed code and never valid (symbol_entry)
      WAVE_SYNC();
      build_table<kDistRoot, T_DIST, 32>(sm->lens, 32, sm->dist_tab, sm->dist_sorted, &sm->dist_cn, lane);
      break;
    }
    if (bp + 14 > total_bits) {
      err = 1;
      break;
    }
    win = in.peek(bp);
    const uint32_t hlit = ((uint32_t)win & 31) + 257, hdist = ((uint32_t)(win >> 5) & 31) + 1,
                   hclen = ((uint32_t)(win >> 10) & 15) + 4;
    bp += 14;
    if (hlit > 286 || hdist > 30) {  // zlib: "too many length or distance symbols"
      err = 1;
      break;
    }
    win = in.peek(bp);  // up to 19 x 3 = 57 bits
    if (lane < 19) {
      // order of the code-length code lengths (RFC 1951 3.2.7): 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
      constexpr uint8_t kOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
      uint32_t pos = 0;  // which transmitted slot carries symbol `lane`
#pragma unroll
      for (int i = 0; i < 19; ++i)
        if (kOrder[i] == lane) pos = (uint32_t)i;
      sm->cl_lens[lane] = pos < hclen ? (uint8_t)((win >> (3 * pos)) & 7) : 0;
    }
    bp += 3 * hclen;
    WAVE_SYNC();
    if (build_table<kClRoot, T_CL, 19>(sm->cl_lens, 19, sm->dist_tab, sm->dist_sorted, &sm->dist_cn, lane)) {
      err = 1;
      break;
    }
    const uint32_t total = hlit + hdist;
    uint32_t i = 0, prev = 0;
    while (i < total) {
      if (bp > total_bits) {
        err = 1;
        break;
      }
      win = in.peek(bp);
      const uint32_t e = U(sm->dist_tab[(uint32_t)win & ((1 << kClRoot) - 1)]);
      if (((e >> 4) & 7) != K_CL) {
        err = 1;
        break;
      }
      const uint32_t n = e & 15, sym = e >> 16;
      uint32_t rep = 1, val = sym;
      bp += n;
      if (sym == 16) {
        if (i == 0) {
          err = 1;
          break;
        }
        rep = 3 + ((uint32_t)(win >> n) & 3);
        val = prev;
        bp += 2;
      } else if (sym == 17) {
        rep = 3 + ((uint32_t)(win >> n) & 7);
        val = 0;
        bp += 3;
      } else if (sym == 18) {
        rep = 11 + ((uint32_t)(win >> n) & 127);
        val = 0;
        bp += 7;
      }
      if (i + rep > total) {
        err = 1;
        break;
      }
      for (uint32_t j = (uint32_t)lane; j < rep; j += 64) sm->lens[i + j] = (uint8_t)val;
      i += rep;
      prev = val;
    }
    if (err) break;
    WAVE_SYNC();
    if (U(sm->lens[256]) == 0) {  // zlib: "invalid code -- missing end-of-block"
      err = 1;
      break;
    }
    if (build_table<kLitRoot, T_LIT, 288>(sm->lens, (int)hlit, sm->lit_tab, sm->lit_sorted, &sm->lit_cn, lane)) {
      err = 1;
      break;
    }
    // the distance lengths follow the literal/length ones; move them to the front for the builder
    uint8_t dl = 0;
    if (lane < 32) dl = (uint32_t)lane < hdist ? sm->lens[hlit + lane] : 0;
    WAVE_SYNC();
    if (lane < 32) sm->lens[lane] = dl;
    WAVE_SYNC();
    if (build_table<kDistRoot, T_DIST, 32>(sm->lens, (int)hdist, sm->dist_tab, sm->dist_sorted, &sm->dist_cn, lane)) err = 1;
  } while (false);
  WAVE_SYNC();
  if (lane == 0) {
    sm->u_bp = bp;
    sm->u_err = err;
    sm->u_last = last;
    sm->u_type = type;
  }
  WAVE_SYNC();
}

// ---- lane-parallel lookup: the symbol that WOULD start at this lane's bit position (lo/hi = the 64 stream bits
// from there), literal/length code, extra bits, distance code and extra bits, all resolved from the two tables.
// Three stages over kSpan windows, so that the windows' table reads are in flight together.
__device__ __forceinline__ void lookup_span(const Smem3* sm, const uint32_t* lo, const uint32_t* hi, uint32_t* out) {
  uint32_t e1[kSpan], e2[kSpan], d32[kSpan], len[kSpan], e1b[kSpan];
#pragma unroll
  for (int r = 0; r < kSpan; ++r) e1[r] = sm->lit_tab[lo[r] & ((1 << kLitRoot) - 1)];
#pragma unroll
  for (int r = 0; r < kSpan; ++r) {
    const uint32_t n1 = e1[r] & 15, eb = (e1[r] >> 8) & 15;
    const uint64_t w1 = (((uint64_t)hi[r] << 32) | lo[r]) >> n1;
    len[r] = (e1[r] >> 16) + ((uint32_t)w1 & ((1u << eb) - 1));
    d32[r] = (uint32_t)(w1 >> eb);  // >= 44 valid bits were left: a distance takes at most 28
    e2[r] = sm->dist_tab[d32[r] & ((1 << kDistRoot) - 1)];
    // behind a literal (eb = 0: d32 = the bits behind its code): the literal / length entry of the NEXT symbol
    if constexpr (kPairLits) e1b[r] = sm->lit_tab[d32[r] & ((1 << kLitRoot) - 1)];
  }
#pragma unroll
  for (int r = 0; r < kSpan; ++r) {
    const uint32_t k1 = (e1[r] >> 4) & 7, n1 = e1[r] & 15, eb = (e1[r] >> 8) & 15;
    const uint32_t k2 = (e2[r] >> 4) & 7, dn = e2[r] & 15, deb = (e2[r] >> 8) & 15;
    const uint32_t dist = (e2[r] >> 16) + ((d32[r] >> dn) & ((1u << deb) - 1));
    // a long or invalid code: the walk stops there and it is decoded serially
    uint32_t res = kStop | (S_SLOW << 7);
    if (k1 == K_LIT) res = pack_lit(n1, e1[r] >> 16);
    if constexpr (kPairLits) {
      if (k1 == K_LIT && ((e1b[r] >> 4) & 7) == K_LIT) res = pack_lit2(n1 + (e1b[r] & 15), e1[r] >> 16, e1b[r] >> 16);
    }
    if (k1 == K_EOB) res = kStop | (S_EOB << 7) | (n1 << 9);
    if (k1 == K_LEN && k2 == K_DIST) res = pack_match(n1 + eb + dn + deb, len[r], dist);
    out[r] = res;
  }
}

#ifndef HCIR_PNG_SLOW_INLINE
#define HCIR_PNG_SLOW_INLINE __noinline__
#endif
// the same for ONE position, serially, with the long codes searched (wave-uniform).  The packed symbol; for the
// end of the block kStop | S_EOB | its code length << 9; ~0: no valid symbol starts here.
__device__ HCIR_PNG_SLOW_INLINE uint32_t lookup_slow(uint32_t lo, uint32_t hi, Smem3* sm, int lane) {
  lo = U(lo);
  hi = U(hi);
  uint32_t e1 = U(sm->lit_tab[lo & ((1 << kLitRoot) - 1)]);
  if (((e1 >> 4) & 7) == K_LONG) e1 = long_code<kLitRoot, T_LIT>(lo, sm->lit_sorted, &sm->lit_cn, lane);
  const uint32_t k1 = (e1 >> 4) & 7, n1 = e1 & 15;
  if (k1 == K_LIT) return pack_lit(n1, e1 >> 16);
  if (k1 == K_EOB) return kStop | (S_EOB << 7) | (n1 << 9);
  if (k1 != K_LEN) return ~0u;
  const uint32_t eb = (e1 >> 8) & 15;
  const uint64_t w1 = (((uint64_t)hi << 32) | lo) >> n1;
  const uint32_t len = (e1 >> 16) + ((uint32_t)w1 & ((1u << eb) - 1));
  const uint32_t d32 = (uint32_t)(w1 >> eb);
  uint32_t e2 = U(sm->dist_tab[d32 & ((1 << kDistRoot) - 1)]);
  if (((e2 >> 4) & 7) == K_LONG) e2 = long_code<kDistRoot, T_DIST>(d32, sm->dist_sorted, &sm->dist_cn, lane);
  if (((e2 >> 4) & 7) != K_DIST) return ~0u;
  const uint32_t dn = e2 & 15, deb = (e2 >> 8) & 15;
  const uint32_t dist = (e2 >> 16) + ((d32 >> dn) & ((1u << deb) - 1));
  return pack_match(n1 + eb + dn + deb, len, dist);
}

// ---- the serial part: the walk from symbol start to symbol start over the four windows of a pass.
// c0..c3: the windows (lane = bit position, value = packed symbol).  In: w, p = window and position (< 64) to start
// at.  Each start's bit is set in its window's mask.  Out: w = 4 and p = the start position inside the next pass, or
// w < 4: the walk landed on a stop entry e (step 64) at position p of window w (its bit is not set).
// Written out: six scalar instructions per symbol and one conditional branch.  Measured on gfx950
// (tools/ubench/walk_latency.hip): v_readlane -> SALU -> v_readlane round trip 29 cycles, a dependent SALU
// instruction 4, a conditional branch 16 NOT taken and 20 taken - the branch count is what this loop is built around.
#ifndef HCIR_PNG_WALK_BRANCHFREE   // default: a conditional branch behind every step (16 cycles even when not taken)
#define HCIR_WALK_STEP(C, M, X)          \
  "v_readlane_b32 %[e], %[" C "], %[p]\n\t" \
  "s_bitset1_b64 %[" M "], %[p]\n\t"        \
  "s_and_b32 %[t], %[e], 127\n\t"           \
  "s_add_u32 %[p], %[p], %[t]\n\t"          \
  "s_cmp_gt_u32 %[p], 63\n\t"               \
  "s_cbranch_scc1 " X "_%=\n\t"
#define HCIR_WALK_STEPS(C, M, X) HCIR_WALK_STEP(C, M, X) HCIR_WALK_STEP(C, M, X) HCIR_WALK_STEP(C, M, X) HCIR_WALK_STEP(C, M, X)
#else
// EXPERIMENT (build flag), measured and rejected: a branch-free step - once the chain has left the window (p >= 64) a
// step changes nothing (entry, mark and advance are selected away), four steps back to back, ONE branch behind them:
// nine scalar instructions per symbol.  Walk 69 -> 79 and 83 -> 105 cycles per symbol, the 880-image launch 45.0 ->
// 52.6 ms (same box, interleaved): the selects lengthen the dependent chain by more than the branch costs.
#define HCIR_WALK_STEP(C, M, X)              \
  "v_readlane_b32 %[e2], %[" C "], %[p]\n\t" \
  "s_and_b32 %[t], %[e2], 127\n\t"           \
  "s_lshl_b64 %[b], 1, %[p]\n\t"             \
  "s_cmp_lt_u32 %[p], 64\n\t"                \
  "s_cselect_b32 %[e], %[e2], %[e]\n\t"      \
  "s_cselect_b32 %[t], %[t], 0\n\t"          \
  "s_cselect_b64 %[b], %[b], 0\n\t"          \
  "s_or_b64 %[" M "], %[" M "], %[b]\n\t"    \
  "s_add_u32 %[p], %[p], %[t]\n\t"
#define HCIR_WALK_STEPS(C, M, X)                                                                     \
  HCIR_WALK_STEP(C, M, X) HCIR_WALK_STEP(C, M, X) HCIR_WALK_STEP(C, M, X) HCIR_WALK_STEP(C, M, X) \
  "s_cmp_gt_u32 %[p], 63\n\t"                                                                     \
  "s_cbranch_scc1 " X "_%=\n\t"
#endif
#define HCIR_WALK_WINDOW(N, C, M, NEXT)                                                                    \
  ".Lw" N "_%=:\n\t" HCIR_WALK_STEPS(C, M, ".Lx" N) "s_branch .Lw" N "_%=\n"                             \
  ".Lx" N "_%=:\n\t"                                                                                       \
  "s_sub_u32 %[p], %[p], 64\n\t"                                                                           \
  "s_bitcmp1_b32 %[e], 6\n\t"                                                                              \
  "s_cbranch_scc0 " NEXT "_%=\n\t"                                                                         \
  "s_bitset0_b64 %[" M "], %[p]\n\t"                                                                       \
  "s_mov_b32 %[w], " N "\n\t"                                                                              \
  "s_branch .Lend_%=\n"
__device__ __forceinline__ void walk4(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t& w, uint32_t& p,
                                      uint64_t& m0, uint64_t& m1, uint64_t& m2, uint64_t& m3, uint32_t& e) {
  uint32_t t, e2;
  uint64_t b;
  asm volatile(
      "s_mov_b64 %[m0], 0\n\ts_mov_b64 %[m1], 0\n\ts_mov_b64 %[m2], 0\n\ts_mov_b64 %[m3], 0\n\t"
      "s_nop 3\n\t"  // p may come from a VALU-written SGPR: four wait states before it selects a lane
      "s_cmp_eq_u32 %[w], 1\n\ts_cbranch_scc1 .Lw1_%=\n\t"
      "s_cmp_eq_u32 %[w], 2\n\ts_cbranch_scc1 .Lw2_%=\n\t"
      "s_cmp_eq_u32 %[w], 3\n\ts_cbranch_scc1 .Lw3_%=\n"
      HCIR_WALK_WINDOW("0", "c0", "m0", ".Lw1") HCIR_WALK_WINDOW("1", "c1", "m1", ".Lw2")
      HCIR_WALK_WINDOW("2", "c2", "m2", ".Lw3") HCIR_WALK_WINDOW("3", "c3", "m3", ".Lpast")
      ".Lpast_%=:\n\t"
      "s_mov_b32 %[w], 4\n"
      ".Lend_%=:\n\t"
      : [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3), [p] "+s"(p), [w] "+s"(w), [e] "=&s"(e),
        [t] "=&s"(t), [e2] "=&s"(e2), [b] "=&s"(b)
      : [c0] "v"(c0), [c1] "v"(c1), [c2] "v"(c2), [c3] "v"(c3)
      : "scc");
}

// inclusive wave scan on the DPP network: four shifts inside the rows of 16, two row broadcasts
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_add(uint32_t s) {
  return s + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_excl_sum(uint32_t v, uint32_t* total) {
  uint32_t s = v;
  s = dpp_add<0x111, 0xf>(s);  // row_shr:1
  s = dpp_add<0x112, 0xf>(s);  // row_shr:2
  s = dpp_add<0x114, 0xf>(s);  // row_shr:4
  s = dpp_add<0x118, 0xf>(s);  // row_shr:8
  s = dpp_add<0x142, 0xa>(s);  // row_bcast:15 into rows 1 and 3
  s = dpp_add<0x143, 0xc>(s);  // row_bcast:31 into rows 2 and 3
  *total = (uint32_t)__builtin_amdgcn_readlane((int)s, 63);  // uniform: the write position stays scalar
  return s - v;
}

__device__ __forceinline__ void flush_units(Smem3* sm, uint8_t* raw, uint32_t& flushed, uint32_t wp, uint32_t need,
                                            int lane) {
  while (flushed + 1024 <= wp && flushed < need) {
    *reinterpret_cast<u32x4*>(raw + flushed + lane * 16) =
        *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(sm->ring + ((flushed + lane * 16) & (kRing - 1)));
    flushed += 1024;
  }
}

// ---- write one batch of symbols (lane k = symbol k, packed) into the ring.
// One match: bytes [mp, mp + len) = bytes from mp - dist on, byte j repeating byte j mod dist when the copy overlaps
// its own output (dist < len).  len <= 64: one lane per byte.
__device__ __forceinline__ void copy_match(Smem3* sm, uint32_t mp, uint32_t len, uint32_t dist, float rdist, int lane,
                                           float lanef) {
  constexpr uint32_t M = kRing - 1;
  uint32_t r = (uint32_t)lane - __umul24((uint32_t)(lanef * rdist), dist);  // lane mod dist; dist >= 64: quotient 0
  r = min(r, r - dist);                                            // the float quotient can be one short
  const uint8_t v = sm->ring[(mp - dist + r) & M];
  const __attribute__((address_space(3))) uint8_t* base = sm->ring;
  // lanes beyond the copy write to a dump byte of their own instead of toggling EXEC (a VALU -> EXEC round trip and a
  // skip branch cost more than the whole copy)
  const uint32_t off = (uint32_t)lane < len ? ((mp + lane) & M) : (uint32_t)(__builtin_offsetof(Smem, dump) + lane);
  const_cast<__attribute__((address_space(3))) uint8_t*>(base)[off] = v;
}

// The rare ordered form: a source more than kRing - tot back could share its ring slot with a byte this batch writes
// later, so the literals go in stream order, between the matches (distances close to 32 KB).
__device__ __noinline__ void resolve_ordered(Smem3* sm, uint32_t sym, uint32_t mypos, uint64_t mm, uint64_t pend,
                                             int lane) {
  constexpr uint32_t M = kRing - 1;
  mm = ((uint64_t)U((uint32_t)(mm >> 32)) << 32) | U((uint32_t)mm);
  pend = ((uint64_t)U((uint32_t)(pend >> 32)) << 32) | U((uint32_t)pend);
  while (mm) {
    const int k = __builtin_ctzll(mm);
    mm &= mm - 1;
    const uint32_t ms = (uint32_t)__builtin_amdgcn_readlane((int)sym, k),
                   mp = (uint32_t)__builtin_amdgcn_readlane((int)mypos, k);
    const uint64_t before = pend & ((1ull << k) - 1);
    if ((before >> lane) & 1) {
      sm->ring[mypos & M] = (uint8_t)(sym >> 9);
      if ((sym >> 17) & 1) sm->ring[(mypos + 1) & M] = (uint8_t)(sym >> 18);
    }
    pend &= ~before;
    __builtin_amdgcn_wave_barrier();
    const uint32_t dist = (ms >> 17) + 1u;
    const float rd = 1.0f / (float)dist;
    int32_t len = (int32_t)((ms >> 9) & 255u) + 3;
    uint32_t at = mp;
    do {
      copy_match(sm, at, (uint32_t)len, dist, rd, lane, (float)lane);
      __builtin_amdgcn_wave_barrier();
      at += 64;
      len -= 64;
    } while (len > 0);
  }
  if ((pend >> lane) & 1) {
    sm->ring[mypos & M] = (uint8_t)(sym >> 9);
    if ((sym >> 17) & 1) sm->ring[(mypos + 1) & M] = (uint8_t)(sym >> 18);
  }
}

// Returns the bytes produced; *bad: a distance reaches before the start of the data.
__device__ __forceinline__ uint32_t resolve(Smem3* sm, uint32_t sym, uint32_t nsym, uint32_t wp, int lane, bool* bad) {
  constexpr uint32_t M = kRing - 1;
  const bool live = (uint32_t)lane < nsym, is_match = live && ((sym >> 7) & 1);
  const bool lit2 = kPairLits && live && !is_match && ((sym >> 17) & 1);   // (a literal's bits 17.. are its own)
  const uint32_t mylen = !live ? 0u : (is_match ? ((sym >> 9) & 255u) + 3u : (lit2 ? 2u : 1u));
  const uint32_t mydist = (sym >> 17) + 1u;
  const float myrd = 1.0f / (float)mydist, lanef = (float)lane;
  uint32_t tot;
  const uint32_t mypos = wp + wave_excl_sum(mylen, &tot);
  *bad = __ballot(is_match && mydist > mypos) != 0;  // PNG has no preset dictionary
  uint64_t mm = __ballot(is_match);
  const uint64_t lits = __ballot(live && !is_match);
  if (__ballot(is_match && mydist + tot > (uint32_t)kRing) != 0) {
    resolve_ordered(sm, sym, mypos, mm, lits, lane);
    return tot;
  }
  if ((lits >> lane) & 1) sm->ring[mypos & M] = (uint8_t)(sym >> 9);  // every literal at once
  if constexpr (kPairLits) {   // second bytes; a lane without one writes its dump byte instead of toggling EXEC
    const uint32_t off2 = lit2 ? ((mypos + 1) & M) : (uint32_t)(__builtin_offsetof(Smem, dump) + lane);
    const __attribute__((address_space(3))) uint8_t* base = sm->ring;
    const_cast<__attribute__((address_space(3))) uint8_t*>(base)[off2] = (uint8_t)(sym >> 18);
  }
  while (mm) {                                                          // every match: lane-parallel copies
    const int k = __builtin_ctzll(mm);
    mm &= mm - 1;
    const uint32_t ms = (uint32_t)__builtin_amdgcn_readlane((int)sym, k);
    uint32_t mp = (uint32_t)__builtin_amdgcn_readlane((int)mypos, k);
    const float rd = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myrd), k));
    const uint32_t dist = (ms >> 17) + 1u;
    int32_t len = (int32_t)((ms >> 9) & 255u) + 3;
    do {  // 64 bytes at a time, front to back: the same copy
      copy_match(sm, mp, (uint32_t)len, dist, rd, lane, lanef);
      __builtin_amdgcn_wave_barrier();
      mp += 64;
      len -= 64;
    } while (len > 0);
  }
  return tot;
}

// The symbols that would start at each of the 256 bit positions from wbase on (lane = position inside a window).
__device__ __forceinline__ void span_windows(Words& in, uint32_t wbase, int lane, const Smem3* sm, uint32_t* ent) {
  const uint32_t k = in.cover(wbase >> 5);
  uint32_t u[2 * kSpan + 3];
  if (k + 2 * kSpan + 3 <= 64) {
#pragma unroll
    for (int i = 0; i < 2 * kSpan + 3; ++i) u[i] = (uint32_t)__builtin_amdgcn_readlane((int)in.v0, (int)(k + i));
  } else {
#pragma unroll
    for (int i = 0; i < 2 * kSpan + 3; ++i) u[i] = in.word(k + i);
  }
  const uint32_t o = (wbase & 31) + (uint32_t)lane, j = o >> 5, sh = o & 31;  // j in {0, 1, 2}
  uint32_t lo[kSpan], hi[kSpan];
#pragma unroll
  for (int r = 0; r < kSpan; ++r) {
    const uint32_t wa = j == 0 ? u[2 * r] : (j == 1 ? u[2 * r + 1] : u[2 * r + 2]);
    const uint32_t wb = j == 0 ? u[2 * r + 1] : (j == 1 ? u[2 * r + 2] : u[2 * r + 3]);
    const uint32_t wc = j == 0 ? u[2 * r + 2] : (j == 1 ? u[2 * r + 3] : u[2 * r + 4]);
    lo[r] = __builtin_amdgcn_alignbit(wb, wa, sh);
    hi[r] = __builtin_amdgcn_alignbit(wc, wb, sh);
  }
  lookup_span(sm, lo, hi, ent);
}

// ---- the wavefronts of an image.  Wave 0 DECODES: block headers and code tables, the lane-parallel lookup, the
// walk, the queue appends.  Wave 1 COPIES: batches of 64 queued symbols into the ring, finished units to HBM.  They meet
// in the symbol queue only (LDS): the decoder publishes q_tail behind its entries, the copier q_head behind its
// reads - LDS operations of one wavefront complete in order, so a counter read after it was written shows the data
// written before it.  Neither wave ever waits inside a barrier for the other; the waits are bounded polls.
// Plain (volatile) LDS accesses between compiler-only fences: no s_waitcnt on the vector-memory counter, which a
// workgroup-scope release would put in front of the store (the decoder's stream prefetch and the copier's write-out
// stores are in flight and have nothing to do with the queue).
__device__ __forceinline__ uint32_t lds_load(const __attribute__((address_space(3))) uint32_t* p) {
  const uint32_t v = *reinterpret_cast<const volatile __attribute__((address_space(3))) uint32_t*>(p);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  return U(v);
}
__device__ __forceinline__ void lds_store(__attribute__((address_space(3))) uint32_t* p, uint32_t v, int lane) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) *reinterpret_cast<volatile __attribute__((address_space(3))) uint32_t*>(p) = v;
}
constexpr uint32_t kPollLimit = 1u << 22;  // polls of ~64 cycles: a wave that waits longer than this gives up (corrupt)

template <bool LW>  // LW: a third wavefront does the lookups (lookup_wave); else this one does its own
__device__ __forceinline__ int decode_wave(const PngBatch& a, Smem3* sm, const uint8_t* stream, uint32_t stream_bytes,
                                           int lane, uint64_t* diag) {
  const uint32_t total_bits = stream_bytes * 8u;
  Words in;
  in.w = reinterpret_cast<const uint32_t*>(stream);
  in.nwords = (stream_bytes + 3) / 4 + 4;  // staged with >= 16 zero bytes behind the stream
  in.lane = lane;
  in.seek(0);
  int err = 0;
  uint32_t bp = 16, tail = 0, head = 0;  // head: the copier's progress as last seen
  bool stop = false;
  STAMP_DECL;
  {  // RFC 1950: CM = 8, window <= 32 KB, header check, no preset dictionary
    const uint64_t h = in.peek(0);
    const uint32_t cmf = (uint32_t)h & 255, flg = ((uint32_t)h >> 8) & 255;
    if ((cmf & 15) != 8 || (cmf >> 4) > 7 || ((cmf << 8) | flg) % 31 != 0 || (flg & 0x20) || stream_bytes < 2) err = 1;
  }
  // free queue slots (one stays free); refreshes the copier's counters when fewer than `want` are known to be free
  auto room_for = [&](uint32_t want) -> uint32_t {
    uint32_t room = kQueue - 1 - (tail - head);
    for (uint32_t polls = 0; room < want && !stop; ++polls) {
      head = lds_load(&sm->q_head);
      stop = lds_load(&sm->q_stop) != 0;
      room = kQueue - 1 - (tail - head);
      if (room >= want || stop) break;
      if (polls > kPollLimit) {
        err = 1;
        stop = true;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    return room;
  };
  uint32_t lk_seq = 0;  // passes requested so far
  auto request_lookup = [&](uint32_t at) {
    lds_store(&sm->lk_wbase, at, lane);
    lds_store(&sm->lk_req, ++lk_seq, lane);
  };
  auto wait_lookup = [&]() -> bool {  // until the last requested pass is in ent_buf
    for (uint32_t polls = 0; lds_load(&sm->lk_ready) != lk_seq; ++polls) {
      if (polls > kPollLimit) {
        err = 1;
        return false;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    return true;
  };
  bool last = false;
  while (!err && !stop && !last) {
    if (lane == 0) sm->u_bp = bp;
    WAVE_SYNC();
    STAMP(0);
    block_header(in.w, in.nwords, total_bits, sm, lane);
    STAMP(1);
    bp = U(sm->u_bp);
    err = (int)U((uint32_t)sm->u_err);
    last = U(sm->u_last) != 0;
    const uint32_t type = U(sm->u_type);
    if (err) break;
    if (type == 0) {  // stored: LEN, ~LEN at the next byte boundary, then LEN bytes: queued as literals, 64 at a time
      bp = (bp + 7) & ~7u;
      if (bp + 32 > total_bits) {
        err = 1;
        break;
      }
      const uint64_t win = in.peek(bp);
      uint32_t len = (uint32_t)win & 0xffff;
      if (len != (~(uint32_t)(win >> 16) & 0xffff)) {
        err = 1;
        break;
      }
      bp += 32;
      uint32_t src = bp >> 3;
      if (src + len > stream_bytes) {
        err = 1;
        break;
      }
      bp += len * 8;
      while (len && !stop && !err) {
        const uint32_t n = len < 64 ? len : 64;
        if (room_for(64) < 64) break;
        if ((uint32_t)lane < n) sm->symq[(tail + lane) & (kQueue - 1)] = pack_lit(0, stream[src + lane]);
        tail += n;
        lds_store(&sm->q_tail, tail, lane);
        src += n;
        len -= n;
      }
      continue;
    }
    // ---- the block's symbols.  Every pass looks up, in parallel, the symbol that would start at each of the next
    // 256 bit positions; the serial part is only the walk from one symbol's start to the next, which marks the
    // starts in a 64-bit mask per window.  Marked entries are appended to the symbol queue by their lanes.
    uint32_t wbase = bp, pos = 0;  // pass origin; next symbol's start relative to it
    bool eob = false;
    if constexpr (LW) {
      if (!wait_lookup()) break;   // a request of the previous block may still be in the works (its tables are gone)
      request_lookup(wbase);
    }
    while (!eob && !err && !stop) {
      if (wbase + pos > total_bits) {  // ran past the end of the stream
        err = 1;
        break;
      }
      // the pass's symbols come from the lookup wavefront, which was asked for them one pass ago; the request for the
      // NEXT pass (a fixed 256 bits on: independent of where this pass's walk ends) goes out before the walk starts
      uint32_t ent[kSpan];
      STAMP(0);
      COUNT(2, 1);
      if constexpr (LW) {
        if (!wait_lookup()) break;
#pragma unroll
        for (int r = 0; r < kSpan; ++r) ent[r] = sm->ent_buf[r][lane];
        request_lookup(wbase + 64 * kSpan);
      } else {
        span_windows(in, wbase, lane, sm, ent);
      }
      STAMP(2);
      // ---- walk segments of the pass: one, plus one behind every serially decoded symbol
      for (bool more = true; more;) {
        uint64_t m0, m1, m2, m3;
        uint32_t w = pos >> 6, p = pos & 63, e;
        walk4(ent[0], ent[1], ent[2], ent[3], w, p, m0, m1, m2, m3, e);
        STAMP(3);
        pos = 64 * w + p;
        bool stopped = w < 4;  // on stop entry e, at pos
        // ---- the marked lanes append their symbols to the queue (a window the walk did not visit has no marks)
        uint32_t n0 = (uint32_t)__popcll(m0), n1 = (uint32_t)__popcll(m1), n2 = (uint32_t)__popcll(m2),
                 n3 = (uint32_t)__popcll(m3);
        // one slot beyond the marks for a serially decoded symbol; at least a window's worth always (n0 <= 64)
        uint32_t want = n0 + n1 + n2 + n3 + 1;
        want = want < 65 ? 65 : (want > kQueue - kBatchMin ? kQueue - kBatchMin : want);  // what the copier surely frees
        const uint32_t room = room_for(want) - 1;
        if (stop) break;
        if (n0 + n1 + n2 + n3 > room) {
          // More starts than the queue takes right now (a pass of 1- and 2-bit codes, or a slow copier).  n0 fits:
          // keep the windows that fit and walk on from the first start of the first one that does not.
          const uint32_t d = n0 + n1 > room ? 1 : (n0 + n1 + n2 > room ? 2 : 3);
          const uint64_t md = d == 1 ? m1 : (d == 2 ? m2 : m3);
          pos = 64 * d + (uint32_t)__builtin_ctzll(md);
          stopped = false;
          if (d <= 1) m1 = 0, n1 = 0;
          if (d <= 2) m2 = 0, n2 = 0;
          m3 = 0, n3 = 0;
        }
        {
          const uint32_t r0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, tail));
          const uint32_t r1 = __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, tail + n0));
          const uint32_t r2 =
              __builtin_amdgcn_mbcnt_hi((uint32_t)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m2, tail + n0 + n1));
          const uint32_t r3 =
              __builtin_amdgcn_mbcnt_hi((uint32_t)(m3 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m3, tail + n0 + n1 + n2));
          // an unmarked lane writes to the dump slot behind the queue instead of toggling EXEC
          sm->symq[(m0 >> lane) & 1 ? (r0 & (kQueue - 1)) : kQueue] = ent[0];
          sm->symq[(m1 >> lane) & 1 ? (r1 & (kQueue - 1)) : kQueue] = ent[1];
          sm->symq[(m2 >> lane) & 1 ? (r2 & (kQueue - 1)) : kQueue] = ent[2];
          sm->symq[(m3 >> lane) & 1 ? (r3 & (kQueue - 1)) : kQueue] = ent[3];
          tail += n0 + n1 + n2 + n3;
        }
        // ---- the entry the walk stopped on
        if (stopped && ((e >> 7) & 3) == S_SLOW) {  // a long code: decode this one symbol serially, queue it
          STAMP(6);
          COUNT(3, 1);
          const uint64_t w64 = in.peek(wbase + pos);
          e = U(lookup_slow((uint32_t)w64, (uint32_t)(w64 >> 32), sm, lane));
          if (e == ~0u) {
            err = 1;
            break;
          }
          if (((e >> 7) & 3) != S_EOB) {
            if (lane == 0) sm->symq[tail & (kQueue - 1)] = e;
            tail += 1;
            pos += e & 63;
          }
          STAMP(7);
        }
        lds_store(&sm->q_tail, tail, lane);
        COUNT(0, n0 + n1 + n2 + n3);
        if (stopped && ((e >> 7) & 3) == S_EOB) {
          eob = true;
          bp = wbase + pos + ((e >> 9) & 15);
        }
        more = !eob && pos < 64 * kSpan;
        STAMP(6);
      }
      wbase += 64 * kSpan;
      pos -= 64 * kSpan;
    }
    if (bp > total_bits) err = 1;
  }
  lds_store(&sm->q_tail, tail, lane);
  lds_store(&sm->q_eos, 1u, lane);
  lds_store(&sm->lk_quit, 1u, lane);
#ifdef HCIR_PNG_STAMPS
  STAMP(0);
  if (lane == 0 && diag) {
    for (int i = 0; i < 8; ++i) diag[i] = st_acc[i];
    diag[8] = st_n[0];
    diag[10] = st_n[2];
    diag[11] = st_n[3];
  }
#endif
  return err;
}

// Wave 2 LOOKS UP: for the pass the decoder asks for (a bit position), the symbol that would start at each of its 256
// bit positions, from the block's tables in LDS, into ent_buf.  It never looks at what the decoder does with them; a
// request made with tables that are being replaced meanwhile gives entries nobody reads.
__device__ __forceinline__ void lookup_wave(Smem3* sm, const uint8_t* stream, uint32_t stream_bytes, int lane) {
  Words in;
  in.w = reinterpret_cast<const uint32_t*>(stream);
  in.nwords = (stream_bytes + 3) / 4 + 4;
  in.lane = lane;
  in.seek(0);
  uint32_t seen = 0;
  for (uint32_t polls = 0;;) {
    const uint32_t quit = lds_load(&sm->lk_quit);  // read BEFORE the request counter: a request seen after "quit" is stale
    const uint32_t req = lds_load(&sm->lk_req);
    if (req == seen) {
      if (quit || ++polls > kPollLimit) break;
      __builtin_amdgcn_s_sleep(1);
      continue;
    }
    polls = 0;
    const uint32_t wbase = lds_load(&sm->lk_wbase);  // stored before the counter
    uint32_t ent[kSpan];
    span_windows(in, wbase, lane, sm, ent);
#pragma unroll
    for (int r = 0; r < kSpan; ++r) sm->ent_buf[r][lane] = ent[r];
    seen = req;
    lds_store(&sm->lk_ready, req, lane);
  }
}

__device__ __forceinline__ int copy_wave(const PngBatch& a, Smem3* sm, uint8_t* raw, uint32_t need, int lane,
                                         uint64_t* diag) {
  int err = 0;
  uint32_t head = 0, wp = 0, flushed = 0;
  STAMP_DECL;
  for (uint32_t polls = 0;;) {
    const uint32_t eos = lds_load(&sm->q_eos);  // read BEFORE the tail: a tail read after "no more" is the final one
    const uint32_t tail = lds_load(&sm->q_tail);
    const uint32_t avail = tail - head;
    if (avail < kBatchMin && !(eos && avail)) {
      if (eos) break;  // drained
      if (++polls > kPollLimit) {
        err = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
      continue;
    }
    polls = 0;
    STAMP(0);
    const uint32_t nsym = avail < 64 ? avail : 64;
    const uint32_t sym = sm->symq[(head + lane) & (kQueue - 1)];
    COUNT(1, (uint32_t)__popcll(__ballot((uint32_t)lane < nsym && ((sym >> 7) & 1))));
    bool bad;
    wp += resolve(sm, sym, nsym, wp, lane, &bad);
    head += nsym;
    lds_store(&sm->q_head, head, lane);
    STAMP(4);
    WAVE_SYNC();
    flush_units(sm, raw, flushed, wp, need, lane);
    STAMP(5);
    if (bad) err = 1;
    if (bad || wp >= need) break;
  }
  lds_store(&sm->q_stop, 1u, lane);  // whatever the reason: the decoder need not go on
  if (!err && wp < need) err = 1;    // the stream ends before the last scanline the window needs (or the decoder failed)
  if (!err && flushed < need) {      // the last, partial unit (the buffer has 1 KB of slack behind `need`)
    WAVE_SYNC();
    *reinterpret_cast<u32x4*>(raw + flushed + lane * 16) =
        *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(sm->ring + ((flushed + lane * 16) & (kRing - 1)));
  }
#ifdef HCIR_PNG_STAMPS
  STAMP(0);
  if (lane == 0 && diag) {
    diag[13] = st_acc[0];  // waiting for symbols
    diag[14] = st_acc[4];  // copying
    diag[15] = st_acc[5];  // write-out
    diag[9] = st_n[1];
  }
#endif
  return err;
}

template <bool LW>
__global__ __launch_bounds__(LW ? 192 : 128) void png_inflate_kernel(PngBatch a) {
  __shared__ __attribute__((aligned(16))) Smem smem;
  Smem3* sm = (Smem3*)&smem;

  const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
  const int64_t img = blockIdx.x;
  const hcir_png_header* hp = reinterpret_cast<const hcir_png_header*>(a.blob) + img;
  const int32_t width = hp->width, height = hp->height, bpp = hp->bpp;
  if (width <= 0) {
    if (threadIdx.x == 0) a.status[img] = HCIR_OK;  // a file the stager rejected: skipped
    return;
  }
  png_host::Win wn;
  png_host::window(width, height, a.win_h, a.win_w, wn);
  const uint32_t need = wn.y1 > 0 && wn.x1 > wn.x0 ? (uint32_t)wn.y1 * (1u + (uint32_t)width * (uint32_t)bpp) : 0u;
  if (need == 0) {
    if (threadIdx.x == 0) a.status[img] = HCIR_OK;
    return;
  }
  if (threadIdx.x == 0) {
    sm->q_tail = sm->q_head = sm->q_eos = sm->q_stop = 0;
    sm->lk_req = sm->lk_ready = sm->lk_quit = sm->lk_wbase = 0;
    sm->errs[0] = sm->errs[1] = 0;
  }
  __syncthreads();
  uint64_t* diag = nullptr;
#ifdef HCIR_PNG_STAMPS
  diag = a.diag ? a.diag + img * 16 : nullptr;
#endif
  int err = 0;
  if (wave == 0)
    err = decode_wave<LW>(a, sm, a.blob + hp->stage_offset, hp->stream_bytes, lane, diag);
  else if (wave == 1)
    err = copy_wave(a, sm, a.raw + (uint64_t)img * a.raw_stride, need, lane, diag);
  else
    lookup_wave(sm, a.blob + hp->stage_offset, hp->stream_bytes, lane);
  if (lane == 0 && wave < 2) sm->errs[wave] = err;
  __syncthreads();
  if (threadIdx.x == 0) a.status[img] = (sm->errs[0] | sm->errs[1]) ? HCIR_ERR_INVALID : HCIR_OK;
}

// ---------------------------------------------------------------------------------------------------------
// lane j <- lane j-1 across the whole wave (lane 0 gets 0)
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

constexpr int kRawPitch = 272;  // bytes of LDS per scanline window: 256 + 16 (a multiple of 16 that is not one of 256)

template <int BPP>
__global__ __launch_bounds__(64) void png_unfilter_kernel(PngBatch a) {
  // [64 x kRawPitch] a 256-byte ring of every lane's scanline (indexed by the byte's offset in the image's buffer
  // & 255), then [x1] the packed pixels of the previous band's last row, then 64 dump words
  extern __shared__ __attribute__((aligned(16))) uint32_t dyn[];
  uint8_t* ring = reinterpret_cast<uint8_t*>(dyn);
  uint32_t* lastrow = dyn + 64 * kRawPitch / 4;
  __shared__ uint8_t pal[768];
  __shared__ int bad_filter;
  const int lane = (int)threadIdx.x;
  const int64_t img = blockIdx.x;
  const hcir_png_header* hp = reinterpret_cast<const hcir_png_header*>(a.blob) + img;
  const int32_t width = hp->width, height = hp->height;
  if (width <= 0 || hp->bpp != BPP) return;
  uint8_t* out = a.out + (uint64_t)img * (uint64_t)a.win_h * (uint64_t)a.win_w * 3u;
  png_host::Win wn;
  png_host::window(width, height, a.win_h, a.win_w, wn);
  const int32_t rows = wn.y1, xe = wn.x1;
  const bool covers = wn.ox == 0 && wn.oy == 0 && wn.x1 - wn.x0 == a.win_w && wn.y1 - wn.y0 == a.win_h;
  const bool failed = a.status[img] != HCIR_OK;
  if (!covers || failed) {  // zero where the window leaves the image (torchvision pads), and a corrupt file's window
    const uint32_t nb = (uint32_t)a.win_h * (uint32_t)a.win_w * 3u;
    for (uint32_t i = (uint32_t)lane; i < nb; i += 64) out[i] = 0;
  }
  if (failed || rows <= 0 || xe <= wn.x0) return;
  const int32_t ctype = hp->color_type;
  if (ctype == 3)
    for (int i = lane; i < 768; i += 64) pal[i] = hp->palette[i];
  if (lane == 0) bad_filter = 0;
  __syncthreads();
  const uint32_t stride = 1u + (uint32_t)width * BPP;
  const uint8_t* raw = a.raw + (uint64_t)img * a.raw_stride;  // 256-B aligned: buffer offsets keep their alignment
  uint8_t* myring = ring + lane * kRawPitch;

  for (int32_t band = 0; band * 64 < rows; ++band) {
    const int32_t r = band * 64 + lane;
    const bool live = r < rows;
    const uint32_t rowoff = (uint32_t)(live ? r : 0) * stride;  // offset of the row's filter byte
    const uint32_t ft = live ? raw[rowoff] : 0u;
    if (ft > 4) bad_filter = 1;
    const int32_t m_sub = -(int32_t)(ft == 1), m_up = -(int32_t)(ft == 2), m_avg = -(int32_t)(ft == 3),
                  m_paeth = -(int32_t)(ft == 4);
    const uint32_t data0 = rowoff + 1;
    // The lane's scanline streams through its ring in aligned 16-B pieces, 64-112 bytes ahead of the pixel it is at
    // (one coalescible 16-B load per four steps instead of twelve byte loads whose 64 addresses each are 64 cache
    // lines: the address unit was what this kernel waited for).  A piece is stored to LDS one group after its load.
    uint32_t fp = data0 & ~15u;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      *reinterpret_cast<u32x4*>(myring + (fp & 255)) = *reinterpret_cast<const u32x4*>(raw + fp);
      fp += 16;
    }
    u32x4 pend = {0, 0, 0, 0};
    uint32_t pend_at = 0;
    bool pending = false;
    uint32_t cur = 0, prv = 0;
    const int32_t steps = xe + 63;
    for (int32_t t0 = 0; t0 < steps; t0 += 4) {
      if (pending) *reinterpret_cast<u32x4*>(myring + pend_at) = pend;
      pending = live && (int32_t)fp < (int32_t)data0 + (t0 + 8 - lane) * BPP + 64;
      if (pending) {
        pend = *reinterpret_cast<const u32x4*>(raw + fp);
        pend_at = fp & 255;
        fp += 16;
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int32_t x = t0 + s - lane;
        const bool act = live && x >= 0 && x < xe;
        // No EXEC toggling except around the output store: every lane reads the previous band's last row (only lane
        // 0 keeps it), every lane writes (lane 63 to the row buffer, the others to a dump word of their own).
        const int32_t xc = x < 0 ? 0 : (x >= xe ? xe - 1 : x);
        const uint32_t lr_up = lastrow[xc], lr_ul = lastrow[xc > 0 ? xc - 1 : 0];
        uint32_t up = wave_shr1(cur), ul = wave_shr1(prv);
        up = lane == 0 ? (band > 0 ? lr_up : 0u) : up;
        ul = lane == 0 ? (band > 0 ? lr_ul : 0u) : ul;
        ul = x <= 0 ? 0u : ul;
        const uint32_t left = x > 0 ? cur : 0u;
        const uint32_t at = data0 + (uint32_t)x * BPP;
        uint32_t res = 0;
#pragma unroll
        for (int c = 0; c < BPP; ++c) {
          const int32_t av = (int32_t)((left >> (8 * c)) & 255), bv = (int32_t)((up >> (8 * c)) & 255),
                        cv = (int32_t)((ul >> (8 * c)) & 255), xv = (int32_t)myring[(at + c) & 255];
          // all five predictors, chosen by bit masks (hipcc turns the ternary chains into EXEC-masked branches:
          // fifteen skip branches per pixel were two thirds of this kernel's time)
          const int32_t pa = abs(bv - cv), pb = abs(av - cv), pc = abs(av + bv - 2 * cv);
          const int32_t ca = -(int32_t)((pa <= pb) & (pa <= pc)), cb = -(int32_t)(pb <= pc);
          const int32_t paeth = (av & ca) | (~ca & ((bv & cb) | (cv & ~cb)));
          const int32_t pred = (av & m_sub) | (bv & m_up) | (((av + bv) >> 1) & m_avg) | (paeth & m_paeth);
          res |= (uint32_t)((xv + pred) & 255) << (8 * c);
        }
        prv = act ? cur : prv;
        cur = act ? res : cur;
        lastrow[(lane == 63 && act) ? x : xe + lane] = res;  // [xe, xe + 64): the dump words
        if (act && r >= wn.y0 && x >= wn.x0) {
          uint8_t* o = out + ((uint64_t)(r - wn.y0 + wn.oy) * (uint32_t)a.win_w + (uint32_t)(x - wn.x0 + wn.ox)) * 3u;
          if (BPP >= 3) {
            o[0] = (uint8_t)res, o[1] = (uint8_t)(res >> 8), o[2] = (uint8_t)(res >> 16);
          } else if (ctype == 3) {
            const uint32_t i = (res & 255) * 3;
            o[0] = pal[i], o[1] = pal[i + 1], o[2] = pal[i + 2];
          } else {
            o[0] = o[1] = o[2] = (uint8_t)res;
          }
        }
      }
    }
    __syncthreads();
  }
  if (bad_filter && lane == 0) a.status[img] = HCIR_ERR_INVALID;
}

struct Plan {
  uint64_t raw_stride;
  int32_t max_x1;
  bool any[5];  // bytes-per-pixel classes present (index = bpp)
  size_t bytes(int64_t b) const { return (size_t)b * raw_stride + (size_t)b * 4 + 512 + (size_t)b * 128 + 256; }
};

int make_plan(const hcir_png_header* hdrs, int64_t b, int32_t win_h, int32_t win_w, Plan* p) {
  uint64_t mx = 0;
  p->max_x1 = 1;
  for (int i = 0; i < 5; ++i) p->any[i] = false;
  for (int64_t i = 0; i < b; ++i) {
    const hcir_png_header& h = hdrs[i];
    if (h.width == 0) continue;
    if (h.width < 0 || h.height <= 0 || h.bpp != png_host::bytes_per_pixel(h.color_type) || h.bpp == 0 ||
        (h.stage_offset & 15) || h.width > png_host::kMaxWidth ||
        (1 + (uint64_t)h.width * h.bpp) * (uint64_t)h.height >= (1ull << 31))
      return HCIR_ERR_INVALID;
    png_host::Win w;
    png_host::window(h.width, h.height, win_h, win_w, w);
    const uint64_t need = w.y1 > 0 ? (uint64_t)w.y1 * (1 + (uint64_t)h.width * h.bpp) : 0;
    mx = need > mx ? need : mx;
    p->max_x1 = w.x1 > p->max_x1 ? w.x1 : p->max_x1;
    p->any[h.bpp] = true;
  }
  p->raw_stride = ((mx + 1023) / 1024) * 1024 + 1024;
  return HCIR_OK;
}

}  // namespace

extern "C" size_t hcir_png_stage_bytes(const uint8_t* file, size_t nbytes) {
  hcir_png_header h;
  png_host::Idat id;
  if (png_host::parse(file, nbytes, 0, &h, &id) != HCIR_OK) return 0;
  return png_host::stage_bound(id);
}

extern "C" int hcir_png_stage(const uint8_t* file, size_t nbytes, int32_t flags, hcir_png_header* hdr, uint8_t* blob,
                              size_t blob_offset, size_t blob_cap, size_t* used) {
  if (!file || !hdr || !blob || !used || (blob_offset & 15)) return HCIR_ERR_INVALID;
  png_host::Idat id;
  const int rc = png_host::parse(file, nbytes, flags, hdr, &id);
  if (rc != HCIR_OK) return rc;
  if (blob_offset > blob_cap || png_host::stage_bound(id) > blob_cap - blob_offset) return HCIR_ERR_WORKSPACE;
  hdr->stage_offset = blob_offset;
  return png_host::stage(file, id, blob + blob_offset, used);
}

extern "C" int hcir_png_stage_batch(const uint8_t* const* files, const size_t* nbytes, int64_t b, int32_t flags,
                                    uint8_t* blob, size_t blob_cap, size_t* blob_used, int32_t* status,
                                    int32_t nthreads) {
  if (!files || !nbytes || !blob_used || !status || b <= 0) return HCIR_ERR_INVALID;
  const size_t hdr_bytes = png_host::align16((size_t)b * sizeof(hcir_png_header));
  auto run = [&](auto&& fn) {
    const int nt = nthreads < 1 ? 1 : (nthreads > 64 ? 64 : nthreads);
    std::atomic<int64_t> next{0};
    auto work = [&]() {
      for (int64_t i = next.fetch_add(1); i < b; i = next.fetch_add(1)) fn(i);
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt && t < b; ++t) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
  };
  // pass 1: the chunk walk without the CRCs gives every file's verdict and size
  std::vector<size_t> bound((size_t)b, 0);
  run([&](int64_t i) {
    hcir_png_header h;
    png_host::Idat id;
    status[i] = files[i] ? png_host::parse(files[i], nbytes[i], 0, &h, &id) : HCIR_ERR_INVALID;
    bound[(size_t)i] = status[i] == HCIR_OK ? png_host::stage_bound(id) : 0;
  });
  std::vector<size_t> off((size_t)b + 1, 0);
  off[0] = hdr_bytes;
  for (int64_t i = 0; i < b; ++i) off[(size_t)i + 1] = off[(size_t)i] + bound[(size_t)i];
  *blob_used = off[(size_t)b];
  if (!blob) return HCIR_OK;
  if (blob_cap < off[(size_t)b]) return HCIR_ERR_WORKSPACE;
  hcir_png_header* hdrs = reinterpret_cast<hcir_png_header*>(blob);
  // pass 2: parse again with the CRCs (when asked for) and copy
  run([&](int64_t i) {
    size_t used = 0;
    if (status[i] == HCIR_OK)
      status[i] = hcir_png_stage(files[i], nbytes[i], flags, &hdrs[i], blob, off[(size_t)i], blob_cap, &used);
    if (status[i] != HCIR_OK) memset(&hdrs[i], 0, sizeof(hcir_png_header));  // width 0: the device skips the image
  });
  return HCIR_OK;
}

extern "C" size_t hcir_png_workspace_bytes(const hcir_png_header* hdrs_host, int64_t b, int32_t win_h, int32_t win_w) {
  Plan p;
  if (!hdrs_host || b <= 0 || win_h <= 0 || win_w <= 0 || make_plan(hdrs_host, b, win_h, win_w, &p) != HCIR_OK) return 0;
  return p.bytes(b);
}

extern "C" int hcir_png_decode_window_u8(const void* blob_dev, const hcir_png_header* hdrs_host, int64_t b, int32_t win_h,
                                         int32_t win_w, uint8_t* out, int32_t* status_dev, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  HCIR_ENTER();
  if (!blob_dev || !hdrs_host || !out || !workspace || b <= 0 || win_h <= 0 || win_w <= 0 || b > (1 << 20) ||
      (int64_t)win_h * win_w > (int64_t(1) << 28))
    return HCIR_ERR_INVALID;
  Plan p;
  const int rc = make_plan(hdrs_host, b, win_h, win_w, &p);
  if (rc != HCIR_OK) return rc;
  if (workspace_bytes < p.bytes(b)) return HCIR_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  PngBatch a{};
  a.blob = static_cast<const uint8_t*>(blob_dev);
  a.b = b;
  a.win_h = win_h;
  a.win_w = win_w;
  a.out = out;
  uint8_t* ws = reinterpret_cast<uint8_t*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  a.raw = ws;
  a.raw_stride = p.raw_stride;
  a.status = status_dev ? status_dev : reinterpret_cast<int32_t*>(ws + (size_t)b * p.raw_stride);
  a.diag = reinterpret_cast<uint64_t*>(ws + (((size_t)b * p.raw_stride + (size_t)b * 4 + 255) & ~(size_t)255));
  // A third wavefront per image does the lookups (png_inflate_kernel<true>): it shortens every image's serial chain.
  // Before the literal pairs a full chip (four images per CU) had no issue slots to spare for it (46.6 against 44.4 ms
  // per 880 files) and launches of more than 600 images ran two wavefronts per image; with pairs the lookup is the
  // largest share of a decoding wavefront's time and the third wavefront wins everywhere (same box, tools/ab_png.py:
  // 256 files 26.0 against 34.8 ms, 880 files 32.4 against 39.3 ms, 1760 files 63.1 against 77.5 ms).
  // -DHCIR_PNG_LW_MAX=<images>: A/B switch (0: never).
#ifndef HCIR_PNG_LW_MAX
#define HCIR_PNG_LW_MAX 0x7fffffff
#endif
  if (b <= HCIR_PNG_LW_MAX)
    hipLaunchKernelGGL(png_inflate_kernel<true>, dim3((unsigned)b), dim3(192), 0, st, a);
  else
    hipLaunchKernelGGL(png_inflate_kernel<false>, dim3((unsigned)b), dim3(128), 0, st, a);
  HCIR_LAUNCH_CHECK();
  const size_t lds = (size_t)64 * kRawPitch + ((size_t)p.max_x1 + 64) * 4;  // + the 64 dump words behind the row
  if (p.any[1]) hipLaunchKernelGGL(png_unfilter_kernel<1>, dim3((unsigned)b), dim3(64), lds, st, a);
  if (p.any[2]) hipLaunchKernelGGL(png_unfilter_kernel<2>, dim3((unsigned)b), dim3(64), lds, st, a);
  if (p.any[3]) hipLaunchKernelGGL(png_unfilter_kernel<3>, dim3((unsigned)b), dim3(64), lds, st, a);
  if (p.any[4]) hipLaunchKernelGGL(png_unfilter_kernel<4>, dim3((unsigned)b), dim3(64), lds, st, a);
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}
