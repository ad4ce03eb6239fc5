// png.hip — PNG decode of CenterCrop windows on the device (include/hcir.h "PNG decode on the device").
//
// Stands where the reference decodes on the host: HP/utils/dataloader.py:28-31 (read_file + decode_image(RGB)) and
// src/models/hair_encoder.py:108,169 (PIL), for the format every hair-region crop it lists is stored in
// (HairPretraining/data/data_train.csv: *_hair.png; assets/hair_region_only/*.png).
//
// Two kernels, one 64-lane wavefront per image each (a batch is hundreds of images; a wave per SIMD):
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

constexpr int kRing = 32768;  // deflate's maximum distance: the ring never needs to be larger (reads of a copy come
                              // before its writes, pending literals are written in stream order)
constexpr int kLitRoot = 10, kDistRoot = 9, kClRoot = 7;
constexpr uint32_t K_INVALID = 0, K_LIT = 1, K_LEN = 2, K_EOB = 3, K_DIST = 4, K_LONG = 5, K_CL = 6;
enum { T_CL = 0, T_LIT = 1, T_DIST = 2 };

struct PngBatch {
  const uint8_t* blob;
  int64_t b;
  int32_t win_h, win_w;
  uint8_t* out;
  int32_t* status;
  uint8_t* raw;
  uint64_t raw_stride;
};

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

struct Canon {  // per table, in LDS: canonical code ranges by length
  uint32_t first[16], count[16], offs[16];
};

// Build one decode table from code lengths lens[0..n) (LDS).  Returns 0, or 1 when the set of lengths is not
// acceptable to zlib's inflate_table: over-subscribed, or incomplete with anything but a single 1-bit code
// (a code-length code must be complete).  ROOT-bit table `tab`, symbols sorted by (length, value) in `sorted`.
template <int ROOT, int TYPE, int MAXN>
__device__ int build_table(const uint8_t* lens, int n, uint32_t* tab, uint16_t* sorted, Canon* cn, int lane) {
  constexpr int kChunks = (MAXN + 63) / 64;
  uint32_t run[16];
#pragma unroll
  for (int l = 0; l < 16; ++l) run[l] = 0;
  uint32_t myrank[kChunks], mylen[kChunks];
#pragma unroll
  for (int c = 0; c < kChunks; ++c) {
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
  __syncthreads();
#pragma unroll
  for (int c = 0; c < kChunks; ++c) {
    const int s = c * 64 + lane;
    if (mylen[c]) sorted[cn->offs[mylen[c]] + myrank[c]] = (uint16_t)s;
  }
  __syncthreads();
  // every ROOT-bit index finds the code it starts with: the first L bits (stream order = most significant code
  // bit first) form the L-bit number c; c is a code of length L iff first[L] <= c < first[L] + count[L]
  for (int e = lane; e < (1 << ROOT); e += 64) {
    uint32_t v = entry(0, maxlen > ROOT ? K_LONG : K_INVALID, 0, 0);
    const uint32_t rev = __brev((uint32_t)e);
#pragma unroll
    for (int L = (ROOT < 15 ? ROOT : 15); L >= 1; --L) {  // descending, so that the shortest match is kept
      const uint32_t c = rev >> (32 - L), idx = c - first[L];
      if (c >= first[L] && idx < run[L]) v = symbol_entry<TYPE>(sorted[offs[L] + idx], (uint32_t)L);
    }
    tab[e] = v;
  }
  __syncthreads();
  return 0;
}

// A code longer than the table's root: canonical range search over lengths ROOT+1..15 (uniform).
template <int ROOT, int TYPE>
__device__ __noinline__ uint32_t long_code(uint32_t bits, const uint16_t* sorted, const Canon* cn) {
  const uint32_t rev = __brev(bits);
  for (int L = ROOT + 1; L < 16; ++L) {
    const uint32_t c = rev >> (32 - L), f = cn->first[L], idx = c - f;
    if (c >= f && idx < cn->count[L]) return symbol_entry<TYPE>(sorted[cn->offs[L] + idx], (uint32_t)L);
  }
  return entry(0, K_INVALID, 0, 0);
}

struct Words {  // the compressed stream, 128 words at a time in two VGPRs
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
  __device__ __forceinline__ uint32_t word(uint32_t k) const {  // k uniform, < 128; v1 (the prefetch) only if needed
    if (k < 64) return __builtin_amdgcn_readlane(v0, k);
    return __builtin_amdgcn_readlane(v1, k - 64);
  }
  // the 64 stream bits from bit position bp on (bp uniform)
  __device__ __forceinline__ uint64_t peek(uint32_t bp) {
    uint32_t k = (bp >> 5) - base;
    if (k >= 64) {  // the look may touch words k .. k+2: keep them inside [base, base+128)
      if (k < 128) {
        v0 = v1;
        base += 64;
        v1 = fetch(base + 64);
      } else {
        seek(bp >> 5);
      }
      k = (bp >> 5) - base;
    }
    uint32_t w0, w1, w2;
    if (k < 62) {
      w0 = __builtin_amdgcn_readlane(v0, k), w1 = __builtin_amdgcn_readlane(v0, k + 1),
      w2 = __builtin_amdgcn_readlane(v0, k + 2);
    } else {
      w0 = word(k), w1 = word(k + 1), w2 = word(k + 2);
    }
    const uint32_t sh = bp & 31;
    const uint64_t lo = ((uint64_t)w1 << 32) | w0;
    return sh ? (lo >> sh) | ((uint64_t)w2 << (64 - sh)) : lo;
  }
};

__device__ __forceinline__ uint32_t wave_excl_sum(uint32_t v, int lane, uint32_t* total) {
  uint32_t s = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(s, o);
    if (lane >= o) s += t;
  }
  *total = __shfl(s, 63);
  return s - v;
}

__global__ __launch_bounds__(64) void png_inflate_kernel(PngBatch a) {
  __shared__ __attribute__((aligned(16))) uint8_t ring[kRing];
  __shared__ uint32_t lit_tab[1 << kLitRoot];
  __shared__ uint32_t dist_tab[1 << kDistRoot];
  __shared__ uint16_t lit_sorted[288], dist_sorted[32];
  __shared__ Canon lit_cn, dist_cn;
  // the code-length code is dead once the lengths are read: it borrows the distance table's storage
  uint32_t* cl_tab = dist_tab;
  uint16_t* cl_sorted = dist_sorted;
  Canon* cl_cnp = &dist_cn;
  __shared__ __attribute__((aligned(4))) uint8_t lens[320 + 16];
  __shared__ __attribute__((aligned(4))) uint8_t cl_lens[20];

  const int lane = (int)threadIdx.x;
  const int64_t img = blockIdx.x;
  const hcir_png_header* hp = reinterpret_cast<const hcir_png_header*>(a.blob) + img;
  const int32_t width = hp->width, height = hp->height, bpp = hp->bpp;
  if (width <= 0) {
    if (lane == 0) a.status[img] = HCIR_OK;  // a file the stager rejected: skipped
    return;
  }
  png_host::Win wn;
  png_host::window(width, height, a.win_h, a.win_w, wn);
  const uint32_t need = wn.y1 > 0 && wn.x1 > wn.x0 ? (uint32_t)wn.y1 * (1u + (uint32_t)width * (uint32_t)bpp) : 0u;
  if (need == 0) {
    if (lane == 0) a.status[img] = HCIR_OK;
    return;
  }
  const uint8_t* stream = a.blob + hp->stage_offset;
  const uint32_t stream_bytes = hp->stream_bytes, total_bits = stream_bytes * 8u;
  uint8_t* raw = a.raw + (uint64_t)img * a.raw_stride;

  Words in;
  in.w = reinterpret_cast<const uint32_t*>(stream);
  in.nwords = (stream_bytes + 3) / 4 + 4;  // staged with >= 16 zero bytes behind the stream
  in.lane = lane;
  in.seek(0);

  int err = 0;
  uint32_t bp = 16, wp = 0, flushed = 0;
  {  // RFC 1950: CM = 8, window <= 32 KB, header check, no preset dictionary
    const uint64_t h = in.peek(0);
    const uint32_t cmf = (uint32_t)h & 255, flg = ((uint32_t)h >> 8) & 255;
    if ((cmf & 15) != 8 || (cmf >> 4) > 7 || ((cmf << 8) | flg) % 31 != 0 || (flg & 0x20) || stream_bytes < 2) err = 1;
  }

  bool last = false, done = false;
  while (!err && !done && !last) {
    if (bp + 3 > total_bits) {
      err = 1;
      break;
    }
    uint64_t win = in.peek(bp);
    last = win & 1;
    const uint32_t type = (uint32_t)(win >> 1) & 3;
    bp += 3;
    if (type == 3) {
      err = 1;
      break;
    }
    if (type == 0) {  // stored: LEN, ~LEN at the next byte boundary, then LEN bytes as they are
      bp = (bp + 7) & ~7u;
      if (bp + 32 > total_bits) {
        err = 1;
        break;
      }
      win = in.peek(bp);
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
      while (len && !done) {
        const uint32_t n = len < 1024 ? len : 1024;
        for (uint32_t j = (uint32_t)lane; j < n; j += 64) ring[(wp + j) & (kRing - 1)] = stream[src + j];
        wp += n;
        src += n;
        len -= n;
        __syncthreads();
        while (flushed + 1024 <= wp && flushed < need) {
          *reinterpret_cast<u32x4*>(raw + flushed + lane * 16) =
              *reinterpret_cast<const u32x4*>(ring + ((flushed + lane * 16) & (kRing - 1)));
          flushed += 1024;
        }
        if (wp >= need) done = true;
      }
      in.seek(bp >> 5);
      continue;
    }
    // ---- code tables of this block
    if (type == 1) {
      for (int s = lane; s < 288; s += 64) lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
      __syncthreads();
      build_table<kLitRoot, T_LIT, 288>(lens, 288, lit_tab, lit_sorted, &lit_cn, lane);
      if (lane < 32) lens[lane] = 5;  // 30 and 31 are part of the fixed code and never valid (symbol_entry)
      __syncthreads();
      build_table<kDistRoot, T_DIST, 32>(lens, 32, dist_tab, dist_sorted, &dist_cn, lane);
    } else {
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
        cl_lens[lane] = pos < hclen ? (uint8_t)((win >> (3 * pos)) & 7) : 0;
      }
      bp += 3 * hclen;
      __syncthreads();
      if (build_table<kClRoot, T_CL, 19>(cl_lens, 19, cl_tab, cl_sorted, cl_cnp, lane)) {
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
        const uint32_t e = __builtin_amdgcn_readfirstlane(cl_tab[(uint32_t)win & ((1 << kClRoot) - 1)]);
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
        for (uint32_t j = (uint32_t)lane; j < rep; j += 64) lens[i + j] = (uint8_t)val;
        i += rep;
        prev = val;
      }
      if (err) break;
      __syncthreads();
      if (lens[256] == 0) {  // zlib: "invalid code -- missing end-of-block"
        err = 1;
        break;
      }
      if (build_table<kLitRoot, T_LIT, 288>(lens, (int)hlit, lit_tab, lit_sorted, &lit_cn, lane)) {
        err = 1;
        break;
      }
      // the distance lengths follow the literal/length ones; move them to the front for the builder
      uint8_t dl = 0;
      if (lane < 32) dl = (uint32_t)lane < hdist ? lens[hlit + lane] : 0;
      __syncthreads();
      if (lane < 32) lens[lane] = dl;
      __syncthreads();
      if (build_table<kDistRoot, T_DIST, 32>(lens, (int)hdist, dist_tab, dist_sorted, &dist_cn, lane)) {
        err = 1;
        break;
      }
    }
    // ---- the block's symbols, in batches of up to 64
    bool eob = false;
    while (!eob && !err && !done) {
      uint32_t sym = 0;  // lane k: symbol k of the batch.  bit 31: match; literal: byte; match: len | (dist-1) << 9
      uint32_t nsym = 0, outlen = 0;
      while (nsym < 64) {
        win = in.peek(bp);
        uint32_t e = __builtin_amdgcn_readfirstlane(lit_tab[(uint32_t)win & ((1 << kLitRoot) - 1)]);
        if (((e >> 4) & 7) == K_LONG) e = long_code<kLitRoot, T_LIT>((uint32_t)win, lit_sorted, &lit_cn);
        const uint32_t kind = (e >> 4) & 7, n = e & 15;
        uint32_t s;
        if (kind == K_LIT) {
          s = e >> 16;
          outlen += 1;
          bp += n;
        } else if (kind == K_LEN) {
          const uint32_t eb = (e >> 8) & 15;
          const uint32_t len = (e >> 16) + ((uint32_t)(win >> n) & ((1u << eb) - 1));
          uint32_t used = n + eb;
          const uint32_t dbits = (uint32_t)(win >> used);
          uint32_t d = __builtin_amdgcn_readfirstlane(dist_tab[dbits & ((1 << kDistRoot) - 1)]);
          if (((d >> 4) & 7) == K_LONG) d = long_code<kDistRoot, T_DIST>(dbits, dist_sorted, &dist_cn);
          if (((d >> 4) & 7) != K_DIST) {
            err = 1;
            break;
          }
          const uint32_t dn = d & 15, deb = (d >> 8) & 15;
          const uint32_t dist = (d >> 16) + ((uint32_t)(win >> (used + dn)) & ((1u << deb) - 1));
          used += dn + deb;
          if (dist > wp + outlen) {  // before the start of the data (PNG has no preset dictionary)
            err = 1;
            break;
          }
          s = 0x80000000u | len | ((dist - 1) << 9);
          outlen += len;
          bp += used;
        } else if (kind == K_EOB) {
          bp += n;
          eob = true;
          break;
        } else {
          err = 1;
          break;
        }
        sym = (uint32_t)lane == nsym ? s : sym;  // park symbol k in lane k
        ++nsym;
        if (wp + outlen >= need) {
          done = true;
          break;
        }
      }
      if (bp > total_bits) err = 1;  // ran past the end of the stream
      if (err) break;
      // ---- resolve the batch
      const bool live = (uint32_t)lane < nsym, is_match = live && (sym >> 31);
      const uint32_t mylen = !live ? 0u : (is_match ? (sym & 511u) : 1u);
      uint32_t tot;
      const uint32_t mypos = wp + wave_excl_sum(mylen, lane, &tot);
      uint64_t pend = __ballot(live && !is_match);  // literals not yet in the ring
      uint64_t mm = __ballot(is_match);
      while (mm) {
        const int k = __builtin_ctzll(mm);
        mm &= mm - 1;
        const uint64_t before = pend & ((1ull << k) - 1);
        if (before) {
          if ((before >> lane) & 1) ring[mypos & (kRing - 1)] = (uint8_t)sym;
          pend &= ~before;
        }
        const uint32_t ms = __builtin_amdgcn_readlane(sym, k), mp = __builtin_amdgcn_readlane(mypos, k);
        const uint32_t len = ms & 511u, dist = ((ms >> 9) & 0xffffu) + 1u, src = mp - dist;
        __builtin_amdgcn_wave_barrier();
        if (dist >= len) {
          for (uint32_t j = (uint32_t)lane; j < len; j += 64) {
            const uint8_t v = ring[(src + j) & (kRing - 1)];
            ring[(mp + j) & (kRing - 1)] = v;
          }
        } else {  // the copy overlaps its own output: byte j repeats byte j mod dist
          const float rd = 1.0f / (float)dist;
          for (uint32_t j = (uint32_t)lane; j < len; j += 64) {
            uint32_t r = j - (uint32_t)((float)j * rd) * dist;
            if (r >= dist) r -= dist;
            const uint8_t v = ring[(src + r) & (kRing - 1)];
            ring[(mp + j) & (kRing - 1)] = v;
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      if ((pend >> lane) & 1) ring[mypos & (kRing - 1)] = (uint8_t)sym;
      wp += tot;
      __syncthreads();
      while (flushed + 1024 <= wp && flushed < need) {
        *reinterpret_cast<u32x4*>(raw + flushed + lane * 16) =
            *reinterpret_cast<const u32x4*>(ring + ((flushed + lane * 16) & (kRing - 1)));
        flushed += 1024;
      }
    }
  }
  if (!err && wp < need) err = 1;  // the stream ends before the last scanline the window needs
  if (!err && flushed < need) {    // the last, partial unit (the buffer has 1 KB of slack behind `need`)
    __syncthreads();
    *reinterpret_cast<u32x4*>(raw + flushed + lane * 16) =
        *reinterpret_cast<const u32x4*>(ring + ((flushed + lane * 16) & (kRing - 1)));
  }
  if (lane == 0) a.status[img] = err ? HCIR_ERR_INVALID : HCIR_OK;
}

// ---------------------------------------------------------------------------------------------------------
// lane j <- lane j-1 across the whole wave (lane 0 gets 0)
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

template <int BPP>
__global__ __launch_bounds__(64) void png_unfilter_kernel(PngBatch a) {
  extern __shared__ uint32_t lastrow[];  // packed pixels of the previous band's last row, [x1]
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
  const uint8_t* raw = a.raw + (uint64_t)img * a.raw_stride;

  for (int32_t band = 0; band * 64 < rows; ++band) {
    const int32_t r = band * 64 + lane;
    const bool live = r < rows;
    const uint8_t* row = raw + (uint64_t)(live ? r : 0) * stride;
    const uint32_t ft = live ? row[0] : 0u;
    if (ft > 4) bad_filter = 1;
    uint32_t cur = 0, prv = 0;
    uint32_t nxt[4];  // raw pixels of the next four steps, loaded four steps ahead
    auto load4 = [&](int32_t t0, uint32_t* dst) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int32_t x = t0 + s - lane;
        uint32_t v = 0;
        if (live && x >= 0 && x < xe) {
          const uint8_t* p = row + 1 + (uint32_t)x * BPP;
#pragma unroll
          for (int c = 0; c < BPP; ++c) v |= (uint32_t)p[c] << (8 * c);
        }
        dst[s] = v;
      }
    };
    load4(0, nxt);
    const int32_t steps = xe + 63;
    for (int32_t t0 = 0; t0 < steps; t0 += 4) {
      uint32_t now[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) now[s] = nxt[s];
      if (t0 + 4 < steps) load4(t0 + 4, nxt);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int32_t x = t0 + s - lane;
        const bool act = live && x >= 0 && x < xe;
        uint32_t up = wave_shr1(cur), ul = wave_shr1(prv);
        if (lane == 0) {
          up = (band > 0 && act) ? lastrow[x] : 0u;
          ul = (band > 0 && act && x > 0) ? lastrow[x - 1] : 0u;
        }
        if (x == 0) ul = 0;
        const uint32_t left = x > 0 ? cur : 0u;
        uint32_t res = 0;
#pragma unroll
        for (int c = 0; c < BPP; ++c) {
          const int32_t av = (int32_t)((left >> (8 * c)) & 255), bv = (int32_t)((up >> (8 * c)) & 255),
                        cv = (int32_t)((ul >> (8 * c)) & 255), xv = (int32_t)((now[s] >> (8 * c)) & 255);
          const int32_t pa = abs(bv - cv), pb = abs(av - cv), pc = abs(av + bv - 2 * cv);
          const int32_t paeth = (pa <= pb && pa <= pc) ? av : (pb <= pc ? bv : cv);
          const int32_t pred = ft == 0 ? 0 : ft == 1 ? av : ft == 2 ? bv : ft == 3 ? ((av + bv) >> 1) : paeth;
          res |= (uint32_t)((xv + pred) & 255) << (8 * c);
        }
        if (act) {
          prv = cur;
          cur = res;
          if (lane == 63) lastrow[x] = res;
          if (r >= wn.y0 && x >= wn.x0) {
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
    }
    __syncthreads();
  }
  if (bad_filter && lane == 0) a.status[img] = HCIR_ERR_INVALID;
}

struct Plan {
  uint64_t raw_stride;
  int32_t max_x1;
  bool any[5];  // bytes-per-pixel classes present (index = bpp)
  size_t bytes(int64_t b) const { return (size_t)b * raw_stride + (size_t)b * 4 + 512; }
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
  hipLaunchKernelGGL(png_inflate_kernel, dim3((unsigned)b), dim3(64), 0, st, a);
  HCIR_LAUNCH_CHECK();
  const size_t lds = (size_t)p.max_x1 * 4;
  if (p.any[1]) hipLaunchKernelGGL(png_unfilter_kernel<1>, dim3((unsigned)b), dim3(64), lds, st, a);
  if (p.any[2]) hipLaunchKernelGGL(png_unfilter_kernel<2>, dim3((unsigned)b), dim3(64), lds, st, a);
  if (p.any[3]) hipLaunchKernelGGL(png_unfilter_kernel<3>, dim3((unsigned)b), dim3(64), lds, st, a);
  if (p.any[4]) hipLaunchKernelGGL(png_unfilter_kernel<4>, dim3((unsigned)b), dim3(64), lds, st, a);
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}
