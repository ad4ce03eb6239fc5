// jpeg_core.h — the arithmetic of the baseline-JPEG path, shared by the HIP kernels (jpeg.hip) and by the
// host-compiled emulation that the CPU tests drive (tests/jpeg_emul.cpp: the same functions, a sequential
// loop in place of the thread grid).  Everything here is plain integer code:
//   * bit reader over the staged stream (32-bit words in big-endian bit order, stuffing / RSTn removed),
//   * one Huffman symbol step of the sequential decoder (DC difference, AC run/size, ZRL, EOB) as a pure
//     function of the decoder state (bit position, block-in-MCU, zigzag index) — which is what makes the
//     self-synchronising subsequence decode verifiable: two chains are merged iff their states are equal,
//   * jpeg_idct_islow (CONST_BITS 13, PASS1_BITS 2), libjpeg's post-IDCT range-limit table as arithmetic,
//   * h2v1 / h2v2 fancy upsampling taps and the YCbCr -> RGB fixed-point tables of libjpeg-turbo.
// oracle/jpeg.py restates the same pipeline independently (bit-serial Huffman, numpy IDCT) and is pinned to
// Pillow's libjpeg-turbo 3.1.4; the GPU path is tested against both.
#pragma once
#include <stdint.h>

#include "../../include/hcir.h"

#if defined(__HIPCC__)
#define JHD __host__ __device__ __forceinline__
#define JHD_COLD __host__ __device__ __attribute__((noinline))
#else
#define JHD inline
#define JHD_COLD inline
#endif

// zigzag index -> natural (row-major) index; 16 extra entries catch a run that overshoots 63 on a
// mis-synchronised chain (as jpeg_natural_order does in libjpeg)
JHD int jpeg_natural(int z) {
  constexpr uint8_t T[80] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33,
                             40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36,
                             29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54,
                             47, 55, 62, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};
  return T[z];
}

// ---- decoder state ---------------------------------------------------------------------------------------
struct JState {
  uint32_t p;        // bit position in the staged stream (always at a symbol boundary, never inside padding)
  uint32_t c;        // block index inside the MCU
  uint32_t z;        // zigzag index of the next coefficient (0: the block's DC symbol comes next)
  uint32_t seg;      // restart segment that contains p (a function of p; kept to avoid searching)
  uint32_t seg_end;  // first bit behind that segment
};

// what two chains compare: equal keys => equal futures
JHD uint64_t jpeg_state_key(const JState& s) { return ((uint64_t)s.p << 16) | (s.c << 8) | s.z; }

struct JStream {
  // stream words, word g holding bits [32 g, 32 g + 32) with the first bit in bit 31, stored INTERLEAVED by
  // subsequence: word j of subsequence t sits at words[j * nx + t] (jpeg_word_addr).  A wave's 64 lanes work on
  // 64 neighbouring subsequences at about the same word offset j, so one wave load touches two or three 128-byte
  // lines instead of 64 (with the stream stored linearly every lane pulled its own line through L1 for 4 useful
  // bytes, and a CU's 1500 live lines no longer fitted any cache: phases ran 3-4x slower than their issue rate)
  const uint32_t* words;
  uint32_t wps, nx;           // words per subsequence, subsequences (incl. the padding tail) = row pitch
  const uint32_t* seg_start;  // [nseg + 1] start bit of every segment; [nseg] = stream_bits
  uint32_t nseg, stream_bits, bpm;
  const hcir_jpeg_lut* luts;      // the four tables' lookup parts (LDS on the device)
  const hcir_jpeg_hufftab* tabs;  // the header's full tables (range search for codes no lookup covers)
  uint32_t dc_sel, ac_sel;        // 4 bits per block of the MCU: which table (a register, not an indexed array)
  uint32_t fast2;                 // every table in use resolves its long codes in the second lookup level
};

JHD void jpeg_stream_tables(const hcir_jpeg_header& h, JStream& J) {
  J.dc_sel = J.ac_sel = 0;
  J.fast2 = 1;
  for (int i = 0; i < h.blocks_per_mcu && i < 8; ++i) {
    const int ci = h.blk_comp[i];
    J.dc_sel |= (uint32_t)h.dc_tab[ci] << (4 * i);
    J.ac_sel |= (uint32_t)h.ac_tab[ci] << (4 * i);
    J.fast2 &= h.huff[h.dc_tab[ci]].lut.use2 & h.huff[h.ac_tab[ci]].lut.use2;
  }
}

// subsequence geometry of a stream decoded by nthreads threads (one place, used by host and device alike)
struct JSubseq {
  uint32_t bits, wps, nact, nx;  // bits per subsequence (multiple of 32, >= 128), words, subsequences, row pitch
};
JHD void jpeg_subseq(uint32_t stream_bits, uint32_t stream_words, uint32_t nthreads, JSubseq& q) {
  uint32_t S = (stream_bits + nthreads - 1) / nthreads;
  S = (S + 31) & ~31u;
  if (S < 128) S = 128;
  q.bits = S;
  q.wps = S / 32;
  q.nact = (stream_bits + S - 1) / S;
  q.nx = (stream_words + q.wps - 1) / q.wps;  // stream_words includes the padding words the reader may touch
}

JHD uint32_t jpeg_word_addr(const JStream& J, uint32_t g) {
  const uint32_t t = g / J.wps;
  return (g - t * J.wps) * J.nx + t;
}

JHD uint32_t jpeg_peek32(const JStream& J, uint32_t p) {
  const uint32_t g = p >> 5, s = p & 31;
  const uint64_t v = ((uint64_t)J.words[jpeg_word_addr(J, g)] << 32) | J.words[jpeg_word_addr(J, g + 1)];
  return (uint32_t)((v << s) >> 32);
}

// The hot loop's view of the stream: four consecutive words in registers.  Symbols are at most 31 bits, so the
// word index advances by 0 or 1 per symbol.  Branch-free: every step selects the shifted or the unshifted window and
// issues ONE load for the word three ahead (the same word again when the index did not move); the loaded word is
// first needed a step later, so its latency stays off the symbol-to-symbol dependency chain.  The interleaved
// address of that word is stepped (one row down, or up to the next subsequence's first row), not divided.  A
// segment jump (the only way the index moves otherwise) calls load() again.
struct JBitWin {
  uint32_t wi, w0, w1, w2, w3;
  uint32_t a3, r3;  // address of word wi + 3 and how many words are left below it in its subsequence
  JHD void load(const JStream& J, uint32_t p) {
    wi = p >> 5;
    w0 = J.words[jpeg_word_addr(J, wi)];
    w1 = J.words[jpeg_word_addr(J, wi + 1)];
    w2 = J.words[jpeg_word_addr(J, wi + 2)];
    const uint32_t t = (wi + 3) / J.wps, j = wi + 3 - t * J.wps;
    a3 = j * J.nx + t;
    r3 = J.wps - 1 - j;
    w3 = J.words[a3];
  }
  JHD uint32_t peek(const JStream& J, uint32_t p) {
    const uint32_t i = p >> 5;
    // (a refill under a branch, so that the new word would first be touched at the NEXT refill, was tried: hipcc
    // rotates the window registers at the loop's back edge and waits for the load in every iteration all the same)
    const bool adv = i != wi;
    w0 = adv ? w1 : w0;
    w1 = adv ? w2 : w1;
    w2 = adv ? w3 : w2;
    const bool wrap = r3 == 0;
    const uint32_t step = wrap ? 1u - (J.wps - 1) * J.nx : J.nx;  // unsigned wrap-around: one column right, wps - 1 rows up
    a3 += adv ? step : 0u;
    r3 = adv ? (wrap ? J.wps - 1 : r3 - 1) : r3;
    w3 = J.words[a3];
    wi = i;
    return (uint32_t)(((((uint64_t)w0 << 32) | w1) << (p & 31)) >> 32);
  }
};

JHD void jpeg_state_at(const JStream& J, uint32_t p, uint32_t c, uint32_t z, JState& s) {
  s.p = p;
  s.c = c;
  s.z = z;
  uint32_t lo = 0, hi = J.nseg;  // last segment whose start is <= p
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (J.seg_start[mid] <= p) lo = mid; else hi = mid;
  }
  s.seg = lo;
  s.seg_end = J.seg_start[lo + 1];
  if (p >= J.stream_bits) {
    s.p = J.stream_bits;
    s.seg = J.nseg;
    s.seg_end = J.stream_bits;
    s.c = s.z = 0;
  }
}

// jdhuff.c's decode in table form.  bits32: next 32 stream bits, first in bit 31; lut: the table of the symbol that
// comes next.  Returns the packed entry (include/hcir.h): codes of up to HCIR_JPEG_LOOK_BITS bits come out of one
// lookup, longer ones out of a second one indexed by the distance of the 16-bit prefix from the first long code
// (canonical codes pack the long ones at the top of the prefix space: 192 prefixes behind 11 bits for the standard
// luminance AC table); a table whose long codes span more than HCIR_JPEG_LOOK2 prefixes falls back to the canonical
// ranges in global memory (the length is the number of limits the prefix has reached).  An invalid prefix (only a
// chain that is not synchronised sees one) decodes as a 16-bit code that advances the zigzag index by one.
JHD uint32_t jpeg_huff_entry(const JStream& J, const hcir_jpeg_lut* lut, uint32_t bits32) {
  uint32_t e = lut->look[bits32 >> (32 - HCIR_JPEG_LOOK_BITS)];
  if (e == 0) {
    const uint32_t v = bits32 >> 16;
    if (lut->use2) {
      e = lut->look2[v & (HCIR_JPEG_LOOK2 - 1)];  // use2: every long code lies in the top LOOK2 prefixes
    } else {
      const hcir_jpeg_hufftab* f = J.tabs + (uint32_t)(lut - J.luts);
      uint32_t l = HCIR_JPEG_LOOK_BITS + 1;
#pragma unroll
      for (int k = HCIR_JPEG_LOOK_BITS + 1; k < 16; ++k) l += v >= f->limit[k] ? 1u : 0u;
      if (v < f->limit[16]) {
        const uint32_t sym = f->vals[(uint32_t)((int32_t)(v >> (16 - l)) + f->valoff[l]) & 255];
        const uint32_t ssss = sym & 15, r = sym >> 4;
        const uint32_t adv = !f->is_ac ? 1u : (ssss ? r + 1 : (r == 15 ? 16u : 0u));
        e = (l + ssss) | (adv << 5) | (l << 10);
      }
    }
    if (e == 0) e = 16u | (1u << 5) | (16u << 10);
  }
  return e;
}

JHD int32_t jpeg_extend(uint32_t v, uint32_t s) {  // HUFF_EXTEND
  return v < (1u << (s - 1)) ? (int32_t)v - (int32_t)((1u << s) - 1) : (int32_t)v;
}

struct JNullSink {
  JHD void coef(uint32_t, uint32_t, uint32_t, uint32_t) {}
  JHD void block_done() {}
  JHD bool finished() const { return false; }
};

// Decodes every symbol that STARTS before bit `limit`; returns the number of blocks completed.  One code path
// for DC and AC symbols (a wave's lanes sit at different places of their blocks): the packed entry says how many
// bits the symbol takes and how far the zigzag index moves; a sink that keeps coefficients gets the position, the
// magnitude size and the bits to extract the value from.  The tables of the current block (DC / AC lookup of its
// component) are two pointers refreshed when a block ends.  After each symbol the state is canonicalised: fewer
// than 8 bits left in the segment and all of them 1 (the encoder's padding — no Huffman code is all ones, ITU
// T.81 Annex C) or the segment exhausted => continue at the next segment's first bit with (c, z) = (0, 0).
template <class Sink>
JHD uint32_t jpeg_decode_span_generic(const JStream& J, JState& s, uint32_t limit, Sink& sink) {
  uint32_t nblk = 0;
  JBitWin bw;
  bw.load(J, s.p);
  const hcir_jpeg_lut* lut_dc = J.luts + ((J.dc_sel >> (4 * s.c)) & 15);
  const hcir_jpeg_lut* lut_ac = J.luts + ((J.ac_sel >> (4 * s.c)) & 15);
  while (s.p < limit && !sink.finished()) {
    const uint32_t bits = bw.peek(J, s.p);
    const uint32_t e = jpeg_huff_entry(J, s.z == 0 ? lut_dc : lut_ac, bits);
    const uint32_t len = e & 31, adv5 = (e >> 5) & 31;
    if (adv5) sink.coef(s.z + adv5 - 1, len - (e >> 10), bits, e >> 10);  // position, magnitude size, bits, code length
    s.z += adv5 ? adv5 : 64u;
    s.p += len;
    if (s.z >= 64) {
      s.z = 0;
      s.c = (s.c + 1 == J.bpm) ? 0 : s.c + 1;
      lut_dc = J.luts + ((J.dc_sel >> (4 * s.c)) & 15);
      lut_ac = J.luts + ((J.ac_sel >> (4 * s.c)) & 15);
      ++nblk;
      sink.block_done();
    }
    if (s.p + 8 > s.seg_end) {
      bool jump = s.p >= s.seg_end;
      if (!jump) {
        const uint32_t rem = s.seg_end - s.p;
        jump = (jpeg_peek32(J, s.p) >> (32 - rem)) == ((1u << rem) - 1);
      }
      if (jump) {
        s.seg += 1;
        s.c = s.z = 0;
        if (s.seg >= J.nseg) {
          s.seg = J.nseg;
          s.p = s.seg_end = J.stream_bits;
        } else {
          s.p = J.seg_start[s.seg];
          s.seg_end = J.seg_start[s.seg + 1];
        }
        lut_dc = J.luts + (J.dc_sel & 15);
        lut_ac = J.luts + (J.ac_sel & 15);
        bw.load(J, s.p);
      }
    }
  }
  return nblk;
}

// The same decode as the loop the kernels actually run (jpeg_decode_span_generic above is its plain statement and
// the path of tables whose long codes do not fit the second lookup level).  What differs is only what costs issue
// slots on a CU whose 32 waves share ONE scalar unit: no divergent branch per symbol.  (1) Both lookup levels are read
// unconditionally and selected; (2) the end of a block is a handful of selects; (3) the padding test leaves the hot
// loop: symbols are decoded up to seven bits before the segment's end (or the span limit), and only there the
// canonicalisation is evaluated; (4) the block's table is picked by a bit-field extract of the selector word.
// kFast is a COMPILE-time choice (the launcher knows every table of the batch): the kernels carry one loop, not both.
template <bool kFast, class Sink>
JHD uint32_t jpeg_decode_span(const JStream& J, JState& s, uint32_t limit, Sink& sink) {
  if (!kFast) return jpeg_decode_span_generic(J, s, limit, sink);
  uint32_t nblk = 0, p = s.p, z = s.z, c4 = 4 * s.c;
  const uint32_t bpm4 = 4 * J.bpm;
  JBitWin bw;
  bw.load(J, p);
  auto step = [&]() {
    const uint32_t bits = bw.peek(J, p);
    const hcir_jpeg_lut* lut = J.luts + (((z == 0 ? J.dc_sel : J.ac_sel) >> c4) & 15);
    const uint32_t e1 = lut->look[bits >> (32 - HCIR_JPEG_LOOK_BITS)];
    const uint32_t e2 = lut->look2[(bits >> 16) & (HCIR_JPEG_LOOK2 - 1)];
    uint32_t e = e1 ? e1 : e2;
    e = e ? e : (16u | (1u << 5) | (16u << 10));
    const uint32_t len = e & 31, adv5 = (e >> 5) & 31;
    if (adv5) sink.coef(z + adv5 - 1, len - (e >> 10), bits, e >> 10);
    z += adv5 ? adv5 : 64u;
    p += len;
    const bool done = z >= 64;
    const uint32_t c4n = c4 + 4 == bpm4 ? 0 : c4 + 4;
    z = done ? 0 : z;
    c4 = done ? c4n : c4;
    nblk += done ? 1u : 0u;
    if (done) sink.block_done();
  };
  for (;;) {
    const uint32_t safe = s.seg_end >= 7 ? s.seg_end - 7 : 0;  // p + 8 > seg_end  <=>  p >= seg_end - 7
    const uint32_t eff = limit < safe ? limit : safe;
    while (p < eff && !sink.finished()) step();
    if (p + 8 > s.seg_end && s.seg < J.nseg) {  // the segment's last bits: padding?
      bool jump = p >= s.seg_end;
      if (!jump) {
        const uint32_t rem = s.seg_end - p;
        jump = (jpeg_peek32(J, p) >> (32 - rem)) == ((1u << rem) - 1);
      }
      if (jump) {
        s.seg += 1;
        c4 = z = 0;
        if (s.seg >= J.nseg) {
          s.seg = J.nseg;
          p = s.seg_end = J.stream_bits;
        } else {
          p = J.seg_start[s.seg];
          s.seg_end = J.seg_start[s.seg + 1];
        }
        bw.load(J, p);
        continue;
      }
      if (p < limit && !sink.finished()) {  // real symbols in the last seven bits
        step();
        continue;
      }
    }
    break;
  }
  s.p = p;
  s.z = z;
  s.c = c4 >> 2;
  return nblk;
}

// ---- window geometry ------------------------------------------------------------------------------------
struct JWin {
  int32_t x0, y0;              // image coordinates of output pixel (0, 0); negative when the image is padded
  int32_t mx0, my0, nmx, nmy;  // MCU rectangle that is reconstructed (empty: nmx == 0)
  int32_t last_mcu;            // scan-order index of the last MCU needed, -1 if none
  int32_t wblocks;             // blocks in the rectangle = nmx * nmy * blocks_per_mcu
};

JHD int32_t jpeg_pyround_half(int32_t num) {  // Python round(num / 2.0): ties to even
  if ((num & 1) == 0) return num / 2;
  const int32_t lo = (num - 1) / 2;
  return (lo & 1) ? lo + 1 : lo;
}

// torchvision center_crop((win_h, win_w)): pad to the window, crop at round((dim' - win) / 2)
JHD void jpeg_window(const hcir_jpeg_header& h, int32_t win_h, int32_t win_w, JWin& w) {
  const int32_t ph = h.height < win_h ? win_h - h.height : 0, pw = h.width < win_w ? win_w - h.width : 0;
  w.y0 = jpeg_pyround_half(h.height + ph - win_h) - ph / 2;
  w.x0 = jpeg_pyround_half(h.width + pw - win_w) - pw / 2;
  // pixels of the image the window shows, widened by one chroma sample for the triangle filter
  int32_t xa = w.x0 - (h.hmax > 1 ? h.hmax : 0), xb = w.x0 + win_w - 1 + (h.hmax > 1 ? h.hmax : 0);
  int32_t ya = w.y0 - (h.vmax > 1 ? h.vmax : 0), yb = w.y0 + win_h - 1 + (h.vmax > 1 ? h.vmax : 0);
  xa = xa < 0 ? 0 : xa;
  ya = ya < 0 ? 0 : ya;
  xb = xb > h.width - 1 ? h.width - 1 : xb;
  yb = yb > h.height - 1 ? h.height - 1 : yb;
  if (xb < xa || yb < ya) {
    w.mx0 = w.my0 = w.nmx = w.nmy = 0;
    w.last_mcu = -1;
    w.wblocks = 0;
    return;
  }
  const int32_t mw = 8 * h.hmax, mh = 8 * h.vmax;
  w.mx0 = xa / mw;
  w.my0 = ya / mh;
  w.nmx = xb / mw - w.mx0 + 1;
  w.nmy = yb / mh - w.my0 + 1;
  w.last_mcu = (w.my0 + w.nmy - 1) * h.mcus_x + w.mx0 + w.nmx - 1;
  w.wblocks = w.nmx * w.nmy * h.blocks_per_mcu;
}

// slot of scan-order block b inside the window's coefficient buffer, -1 when its MCU lies outside
JHD int32_t jpeg_window_slot(const hcir_jpeg_header& h, const JWin& w, uint32_t b) {
  const uint32_t mcu = b / (uint32_t)h.blocks_per_mcu, blk = b - mcu * (uint32_t)h.blocks_per_mcu;
  const int32_t my = (int32_t)(mcu / (uint32_t)h.mcus_x), mx = (int32_t)(mcu - (uint32_t)my * (uint32_t)h.mcus_x);
  const int32_t ry = my - w.my0, rx = mx - w.mx0;
  if (ry < 0 || ry >= w.nmy || rx < 0 || rx >= w.nmx) return -1;
  return (ry * w.nmx + rx) * h.blocks_per_mcu + (int32_t)blk;
}

// ---- jidctint.c: jpeg_idct_islow ------------------------------------------------------------------------
JHD int32_t jpeg_descale(uint32_t x, int n) { return (int32_t)(x + (1u << (n - 1))) >> n; }

// one 8-point pass; in/out strides let the same code run down columns (pass 1) and along rows (pass 2).
// All products and sums in uint32_t (two's-complement wrap, what the device's 32-bit ALU does anyway): with 16-bit
// quantisation tables or corrupt coefficients the signed forms overflow, which is undefined behaviour in the
// host-compiled emulation (tests/jpeg_emul.cpp) - found by a UBSan fuzz of that file.
JHD void jpeg_idct8(const int32_t* in, int istride, int32_t* out, int ostride, int shift) {
  typedef uint32_t U;
  constexpr U F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299,
              F1_847 = 15137, F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
  U z2 = (U)in[2 * istride], z3 = (U)in[6 * istride];
  U z1 = (z2 + z3) * F0_541;
  const U tmp2 = z1 - z3 * F1_847;
  const U tmp3 = z1 + z2 * F0_765;
  z2 = (U)in[0];
  z3 = (U)in[4 * istride];
  const U tmp0 = (z2 + z3) << 13;
  const U tmp1 = (z2 - z3) << 13;
  const U tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  U t0 = (U)in[7 * istride], t1 = (U)in[5 * istride], t2 = (U)in[3 * istride], t3 = (U)in[1 * istride];
  z1 = t0 + t3;
  z2 = t1 + t2;
  z3 = t0 + t2;
  U z4 = t1 + t3;
  const U z5 = (z3 + z4) * F1_175;
  t0 *= F0_298;
  t1 *= F2_053;
  t2 *= F3_072;
  t3 *= F1_501;
  z1 = 0u - z1 * F0_899;
  z2 = 0u - z2 * F2_562;
  z3 = z5 - z3 * F1_961;
  z4 = z5 - z4 * F0_390;
  t0 += z1 + z3;
  t1 += z2 + z4;
  t2 += z2 + z3;
  t3 += z1 + z4;
  out[0 * ostride] = jpeg_descale(tmp10 + t3, shift);
  out[7 * ostride] = jpeg_descale(tmp10 - t3, shift);
  out[1 * ostride] = jpeg_descale(tmp11 + t2, shift);
  out[6 * ostride] = jpeg_descale(tmp11 - t2, shift);
  out[2 * ostride] = jpeg_descale(tmp12 + t1, shift);
  out[5 * ostride] = jpeg_descale(tmp12 - t1, shift);
  out[3 * ostride] = jpeg_descale(tmp13 + t0, shift);
  out[4 * ostride] = jpeg_descale(tmp13 - t0, shift);
}

// range_limit[x & RANGE_MASK] with the post-IDCT table of jdmaster.c prepare_range_limit_table
JHD uint8_t jpeg_range_limit(int32_t v) {
  const uint32_t i = (uint32_t)v & 1023u;
  return (uint8_t)(i < 128 ? i + 128 : (i < 512 ? 255 : (i < 896 ? 0 : i - 896)));
}

// ---- jdcolor.c: ycc_rgb_convert ---------------------------------------------------------------------------
JHD uint8_t jpeg_clamp8(int32_t v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

JHD void jpeg_ycc_rgb(int32_t y, int32_t cb, int32_t cr, uint8_t* rgb) {
  constexpr int32_t FIX_1_40200 = 91881, FIX_1_77200 = 116130, FIX_0_71414 = 46802, FIX_0_34414 = 22554;
  const int32_t xb = cb - 128, xr = cr - 128;
  rgb[0] = jpeg_clamp8(y + ((FIX_1_40200 * xr + 32768) >> 16));
  rgb[1] = jpeg_clamp8(y + ((-FIX_0_34414 * xb + 32768 - FIX_0_71414 * xr) >> 16));
  rgb[2] = jpeg_clamp8(y + ((FIX_1_77200 * xb + 32768) >> 16));
}

// ---- jdsample.c: one output sample of a chroma plane at full-resolution pixel (x, y) -------------------------
// plane: window plane of the component (pitch `pitch`, origin (px0, py0) in the component's own grid);
// dw x dh: the component's REAL size (ceil(width * h / hmax)): neighbours clamp to it, which is what libjpeg's
// edge handling (first / last column special cases, replicated context rows) amounts to.
JHD int32_t jpeg_upsampled(const uint8_t* plane, int32_t pitch, int32_t px0, int32_t py0, int32_t dw, int32_t dh,
                           int32_t fx, int32_t fy, int32_t x, int32_t y) {
  if (fx == 1 && fy == 1) return plane[(y - py0) * pitch + (x - px0)];
  const int32_t cx = x >> 1, cy = fy == 2 ? y >> 1 : y;
  if (dw <= 2) return plane[(cy - py0) * pitch + (cx - px0)];  // jinit_upsampler: no fancy path, box replication
  int32_t cxn = (x & 1) ? cx + 1 : cx - 1;
  cxn = cxn < 0 ? 0 : (cxn > dw - 1 ? dw - 1 : cxn);
  const uint8_t* r0 = plane + (cy - py0) * pitch - px0;
  if (fy == 1) {
    const int32_t cur = r0[cx], oth = r0[cxn];
    return (x & 1) ? (3 * cur + oth + 2) >> 2 : (3 * cur + oth + 1) >> 2;
  }
  int32_t cyn = (y & 1) ? cy + 1 : cy - 1;
  cyn = cyn < 0 ? 0 : (cyn > dh - 1 ? dh - 1 : cyn);
  const uint8_t* r1 = plane + (cyn - py0) * pitch - px0;
  const int32_t thiscol = 3 * r0[cx] + r1[cx], othcol = 3 * r0[cxn] + r1[cxn];
  return (x & 1) ? (3 * thiscol + othcol + 7) >> 4 : (3 * thiscol + othcol + 8) >> 4;
}

// ---- the write pass's sink: DC differences of every block up to the last one needed, AC coefficients of the
// blocks whose MCU lies in the window rectangle -------------------------------------------------------------------
struct JWriteSink {
  const hcir_jpeg_header* h;
  const JWin* w;
  int16_t* dcdiff;  // [last_block + 1] in scan order
  int16_t* coefs;   // [wblocks][64] natural order, zero-filled by the caller
  uint32_t b;       // scan-order index of the block being decoded
  uint32_t last_block;
  int32_t slot;     // its place in the window buffer, -1 outside
  int32_t blk, mx, my;  // block inside the MCU, MCU column / row: stepped, not divided, from block to block
  JHD void place() {
    const int32_t ry = my - w->my0, rx = mx - w->mx0;
    slot = (b <= last_block && ry >= 0 && ry < w->nmy && rx >= 0 && rx < w->nmx)
               ? (ry * w->nmx + rx) * h->blocks_per_mcu + blk : -1;
  }
  JHD void begin(uint32_t b0) {
    b = b0;
    const uint32_t mcu = b0 / (uint32_t)h->blocks_per_mcu;
    blk = (int32_t)(b0 - mcu * (uint32_t)h->blocks_per_mcu);
    my = (int32_t)(mcu / (uint32_t)h->mcus_x);
    mx = (int32_t)(mcu - (uint32_t)my * (uint32_t)h->mcus_x);
    place();
  }
  JHD void coef(uint32_t z, uint32_t ssss, uint32_t bits, uint32_t nb) {
    const int32_t v = ssss ? jpeg_extend((bits << nb) >> (32 - ssss), ssss) : 0;
    if (z == 0) {
      if (b <= last_block) dcdiff[b] = (int16_t)v;
    } else if (slot >= 0 && ssss) {  // ssss == 0 behind the DC symbol is ZRL: nothing to store
      coefs[slot * 64 + jpeg_natural((int)z)] = (int16_t)v;
    }
  }
  JHD void block_done() {
    ++b;
    if (++blk == h->blocks_per_mcu) {
      blk = 0;
      if (++mx == h->mcus_x) {
        mx = 0;
        ++my;
      }
    }
    place();
  }
  JHD bool finished() const { return b > last_block; }
};

// one block: dequantise, two islow passes, range limit; out = 8 rows of 8 samples at `pitch`
JHD void jpeg_idct_block(const int16_t* coef, const uint16_t* q, uint8_t* out, int32_t pitch) {
  int32_t ws[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) ws[i] = (int32_t)coef[i] * (int32_t)q[i];
#pragma unroll
  for (int c = 0; c < 8; ++c) jpeg_idct8(ws + c, 8, ws + c, 8, 13 - 2);       // columns, keep PASS1_BITS
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    int32_t px[8];
    jpeg_idct8(ws + 8 * r, 1, px, 1, 13 + 2 + 3);                              // rows, /8 and drop PASS1_BITS
#pragma unroll
    for (int c = 0; c < 8; ++c) out[r * pitch + c] = jpeg_range_limit(px[c]);
  }
}

// geometry of component ci's window plane
struct JPlane {
  int32_t pitch, rows;  // samples
  int32_t px0, py0;     // origin in the component's own sample grid
  int32_t dw, dh;       // real size of the component (neighbour clamp)
  int32_t fx, fy;       // upsampling factors to full resolution
};
JHD void jpeg_plane(const hcir_jpeg_header& h, const JWin& w, int ci, JPlane& p) {
  p.pitch = w.nmx * h.hs[ci] * 8;
  p.rows = w.nmy * h.vs[ci] * 8;
  p.px0 = w.mx0 * h.hs[ci] * 8;
  p.py0 = w.my0 * h.vs[ci] * 8;
  p.dw = (h.width * h.hs[ci] + h.hmax - 1) / h.hmax;
  p.dh = (h.height * h.vs[ci] + h.vmax - 1) / h.vmax;
  p.fx = h.hmax / h.hs[ci];
  p.fy = h.vmax / h.vs[ci];
}
// where block `slot` of the coefficient buffer lands: component and top-left sample inside that component's plane
JHD void jpeg_block_place(const hcir_jpeg_header& h, const JWin& w, int32_t slot, int& ci, int32_t& sx, int32_t& sy) {
  const int32_t wm = slot / h.blocks_per_mcu, blk = slot - wm * h.blocks_per_mcu;
  const int32_t ry = wm / w.nmx, rx = wm - ry * w.nmx;
  ci = h.blk_comp[blk];
  int32_t first = 0;  // index of the component's first block inside the MCU
  for (int c = 0; c < ci; ++c) first += h.hs[c] * h.vs[c];
  const int32_t k = blk - first, by = k / h.hs[ci], bx = k - by * h.hs[ci];
  sx = (rx * h.hs[ci] + bx) * 8;
  sy = (ry * h.vs[ci] + by) * 8;
}
