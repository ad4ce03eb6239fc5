// jpeg_stage.h — HOST side of the JPEG path: marker walk + staging copy (hcir_jpeg_stage).  Plain C++ (no HIP),
// so that the CPU tests compile the very same parser (tests/jpeg_emul.cpp).
//
// What the reference's decoders do here is libjpeg's jdmarker.c (read_markers, get_sof, get_dht, get_dqt,
// get_dri, get_sos) and jdhuff.c's jpeg_make_d_derived_tbl (here: a 10-bit lookahead table and the canonical ranges); behind HP/utils/dataloader.py:28-31
// (torchvision.io.decode_image) and src/models/hair_encoder.py:108 (PIL).  The staging copy replaces the
// byte-at-a-time unstuffing of jdhuff.c's fill_bit_buffer: one pass that drops the zero after every FF and the
// RSTn markers, records where every restart segment starts, and packs the bytes into 32-bit words whose bit 31
// is the first stream bit (the device reads two words and shifts).
#pragma once
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../include/hcir.h"

namespace jpeg_host {

inline uint32_t be16(const uint8_t* p) { return ((uint32_t)p[0] << 8) | p[1]; }

static const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// packed lookup entry of a symbol (include/hcir.h): consumed bits | zigzag advance << 5 | code length << 10
inline uint16_t pack_entry(int is_ac, int len, uint8_t sym) {
  const int ssss = sym & 15, r = sym >> 4;
  const int adv = !is_ac ? 1 : (ssss ? r + 1 : (r == 15 ? 16 : 0));
  return (uint16_t)((len + ssss) | (adv << 5) | (len << 10));
}

// jpeg_make_d_derived_tbl: canonical codes from BITS / HUFFVAL; HCIR_JPEG_LOOK_BITS-bit lookahead table
inline int derive_table(int is_ac, const uint8_t bits[16], const uint8_t* vals, int nvals, hcir_jpeg_hufftab* t) {
  memset(t, 0, sizeof(*t));
  t->is_ac = (uint32_t)is_ac;
  uint8_t size[257];
  uint32_t code_of[257];
  int p = 0;
  for (int l = 1; l <= 16; ++l)
    for (int i = 0; i < bits[l - 1]; ++i) {
      if (p >= 256) return HCIR_ERR_INVALID;
      size[p++] = (uint8_t)l;
    }
  if (p != nvals) return HCIR_ERR_INVALID;
  if (!is_ac)
    for (int i = 0; i < nvals; ++i)
      if (vals[i] > 15) return HCIR_ERR_INVALID;  // libjpeg: JERR_BAD_HUFF_TABLE (a DC category above 15)
  uint32_t code = 0;
  int si = p ? size[0] : 0, k = 0;
  while (k < p) {
    while (k < p && size[k] == si) code_of[k++] = code++;
    if (code > (1u << si)) return HCIR_ERR_INVALID;  // more codes of this length than fit
    code <<= 1;
    ++si;
  }
  // canonical codes of length l fill [first_l, limit_l) of the left-aligned 16-bit prefix space, lengths ascending
  k = 0;
  uint32_t lim = 0;
  t->limit[0] = 0;
  for (int l = 1; l <= 16; ++l) {
    if (bits[l - 1]) {
      t->valoff[l] = k - (int32_t)code_of[k];
      k += bits[l - 1];
      lim = (code_of[k - 1] + 1) << (16 - l);
    }
    t->limit[l] = lim;
  }
  t->limit[17] = 0x10000;
  k = 0;
  for (int l = 1; l <= 16; ++l)
    for (int i = 0; i < bits[l - 1]; ++i, ++k) {
      if (l > HCIR_JPEG_LOOK_BITS) continue;
      const uint32_t first = code_of[k] << (HCIR_JPEG_LOOK_BITS - l);
      for (uint32_t j = 0; j < (1u << (HCIR_JPEG_LOOK_BITS - l)); ++j) t->lut.look[first + j] = pack_entry(is_ac, l, vals[k]);
    }
  // second level: every longer code lies in [limit[LOOK_BITS], 65536) of the 16-bit prefix space
  t->lut.base2 = t->limit[HCIR_JPEG_LOOK_BITS];
  t->lut.use2 = t->lut.base2 >= 0x10000u - HCIR_JPEG_LOOK2 ? 1u : 0u;  // look2 is indexed by the prefix's low bits
  if (t->lut.use2) {
    k = 0;
    for (int l = 1; l <= 16; ++l)
      for (int i = 0; i < bits[l - 1]; ++i, ++k) {
        if (l <= HCIR_JPEG_LOOK_BITS) continue;
        const uint32_t first = (code_of[k] << (16 - l)) - (0x10000u - HCIR_JPEG_LOOK2);
        for (uint32_t j = 0; j < (1u << (16 - l)); ++j) t->lut.look2[first + j] = pack_entry(is_ac, l, vals[k]);
      }
  }
  memcpy(t->vals, vals, (size_t)nvals);
  return HCIR_OK;
}

struct Scan {
  const uint8_t* data;  // first entropy-coded byte
  size_t nbytes;        // upper bound (the rest of the file); the segment ends at the first marker that is neither RSTn nor a stuffed FF
};

// Marker walk.  Fills everything of *h except the stream fields; *scan = the entropy-coded bytes.
inline int parse(const uint8_t* f, size_t n, hcir_jpeg_header* h, Scan* scan) {
  if (!f || n < 4 || f[0] != 0xFF || f[1] != 0xD8) return HCIR_ERR_INVALID;
  memset(h, 0, sizeof(*h));
  uint16_t qt[4][64];
  bool have_qt[4] = {false, false, false, false}, have_ht[4] = {false, false, false, false}, have_sof = false;
  int comp_id[3] = {0, 0, 0}, comp_tq[3] = {0, 0, 0};
  bool saw_jfif = false, saw_adobe = false;
  int adobe_transform = 0;
  size_t i = 2;
  for (;;) {
    if (i + 4 > n || f[i] != 0xFF) return HCIR_ERR_INVALID;
    while (i + 1 < n && f[i + 1] == 0xFF) ++i;  // fill bytes
    if (i + 4 > n) return HCIR_ERR_INVALID;
    const uint8_t m = f[i + 1];
    i += 2;
    if (m == 0xD9 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) return HCIR_ERR_INVALID;  // EOI / RST / TEM before SOS
    const size_t len = be16(f + i);
    if (len < 2 || i + len > n) return HCIR_ERR_INVALID;
    const uint8_t* s = f + i + 2;
    const size_t sl = len - 2;
    if (m == 0xE0 && sl >= 14 && !memcmp(s, "JFIF\0", 5)) {
      saw_jfif = true;
    } else if (m == 0xEE && sl >= 12 && !memcmp(s, "Adobe", 5)) {
      saw_adobe = true;
      adobe_transform = s[11];
    } else if (m == 0xDB) {
      size_t j = 0;
      while (j < sl) {
        const int pq = s[j] >> 4, tq = s[j] & 15;
        if (tq > 3 || pq > 1) return HCIR_ERR_INVALID;
        if (j + 1 + (pq ? 128 : 64) > sl) return HCIR_ERR_INVALID;
        for (int k = 0; k < 64; ++k)
          qt[tq][kZigzag[k]] = pq ? (uint16_t)be16(s + j + 1 + 2 * k) : s[j + 1 + k];
        have_qt[tq] = true;
        j += 1 + (pq ? 128 : 64);
      }
    } else if (m == 0xC4) {
      size_t j = 0;
      while (j < sl) {
        if (j + 17 > sl) return HCIR_ERR_INVALID;
        const int tc = s[j] >> 4, th = s[j] & 15;
        int cnt = 0;
        for (int k = 0; k < 16; ++k) cnt += s[j + 1 + k];
        if (tc > 1 || cnt > 256 || j + 17 + cnt > sl) return HCIR_ERR_INVALID;
        if (th > 1) return HCIR_ERR_UNSUPPORTED;  // baseline allows table ids 0 and 1
        const int rc = derive_table(tc, s + j + 1, s + j + 17, cnt, &h->huff[tc * 2 + th]);
        if (rc != HCIR_OK) return rc;
        have_ht[tc * 2 + th] = true;
        j += 17 + cnt;
      }
    } else if (m == 0xC0 || m == 0xC1) {
      if (sl < 6 || have_sof) return HCIR_ERR_INVALID;
      if (s[0] != 8) return HCIR_ERR_UNSUPPORTED;
      h->height = (int32_t)be16(s + 1);
      h->width = (int32_t)be16(s + 3);
      h->ncomp = s[5];
      if (h->height == 0 || h->width == 0) return HCIR_ERR_UNSUPPORTED;  // DNL-defined height
      if (h->ncomp != 1 && h->ncomp != 3) return HCIR_ERR_UNSUPPORTED;
      if (sl < (size_t)(6 + 3 * h->ncomp)) return HCIR_ERR_INVALID;
      for (int c = 0; c < h->ncomp; ++c) {
        comp_id[c] = s[6 + 3 * c];
        h->hs[c] = s[7 + 3 * c] >> 4;
        h->vs[c] = s[7 + 3 * c] & 15;
        comp_tq[c] = s[8 + 3 * c];
        if (comp_tq[c] > 3) return HCIR_ERR_INVALID;
      }
      have_sof = true;
    } else if (m == 0xC2 || m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) {
      return HCIR_ERR_UNSUPPORTED;  // progressive, lossless, arithmetic, hierarchical
    } else if (m == 0xCC) {
      return HCIR_ERR_UNSUPPORTED;  // DAC: arithmetic conditioning
    } else if (m == 0xDD) {
      if (sl < 2) return HCIR_ERR_INVALID;
      h->restart_interval = (int32_t)be16(s);
    } else if (m == 0xDA) {
      if (!have_sof || sl < 1) return HCIR_ERR_INVALID;
      const int ns = s[0];
      if (ns != h->ncomp) return HCIR_ERR_UNSUPPORTED;  // non-interleaved / multi-scan
      if (sl < (size_t)(1 + 2 * ns + 3)) return HCIR_ERR_INVALID;
      for (int k = 0; k < ns; ++k) {
        if (s[1 + 2 * k] != comp_id[k]) return HCIR_ERR_UNSUPPORTED;  // components out of frame order
        const int td = s[2 + 2 * k] >> 4, ta = s[2 + 2 * k] & 15;
        if (td > 1 || ta > 1) return HCIR_ERR_UNSUPPORTED;
        if (!have_ht[td] || !have_ht[2 + ta]) return HCIR_ERR_INVALID;
        h->dc_tab[k] = (uint8_t)td;
        h->ac_tab[k] = (uint8_t)(2 + ta);
      }
      if (s[1 + 2 * ns] != 0 || s[2 + 2 * ns] != 63 || s[3 + 2 * ns] != 0) return HCIR_ERR_UNSUPPORTED;
      i += len;
      break;
    }
    i += len;
  }
  // colour space of a three-component file as libjpeg decides it (jdapimin.c default_decompress_parms): JFIF -> YCbCr;
  // else Adobe transform 0 -> RGB; else component ids 'R','G','B' -> RGB.  RGB files are not converted by libjpeg:
  // the device path (which always converts) leaves them to the host decoder
  if (h->ncomp == 3 && !saw_jfif) {
    if (saw_adobe ? adobe_transform == 0 : (comp_id[0] == 'R' && comp_id[1] == 'G' && comp_id[2] == 'B'))
      return HCIR_ERR_UNSUPPORTED;
  }
  // frame geometry
  if (h->ncomp == 1) {
    h->hs[0] = h->vs[0] = 1;  // a one-component scan is not interleaved: its MCU is one block (T.81 A.2.2)
  } else {
    const bool ok = h->hs[1] == 1 && h->vs[1] == 1 && h->hs[2] == 1 && h->vs[2] == 1 &&
                    ((h->hs[0] == 1 && h->vs[0] == 1) || (h->hs[0] == 2 && h->vs[0] == 1) ||
                     (h->hs[0] == 2 && h->vs[0] == 2));
    if (!ok) return HCIR_ERR_UNSUPPORTED;
  }
  h->hmax = h->hs[0];
  h->vmax = h->vs[0];
  h->mcus_x = (h->width + 8 * h->hmax - 1) / (8 * h->hmax);
  h->mcus_y = (h->height + 8 * h->vmax - 1) / (8 * h->vmax);
  int nb = 0;
  for (int c = 0; c < h->ncomp; ++c) {
    if (!have_qt[comp_tq[c]]) return HCIR_ERR_INVALID;
    memcpy(h->quant[c], qt[comp_tq[c]], sizeof(qt[0]));
    for (int k = 0; k < h->hs[c] * h->vs[c]; ++k) h->blk_comp[nb++] = (uint8_t)c;
  }
  h->blocks_per_mcu = nb;
  const int64_t mcus = (int64_t)h->mcus_x * h->mcus_y;
  if (mcus * nb >= (int64_t(1) << 31)) return HCIR_ERR_UNSUPPORTED;
  h->nsegments = h->restart_interval > 0 ? (int32_t)((mcus + h->restart_interval - 1) / h->restart_interval) : 1;
  // entropy-coded segment: from here to the first FF that is followed by neither 00, RSTn nor another FF.  The
  // parse does not look for that end (a pass over the whole file); stage() stops there, nbytes is the upper bound.
  scan->data = f + i;
  scan->nbytes = n - i;
  return HCIR_OK;
}

inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

// upper bound of the staged size: every scan byte kept, + 2 pad words, + the segment table
inline size_t stage_bound(const hcir_jpeg_header& h, const Scan& s) {
  return align16(((s.nbytes + 3) / 4 + 4) * 4) + align16(((size_t)h.nsegments + 1) * 4);
}

// The staging copy.  dst must hold stage_bound() bytes.  Returns HCIR_OK and fills stream_bits / stream_words.
inline int stage(hcir_jpeg_header* h, const Scan& s, uint8_t* dst, size_t* used) {
  std::vector<uint32_t> seg;
  seg.reserve((size_t)h->nsegments + 1);
  const uint8_t* p = s.data;
  const uint8_t* end = s.data + s.nbytes;
  size_t nout = 0;  // bytes emitted (in stream order; the words are byte-swapped in one pass at the end)
  seg.push_back(0);
  while (p < end) {
    const uint8_t* q = (const uint8_t*)memchr(p, 0xFF, (size_t)(end - p));
    const size_t run = (size_t)((q ? q : end) - p);
    memcpy(dst + nout, p, run);
    nout += run;
    if (!q || q + 1 >= end) break;  // a lone FF at the very end belongs to the next marker
    const uint8_t nb = q[1];
    if (nb == 0x00) {
      dst[nout++] = 0xFF;
      p = q + 2;
    } else if (nb >= 0xD0 && nb <= 0xD7) {
      if ((int64_t)seg.size() >= h->nsegments) return HCIR_ERR_INVALID;  // more RSTn than the frame has intervals
      seg.push_back((uint32_t)(nout * 8));
      p = q + 2;
    } else if (nb == 0xFF) {
      p = q + 1;  // fill byte in front of a marker
    } else {
      break;      // a marker: the entropy-coded segment ends here
    }
  }
  if (nout * 8 >= (uint64_t(1) << 31)) return HCIR_ERR_UNSUPPORTED;
  h->stream_bits = (uint32_t)(nout * 8);
  // a stream with fewer RSTn than intervals (truncated file): the missing segments are empty
  while ((int64_t)seg.size() <= h->nsegments) seg.push_back(h->stream_bits);
  // pad with 1-bits: the last partial word plus four whole words (the reader keeps four words in registers)
  const size_t padded = (nout + 3) / 4 * 4 + 16;
  memset(dst + nout, 0xFF, padded - nout);
  nout = padded;
  uint32_t* w = reinterpret_cast<uint32_t*>(dst);
  for (size_t i = 0; i < nout / 4; ++i) w[i] = __builtin_bswap32(w[i]);  // bit 31 = first stream bit
  h->stream_words = (uint32_t)(nout / 4);
  // segment table right behind the words (16-byte aligned): the device finds it from stream_words
  memcpy(dst + align16(nout), seg.data(), seg.size() * 4);
  *used = align16(nout) + align16(seg.size() * 4);
  return HCIR_OK;
}

}  // namespace jpeg_host
