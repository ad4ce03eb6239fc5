/* oracle/png_oracle.c — TEST INFRASTRUCTURE ONLY (never linked or loaded by the product path).
 *
 * CPU restatement of what the reference's loader does to a PNG file before `knn_transform`:
 *   HairPretraining/utils/dataloader.py:28-31   read_file -> torchvision.io.decode_image(img_bytes, mode=RGB)
 *   src/models/hair_encoder.py:108,169          PIL.Image.open(path).convert('RGB')
 * Both hand the bytes to libpng / Pillow's PngImagePlugin + zlib, third-party code that is NOT under
 * /root/reference.  What is restated here is their published format: the PNG chunk walk and the five scanline
 * filters (PNG specification, 2nd ed., sections 5 and 9), the zlib wrapper (RFC 1950) and inflate (RFC 1951).
 * Written to be obviously right rather than fast, and on purpose structured differently from csrc/png.hip:
 * codes are decoded one BIT at a time against the canonical first-code / count arrays (no lookup tables),
 * output goes to one flat buffer, rows are unfiltered one byte at a time.
 *
 * PIN: tests/test_png_host.py checks png_oracle_inflate byte for byte against zlib.decompress (Python's zlib,
 * 1.2.11 in this image) and png_oracle_decode against Pillow 12.2's Image.open(...).convert("RGB") on the four
 * assets/hair_region_only files (tests/golden/png_streams.npz) and on seeded synthetic files.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { OK = 0, TRUNCATED_AT_CAP = 1, E_CORRUPT = -1, E_UNSUPPORTED = -2, E_INPUT = -3 };

typedef struct {
  const uint8_t* in;
  size_t n, byte;
  int bit; /* next bit inside in[byte], 0 = least significant */
  int over;
} bits_t;

static unsigned getbit(bits_t* s) {
  if (s->byte >= s->n) {
    s->over = 1;
    return 0;
  }
  unsigned b = (s->in[s->byte] >> s->bit) & 1u;
  if (++s->bit == 8) {
    s->bit = 0;
    ++s->byte;
  }
  return b;
}
/* RFC 1951 3.1.1: data elements other than Huffman codes are packed starting from the least significant bit */
static unsigned getbits(bits_t* s, int n) {
  unsigned v = 0;
  for (int i = 0; i < n; ++i) v |= getbit(s) << i;
  return v;
}

typedef struct {
  uint16_t count[16];
  uint16_t sym[320];
  int nsym_nonzero, maxlen;
} huff_t;

/* RFC 1951 3.2.2: canonical code from the code lengths.  Returns <0 over-subscribed, >0 incomplete, 0 complete. */
static int build(huff_t* h, const uint8_t* len, int n) {
  uint16_t offs[16];
  memset(h->count, 0, sizeof(h->count));
  for (int i = 0; i < n; ++i) h->count[len[i]]++;
  h->nsym_nonzero = n - h->count[0];
  h->maxlen = 0;
  for (int l = 1; l < 16; ++l)
    if (h->count[l]) h->maxlen = l;
  int left = 1;
  for (int l = 1; l < 16; ++l) {
    left <<= 1;
    left -= h->count[l];
    if (left < 0) return -1;
  }
  offs[1] = 0;
  for (int l = 1; l < 15; ++l) offs[l + 1] = offs[l] + h->count[l];
  for (int i = 0; i < n; ++i)
    if (len[i]) h->sym[offs[len[i]]++] = (uint16_t)i;
  return left;
}

/* Huffman codes are packed most significant bit first (RFC 1951 3.1.1): read a bit, extend the code, compare
 * with the range of codes of this length. */
static int decode(bits_t* s, const huff_t* h) {
  int code = 0, first = 0, index = 0;
  for (int l = 1; l < 16; ++l) {
    code |= (int)getbit(s);
    int cnt = h->count[l];
    if (code - cnt < first) return h->sym[index + (code - first)];
    index += cnt;
    first += cnt;
    first <<= 1;
    code <<= 1;
  }
  return -1; /* ran out of codes */
}

static const uint16_t LBASE[29] = {3,  4,  5,  6,  7,  8,  9,  10, 11,  13,  15,  17,  19,  23, 27,
                                   31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DBASE[30] = {1,   2,   3,   4,   5,   7,    9,    13,   17,   25,   33,   49,   65,    97,    129,
                                   193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

/* stats[] (optional, 16 x uint64): 0 stored blocks, 1 fixed, 2 dynamic, 3 literals, 4 matches, 5 match bytes,
 * 6 matches with dist < len, 7 dist <= 4096, 8 dist <= 8192, 9 dist <= 16384, 10 dist > 16384, 11 long litlen
 * codes (> 10 bits), 12 consumed input bytes at the point of return, 13 / 14 codes > 11 / > 12 bits. */
static int codes(bits_t* s, uint8_t* out, size_t cap, size_t* pos, const huff_t* lc, const huff_t* dc, uint64_t* st) {
  for (;;) {
    size_t b0 = s->byte * 8 + s->bit;
    int sym = decode(s, lc);
    if (s->over) return E_CORRUPT;
    if (sym < 0) return E_CORRUPT;
    if (st && (s->byte * 8 + s->bit) - b0 > 10) st[11]++;
    if (st && (s->byte * 8 + s->bit) - b0 > 11) st[13]++;
    if (st && (s->byte * 8 + s->bit) - b0 > 12) st[14]++;
    if (sym < 256) {
      if (*pos >= cap) return TRUNCATED_AT_CAP;
      out[(*pos)++] = (uint8_t)sym;
      if (st) st[3]++;
    } else if (sym == 256) {
      return OK;
    } else {
      sym -= 257;
      if (sym >= 29) return E_CORRUPT;
      unsigned len = LBASE[sym] + getbits(s, LEXT[sym]);
      int ds = decode(s, dc);
      if (ds < 0 || ds >= 30) return E_CORRUPT;
      unsigned dist = DBASE[ds] + getbits(s, DEXT[ds]);
      if (s->over) return E_CORRUPT;
      if (dist > *pos) return E_CORRUPT; /* before the start of the output: no preset dictionary in PNG */
      if (st) {
        st[4]++;
        st[5] += len;
        if (dist < len) st[6]++;
        if (dist <= 4096) st[7]++;
        else if (dist <= 8192) st[8]++;
        else if (dist <= 16384) st[9]++;
        else st[10]++;
      }
      for (unsigned i = 0; i < len; ++i) {
        if (*pos >= cap) return TRUNCATED_AT_CAP;
        out[*pos] = out[*pos - dist];
        ++*pos;
      }
    }
  }
}

static int inflate_raw(bits_t* s, uint8_t* out, size_t cap, size_t* pos, uint64_t* st) {
  huff_t lc, dc;
  uint8_t len[320];
  for (;;) {
    unsigned last = getbit(s), type = getbits(s, 2);
    if (s->over) return E_CORRUPT;
    int rc;
    if (type == 0) {
      if (s->bit) {
        s->bit = 0;
        ++s->byte;
      }
      if (s->byte + 4 > s->n) return E_CORRUPT;
      unsigned l = s->in[s->byte] | (s->in[s->byte + 1] << 8), nl = s->in[s->byte + 2] | (s->in[s->byte + 3] << 8);
      if (l != (~nl & 0xffff)) return E_CORRUPT;
      s->byte += 4;
      if (s->byte + l > s->n) return E_CORRUPT;
      if (st) st[0]++;
      for (unsigned i = 0; i < l; ++i) {
        if (*pos >= cap) return TRUNCATED_AT_CAP;
        out[(*pos)++] = s->in[s->byte++];
      }
      rc = OK;
    } else if (type == 1) {
      int i = 0;
      for (; i < 144; ++i) len[i] = 8;
      for (; i < 256; ++i) len[i] = 9;
      for (; i < 280; ++i) len[i] = 7;
      for (; i < 288; ++i) len[i] = 8;
      build(&lc, len, 288);
      for (i = 0; i < 30; ++i) len[i] = 5;
      build(&dc, len, 30);
      if (st) st[1]++;
      rc = codes(s, out, cap, pos, &lc, &dc, st);
    } else if (type == 2) {
      static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
      int nlen = (int)getbits(s, 5) + 257, ndist = (int)getbits(s, 5) + 1, ncode = (int)getbits(s, 4) + 4;
      if (nlen > 286 || ndist > 30) return E_CORRUPT; /* zlib: "too many length or distance symbols" */
      int i = 0;
      for (; i < ncode; ++i) len[order[i]] = (uint8_t)getbits(s, 3);
      for (; i < 19; ++i) len[order[i]] = 0;
      if (build(&lc, len, 19) != 0) return E_CORRUPT; /* zlib requires a complete code-length code */
      i = 0;
      while (i < nlen + ndist) {
        int sym = decode(s, &lc);
        if (sym < 0 || s->over) return E_CORRUPT;
        if (sym < 16) {
          len[i++] = (uint8_t)sym;
        } else {
          int prev = 0, rep;
          if (sym == 16) {
            if (i == 0) return E_CORRUPT;
            prev = len[i - 1];
            rep = 3 + (int)getbits(s, 2);
          } else if (sym == 17) {
            rep = 3 + (int)getbits(s, 3);
          } else {
            rep = 11 + (int)getbits(s, 7);
          }
          if (i + rep > nlen + ndist) return E_CORRUPT;
          while (rep--) len[i++] = (uint8_t)prev;
        }
      }
      if (s->over) return E_CORRUPT;
      if (len[256] == 0) return E_CORRUPT; /* no end-of-block code */
      int e = build(&lc, len, nlen);
      if (e < 0 || (e > 0 && lc.maxlen != 1)) return E_CORRUPT; /* zlib inflate_table: incomplete only with one 1-bit code */
      e = build(&dc, len + nlen, ndist);
      if (e < 0 || (e > 0 && dc.maxlen > 1)) return E_CORRUPT;
      if (st) st[2]++;
      rc = codes(s, out, cap, pos, &lc, &dc, st);
    } else {
      return E_CORRUPT;
    }
    if (rc != OK) return rc;
    if (last) return OK;
  }
}

/* zlib stream (RFC 1950) -> out.  Returns OK after the final block with the Adler-32 verified; TRUNCATED_AT_CAP
 * when `cap` bytes were produced before the end (what a window decoder asks for); <0 on a corrupt stream. */
int png_oracle_inflate(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* produced, uint64_t* stats) {
  *produced = 0;
  if (n < 2) return E_CORRUPT;
  if ((in[0] & 15) != 8 || (in[0] >> 4) > 7 || ((in[0] << 8) | in[1]) % 31 != 0 || (in[1] & 0x20)) return E_CORRUPT;
  bits_t s = {in, n, 2, 0, 0};
  int rc = inflate_raw(&s, out, cap, produced, stats);
  if (stats) stats[12] = s.byte;
  if (rc != OK) return rc;
  if (s.bit) {
    s.bit = 0;
    ++s.byte;
  }
  if (s.byte + 4 > n) return E_CORRUPT;
  uint32_t a = 1, b = 0;
  for (size_t i = 0; i < *produced; ++i) {
    a = (a + out[i]) % 65521u;
    b = (b + a) % 65521u;
  }
  uint32_t want = ((uint32_t)in[s.byte] << 24) | ((uint32_t)in[s.byte + 1] << 16) | ((uint32_t)in[s.byte + 2] << 8) | in[s.byte + 3];
  return ((b << 16) | a) == want ? OK : E_CORRUPT;
}

static uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

static uint32_t crc32_bitwise(const uint8_t* p, size_t n) { /* PNG spec annex D, without the table */
  uint32_t c = 0xffffffffu;
  for (size_t i = 0; i < n; ++i) {
    c ^= p[i];
    for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
  }
  return c ^ 0xffffffffu;
}

/* Chunk walk.  info[0..4] = width, height, bit depth, colour type, interlace; idat (caller's buffer of at least
 * `n` bytes) receives the concatenated IDAT payloads; palette[768].  check_crc: verify every chunk's CRC. */
int png_oracle_parse(const uint8_t* f, size_t n, int32_t* info, uint8_t* idat, size_t* idat_len, uint8_t* palette,
                     int check_crc) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 13, 10, 26, 10};
  if (n < 8 || memcmp(f, sig, 8)) return E_INPUT;
  size_t i = 8;
  int seen_ihdr = 0, seen_iend = 0;
  *idat_len = 0;
  memset(palette, 0, 768);
  while (i + 12 <= n) {
    uint32_t len = be32(f + i);
    const uint8_t* type = f + i + 4;
    if (len > n - i - 12) return E_CORRUPT;
    if (check_crc && crc32_bitwise(f + i + 4, len + 4) != be32(f + i + 8 + len)) return E_CORRUPT;
    const uint8_t* d = f + i + 8;
    if (!memcmp(type, "IHDR", 4)) {
      if (len != 13) return E_CORRUPT;
      info[0] = (int32_t)be32(d);
      info[1] = (int32_t)be32(d + 4);
      info[2] = d[8];
      info[3] = d[9];
      info[4] = d[12];
      if (d[10] != 0 || d[11] != 0) return E_CORRUPT;
      seen_ihdr = 1;
    } else if (!seen_ihdr) {
      return E_CORRUPT;
    } else if (!memcmp(type, "PLTE", 4)) {
      if (len > 768 || len % 3) return E_CORRUPT;
      memcpy(palette, d, len);
    } else if (!memcmp(type, "IDAT", 4)) {
      memcpy(idat + *idat_len, d, len);
      *idat_len += len;
    } else if (!memcmp(type, "IEND", 4)) {
      seen_iend = 1;
      break;
    }
    i += 12 + (size_t)len;
  }
  (void)seen_iend; /* Pillow decodes the pixels of a file without IEND as long as the IDAT stream is whole */
  return seen_ihdr && *idat_len ? OK : E_CORRUPT;
}

static int paeth(int a, int b, int c) { /* PNG spec 9.4 */
  int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
  if (pa <= pb && pa <= pc) return a;
  if (pb <= pc) return b;
  return c;
}

/* Filtered scanlines -> RGB8 (what Image.convert("RGB") / decode_image(mode=RGB) return): grey is replicated, alpha is
 * dropped, palette indices go through PLTE.  8-bit samples, no interlace.  raw: rows x (1 + w*bpp) bytes, IN PLACE. */
int png_oracle_unfilter_rgb(uint8_t* raw, int w, int rows, int color_type, const uint8_t* palette, uint8_t* rgb) {
  int bpp = color_type == 2 ? 3 : color_type == 6 ? 4 : color_type == 4 ? 2 : 1;
  size_t stride = 1 + (size_t)w * bpp;
  for (int y = 0; y < rows; ++y) {
    uint8_t* cur = raw + y * stride + 1;
    const uint8_t* up = y ? raw + (y - 1) * stride + 1 : NULL;
    int ft = cur[-1];
    if (ft > 4) return E_CORRUPT;
    for (size_t i = 0; i < (size_t)w * bpp; ++i) {
      int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)bpp) ? up[i - bpp] : 0;
      int pred = ft == 0 ? 0 : ft == 1 ? a : ft == 2 ? b : ft == 3 ? (a + b) >> 1 : paeth(a, b, c);
      cur[i] = (uint8_t)(cur[i] + pred);
    }
    for (int x = 0; x < w; ++x) {
      uint8_t* o = rgb + ((size_t)y * w + x) * 3;
      const uint8_t* p = cur + (size_t)x * bpp;
      if (color_type == 2 || color_type == 6) {
        o[0] = p[0], o[1] = p[1], o[2] = p[2];
      } else if (color_type == 3) {
        o[0] = palette[p[0] * 3], o[1] = palette[p[0] * 3 + 1], o[2] = palette[p[0] * 3 + 2];
      } else {
        o[0] = o[1] = o[2] = p[0];
      }
    }
  }
  return OK;
}
