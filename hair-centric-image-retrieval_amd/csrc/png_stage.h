// png_stage.h — HOST side of the PNG path: chunk walk + staging copy (hcir_png_stage).  Plain C++ (no HIP), so
// that the CPU tests compile the very same parser (tests/png_emul.cpp).
//
// What the reference's decoders do here is libpng's png_read_info / Pillow's PngImagePlugin.PngStream (IHDR,
// PLTE, IDAT ... with a CRC-32 per chunk), behind HP/utils/dataloader.py:28-31 (torchvision.io.decode_image) and
// src/models/hair_encoder.py:108 (PIL).  The staging copy concatenates the IDAT payloads: the zlib stream the
// device inflates, little-endian bit order as stored (RFC 1951 3.1.1), so words are staged as they are.
#pragma once
#include <stdint.h>
#include <string.h>

#include "../../include/hcir.h"

#if defined(__HIPCC__)
#define PHD __host__ __device__ inline
#else
#define PHD inline
#endif

namespace png_host {

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

// CRC-32 (PNG annex D polynomial), eight table lookups per 8 bytes
struct CrcTables {
  uint32_t t[8][256];
  CrcTables() {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
      t[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int k = 1; k < 8; ++k) t[k][i] = t[0][t[k - 1][i] & 255] ^ (t[k - 1][i] >> 8);
  }
};
inline uint32_t crc32(const uint8_t* p, size_t n) {
  static const CrcTables T;
  uint32_t c = 0xffffffffu;
  while (n && ((uintptr_t)p & 7)) {
    c = T.t[0][(c ^ *p++) & 255] ^ (c >> 8);
    --n;
  }
  while (n >= 8) {
    uint64_t v;
    memcpy(&v, p, 8);
    v ^= c;
    c = T.t[7][v & 255] ^ T.t[6][(v >> 8) & 255] ^ T.t[5][(v >> 16) & 255] ^ T.t[4][(v >> 24) & 255] ^
        T.t[3][(v >> 32) & 255] ^ T.t[2][(v >> 40) & 255] ^ T.t[1][(v >> 48) & 255] ^ T.t[0][v >> 56];
    p += 8;
    n -= 8;
  }
  while (n--) c = T.t[0][(c ^ *p++) & 255] ^ (c >> 8);
  return c ^ 0xffffffffu;
}

constexpr int32_t kMaxWidth = 8192;

struct Idat {  // where the IDAT payloads lie in the file
  static const int kMax = 4096;
  uint32_t off[kMax], len[kMax];
  int n;
  size_t total;
};

PHD int bytes_per_pixel(int color_type) {
  switch (color_type) {
    case 0: return 1;
    case 2: return 3;
    case 3: return 1;
    case 4: return 2;
    case 6: return 4;
  }
  return 0;
}

// Chunk walk.  HCIR_ERR_INVALID: not a PNG / broken chunk structure / CRC mismatch; HCIR_ERR_UNSUPPORTED: a valid
// PNG outside the device subset.
inline int parse(const uint8_t* f, size_t n, int32_t flags, hcir_png_header* h, Idat* id) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 13, 10, 26, 10};
  memset(h, 0, sizeof(*h));
  id->n = 0;
  id->total = 0;
  if (!f || n < 8 + 25 || memcmp(f, sig, 8)) return HCIR_ERR_INVALID;
  size_t i = 8;
  bool ihdr = false, plte = false;
  int depth = 0, interlace = 0;
  while (i + 12 <= n) {
    const uint32_t len = be32(f + i);
    const uint8_t* type = f + i + 4;
    if ((size_t)len > n - i - 12) return HCIR_ERR_INVALID;  // truncated file
    const uint8_t* d = f + i + 8;
    if ((flags & HCIR_PNG_VERIFY_CRC) && crc32(type, (size_t)len + 4) != be32(d + len)) return HCIR_ERR_INVALID;
    if (!memcmp(type, "IHDR", 4)) {
      if (len != 13 || ihdr) return HCIR_ERR_INVALID;
      ihdr = true;
      const uint32_t w = be32(d), hh = be32(d + 4);
      depth = d[8];
      h->color_type = d[9];
      interlace = d[12];
      if (w == 0 || hh == 0 || w > 0x7fffffffu || hh > 0x7fffffffu || d[10] != 0 || d[11] != 0) return HCIR_ERR_INVALID;
      h->width = (int32_t)w;
      h->height = (int32_t)hh;
    } else if (!ihdr) {
      return HCIR_ERR_INVALID;
    } else if (!memcmp(type, "PLTE", 4)) {
      if (len == 0 || len > 768 || len % 3) return HCIR_ERR_INVALID;
      memcpy(h->palette, d, len);
      plte = true;
    } else if (!memcmp(type, "IDAT", 4)) {
      if (len) {
        if (id->n == Idat::kMax) return HCIR_ERR_UNSUPPORTED;
        id->off[id->n] = (uint32_t)(d - f);
        id->len[id->n++] = len;
        id->total += len;
      }
    } else if (!memcmp(type, "IEND", 4)) {
      break;
    }
    i += 12 + (size_t)len;
  }
  if (!ihdr || id->total == 0) return HCIR_ERR_INVALID;
  h->bpp = bytes_per_pixel(h->color_type);
  if (depth != 8 || interlace != 0 || h->bpp == 0) return HCIR_ERR_UNSUPPORTED;
  if (h->color_type == 3 && !plte) return HCIR_ERR_INVALID;
  // the device keeps byte positions in 32 bits and one row of packed pixels (4 B each) in LDS
  const uint64_t row = 1 + (uint64_t)h->width * h->bpp;
  if (h->width > kMaxWidth || row * (uint64_t)h->height >= (1ull << 31) || id->total >= (1ull << 31))
    return HCIR_ERR_UNSUPPORTED;
  h->stream_bytes = (uint32_t)id->total;
  return HCIR_OK;
}

inline size_t stage_bound(const Idat& id) { return align16(id.total) + 16; }

inline int stage(const uint8_t* f, const Idat& id, uint8_t* dst, size_t* used) {
  size_t o = 0;
  for (int k = 0; k < id.n; ++k) {
    memcpy(dst + o, f + id.off[k], id.len[k]);
    o += id.len[k];
  }
  const size_t end = stage_bound(id);
  memset(dst + o, 0, end - o);  // the bit reader may look (not consume) past the end
  *used = end;
  return HCIR_OK;
}

// CenterCrop((win_h, win_w)) of an h x w image, as torchvision pads then crops (same arithmetic as jpeg_window)
struct Win {
  int32_t x0, x1, y0, y1;  // image columns / rows the window touches [x0, x1) x [y0, y1)
  int32_t ox, oy;          // where (x0, y0) lands inside the window
};
PHD int32_t round_half_even_div2(int32_t v) {  // python round(v / 2.0)
  const int32_t q = v >> 1;
  return (v & 1) ? q + (q & 1) : q;
}
PHD void window(int32_t w, int32_t h, int32_t win_h, int32_t win_w, Win& o) {
  const int32_t pw = win_w > w ? win_w - w : 0, ph = win_h > h ? win_h - h : 0;
  const int32_t pl = pw / 2, pt = ph / 2;             // torchvision pads left/top with pad // 2
  const int32_t W = w + pw, H = h + ph;               // padded size
  const int32_t left = round_half_even_div2(W - win_w), top = round_half_even_div2(H - win_h);
  // padded column c = image column c - pl
  int32_t x0 = left - pl, x1 = left + win_w - pl, y0 = top - pt, y1 = top + win_h - pt;
  o.ox = x0 < 0 ? -x0 : 0;
  o.oy = y0 < 0 ? -y0 : 0;
  o.x0 = x0 < 0 ? 0 : x0;
  o.y0 = y0 < 0 ? 0 : y0;
  o.x1 = x1 > w ? w : x1;
  o.y1 = y1 > h ? h : y1;
}

}  // namespace png_host
