// CPU fuzz of the PNG stager (csrc/png_stage.h: chunk walk, CRC-32, staging copy), built with
// -fsanitize=address,undefined by tests/test_png_host.py: mutated and truncated files must be parsed or rejected without
// a memory error, and whatever is accepted must stage inside its bound.
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../hair-centric-image-retrieval_amd/csrc/png_stage.h"

static uint64_t rng_state = 88172645463325252ull;
static uint32_t rnd() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return (uint32_t)(rng_state >> 11);
}

extern "C" int png_stage_fuzz(const uint8_t* file, size_t n, int iterations, int64_t* accepted) {
  std::vector<uint8_t> f;
  png_host::Idat* id = new png_host::Idat;
  *accepted = 0;
  for (int it = 0; it < iterations; ++it) {
    f.assign(file, file + n);
    const int kind = rnd() % 5;
    if (kind == 0) {
      f.resize(rnd() % (n + 1));  // truncated anywhere
    } else if (kind == 1) {
      for (int k = 0; k < 4; ++k) f[rnd() % n] = (uint8_t)rnd();
    } else if (kind == 2 && n > 40) {  // a chunk length field damaged
      const size_t at = 8 + rnd() % 32;
      f[at] = (uint8_t)rnd();
    } else if (kind == 3) {
      const size_t at = rnd() % n, len = rnd() % 64;
      f.erase(f.begin() + at, f.begin() + (at + len < n ? at + len : n));
    } else {
      const size_t at = rnd() % n;
      f.insert(f.begin() + at, (size_t)(rnd() % 64), (uint8_t)rnd());
    }
    hcir_png_header h;
    for (int flags = 0; flags < 2; ++flags) {
      const int rc = png_host::parse(f.data(), f.size(), flags, &h, id);
      if (rc != HCIR_OK) continue;
      ++*accepted;
      std::vector<uint8_t> blob(png_host::stage_bound(*id));
      size_t used = 0;
      if (png_host::stage(f.data(), *id, blob.data(), &used) != HCIR_OK || used != blob.size()) return 1;
      png_host::Win w;
      png_host::window(h.width, h.height, 224, 224, w);
      if (w.x0 < 0 || w.y0 < 0 || w.x1 > h.width || w.y1 > h.height) return 2;
    }
  }
  delete id;
  return 0;
}
