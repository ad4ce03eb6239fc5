// jpeg.hip — baseline-JPEG decode of CenterCrop windows on the device (include/hcir.h "Baseline-JPEG
// decode").  Replaces torchvision.io.decode_image / PIL in front of knn_transform
// (HP/utils/dataloader.py:28-31, HP/utils/transform.py:11, src/models/hair_encoder.py:108,169).
//
// Three kernels per batch, all integer work, none of it shaped like a GEMM:
//
//   jpeg_huffman_kernel   one workgroup per image, one thread per fixed-size SUBSEQUENCE of the staged stream.
//       phase 0  every thread decodes its subsequence from (block 0 of an MCU, DC next) — right for thread 0 and at
//                restart boundaries, a guess elsewhere; Huffman streams re-synchronise by themselves, so the state
//                (bit, block-in-MCU, zigzag index) at the subsequence's end is usually already the true one.
//       sync     lock step: in iteration k thread t decodes subsequence t + k from ITS chain and compares its end state
//                with the one recorded there; equal => the two chains have merged and t stops, else it overwrites
//                the record (and always the subsequence's block count) and goes on.  Thread 0's chain is the
//                sequential decoder, so by induction every record ends up being the sequential decoder's state
//                (Weissenberger & Schmidt, "Massively Parallel Huffman Decoding on GPUs", ICPP'18; the JPEG state
//                triple from their 2021 follow-up).  Worst case (never re-synchronising) degenerates to a sequential
//                decode by thread 0; the sample images need 6–9 iterations of 512–864-bit subsequences.
//       write    exclusive scan of the block counts = first block of every subsequence; each thread re-decodes its
//                subsequence from the verified state and stores DC differences (all blocks up to the last MCU the
//                window needs) and AC coefficients (blocks of the window's MCU rectangle only).
//       DC       segmented (restart intervals) prefix sum of the differences per component, written into the
//                coefficient buffer.
//   jpeg_idct_kernel      thread per 8x8 block: dequantise, jidctint.c islow, range limit -> component planes.
//   jpeg_color_kernel     thread per output pixel: h2v1 / h2v2 fancy upsampling taps + YCbCr -> RGB, zero outside
//                         the image (CenterCrop pads).
//
// Bounds: HBM / latency.  Algorithmic bytes per image = staged stream (66–110 KB for the 1024^2 samples) in, 3 * 224^2
// out; the coefficient / plane round trip of the window (0.3 MB) stays in L2 / Infinity Cache.
#include <atomic>
#include <thread>
#include <vector>

#include "common.h"
#include "jpeg_core.h"
#include "jpeg_stage.h"

namespace {

// what the IDCT / colour kernels need of an image, computed once by its Huffman workgroup
struct JGeom {
  JWin w;
  JPlane pl[3];
  int32_t plane_off[3];  // byte offset of each component's plane inside the image's plane buffer
  int32_t ncomp, width, height;
  int32_t rounds;        // hand-over rounds its synchronisation took (diagnostic)
  uint32_t stamp[6];     // -DHCIR_JPEG_STAMPS: s_memrealtime (100 MHz) at the phase boundaries of the Huffman kernel
};

struct JpegBatch {
  const uint8_t* blob;  // device: headers, then streams
  int64_t b;
  int32_t win_h, win_w;
  uint8_t* out;
  int32_t* status;
  int16_t* coef;    // [b][coef_stride]
  int16_t* dcdiff;  // [b][dc_stride]
  uint8_t* planes;  // [b][plane_stride]
  JGeom* geom;      // [b]
  uint32_t* ilv;    // [b][ilv_stride] streams re-laid interleaved by subsequence
  int64_t coef_stride, dc_stride, plane_stride, ilv_stride;  // elements
  int32_t huff_threads;
};

__device__ __forceinline__ const hcir_jpeg_header* hdr_of(const JpegBatch& a, int64_t img) {
  return reinterpret_cast<const hcir_jpeg_header*>(a.blob) + img;
}

// inclusive scan of one uint32 per thread over the workgroup (blockDim.x a multiple of 64, <= 1024); the
// workgroup's total is left in wave_tot[16].  Two barriers.
__device__ __forceinline__ uint32_t block_inclusive_scan(uint32_t v, uint32_t* wave_tot /* LDS [17] */) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t u = __shfl_up(v, o);
    if (lane >= o) v += u;
  }
  if (lane == 63) wave_tot[wv] = v;
  __syncthreads();
  uint32_t add = 0;
  for (int i = 0; i < wv; ++i) add += wave_tot[i];
  if (threadIdx.x == blockDim.x - 1) wave_tot[16] = v + add;
  __syncthreads();
  return v + add;
}

struct JChain {  // a live chain between two rounds of the synchronisation
  uint32_t p, cz, m, seg;
};

struct DcCarry {
  int32_t s[3];
  int32_t flag;  // a restart boundary lies inside (the sums restart there)
};

// Re-lays an image's stream for the Huffman kernel: linear word g -> ilv[(g % wps) * nx + g / wps] (jpeg_word_addr),
// through 32 x 32-word LDS tiles so that both the read (128 B per subsequence row) and the write (128 B = the same
// word of 32 neighbouring subsequences) are whole lines.  Words behind the staged stream read as 1-bits.
__global__ __launch_bounds__(256) void jpeg_interleave_kernel(JpegBatch a) {
  __shared__ uint32_t tile[32][33];
  const int64_t img = blockIdx.y;
  const hcir_jpeg_header* h = hdr_of(a, img);
  if (h->width == 0 || h->stream_bits == 0) return;
  JSubseq q;
  jpeg_subseq(h->stream_bits, h->stream_words, (uint32_t)a.huff_threads, q);
  const uint32_t tiles_j = (q.wps + 31) / 32, tiles_t = (q.nx + 31) / 32;
  const uint32_t* src = reinterpret_cast<const uint32_t*>(a.blob + h->stage_offset);
  uint32_t* dst = a.ilv + img * a.ilv_stride;
  const uint32_t x = threadIdx.x & 31, y0 = threadIdx.x >> 5;
  for (uint32_t tl = blockIdx.x; tl < tiles_j * tiles_t; tl += gridDim.x) {
    const uint32_t t0 = (tl / tiles_j) * 32, j0 = (tl % tiles_j) * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t r = y0 + 8 * k, t = t0 + r, j = j0 + x;
      const uint64_t g = (uint64_t)t * q.wps + j;
      tile[r][x] = (t < q.nx && j < q.wps && g < h->stream_words) ? src[g] : 0xFFFFFFFFu;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t r = y0 + 8 * k, j = j0 + r, t = t0 + x;
      if (j < q.wps && t < q.nx) dst[(uint64_t)j * q.nx + t] = tile[x][r];
    }
    __syncthreads();
  }
}

// kT threads = subsequences per image.  1024: two workgroups per CU, shortest subsequences (small batches: latency);
// 512: four per CU (33 KB of LDS each), so a batch of ~1000 images is resident at once and one image's serial tail
// (the few chains that take many rounds to merge) runs under the others' bulk phases.
template <int kT, bool kFast>
__global__ __launch_bounds__(kT, 8) void jpeg_huffman_kernel(JpegBatch a) {
  // LDS: the header up to its Huffman tables (geometry, selectors, quantisation) and the LOOKUP part of the four
  // tables; the canonical ranges behind them stay in global memory (only tables whose long codes overflow the
  // second lookup level ever read them)
  constexpr int kHeadWords = (int)(offsetof(hcir_jpeg_header, huff) / 4);
  constexpr int kLutWords = (int)(sizeof(hcir_jpeg_lut) / 4);
  __shared__ uint32_t sh_raw[kHeadWords];
  __shared__ hcir_jpeg_lut luts[4];
  __shared__ uint64_t sinfo[kT];
  __shared__ uint32_t cnt[kT];
  __shared__ uint32_t wave_tot[17];
  __shared__ DcCarry wave_carry[16];
  __shared__ JChain queue[kT / 4];
  const hcir_jpeg_header& sh = *reinterpret_cast<const hcir_jpeg_header*>(sh_raw);  // fields in front of .huff only
  const int tid = threadIdx.x, T = kT;
  const int64_t img = blockIdx.x;
  {
    const hcir_jpeg_header* gh = hdr_of(a, img);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(gh);
    for (int i = tid; i < kHeadWords; i += T) sh_raw[i] = src[i];
    for (int i = tid; i < 4 * kLutWords; i += T)
      reinterpret_cast<uint32_t*>(&luts[i / kLutWords])[i % kLutWords] =
          reinterpret_cast<const uint32_t*>(&gh->huff[i / kLutWords].lut)[i % kLutWords];
  }
  __syncthreads();
  JWin w;
  jpeg_window(sh, a.win_h, a.win_w, w);
  if (tid == 0) {
    JGeom* g = a.geom + img;
    g->w = w;
    int32_t off = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      JPlane pl;
      jpeg_plane(sh, w, c < sh.ncomp ? c : 0, pl);
      g->pl[c] = pl;
      g->plane_off[c] = off;
      off += pl.pitch * pl.rows;
    }
    g->ncomp = sh.ncomp;
    g->width = sh.width;
    g->height = sh.height;
    g->rounds = 0;
  }
#ifdef HCIR_JPEG_STAMPS
#define JSTAMP(i) if (tid == 0) a.geom[img].stamp[i] = (uint32_t)__builtin_amdgcn_s_memrealtime()
#else
#define JSTAMP(i)
#endif
  JSTAMP(0);
  if (w.last_mcu < 0 || sh.stream_bits == 0) {  // window off the image (all padding) / empty scan
    if (tid == 0 && a.status) a.status[img] = w.last_mcu < 0 ? HCIR_OK : HCIR_ERR_INVALID;
    return;
  }
  JSubseq sq;
  jpeg_subseq(sh.stream_bits, sh.stream_words, (uint32_t)T, sq);
  JStream J;
  J.words = a.ilv + img * a.ilv_stride;
  J.wps = sq.wps;
  J.nx = sq.nx;
  J.seg_start = reinterpret_cast<const uint32_t*>(a.blob + sh.stage_offset) + (((size_t)sh.stream_words * 4 + 15) / 16) * 4;
  J.nseg = (uint32_t)sh.nsegments;
  J.stream_bits = sh.stream_bits;
  J.bpm = (uint32_t)sh.blocks_per_mcu;
  jpeg_stream_tables(sh, J);
  J.luts = luts;
  J.tabs = hdr_of(a, img)->huff;
  const uint32_t S = sq.bits, nact = sq.nact;
  auto limit = [&](uint32_t m) {
    const uint64_t l = (uint64_t)(m + 1) * S;
    return (uint32_t)(l < J.stream_bits ? l : J.stream_bits);
  };
  JNullSink null_sink;
  JState st;
  // ---- phase 0: speculative decode ----
  if ((uint32_t)tid < nact) {
    jpeg_state_at(J, (uint32_t)tid * S, 0, 0, st);
    cnt[tid] = jpeg_decode_span<kFast>(J, st, limit((uint32_t)tid), null_sink);
    sinfo[tid] = jpeg_state_key(st);
  }
  JSTAMP(1);
  // ---- synchronisation: every chain verifies the subsequences behind its own, one per round, in lock step ----
  // Round 1 runs in place (every thread decodes the subsequence behind its own).  Most chains have merged by
  // then; the survivors are few and scattered over the waves, so from round 2 on they are COMPACTED into an LDS
  // queue and the first `live` threads take one each: a round costs the issue slots of ceil(live / 64) waves, not
  // of every wave that still holds one unfinished lane.  One subsequence per live chain per round, so the chains
  // through a subsequence still arrive in order of decreasing origin (the last writer is the verified one).
  uint32_t m_next = (uint32_t)tid + 1;
  bool live = m_next < nact;
  __syncthreads();
  for (uint32_t round = 1; round <= nact; ++round) {  // bounded: thread 0's chain alone ends after nact - 1 rounds
    if (live) {
      const uint32_t nb = jpeg_decode_span<kFast>(J, st, limit(m_next), null_sink);
      const uint64_t key = jpeg_state_key(st);
      if (sinfo[m_next] == key) live = false; else sinfo[m_next] = key;
      cnt[m_next] = nb;  // the chain with the smallest origin through m decodes it last, and it is the verified one
      if (++m_next >= nact) live = false;
    }
    const uint32_t incl = block_inclusive_scan(live ? 1u : 0u, wave_tot);  // two barriers: orders the LDS traffic
    const uint32_t nlive = wave_tot[16];
    if (round == 1) { JSTAMP(2); }
    if (tid == 0) a.geom[img].rounds = (int32_t)round;
    if (nlive == 0) break;
    if (nlive <= (uint32_t)(kT / 4)) {  // few survivors: pack them into the first waves (else carry on in place)
      if (live) queue[incl - 1] = JChain{st.p, (st.c << 8) | st.z, m_next, st.seg};
      __syncthreads();
      live = (uint32_t)tid < nlive;
      if (live) {
        const JChain q = queue[tid];
        st.p = q.p;
        st.c = q.cz >> 8;
        st.z = q.cz & 255u;
        st.seg = q.seg;
        st.seg_end = q.seg >= J.nseg ? J.stream_bits : J.seg_start[q.seg + 1];
        m_next = q.m;
      }
    }
    __syncthreads();
  }
  __syncthreads();
  JSTAMP(3);
  // ---- first block of every subsequence ----
  const uint32_t mine = (uint32_t)tid < nact ? cnt[tid] : 0;
  const uint32_t incl = block_inclusive_scan(mine, wave_tot);
  const uint32_t first = incl - mine;
  const uint32_t total_blocks = (uint32_t)sh.mcus_x * (uint32_t)sh.mcus_y * (uint32_t)sh.blocks_per_mcu;
  if ((uint32_t)tid == nact - 1 && a.status) a.status[img] = incl == total_blocks ? HCIR_OK : HCIR_ERR_INVALID;
  // ---- write pass ----
  const uint32_t last_block = (uint32_t)(w.last_mcu + 1) * (uint32_t)sh.blocks_per_mcu - 1;
  int16_t* dcdiff = a.dcdiff + img * a.dc_stride;
  int16_t* coef = a.coef + img * a.coef_stride;
  if ((uint32_t)tid < nact && first <= last_block) {
    if (tid == 0) {
      jpeg_state_at(J, 0, 0, 0, st);
    } else {
      const uint64_t key = sinfo[tid - 1];
      jpeg_state_at(J, (uint32_t)(key >> 16), (uint32_t)(key >> 8) & 255u, (uint32_t)key & 255u, st);
    }
    JWriteSink sink{&sh, &w, dcdiff, coef, 0, last_block, -1, 0, 0, 0};
    sink.begin(first);
    jpeg_decode_span<kFast>(J, st, limit((uint32_t)tid), sink);
  }
  __syncthreads();  // this workgroup's dcdiff stores are visible to its own threads behind the barrier
  JSTAMP(4);
  // ---- DC prediction: segmented prefix sums over the MCUs [0, last_mcu] ----
  const uint32_t nmcu = (uint32_t)w.last_mcu + 1, bpm = (uint32_t)sh.blocks_per_mcu;
  const uint32_t chunk = (nmcu + (uint32_t)T - 1) / (uint32_t)T;
  const uint32_t ri = sh.restart_interval > 0 ? (uint32_t)sh.restart_interval : 0xFFFFFFFFu;
  const uint32_t m0 = (uint32_t)tid * chunk, m1 = m0 + chunk < nmcu ? m0 + chunk : nmcu;
  DcCarry c{{0, 0, 0}, 0};
  for (uint32_t m = m0; m < m1; ++m) {
    if (m % ri == 0) c = DcCarry{{0, 0, 0}, 1};
    for (uint32_t k = 0; k < bpm; ++k) c.s[sh.blk_comp[k]] += dcdiff[m * bpm + k];
  }
  // segmented inclusive scan of the chunk sums: shuffles inside a wave, the waves' totals through LDS
  auto combine = [](const DcCarry& left, const DcCarry& right) {
    if (right.flag) return right;
    return DcCarry{{left.s[0] + right.s[0], left.s[1] + right.s[1], left.s[2] + right.s[2]}, left.flag};
  };
  const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    DcCarry u;
    u.s[0] = __shfl_up(c.s[0], o);
    u.s[1] = __shfl_up(c.s[1], o);
    u.s[2] = __shfl_up(c.s[2], o);
    u.flag = __shfl_up(c.flag, o);
    if (lane >= o) c = combine(u, c);
  }
  if (lane == 63) wave_carry[wv] = c;
  __syncthreads();
  DcCarry before{{0, 0, 0}, 0};  // everything in front of this wave
  for (int i = 0; i < wv; ++i) before = combine(before, wave_carry[i]);
  DcCarry excl;                  // everything in front of this thread
  excl.s[0] = __shfl_up(c.s[0], 1);
  excl.s[1] = __shfl_up(c.s[1], 1);
  excl.s[2] = __shfl_up(c.s[2], 1);
  excl.flag = __shfl_up(c.flag, 1);
  excl = lane == 0 ? before : combine(before, excl);
  int32_t pred[3] = {excl.s[0], excl.s[1], excl.s[2]};
  for (uint32_t m = m0; m < m1; ++m) {
    if (m % ri == 0) pred[0] = pred[1] = pred[2] = 0;
    for (uint32_t k = 0; k < bpm; ++k) {
      const int ci = sh.blk_comp[k];
      pred[ci] += dcdiff[m * bpm + k];
      const int32_t slot = jpeg_window_slot(sh, w, m * bpm + k);
      if (slot >= 0) coef[(int64_t)slot * 64] = (int16_t)pred[ci];
    }
  }
  JSTAMP(5);
}

__global__ __launch_bounds__(256) void jpeg_idct_kernel(JpegBatch a) {
  __shared__ uint16_t quant[3][64];
  __shared__ JGeom g;
  __shared__ uint8_t blk_comp[12];
  __shared__ int32_t hs[3], vs[3], bpm;
  const int64_t img = blockIdx.y;
  const hcir_jpeg_header* h = hdr_of(a, img);
  if (threadIdx.x < 96) reinterpret_cast<uint32_t*>(&quant[0][0])[threadIdx.x] = reinterpret_cast<const uint32_t*>(&h->quant[0][0])[threadIdx.x];
  if (threadIdx.x >= 128 && threadIdx.x < 128 + sizeof(JGeom) / 4)
    reinterpret_cast<uint32_t*>(&g)[threadIdx.x - 128] = reinterpret_cast<const uint32_t*>(a.geom + img)[threadIdx.x - 128];
  if (threadIdx.x >= 96 && threadIdx.x < 108) blk_comp[threadIdx.x - 96] = h->blk_comp[threadIdx.x - 96];
  if (threadIdx.x >= 108 && threadIdx.x < 111) {
    hs[threadIdx.x - 108] = h->hs[threadIdx.x - 108];
    vs[threadIdx.x - 108] = h->vs[threadIdx.x - 108];
  }
  if (threadIdx.x == 111) bpm = h->blocks_per_mcu;
  __syncthreads();
  const int32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= g.w.wblocks) return;
  // where the block lands (jpeg_block_place on the LDS copies)
  const int32_t wm = slot / bpm, blk = slot - wm * bpm;
  const int32_t ry = wm / g.w.nmx, rx = wm - ry * g.w.nmx;
  const int ci = blk_comp[blk];
  int32_t first = 0;
  for (int c = 0; c < ci; ++c) first += hs[c] * vs[c];
  const int32_t k = blk - first, by = k / hs[ci], bx = k - by * hs[ci];
  const int32_t sx = (rx * hs[ci] + bx) * 8, sy = (ry * vs[ci] + by) * 8;
  const int32_t pitch = g.pl[ci].pitch;
  uint8_t* plane = a.planes + img * a.plane_stride + g.plane_off[ci];
  // 64 coefficients: eight 16-byte loads
  int16_t cf[64];
  const uint4* src = reinterpret_cast<const uint4*>(a.coef + img * a.coef_stride + (int64_t)slot * 64);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint4 v = src[i];
    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      cf[i * 8 + 2 * j] = (int16_t)(u[j] & 0xFFFF);
      cf[i * 8 + 2 * j + 1] = (int16_t)(u[j] >> 16);
    }
  }
  uint8_t px[64];
  jpeg_idct_block(cf, quant[ci], px, 8);
  uint8_t* dst = plane + (int64_t)sy * pitch + sx;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    uint2 v;
    v.x = px[r * 8] | (px[r * 8 + 1] << 8) | (px[r * 8 + 2] << 16) | ((uint32_t)px[r * 8 + 3] << 24);
    v.y = px[r * 8 + 4] | (px[r * 8 + 5] << 8) | (px[r * 8 + 6] << 16) | ((uint32_t)px[r * 8 + 7] << 24);
    *reinterpret_cast<uint2*>(dst + (int64_t)r * pitch) = v;  // sx and pitch are multiples of 8
  }
}

// kPix output pixels of one row per thread (4: twelve contiguous bytes leave as three dword stores; needs win_w % 4 == 0)
template <int kPix>
__global__ __launch_bounds__(256) void jpeg_color_kernel(JpegBatch a) {
  __shared__ JGeom g;
  const int64_t img = blockIdx.y;
  if (threadIdx.x < sizeof(JGeom) / 4)
    reinterpret_cast<uint32_t*>(&g)[threadIdx.x] = reinterpret_cast<const uint32_t*>(a.geom + img)[threadIdx.x];
  __syncthreads();
  const int32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const int32_t per_row = a.win_w / kPix;
  if (t >= a.win_h * per_row) return;
  const int32_t oy = t / per_row, ox0 = (t - oy * per_row) * kPix;
  const int32_t y = g.w.y0 + oy;
  const uint8_t* base = a.planes + img * a.plane_stride;
  uint8_t rgb[kPix * 3];
#pragma unroll
  for (int i = 0; i < kPix; ++i) {
    const int32_t x = g.w.x0 + ox0 + i;
    uint8_t* o = rgb + 3 * i;
    o[0] = o[1] = o[2] = 0;
    if (x >= 0 && y >= 0 && x < g.width && y < g.height) {
      const JPlane& p0 = g.pl[0];
      const int32_t yy = jpeg_upsampled(base + g.plane_off[0], p0.pitch, p0.px0, p0.py0, p0.dw, p0.dh, p0.fx, p0.fy, x, y);
      if (g.ncomp == 1) {
        o[0] = o[1] = o[2] = (uint8_t)yy;
      } else {
        const JPlane& p1 = g.pl[1];
        const JPlane& p2 = g.pl[2];
        const int32_t cb = jpeg_upsampled(base + g.plane_off[1], p1.pitch, p1.px0, p1.py0, p1.dw, p1.dh, p1.fx, p1.fy, x, y);
        const int32_t cr = jpeg_upsampled(base + g.plane_off[2], p2.pitch, p2.px0, p2.py0, p2.dw, p2.dh, p2.fx, p2.fy, x, y);
        jpeg_ycc_rgb(yy, cb, cr, o);
      }
    }
  }
  uint8_t* o = a.out + (img * (int64_t)a.win_h * a.win_w + (int64_t)oy * a.win_w + ox0) * 3;
  if (kPix == 4) {
    uint32_t* o32 = reinterpret_cast<uint32_t*>(o);  // (pixel index multiple of 4) * 3 bytes: 4-byte aligned
#pragma unroll
    for (int j = 0; j < 3; ++j)
      o32[j] = rgb[4 * j] | (rgb[4 * j + 1] << 8) | (rgb[4 * j + 2] << 16) | ((uint32_t)rgb[4 * j + 3] << 24);
  } else {
#pragma unroll
    for (int j = 0; j < kPix * 3; ++j) o[j] = rgb[j];
  }
}

struct Plan {
  int64_t coef_stride, dc_stride, plane_stride, ilv_stride;  // elements per image (int16, int16, uint8, uint32)
  int32_t max_wblocks, huff_threads;
  bool fast;  // every table in use of every image resolves its long codes in the second lookup level
  int64_t max_words;
  size_t bytes(int64_t b) const {
    return (size_t)(b * (coef_stride * 2 + dc_stride * 2 + plane_stride + ilv_stride * 4 + (int64_t)sizeof(JGeom))) + 1024;
  }
};

int make_plan(const hcir_jpeg_header* hdrs, int64_t b, int32_t win_h, int32_t win_w, Plan* p) {
  int64_t mc = 0, md = 0, mp = 0, mi = 0, mwords = 0;
  int32_t mw = 0;
  // 512 threads: four workgroups per CU, a batch of ~1000 images resident at once; small batches take the shorter
  // subsequences of 1024 threads (latency)
  p->huff_threads = b >= 384 ? 512 : 1024;
  p->fast = true;
  for (int64_t i = 0; i < b; ++i) {
    const hcir_jpeg_header& h = hdrs[i];
    if (h.width == 0) continue;  // placeholder of a file the stager rejected: skipped, its window stays zero
    if (h.width <= 0 || h.height <= 0 || (h.ncomp != 1 && h.ncomp != 3) || h.blocks_per_mcu < 1 ||
        h.blocks_per_mcu > 6 || h.nsegments < 1 || h.mcus_x < 1 || h.mcus_y < 1)
      return HCIR_ERR_INVALID;
    JWin w;
    jpeg_window(h, win_h, win_w, w);
    if (w.last_mcu < 0) continue;
    JStream js;
    jpeg_stream_tables(h, js);
    p->fast = p->fast && js.fast2;
    JSubseq q;
    jpeg_subseq(h.stream_bits, h.stream_words, (uint32_t)p->huff_threads, q);
    mi = (int64_t)q.nx * q.wps > mi ? (int64_t)q.nx * q.wps : mi;
    mwords = h.stream_words > mwords ? h.stream_words : mwords;
    mw = w.wblocks > mw ? w.wblocks : mw;
    mc = (int64_t)w.wblocks * 64 > mc ? (int64_t)w.wblocks * 64 : mc;
    const int64_t nd = ((int64_t)w.last_mcu + 1) * h.blocks_per_mcu;
    md = nd > md ? nd : md;
    int64_t np = 0;
    for (int c = 0; c < h.ncomp; ++c) {
      JPlane pl;
      jpeg_plane(h, w, c, pl);
      np += (int64_t)pl.pitch * pl.rows;
    }
    mp = np > mp ? np : mp;
  }
  p->coef_stride = (mc + 63) / 64 * 64;
  p->dc_stride = (md + 63) / 64 * 64;
  p->plane_stride = (mp + 127) / 128 * 128;
  p->ilv_stride = (mi + 63) / 64 * 64;
  p->max_wblocks = mw;
  p->max_words = mwords;
  return HCIR_OK;
}

}  // namespace

extern "C" size_t hcir_jpeg_stage_bytes(const uint8_t* file, size_t nbytes) {
  hcir_jpeg_header h;
  jpeg_host::Scan sc;
  if (jpeg_host::parse(file, nbytes, &h, &sc) != HCIR_OK) return 0;
  return jpeg_host::stage_bound(h, sc);
}

extern "C" int hcir_jpeg_stage(const uint8_t* file, size_t nbytes, hcir_jpeg_header* hdr, uint8_t* blob,
                               size_t blob_offset, size_t blob_cap, size_t* used) {
  if (!file || !hdr || !blob || !used || (blob_offset & 15)) return HCIR_ERR_INVALID;
  jpeg_host::Scan sc;
  const int rc = jpeg_host::parse(file, nbytes, hdr, &sc);
  if (rc != HCIR_OK) return rc;
  if (blob_offset > blob_cap || jpeg_host::stage_bound(*hdr, sc) > blob_cap - blob_offset) return HCIR_ERR_WORKSPACE;
  hdr->stage_offset = blob_offset;
  return jpeg_host::stage(hdr, sc, blob + blob_offset, used);
}

extern "C" size_t hcir_jpeg_workspace_bytes(const hcir_jpeg_header* hdrs_host, int64_t b, int32_t win_h, int32_t win_w) {
  Plan p;
  if (!hdrs_host || b <= 0 || win_h <= 0 || win_w <= 0 || make_plan(hdrs_host, b, win_h, win_w, &p) != HCIR_OK) return 0;
  return p.bytes(b);
}

extern "C" int hcir_jpeg_decode_window_u8(const void* blob_dev, const hcir_jpeg_header* hdrs_host, int64_t b,
                                          int32_t win_h, int32_t win_w, uint8_t* out, int32_t* status_dev,
                                          void* workspace, size_t workspace_bytes, void* stream) {
  HCIR_ENTER();
  if (!blob_dev || !hdrs_host || !out || !workspace || b <= 0 || win_h <= 0 || win_w <= 0 || b > 65535 ||
      (int64_t)win_h * win_w > (int64_t(1) << 30))
    return HCIR_ERR_INVALID;
  Plan p;
  const int rc = make_plan(hdrs_host, b, win_h, win_w, &p);
  if (rc != HCIR_OK) return rc;
  const size_t need = p.bytes(b);
  if (workspace_bytes < need) return HCIR_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  JpegBatch a{};
  a.blob = static_cast<const uint8_t*>(blob_dev);
  a.b = b;
  a.win_h = win_h;
  a.win_w = win_w;
  a.out = out;
  a.status = status_dev;
  uint8_t* ws = reinterpret_cast<uint8_t*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  a.geom = reinterpret_cast<JGeom*>(ws);  // first: tools/diag_jpeg.py reads the records back
  a.coef = reinterpret_cast<int16_t*>(ws + (((size_t)b * sizeof(JGeom) + 255) & ~(size_t)255));
  a.dcdiff = a.coef + b * p.coef_stride;
  a.planes = reinterpret_cast<uint8_t*>(a.dcdiff + b * p.dc_stride);
  a.ilv = reinterpret_cast<uint32_t*>(((uintptr_t)(a.planes + b * p.plane_stride) + 255) & ~(uintptr_t)255);
  a.ilv_stride = p.ilv_stride;
  a.huff_threads = p.huff_threads;
  a.coef_stride = p.coef_stride;
  a.dc_stride = p.dc_stride;
  a.plane_stride = p.plane_stride;
  // AC coefficients are stored sparsely: the buffer starts as zeros
  if (p.coef_stride && hipMemsetAsync(a.coef, 0, (size_t)(b * p.coef_stride * 2), st) != hipSuccess) return HCIR_ERR_LAUNCH;
  if (p.max_words > 0) {
    const int64_t tiles = hcir_cdiv(p.max_words, 1024) + 64;
    hipLaunchKernelGGL(jpeg_interleave_kernel, dim3((unsigned)(tiles < 256 ? tiles : 256), (unsigned)b), dim3(256), 0, st, a);
    HCIR_LAUNCH_CHECK();
  }
  if (p.huff_threads == 512) {
    if (p.fast) hipLaunchKernelGGL((jpeg_huffman_kernel<512, true>), dim3((unsigned)b), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((jpeg_huffman_kernel<512, false>), dim3((unsigned)b), dim3(512), 0, st, a);
  } else {
    if (p.fast) hipLaunchKernelGGL((jpeg_huffman_kernel<1024, true>), dim3((unsigned)b), dim3(1024), 0, st, a);
    else hipLaunchKernelGGL((jpeg_huffman_kernel<1024, false>), dim3((unsigned)b), dim3(1024), 0, st, a);
  }
  HCIR_LAUNCH_CHECK();
  if (p.max_wblocks > 0) {
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)hcir_cdiv(p.max_wblocks, 256), (unsigned)b), dim3(256), 0, st, a);
    HCIR_LAUNCH_CHECK();
  }
  if (win_w % 4 == 0)
    hipLaunchKernelGGL(jpeg_color_kernel<4>, dim3((unsigned)hcir_cdiv((int64_t)win_h * (win_w / 4), 256), (unsigned)b),
                       dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(jpeg_color_kernel<1>, dim3((unsigned)hcir_cdiv((int64_t)win_h * win_w, 256), (unsigned)b), dim3(256),
                       0, st, a);
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}

// HOST: stage a whole batch with a few worker threads (the per-file work is a memchr walk and a copy).
// blob == NULL: only *blob_used (bytes needed: headers + every stream's bound) and status[] are produced.
extern "C" int hcir_jpeg_stage_batch(const uint8_t* const* files, const size_t* nbytes, int64_t b, uint8_t* blob,
                                     size_t blob_cap, size_t* blob_used, int32_t* status, int32_t nthreads) {
  if (!files || !nbytes || !blob_used || !status || b <= 0) return HCIR_ERR_INVALID;
  const size_t hdr_bytes = jpeg_host::align16((size_t)b * sizeof(hcir_jpeg_header));
  const bool too_small = blob && blob_cap < hdr_bytes;   // not even the headers fit: size it like the sizing call
  const bool fill = blob != nullptr && !too_small;
  std::vector<size_t> off((size_t)b + 1, 0);
  std::vector<hcir_jpeg_header> tmp;
  if (!fill) tmp.resize((size_t)b);
  hcir_jpeg_header* hdrs = fill ? reinterpret_cast<hcir_jpeg_header*>(blob) : tmp.data();
  std::vector<jpeg_host::Scan> scans((size_t)b);
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
  // pass 1: marker walk (headers and tables only - the entropy-coded bytes are not touched), per-file bound.  With a
  // blob the headers are parsed in place, once.
  std::vector<size_t> bound((size_t)b, 0);
  run([&](int64_t i) {
    status[i] = files[i] ? jpeg_host::parse(files[i], nbytes[i], &hdrs[i], &scans[(size_t)i]) : HCIR_ERR_INVALID;
    bound[(size_t)i] = status[i] == HCIR_OK ? jpeg_host::stage_bound(hdrs[i], scans[(size_t)i]) : 0;
  });
  off[0] = hdr_bytes;
  for (int64_t i = 0; i < b; ++i) off[(size_t)i + 1] = off[(size_t)i] + bound[(size_t)i];
  *blob_used = off[(size_t)b];
  if (!fill) return too_small ? HCIR_ERR_WORKSPACE : HCIR_OK;
  if (blob_cap < off[(size_t)b]) return HCIR_ERR_WORKSPACE;
  // pass 2: the staging copies
  run([&](int64_t i) {
    if (status[i] == HCIR_OK) {
      size_t used = 0;
      hdrs[i].stage_offset = off[(size_t)i];
      status[i] = jpeg_host::stage(&hdrs[i], scans[(size_t)i], blob + off[(size_t)i], &used);
    }
    // width 0: the device skips the image (its window stays zero)
    if (status[i] != HCIR_OK) memset(&hdrs[i], 0, sizeof(hcir_jpeg_header));
  });
  return HCIR_OK;
}
