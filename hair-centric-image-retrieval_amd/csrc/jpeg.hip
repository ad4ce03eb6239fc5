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

constexpr int kHuffThreads = 1024;

struct JpegBatch {
  const uint8_t* blob;  // device: headers, then streams
  int64_t b;
  int32_t win_h, win_w;
  uint8_t* out;
  int32_t* status;
  int16_t* coef;    // [b][coef_stride]
  int16_t* dcdiff;  // [b][dc_stride]
  uint8_t* planes;  // [b][plane_stride]
  int64_t coef_stride, dc_stride, plane_stride;  // elements
};

__device__ __forceinline__ const hcir_jpeg_header* hdr_of(const JpegBatch& a, int64_t img) {
  return reinterpret_cast<const hcir_jpeg_header*>(a.blob) + img;
}

// inclusive scan of one uint32 per thread over the workgroup (blockDim.x a multiple of 64, <= 1024)
__device__ __forceinline__ uint32_t block_inclusive_scan(uint32_t v, uint32_t* wave_tot /* LDS [16] */) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t u = __shfl_up(v, o);
    if (lane >= o) v += u;
  }
  if (lane == 63) wave_tot[wv] = v;
  __syncthreads();
  uint32_t add = 0;
  for (int i = 0; i < wv; ++i) add += wave_tot[i];
  (void)nw;
  __syncthreads();
  return v + add;
}

struct DcCarry {
  int32_t s[3];
  int32_t flag;  // a restart boundary lies inside (the sums restart there)
};

__global__ __launch_bounds__(kHuffThreads) void jpeg_huffman_kernel(JpegBatch a) {
  __shared__ hcir_jpeg_header sh;
  __shared__ uint64_t sinfo[kHuffThreads];
  __shared__ uint32_t cnt[kHuffThreads];
  __shared__ uint32_t wave_tot[16];
  __shared__ DcCarry carry[2][kHuffThreads];
  const int tid = threadIdx.x, T = blockDim.x;
  const int64_t img = blockIdx.x;
  {  // header -> LDS (4-byte pieces; sizeof is a multiple of 8)
    const uint32_t* src = reinterpret_cast<const uint32_t*>(hdr_of(a, img));
    uint32_t* dst = reinterpret_cast<uint32_t*>(&sh);
    for (int i = tid; i < (int)(sizeof(hcir_jpeg_header) / 4); i += T) dst[i] = src[i];
  }
  __syncthreads();
  JWin w;
  jpeg_window(sh, a.win_h, a.win_w, w);
  if (w.last_mcu < 0 || sh.stream_bits == 0) {  // window off the image (all padding) / empty scan
    if (tid == 0 && a.status) a.status[img] = w.last_mcu < 0 ? HCIR_OK : HCIR_ERR_INVALID;
    return;
  }
  JStream J;
  J.words = reinterpret_cast<const uint32_t*>(a.blob + sh.stage_offset);
  J.seg_start = J.words + (((size_t)sh.stream_words * 4 + 15) / 16) * 4;
  J.nseg = (uint32_t)sh.nsegments;
  J.stream_bits = sh.stream_bits;
  J.bpm = (uint32_t)sh.blocks_per_mcu;
  jpeg_stream_tables(sh, J);
  uint32_t S = (J.stream_bits + (uint32_t)T - 1) / (uint32_t)T;
  S = (S + 31) & ~31u;
  if (S < 128) S = 128;
  const uint32_t nact = (J.stream_bits + S - 1) / S;
  auto limit = [&](uint32_t m) {
    const uint64_t l = (uint64_t)(m + 1) * S;
    return (uint32_t)(l < J.stream_bits ? l : J.stream_bits);
  };
  JNullSink null_sink;
  JState st;
  // ---- phase 0: speculative decode ----
  if ((uint32_t)tid < nact) {
    jpeg_state_at(J, (uint32_t)tid * S, 0, 0, st);
    cnt[tid] = jpeg_decode_span(J, st, limit((uint32_t)tid), null_sink);
    sinfo[tid] = jpeg_state_key(st);
  }
  // ---- synchronisation: every active thread verifies the subsequences behind its own ----
  bool done = (uint32_t)tid >= nact;
  for (uint32_t k = 1; k <= nact; ++k) {  // bounded: thread 0 alone needs at most nact - 1 iterations
    const uint32_t m = (uint32_t)tid + k;
    if (!done && m >= nact) done = true;
    if (!__syncthreads_or(!done)) break;  // also orders the previous iteration's LDS writes before these reads
    if (!done) {
      const uint32_t nb = jpeg_decode_span(J, st, limit(m), null_sink);
      const uint64_t key = jpeg_state_key(st);
      if (sinfo[m] == key) done = true; else sinfo[m] = key;
      cnt[m] = nb;  // the chain with the smallest origin through m decodes it last, and it is the verified one
    }
  }
  __syncthreads();
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
    JWriteSink sink{&sh, &w, dcdiff, coef, 0, last_block, -1};
    sink.begin(first);
    jpeg_decode_span(J, st, limit((uint32_t)tid), sink);
  }
  __syncthreads();  // this workgroup's dcdiff stores are visible to its own threads behind the barrier
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
  carry[0][tid] = c;
  __syncthreads();
  int cur = 0;
  for (int o = 1; o < T; o <<= 1) {  // Hillis-Steele with the segmented combine
    DcCarry v = carry[cur][tid];
    if (tid >= o && !v.flag) {
      const DcCarry u = carry[cur][tid - o];
      v.s[0] += u.s[0];
      v.s[1] += u.s[1];
      v.s[2] += u.s[2];
      v.flag = u.flag;
    }
    carry[cur ^ 1][tid] = v;
    cur ^= 1;
    __syncthreads();
  }
  int32_t pred[3] = {0, 0, 0};
  if (tid > 0) {
    const DcCarry u = carry[cur][tid - 1];
    pred[0] = u.s[0];
    pred[1] = u.s[1];
    pred[2] = u.s[2];
  }
  for (uint32_t m = m0; m < m1; ++m) {
    if (m % ri == 0) pred[0] = pred[1] = pred[2] = 0;
    for (uint32_t k = 0; k < bpm; ++k) {
      const int ci = sh.blk_comp[k];
      pred[ci] += dcdiff[m * bpm + k];
      const int32_t slot = jpeg_window_slot(sh, w, m * bpm + k);
      if (slot >= 0) coef[(int64_t)slot * 64] = (int16_t)pred[ci];
    }
  }
}

__global__ __launch_bounds__(256) void jpeg_idct_kernel(JpegBatch a) {
  __shared__ hcir_jpeg_header sh;  // geometry + quantisation tables (the Huffman tables ride along: one copy loop)
  const int64_t img = blockIdx.y;
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(hdr_of(a, img));
    uint32_t* dst = reinterpret_cast<uint32_t*>(&sh);
    constexpr int kWords = (int)(offsetof(hcir_jpeg_header, huff) / 4);
    for (int i = threadIdx.x; i < kWords; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  JWin w;
  jpeg_window(sh, a.win_h, a.win_w, w);
  const int32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= w.wblocks) return;
  int ci;
  int32_t sx, sy;
  jpeg_block_place(sh, w, slot, ci, sx, sy);
  JPlane pl;
  uint8_t* plane = a.planes + img * a.plane_stride;
  for (int c = 0; c < ci; ++c) {
    jpeg_plane(sh, w, c, pl);
    plane += (int64_t)pl.pitch * pl.rows;
  }
  jpeg_plane(sh, w, ci, pl);
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
  jpeg_idct_block(cf, sh.quant[ci], px, 8);
  uint8_t* dst = plane + (int64_t)sy * pl.pitch + sx;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    uint2 v;
    v.x = px[r * 8] | (px[r * 8 + 1] << 8) | (px[r * 8 + 2] << 16) | ((uint32_t)px[r * 8 + 3] << 24);
    v.y = px[r * 8 + 4] | (px[r * 8 + 5] << 8) | (px[r * 8 + 6] << 16) | ((uint32_t)px[r * 8 + 7] << 24);
    *reinterpret_cast<uint2*>(dst + (int64_t)r * pl.pitch) = v;  // sx and pitch are multiples of 8
  }
}

__global__ __launch_bounds__(256) void jpeg_color_kernel(JpegBatch a) {
  __shared__ hcir_jpeg_header sh;
  const int64_t img = blockIdx.y;
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(hdr_of(a, img));
    uint32_t* dst = reinterpret_cast<uint32_t*>(&sh);
    constexpr int kWords = (int)(offsetof(hcir_jpeg_header, quant) / 4);
    for (int i = threadIdx.x; i < kWords; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  const int32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= a.win_h * a.win_w) return;
  JWin w;
  jpeg_window(sh, a.win_h, a.win_w, w);
  const int32_t oy = p / a.win_w, ox = p - oy * a.win_w;
  const int32_t x = w.x0 + ox, y = w.y0 + oy;
  uint8_t rgb[3] = {0, 0, 0};
  if (x >= 0 && y >= 0 && x < sh.width && y < sh.height) {
    const uint8_t* plane = a.planes + img * a.plane_stride;
    JPlane pl;
    jpeg_plane(sh, w, 0, pl);
    const int32_t yy = jpeg_upsampled(plane, pl.pitch, pl.px0, pl.py0, pl.dw, pl.dh, pl.fx, pl.fy, x, y);
    if (sh.ncomp == 1) {
      rgb[0] = rgb[1] = rgb[2] = (uint8_t)yy;
    } else {
      plane += (int64_t)pl.pitch * pl.rows;
      jpeg_plane(sh, w, 1, pl);
      const int32_t cb = jpeg_upsampled(plane, pl.pitch, pl.px0, pl.py0, pl.dw, pl.dh, pl.fx, pl.fy, x, y);
      plane += (int64_t)pl.pitch * pl.rows;
      jpeg_plane(sh, w, 2, pl);
      const int32_t cr = jpeg_upsampled(plane, pl.pitch, pl.px0, pl.py0, pl.dw, pl.dh, pl.fx, pl.fy, x, y);
      jpeg_ycc_rgb(yy, cb, cr, rgb);
    }
  }
  uint8_t* o = a.out + (img * (int64_t)a.win_h * a.win_w + p) * 3;
  o[0] = rgb[0];
  o[1] = rgb[1];
  o[2] = rgb[2];
}

struct Plan {
  int64_t coef_stride, dc_stride, plane_stride;  // elements per image (int16, int16, uint8), 128-byte multiples
  int32_t max_wblocks;
};

int make_plan(const hcir_jpeg_header* hdrs, int64_t b, int32_t win_h, int32_t win_w, Plan* p) {
  int64_t mc = 0, md = 0, mp = 0;
  int32_t mw = 0;
  for (int64_t i = 0; i < b; ++i) {
    const hcir_jpeg_header& h = hdrs[i];
    if (h.width == 0) continue;  // placeholder of a file the stager rejected: skipped, its window stays zero
    if (h.width <= 0 || h.height <= 0 || (h.ncomp != 1 && h.ncomp != 3) || h.blocks_per_mcu < 1 ||
        h.blocks_per_mcu > 6 || h.nsegments < 1 || h.mcus_x < 1 || h.mcus_y < 1)
      return HCIR_ERR_INVALID;
    JWin w;
    jpeg_window(h, win_h, win_w, w);
    if (w.last_mcu < 0) continue;
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
  p->max_wblocks = mw;
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
  return (size_t)(b * (p.coef_stride * 2 + p.dc_stride * 2 + p.plane_stride)) + 256;
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
  const size_t need = (size_t)(b * (p.coef_stride * 2 + p.dc_stride * 2 + p.plane_stride)) + 256;
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
  a.coef = reinterpret_cast<int16_t*>(ws);
  a.dcdiff = a.coef + b * p.coef_stride;
  a.planes = reinterpret_cast<uint8_t*>(a.dcdiff + b * p.dc_stride);
  a.coef_stride = p.coef_stride;
  a.dc_stride = p.dc_stride;
  a.plane_stride = p.plane_stride;
  // AC coefficients are stored sparsely: the buffer starts as zeros
  if (p.coef_stride && hipMemsetAsync(a.coef, 0, (size_t)(b * p.coef_stride * 2), st) != hipSuccess) return HCIR_ERR_LAUNCH;
  hipLaunchKernelGGL(jpeg_huffman_kernel, dim3((unsigned)b), dim3(kHuffThreads), 0, st, a);
  HCIR_LAUNCH_CHECK();
  if (p.max_wblocks > 0) {
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)hcir_cdiv(p.max_wblocks, 256), (unsigned)b), dim3(256), 0, st, a);
    HCIR_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(jpeg_color_kernel, dim3((unsigned)hcir_cdiv((int64_t)win_h * win_w, 256), (unsigned)b), dim3(256), 0,
                     st, a);
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}

// HOST: stage a whole batch with a few worker threads (the per-file work is a memchr walk and a copy).
// blob == NULL: only *blob_used (bytes needed: headers + every stream's bound) and status[] are produced.
extern "C" int hcir_jpeg_stage_batch(const uint8_t* const* files, const size_t* nbytes, int64_t b, uint8_t* blob,
                                     size_t blob_cap, size_t* blob_used, int32_t* status, int32_t nthreads) {
  if (!files || !nbytes || !blob_used || !status || b <= 0) return HCIR_ERR_INVALID;
  const size_t hdr_bytes = jpeg_host::align16((size_t)b * sizeof(hcir_jpeg_header));
  std::vector<size_t> off((size_t)b + 1, 0);
  std::vector<hcir_jpeg_header> tmp;
  if (!blob) tmp.resize((size_t)b);
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
  // pass 1: parse, per-file bound
  std::vector<size_t> bound((size_t)b, 0);
  run([&](int64_t i) {
    hcir_jpeg_header h;
    jpeg_host::Scan sc;
    status[i] = files[i] ? jpeg_host::parse(files[i], nbytes[i], &h, &sc) : HCIR_ERR_INVALID;
    bound[(size_t)i] = status[i] == HCIR_OK ? jpeg_host::stage_bound(h, sc) : 0;
  });
  off[0] = hdr_bytes;
  for (int64_t i = 0; i < b; ++i) off[(size_t)i + 1] = off[(size_t)i] + bound[(size_t)i];
  *blob_used = off[(size_t)b];
  if (!blob) return HCIR_OK;
  if (blob_cap < off[(size_t)b]) return HCIR_ERR_WORKSPACE;
  hcir_jpeg_header* hdrs = reinterpret_cast<hcir_jpeg_header*>(blob);
  run([&](int64_t i) {
    if (status[i] != HCIR_OK) {
      memset(&hdrs[i], 0, sizeof(hcir_jpeg_header));  // width 0: the device skips the image (its window stays zero)
      return;
    }
    size_t used = 0;
    status[i] = hcir_jpeg_stage(files[i], nbytes[i], &hdrs[i], blob, off[(size_t)i], blob_cap, &used);
    if (status[i] != HCIR_OK) memset(&hdrs[i], 0, sizeof(hcir_jpeg_header));
  });
  return HCIR_OK;
}
