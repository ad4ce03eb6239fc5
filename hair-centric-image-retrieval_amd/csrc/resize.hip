// resize.hip — Pillow's bicubic Image.resize + CenterCrop on the device (include/hcir.h "Resize(224, bicubic)").
//
// Stands where src/models/hair_encoder.py:44-48 runs transforms.Resize(224, interpolation=3) -> CenterCrop(224) on
// the host (torchvision -> PIL.Image.resize(BICUBIC) -> libImaging/Resample.c).  The coefficient tables are
// Pillow's, computed on the host in double (precompute_coeffs, normalize_coeffs_8bpc: PRECISION_BITS = 22); the two
// passes are its integer loops (ImagingResampleHorizontal_8bpc / Vertical_8bpc): accumulator starts at 1 << 21,
// result clip8(ss >> 22), 8-bit intermediate between the passes.  One thread per output pixel; only the window's
// columns (horizontal pass) and rows (vertical pass) are produced.  HBM-bound and small: a 1024^2 source is read once.
#include <math.h>

#include <vector>

#include "common.h"

namespace {

constexpr int kPrecisionBits = 32 - 8 - 2;

inline double bicubic_filter(double x) {  // Resample.c, a = -0.5
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

inline int ksize_of(int32_t in_size, int32_t out_size) {
  double filterscale = (double)in_size / out_size;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 2.0 * filterscale;
  return (int)ceil(support) * 2 + 1;
}

struct ResizeBatch {
  const uint8_t* src;
  const int32_t* coef;
  const hcir_resize_job* jobs;
  uint8_t* tmp;
  uint64_t tmp_stride;  // bytes per image
  uint8_t* out;
  int32_t win_h, win_w;
};

__device__ __forceinline__ uint32_t clip8(int32_t ss) {
  const int32_t v = ss >> kPrecisionBits;
  return (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// rows of the source the vertical pass of the window reads: [r0, r1)
__device__ __forceinline__ void source_rows(const hcir_resize_job& j, const int32_t* coef, int32_t win_h, int32_t& r0,
                                            int32_t& r1) {
  int32_t y0 = j.crop_top < 0 ? 0 : j.crop_top, y1 = j.crop_top + win_h > j.out_h ? j.out_h : j.crop_top + win_h;
  if (y1 <= y0) {
    r0 = r1 = 0;
    return;
  }
  if (j.coef_v < 0) {
    r0 = y0;
    r1 = y1;
    return;
  }
  const int32_t* bv = coef + j.coef_v;
  r0 = bv[2 * y0];
  r1 = bv[2 * (y1 - 1)] + bv[2 * (y1 - 1) + 1];
}

__global__ __launch_bounds__(256) void resize_horizontal_kernel(ResizeBatch a) {
  const hcir_resize_job j = a.jobs[blockIdx.y];
  int32_t r0, r1;
  source_rows(j, a.coef, a.win_h, r0, r1);
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int32_t row = (int32_t)(t / a.win_w), x = (int32_t)(t - (int64_t)row * a.win_w);
  if (row >= r1 - r0) return;
  const int32_t xx = j.crop_left + x;
  uint8_t* o = a.tmp + (uint64_t)blockIdx.y * a.tmp_stride + ((uint64_t)row * a.win_w + x) * 3;
  if (xx < 0 || xx >= j.out_w) {
    o[0] = o[1] = o[2] = 0;
    return;
  }
  const uint8_t* s = a.src + j.src_offset + (uint64_t)(r0 + row) * j.src_pitch;
  if (j.coef_h < 0) {
    o[0] = s[xx * 3], o[1] = s[xx * 3 + 1], o[2] = s[xx * 3 + 2];
    return;
  }
  const int32_t* bh = a.coef + j.coef_h;
  const int32_t xmin = bh[2 * xx], n = bh[2 * xx + 1];
  const int32_t* k = bh + 2 * j.out_w + (int64_t)xx * j.ksize_h;
  int32_t s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
  const uint8_t* p = s + (int64_t)xmin * 3;
  for (int32_t i = 0; i < n; ++i) {
    const int32_t c = k[i];
    s0 += (int32_t)p[3 * i] * c;
    s1 += (int32_t)p[3 * i + 1] * c;
    s2 += (int32_t)p[3 * i + 2] * c;
  }
  o[0] = (uint8_t)clip8(s0), o[1] = (uint8_t)clip8(s1), o[2] = (uint8_t)clip8(s2);
}

__global__ __launch_bounds__(256) void resize_vertical_kernel(ResizeBatch a) {
  const hcir_resize_job j = a.jobs[blockIdx.y];
  int32_t r0, r1;
  source_rows(j, a.coef, a.win_h, r0, r1);
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int32_t y = (int32_t)(t / a.win_w), x = (int32_t)(t - (int64_t)y * a.win_w);
  if (y >= a.win_h) return;
  uint8_t* o = a.out + ((uint64_t)blockIdx.y * a.win_h * a.win_w + (uint64_t)y * a.win_w + x) * 3;
  const int32_t yy = j.crop_top + y;
  if (yy < 0 || yy >= j.out_h) {
    o[0] = o[1] = o[2] = 0;
    return;
  }
  const uint8_t* tmp = a.tmp + (uint64_t)blockIdx.y * a.tmp_stride + (uint64_t)x * 3;
  const uint64_t pitch = (uint64_t)a.win_w * 3;
  if (j.coef_v < 0) {
    const uint8_t* p = tmp + (uint64_t)(yy - r0) * pitch;
    o[0] = p[0], o[1] = p[1], o[2] = p[2];
    return;
  }
  const int32_t* bv = a.coef + j.coef_v;
  const int32_t ymin = bv[2 * yy], n = bv[2 * yy + 1];
  const int32_t* k = bv + 2 * j.out_h + (int64_t)yy * j.ksize_v;
  int32_t s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
  const uint8_t* p = tmp + (uint64_t)(ymin - r0) * pitch;
  for (int32_t i = 0; i < n; ++i) {
    const int32_t c = k[i];
    s0 += (int32_t)p[0] * c;
    s1 += (int32_t)p[1] * c;
    s2 += (int32_t)p[2] * c;
    p += pitch;
  }
  o[0] = (uint8_t)clip8(s0), o[1] = (uint8_t)clip8(s1), o[2] = (uint8_t)clip8(s2);
}

// host copy of source_rows() over the host coefficient tables is not available (they live in the caller's blob):
// the intermediate is sized for every source row
int64_t max_rows(const hcir_resize_job* jobs, int64_t b) {
  int64_t m = 0;
  for (int64_t i = 0; i < b; ++i) m = jobs[i].src_h > m ? jobs[i].src_h : m;
  return m;
}

}  // namespace

extern "C" int32_t hcir_resize_bicubic_ksize(int32_t in_size, int32_t out_size) {
  if (in_size <= 0 || out_size <= 0) return 0;
  return ksize_of(in_size, out_size);
}

// Pillow's precompute_coeffs (box = the whole axis) followed by normalize_coeffs_8bpc.  The statements follow
// Resample.c one for one: the rounding of every double operation is what makes the tables equal.
extern "C" int hcir_resize_bicubic_coeffs(int32_t in_size, int32_t out_size, int32_t* bounds, int32_t* kk) {
  if (in_size <= 0 || out_size <= 0 || !bounds || !kk) return HCIR_ERR_INVALID;
  const int inSize = in_size, outSize = out_size;
  const double in0 = 0.0, in1 = (double)(float)in_size;  // the box travels as float through Image.resize
  double support, scale, filterscale;
  filterscale = scale = (double)(in1 - in0) / outSize;
  if (filterscale < 1.0) filterscale = 1.0;
  support = 2.0 * filterscale;
  const int ksize = (int)ceil(support) * 2 + 1;
  std::vector<double> kbuf((size_t)ksize);
  for (int xx = 0; xx < outSize; xx++) {
    const double center = in0 + (xx + 0.5) * scale;
    double ww = 0.0;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > inSize) xmax = inSize;
    xmax -= xmin;
    double* k = kbuf.data();
    int x;
    for (x = 0; x < xmax; x++) {
      const double w = bicubic_filter((x + xmin - center + 0.5) * ss);
      k[x] = w;
      ww += w;
    }
    for (x = 0; x < xmax; x++)
      if (ww != 0.0) k[x] /= ww;
    for (; x < ksize; x++) k[x] = 0;
    bounds[xx * 2 + 0] = xmin;
    bounds[xx * 2 + 1] = xmax;
    int32_t* ko = kk + (size_t)xx * ksize;
    for (x = 0; x < ksize; x++)
      ko[x] = k[x] < 0 ? (int32_t)(-0.5 + k[x] * (1 << kPrecisionBits)) : (int32_t)(0.5 + k[x] * (1 << kPrecisionBits));
  }
  return HCIR_OK;
}

extern "C" size_t hcir_resize_crop_workspace_bytes(const hcir_resize_job* jobs_host, int64_t b, int32_t win_h, int32_t win_w) {
  if (!jobs_host || b <= 0 || win_h <= 0 || win_w <= 0) return 0;
  return (size_t)b * (((size_t)max_rows(jobs_host, b) * win_w * 3 + 255) & ~(size_t)255) + 256;
}

extern "C" int hcir_resize_crop_bicubic_u8(const uint8_t* src, const int32_t* coef, const hcir_resize_job* jobs_dev,
                                           const hcir_resize_job* jobs_host, int64_t b, int32_t win_h, int32_t win_w,
                                           uint8_t* out, void* workspace, size_t workspace_bytes, void* stream) {
  HCIR_ENTER();
  if (!src || !jobs_dev || !jobs_host || !out || !workspace || b <= 0 || b > 65535 || win_h <= 0 || win_w <= 0)
    return HCIR_ERR_INVALID;
  for (int64_t i = 0; i < b; ++i) {
    const hcir_resize_job& j = jobs_host[i];
    if (j.src_h <= 0 || j.src_w <= 0 || j.out_h <= 0 || j.out_w <= 0 || j.src_pitch < (int64_t)j.src_w * 3 ||
        (j.coef_h < 0 && j.out_w != j.src_w) || (j.coef_v < 0 && j.out_h != j.src_h) ||
        ((j.coef_h >= 0 || j.coef_v >= 0) && !coef))
      return HCIR_ERR_INVALID;
  }
  if (workspace_bytes < hcir_resize_crop_workspace_bytes(jobs_host, b, win_h, win_w)) return HCIR_ERR_WORKSPACE;
  ResizeBatch a{};
  a.src = src;
  a.coef = coef;
  a.jobs = jobs_dev;
  a.tmp = reinterpret_cast<uint8_t*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  const int64_t rows = max_rows(jobs_host, b);
  a.tmp_stride = ((uint64_t)rows * win_w * 3 + 255) & ~(uint64_t)255;
  a.out = out;
  a.win_h = win_h;
  a.win_w = win_w;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(resize_horizontal_kernel, dim3((unsigned)hcir_cdiv(rows * win_w, 256), (unsigned)b), dim3(256), 0, st, a);
  HCIR_LAUNCH_CHECK();
  hipLaunchKernelGGL(resize_vertical_kernel, dim3((unsigned)hcir_cdiv((int64_t)win_h * win_w, 256), (unsigned)b), dim3(256), 0,
                     st, a);
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}
