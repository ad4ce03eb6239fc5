// transform.hip — knn_transform on the device (HP/utils/transform.py:10-14):
//   CenterCrop(size) -> ToTensor (/255, HWC -> CHW) -> Normalize(mean, std)
// over a batch of decoded RGB8 images of one size.  One thread per output pixel (3 channels):
// reads 3 consecutive bytes, writes one float into each of the three channel planes (coalesced
// per plane).  HBM-bound: 3 B in, 12 B out per pixel.
//
// Bit-exact with torchvision's arithmetic: fp32(u8) / 255 (IEEE division), then (x - mean) / std
// with fp32 mean / std (IEEE subtraction and division; hipcc keeps fp32 division correctly rounded).
// CenterCrop on an image smaller than the window pads with zeros BEFORE ToTensor, i.e. the padded
// pixels are the byte 0 and come out as (0 - mean) / std: torchvision semantics
// (crop top = round((H - size) / 2), pad = (size - H) // 2 on the top / left).
#include "common.h"

namespace {

struct TransformArgs {
  const uint8_t* img;  // [B][H][W][3]
  float* out;          // [B][3][size][size]
  int64_t b;
  int h, w, size;
  int top, left;       // crop origin in the (possibly padded) image
  int pad_t, pad_l;    // zero padding added on the top / left when the image is smaller than the window
  float mean[3], std[3];
};

__global__ void knn_transform_kernel(TransformArgs a) {
  const int64_t npix = (int64_t)a.size * a.size;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.b * npix) return;
  const int64_t bi = t / npix;
  const int p = (int)(t % npix);
  const int oy = p / a.size, ox = p % a.size;
  // coordinates in the padded image, then in the real one
  const int y = oy + a.top - a.pad_t, x = ox + a.left - a.pad_l;
  uint8_t px[3] = {0, 0, 0};
  if (y >= 0 && y < a.h && x >= 0 && x < a.w) {
    const uint8_t* s = a.img + ((bi * a.h + y) * (int64_t)a.w + x) * 3;
    px[0] = s[0];
    px[1] = s[1];
    px[2] = s[2];
  }
  float* o = a.out + bi * 3 * npix + p;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v = (float)px[c] / 255.0f;
    o[c * npix] = (v - a.mean[c]) / a.std[c];
  }
}

}  // namespace

extern "C" int hcir_knn_transform_u8(const uint8_t* img, int64_t b, int32_t h, int32_t w, int32_t size,
                                     const float* mean3, const float* std3, float* out, void* stream) {
  HCIR_ENTER();
  if (!img || !out || !mean3 || !std3 || b <= 0 || h <= 0 || w <= 0 || size <= 0) return HCIR_ERR_INVALID;
  if (b * (int64_t)size * size > (int64_t(1) << 40)) return HCIR_ERR_INVALID;
  TransformArgs a{};
  a.img = img;
  a.out = out;
  a.b = b;
  a.h = h;
  a.w = w;
  a.size = size;
  // torchvision.transforms.functional.center_crop: pad to at least the window, then crop at
  // round((H' - size) / 2) of the padded size H' (Python round: half to even; the numerator is an integer,
  // so a half occurs for odd differences: round(x.5) -> even)
  const int ph = h < size ? size - h : 0, pw = w < size ? size - w : 0;
  a.pad_t = ph / 2;
  a.pad_l = pw / 2;
  const int hp = h + ph, wp = w + pw;
  auto pyround_half = [](int num) {  // round(num / 2.0) with ties to even
    if ((num & 1) == 0) return num / 2;
    const int lo = (num - 1) / 2;    // num/2 = lo + 0.5
    return (lo & 1) ? lo + 1 : lo;
  };
  a.top = pyround_half(hp - size);
  a.left = pyround_half(wp - size);
  for (int c = 0; c < 3; ++c) {
    a.mean[c] = mean3[c];
    a.std[c] = std3[c];
    if (!(a.std[c] != 0.f)) return HCIR_ERR_INVALID;
  }
  const int64_t n = b * (int64_t)size * size;
  hipLaunchKernelGGL(knn_transform_kernel, dim3((unsigned)hcir_cdiv(n, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a);
  HCIR_LAUNCH_CHECK();
  return HCIR_OK;
}
