// GPU pre-processing of the regressor input: crop around the person, bilinear resize to the
// network resolution, clamp, normalise, HWC uint8 -> NCHW float32 -- one kernel for a ragged
// batch of full images.
//
// Replaces the CPU path of the reference's data pipeline for one sample:
//   read_img (/255, clip)                  regressor/human_shape/utils/img_utils.py:12-64
//   crop(): window copy with zero fill     regressor/human_shape/utils/transf_utils.py:53-84
//           cv2.resize(..., INTER_LINEAR)  transf_utils.py:95   (third-party OpenCV; its
//           float32 bilinear path is restated: half-pixel centres, source index clamped to
//           the window, horizontal pass then vertical pass)
//   ToTensor, Normalize (clamp to [0,1], (x - mean) / std)
//                                          data/transforms/transforms.py:613-624,710-733
// The integer crop window (ul, br) is computed on the host exactly like transf_utils.transform.
#include "common.h"

namespace shapy {

struct CropK {
  const unsigned char *images;
  const long long *img_off;
  const int *img_hw, *boxes;
  float *out;
  int S;
  float mean[3], inv_std[3];
};

__global__ __launch_bounds__(256) void crop_resize_normalize_kernel(CropK k) {
  const int b = blockIdx.y;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= k.S * k.S) return;
  const int dy = pix / k.S, dx = pix % k.S;
  const int H = k.img_hw[b * 2], W = k.img_hw[b * 2 + 1];
  const int ulx = k.boxes[b * 4], uly = k.boxes[b * 4 + 1];
  const int cw = k.boxes[b * 4 + 2] - ulx, ch = k.boxes[b * 4 + 3] - uly;
  const unsigned char *img = k.images + k.img_off[b];
  // cv2 INTER_LINEAR source coordinates (computed in double, like OpenCV)
  double fxd = (dx + 0.5) * ((double)cw / k.S) - 0.5, fyd = (dy + 0.5) * ((double)ch / k.S) - 0.5;
  int sx = (int)floor(fxd), sy = (int)floor(fyd);
  float fx = (float)(fxd - sx), fy = (float)(fyd - sy);
  if (sx < 0) { fx = 0.f; sx = 0; }
  if (sx >= cw - 1) { fx = 0.f; sx = cw - 1; }
  if (sy < 0) { fy = 0.f; sy = 0; }
  if (sy >= ch - 1) { fy = 0.f; sy = ch - 1; }
  const int sx1 = min(sx + 1, cw - 1), sy1 = min(sy + 1, ch - 1);
  auto px = [&](int y, int x, int c) -> float {
    const int iy = y + uly, ix = x + ulx;                 // window -> full image, zero fill
    if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) return 0.f;
    return fminf(fmaxf((float)img[((long)iy * W + ix) * 3 + c] / 255.0f, 0.f), 1.f);
  };
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float r0 = px(sy, sx, c) * (1.f - fx) + px(sy, sx1, c) * fx;
    const float r1 = px(sy1, sx, c) * (1.f - fx) + px(sy1, sx1, c) * fx;
    float v = r0 * (1.f - fy) + r1 * fy;
    v = fminf(fmaxf(v, 0.f), 1.f);
    k.out[(((long)b * 3 + c) * k.S + dy) * k.S + dx] = (v - k.mean[c]) * k.inv_std[c];
  }
}

}  // namespace shapy

extern "C" int shapy_crop_resize_normalize_u8(const unsigned char *images, const int64_t *img_off,
                                              const int32_t *img_hw, const int32_t *boxes,
                                              float *out, int B, int S, const float *mean_host,
                                              const float *std_host, void *stream) {
  if (B <= 0) return SHAPY_OK;
  if (S <= 0) return SHAPY_EINVAL;
  shapy::CropK k;
  k.images = images; k.img_off = (const long long *)img_off; k.img_hw = img_hw; k.boxes = boxes;
  k.out = out; k.S = S;
  for (int c = 0; c < 3; ++c) { k.mean[c] = mean_host[c]; k.inv_std[c] = 1.0f / std_host[c]; }
  hipLaunchKernelGGL(shapy::crop_resize_normalize_kernel, dim3((S * S + 255) / 256, B), dim3(256), 0,
                     (hipStream_t)stream, k);
  return (int)hipGetLastError();
}
