// extern "C" entry points that are thin wrappers over the C++ launchers.
#include "common.h"

namespace shapy {
int hrnet_run_f32(const ShapyOp *ops, int n_ops, const float *weights, const float *input,
                  float *ws, int64_t ws_per_img, float *features_out, int B, int H, int W,
                  int multi_stream, hipStream_t main);
}

extern "C" int shapy_abi_version(void) { return 1; }
extern "C" const char *shapy_build_arch(void) { return "gfx950"; }

extern "C" int shapy_conv2d_f32(const ShapyConv *d, void *stream) {
  if (!d) return SHAPY_EINVAL;
  return shapy::conv2d_f32(*d, (hipStream_t)stream);
}

extern "C" int shapy_hrnet_run_f32(const ShapyOp *ops, int n_ops, const float *weights,
                                   const float *input_nchw, float *workspace,
                                   int64_t ws_floats_per_image, float *features_out, int B, int H,
                                   int W, int multi_stream, void *stream) {
  if (!ops || n_ops <= 0 || B <= 0 || (H % 32) || (W % 32)) return SHAPY_EINVAL;
  return shapy::hrnet_run_f32(ops, n_ops, weights, input_nchw, workspace, ws_floats_per_image,
                              features_out, B, H, W, multi_stream, (hipStream_t)stream);
}
