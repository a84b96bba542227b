// extern "C" entry points that are thin wrappers over the C++ launchers.
#include "common.h"

namespace shapy {
int hrnet_run(const ShapyOp *ops, int n_ops, const void *weights, const float *input, void *ws,
              int64_t ws_per_img, float *features_out, int B, int H, int W, int multi_stream,
              int dtype, hipStream_t main);
}

extern "C" int shapy_abi_version(void) { return 2; }
extern "C" const char *shapy_build_arch(void) { return "gfx950"; }

extern "C" int shapy_conv2d(const ShapyConv *d, void *stream) {
  if (!d) return SHAPY_EINVAL;
  return shapy::conv2d(*d, (hipStream_t)stream);
}

extern "C" int shapy_hrnet_run(const ShapyOp *ops, int n_ops, const void *weights,
                               const float *input_nchw, void *workspace,
                               int64_t ws_elems_per_image, float *features_out, int B, int H, int W,
                               int multi_stream, int dtype, void *stream) {
  if (!ops || n_ops <= 0 || B <= 0 || (H % 32) || (W % 32)) return SHAPY_EINVAL;
  if (dtype != SHAPY_DTYPE_F32 && dtype != SHAPY_DTYPE_BF16 && dtype != SHAPY_DTYPE_F32X6)
    return SHAPY_EINVAL;
  return shapy::hrnet_run(ops, n_ops, weights, input_nchw, workspace, ws_elems_per_image,
                          features_out, B, H, W, multi_stream, dtype, (hipStream_t)stream);
}
