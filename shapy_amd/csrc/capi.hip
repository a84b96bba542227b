// extern "C" entry points that are thin wrappers over the C++ launchers.
#include "common.h"

namespace shapy {
int hrnet_run(const ShapyOp *ops, int n_ops, const void *weights, const float *input, void *ws,
              int64_t ws_per_img, int32_t *counters, int64_t cnt_per_img, float *features_out, int B,
              int H, int W, int multi_stream, int dtype, hipStream_t main);
int hrnet_lane_stream(int lane, hipStream_t *out);
}

extern "C" int shapy_abi_version(void) { return 8; }
extern "C" const char *shapy_build_arch(void) { return "gfx950"; }

extern "C" int shapy_conv2d(const ShapyConv *d, void *stream) {
  if (!d) return SHAPY_EINVAL;
  return shapy::conv2d(*d, (hipStream_t)stream);
}

extern "C" int shapy_conv2d_group(const ShapyConv *descs, int n, void *stream) {
  if (!descs) return SHAPY_EINVAL;
  return shapy::conv2d_group(descs, n, (hipStream_t)stream);
}

extern "C" int shapy_hrnet_run(const ShapyOp *ops, int n_ops, const void *weights,
                               const float *input_nchw, void *workspace,
                               int64_t ws_elems_per_image, int32_t *counters, int64_t cnt_per_image,
                               float *features_out, int B, int H, int W, int multi_stream, int dtype,
                               void *stream) {
  if (!ops || n_ops <= 0 || B <= 0 || (H % 32) || (W % 32)) return SHAPY_EINVAL;
  if (dtype != SHAPY_DTYPE_F32 && dtype != SHAPY_DTYPE_BF16 && dtype != SHAPY_DTYPE_F32X6)
    return SHAPY_EINVAL;
  if (cnt_per_image < 0 || (cnt_per_image > 0 && !counters)) return SHAPY_EINVAL;
  return shapy::hrnet_run(ops, n_ops, weights, input_nchw, workspace, ws_elems_per_image, counters,
                          cnt_per_image, features_out, B, H, W, multi_stream, dtype, (hipStream_t)stream);
}

extern "C" int shapy_hrnet_lane_stream(int lane, void **stream_out) {
  return shapy::hrnet_lane_stream(lane, reinterpret_cast<hipStream_t *>(stream_out));
}

// ---- the same op list captured once into a hipGraph and replayed --------------------------
// 330 launches + their fork/join events cost ~1 ms of host time per forward; for small
// batches (the demo runs one person at a time) that is longer than the GPU work.  Capture
// records the launches of shapy::hrnet_run -- including the side-stream branches, which join
// the capture through the fork event -- into one executable graph with the pointers baked in.
struct HrnetGraph {
  hipGraphExec_t exec = nullptr;
};

extern "C" int shapy_hrnet_graph_create(const ShapyOp *ops, int n_ops, const void *weights,
                                        const float *input_nchw, void *workspace,
                                        int64_t ws_elems_per_image, int32_t *counters,
                                        int64_t cnt_per_image, float *features_out, int B, int H,
                                        int W, int multi_stream, int dtype, void **graph_out) {
  if (!graph_out) return SHAPY_EINVAL;
  *graph_out = nullptr;
  if (!ops || n_ops <= 0 || B <= 0 || (H % 32) || (W % 32)) return SHAPY_EINVAL;
  if (dtype != SHAPY_DTYPE_F32 && dtype != SHAPY_DTYPE_BF16 && dtype != SHAPY_DTYPE_F32X6)
    return SHAPY_EINVAL;
  if (cnt_per_image < 0 || (cnt_per_image > 0 && !counters)) return SHAPY_EINVAL;
  hipStream_t cs = nullptr;
  SHAPY_HIP_TRY(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
  // a plain (uncaptured) pass first: creates the side streams / events and validates the ops
  int rc = shapy::hrnet_run(ops, n_ops, weights, input_nchw, workspace, ws_elems_per_image, counters,
                            cnt_per_image, features_out, B, H, W, multi_stream, dtype, cs);
  if (rc == 0) rc = (int)hipStreamSynchronize(cs);
  hipGraph_t graph = nullptr;
  if (rc == 0) rc = (int)hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
  if (rc == 0) {
    rc = shapy::hrnet_run(ops, n_ops, weights, input_nchw, workspace, ws_elems_per_image, counters,
                          cnt_per_image, features_out, B, H, W, multi_stream, dtype, cs);
    const int rc2 = (int)hipStreamEndCapture(cs, &graph);
    if (rc == 0) rc = rc2;
  }
  HrnetGraph *g = nullptr;
  if (rc == 0) {
    g = new HrnetGraph();
    rc = (int)hipGraphInstantiate(&g->exec, graph, nullptr, nullptr, 0);
  }
  if (graph) (void)hipGraphDestroy(graph);
  (void)hipStreamDestroy(cs);
  if (rc != 0) {
    delete g;
    return rc;
  }
  *graph_out = g;
  return SHAPY_OK;
}

extern "C" int shapy_hrnet_graph_launch(void *graph, void *stream) {
  if (!graph) return SHAPY_EINVAL;
  SHAPY_HIP_TRY(hipGraphLaunch(static_cast<HrnetGraph *>(graph)->exec, (hipStream_t)stream));
  return SHAPY_OK;
}

extern "C" int shapy_hrnet_graph_destroy(void *graph) {
  if (!graph) return SHAPY_OK;
  HrnetGraph *g = static_cast<HrnetGraph *>(graph);
  if (g->exec) (void)hipGraphExecDestroy(g->exec);
  delete g;
  return SHAPY_OK;
}
