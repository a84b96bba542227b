// extern "C" entry points that are thin wrappers over the C++ launchers.
#include "common.h"

namespace shapy {
int hrnet_run(const ShapyOp *ops, int n_ops, const void *weights, const float *input, void *ws,
              int64_t ws_per_img, float *features_out, int B, int H, int W, int multi_stream,
              int dtype, hipStream_t main);
}

extern "C" int shapy_abi_version(void) { return 7; }
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
                               int64_t ws_elems_per_image, float *features_out, int B, int H, int W,
                               int multi_stream, int dtype, void *stream) {
  if (!ops || n_ops <= 0 || B <= 0 || (H % 32) || (W % 32)) return SHAPY_EINVAL;
  if (dtype != SHAPY_DTYPE_F32 && dtype != SHAPY_DTYPE_BF16 && dtype != SHAPY_DTYPE_F32X6)
    return SHAPY_EINVAL;
  return shapy::hrnet_run(ops, n_ops, weights, input_nchw, workspace, ws_elems_per_image,
                          features_out, B, H, W, multi_stream, dtype, (hipStream_t)stream);
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
                                        int64_t ws_elems_per_image, float *features_out, int B,
                                        int H, int W, int multi_stream, int dtype,
                                        void **graph_out) {
  if (!graph_out) return SHAPY_EINVAL;
  *graph_out = nullptr;
  if (!ops || n_ops <= 0 || B <= 0 || (H % 32) || (W % 32)) return SHAPY_EINVAL;
  if (dtype != SHAPY_DTYPE_F32 && dtype != SHAPY_DTYPE_BF16 && dtype != SHAPY_DTYPE_F32X6)
    return SHAPY_EINVAL;
  hipStream_t cs = nullptr;
  SHAPY_HIP_TRY(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
  // a plain (uncaptured) pass first: creates the side streams / events and validates the ops
  int rc = shapy::hrnet_run(ops, n_ops, weights, input_nchw, workspace, ws_elems_per_image,
                            features_out, B, H, W, multi_stream, dtype, cs);
  if (rc == 0) rc = (int)hipStreamSynchronize(cs);
  hipGraph_t graph = nullptr;
  if (rc == 0) rc = (int)hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
  if (rc == 0) {
    rc = shapy::hrnet_run(ops, n_ops, weights, input_nchw, workspace, ws_elems_per_image,
                          features_out, B, H, W, multi_stream, dtype, cs);
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

// ---- the EVENT-DRIVEN plan as an explicitly built graph (no multi-stream capture) -------------
// Capturing the event-driven multi-stream forward crashes inside graph creation on ROCm 7.2.  Here
// the op list is captured ONCE on a single stream (a linear chain of kernel nodes: that capture
// works), the kernel nodes' parameters are read back, and a second graph is built from them by
// hand whose edges are exactly the executor's ordering rules (csrc/hrnet_ops.hip): the ops of a
// lane in plan order, an op's sig / wait event slots as edges from the producing op, a barrier as
// an empty join node that every lane continues from.  The workspace packing of the plan
// (hrnet.py: _Plan.happens_before / allocate) assumes those rules and nothing else.
// Failure codes -1001 .. -10NN number the graph API calls below in source order (diagnostics).
#include <vector>

extern "C" int shapy_hrnet_graph_create_explicit(const ShapyOp *ops, int n_ops, const void *weights,
                                                 const float *input_nchw, void *workspace,
                                                 int64_t ws_elems_per_image, float *features_out,
                                                 int B, int H, int W, int dtype, void **graph_out) {
  if (!graph_out) return SHAPY_EINVAL;
  *graph_out = nullptr;
  if (!ops || n_ops <= 0 || B <= 0 || (H % 32) || (W % 32)) return SHAPY_EINVAL;
  if (dtype != SHAPY_DTYPE_F32 && dtype != SHAPY_DTYPE_BF16 && dtype != SHAPY_DTYPE_F32X6)
    return SHAPY_EINVAL;
  // launch units in plan order: single ops and launch groups
  struct Unit { int first, n; };
  std::vector<Unit> units;
  for (int i = 0; i < n_ops;) {
    const int n = (ops[i].type == SHAPY_OP_CONV && ops[i].group > 1) ? ops[i].group : 1;
    if (i + n > n_ops || ops[i].lane < 0 || ops[i].lane > 6) return SHAPY_EINVAL;
    units.push_back({i, n});
    i += n;
  }
  hipStream_t cs = nullptr;
  SHAPY_HIP_TRY(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
  // a plain single-stream pass first (validates the ops, loads the code objects), then the capture
  int rc = shapy::hrnet_run(ops, n_ops, weights, input_nchw, workspace, ws_elems_per_image,
                            features_out, B, H, W, 0, dtype, cs);
  if (rc == 0) rc = (int)hipStreamSynchronize(cs);
  hipGraph_t chain = nullptr, dag = nullptr;
  if (rc == 0) rc = (int)hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
  if (rc == 0) {
    rc = shapy::hrnet_run(ops, n_ops, weights, input_nchw, workspace, ws_elems_per_image,
                          features_out, B, H, W, 0, dtype, cs);
    const int rc2 = (int)hipStreamEndCapture(cs, &chain);
    if (rc == 0) rc = rc2;
  }
  (void)hipStreamDestroy(cs);
  HrnetGraph *g = nullptr;
  auto fail = [&](int code) {
    if (chain) (void)hipGraphDestroy(chain);
    if (dag) (void)hipGraphDestroy(dag);
    delete g;
    return code;
  };
  if (rc) return fail(rc);
  // the captured chain, node by node in execution order
  std::vector<hipGraphNode_t> order;
  {
    size_t n_root = 0;
    if (hipGraphGetRootNodes(chain, nullptr, &n_root) != hipSuccess || n_root != 1)
      return fail(-1001);
    hipGraphNode_t cur = nullptr;
    if (hipGraphGetRootNodes(chain, &cur, &n_root) != hipSuccess) return fail(-1002);
    while (cur) {
      order.push_back(cur);
      size_t n_dep = 0;
      if (hipGraphNodeGetDependentNodes(cur, nullptr, &n_dep) != hipSuccess) return fail(-1003);
      if (n_dep == 0) break;
      if (n_dep != 1) return fail(-1004);
      hipGraphNode_t nxt = nullptr;
      if (hipGraphNodeGetDependentNodes(cur, &nxt, &n_dep) != hipSuccess) return fail(-1005);
      cur = nxt;
    }
  }
  if (order.size() != units.size()) return fail(-1006);      // one kernel per launch unit
  if (hipGraphCreate(&dag, 0) != hipSuccess) return fail(-1007);
  hipGraphNode_t lane_tail[7] = {}, ev_node[64] = {};
  for (size_t u = 0; u < units.size(); ++u) {
    const ShapyOp &o = ops[units[u].first];
    hipGraphNodeType ty;
    if (hipGraphNodeGetType(order[u], &ty) != hipSuccess || ty != hipGraphNodeTypeKernel)
      return fail(-1008);
    hipKernelNodeParams kp;
    if (hipGraphKernelNodeGetParams(order[u], &kp) != hipSuccess) return fail(-1009);
    std::vector<hipGraphNode_t> deps;
    auto add_dep = [&](hipGraphNode_t n) {
      if (!n) return;
      for (hipGraphNode_t d : deps)
        if (d == n) return;
      deps.push_back(n);
    };
    if (o.barrier_before) {
      std::vector<hipGraphNode_t> tails;
      for (int l = 0; l < 7; ++l) {
        bool dup = !lane_tail[l];
        for (hipGraphNode_t t : tails) dup |= t == lane_tail[l];
        if (!dup) tails.push_back(lane_tail[l]);
      }
      if (!tails.empty()) {
        hipGraphNode_t join = nullptr;
        if (hipGraphAddEmptyNode(&join, dag, tails.data(), tails.size()) != hipSuccess)
          return fail(-1010);
        for (int l = 0; l < 7; ++l) lane_tail[l] = join;
      }
    }
    // lane order; a side lane that has not run yet forks from the main lane's position
    add_dep(lane_tail[o.lane] ? lane_tail[o.lane] : lane_tail[0]);
    for (int m = 0; m < units[u].n; ++m)
      for (int w = 0; w < 3; ++w) {
        const int e = ops[units[u].first + m].wait[w];
        if (e >= 64) return fail(-1011);
        if (e >= 0) add_dep(ev_node[e]);
      }
    hipGraphNode_t node = nullptr;
    if (hipGraphAddKernelNode(&node, dag, deps.data(), deps.size(), &kp) != hipSuccess)
      return fail(-1012);
    lane_tail[o.lane] = node;
    for (int m = 0; m < units[u].n; ++m) {
      const int sgl = ops[units[u].first + m].sig;
      if (sgl >= 64) return fail(-1013);
      if (sgl >= 0) ev_node[sgl] = node;
    }
  }
  g = new HrnetGraph();
  if (hipGraphInstantiate(&g->exec, dag, nullptr, nullptr, 0) != hipSuccess) return fail(-1014);
  (void)hipGraphDestroy(chain);
  (void)hipGraphDestroy(dag);
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
