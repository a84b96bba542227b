// LBVH path of the mesh-mesh intersection operator (large query meshes) -- see bvh.hip notes.
#include "tri_tri.h"

namespace shapy {

size_t mesh_to_mesh_bvh_workspace(int B, int Q, int F, int MC) { return 16; }

int mesh_to_mesh_bvh(const float *query, const float *target, int B, int Q, int F, int MC,
                     long long *faces_out, float *bcs_out, void *ws, size_t ws_bytes,
                     int *overflow, hipStream_t s) {
  return SHAPY_EINVAL;
}

}  // namespace shapy
