// Virtual measurements on gfx950.
//
//  * shapy_mesh_to_mesh_f32, scan path: the reference operator
//    mesh_mesh_intersect_cuda.mesh_to_mesh_forward (mesh_mesh_intersect.cpp:36-64,
//    mesh_mesh_intersect_cuda_op.cu:969-1079) for SMALL query meshes.  SHAPY only ever
//    calls it with a 2-triangle plane quad against the 20,908-triangle body
//    (body_measurements.py:86-97,137-139): building a tree to answer two queries costs more
//    than streaming the 752 KB of target triangles once, so for Q <= SCAN_MAX_Q each
//    (mesh, query triangle) pair gets one workgroup that scans the targets in index order
//    (HBM/L2-bound, coalesced) and compacts hits with wave ballots.  Larger query meshes go
//    through the LBVH in bvh.hip.
//  * shapy_body_measure_f32: BodyMeasurements.forward (body_measurements.py:99-246) fused:
//    one scan over the faces of v_shaped computes the signed-volume partial sums and the
//    plane/triangle hits of all three planes (no [B,F,3,3] triangle tensor is materialised),
//    then one workgroup per (mesh, plane) sorts the <= 2*max_coll points in LDS and walks the
//    2-D convex hull (monotone chain with exact float64 orientation tests) -- replacing the
//    per-mesh scipy/Qhull call and the D2H copy in front of it.
//
// This translation unit is compiled with -ffp-contract=off so that every float32 decision
// (SAT tolerance tests, barycentric range tests) is bit-identical with the CPU oracle.
#include "tri_tri.h"

namespace shapy {

constexpr int SCAN_THREADS = 256;

__device__ __forceinline__ Tri load_tri(const float *p) {
  Tri t;
  t.v0 = v3(p[0], p[1], p[2]);
  t.v1 = v3(p[3], p[4], p[5]);
  t.v2 = v3(p[6], p[7], p[8]);
  return t;
}

// block-wide exclusive prefix of a 0/1 flag in thread order; returns the block total
__device__ __forceinline__ int block_rank(bool flag, int &total, int *wave_cnt) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long m = __ballot(flag);
  const int below = __popcll(m & ((1ull << lane) - 1ull));
  if (lane == 0) wave_cnt[wave] = __popcll(m);
  __syncthreads();
  int off = 0, tot = 0;
  for (int w = 0; w < SCAN_THREADS / 64; ++w) {
    const int c = wave_cnt[w];
    if (w < wave) off += c;
    tot += c;
  }
  __syncthreads();
  total = tot;
  return off + below;
}

__global__ __launch_bounds__(SCAN_THREADS) void mesh_to_mesh_scan_kernel(
    const float *__restrict__ query, const float *__restrict__ target, int Q, int F, int MC,
    long long *__restrict__ faces_out, float *__restrict__ bcs_out, int *__restrict__ overflow) {
  __shared__ int wave_cnt[SCAN_THREADS / 64];
  const int q = blockIdx.x, b = blockIdx.y;
  const Tri qt = load_tri(query + ((long)b * Q + q) * 9);
  const float *tb = target + (long)b * F * 9;
  long long *fo = faces_out + ((long)b * Q + q) * MC;
  float *bo = bcs_out + ((long)b * Q + q) * MC * 6;
  int base = 0;
  for (int f0 = 0; f0 < F; f0 += SCAN_THREADS) {
    const int f = f0 + threadIdx.x;
    bool hit = false;
    Tri tt;
    if (f < F) {
      tt = load_tri(tb + (long)f * 9);
      hit = aabb_overlap(qt, tt) && tri_tri_sat(qt, tt);
    }
    int total;
    const int slot = base + block_rank(hit, total, wave_cnt);
    if (hit) {
      if (slot < MC) {
        V3 bc;
        const bool ok = tri_tri_point(qt, tt, bc);
        fo[slot] = f;
        if (ok) {
          float *o = bo + (long)slot * 6;
          o[0] = bc.x; o[1] = bc.y; o[2] = bc.z;
          o[3] = bc.x; o[4] = bc.y; o[5] = bc.z;
        }
      } else if (overflow) {
        atomicAdd(overflow, 1);
      }
    }
    base += total;
  }
}

// ------------------------------------------------------------------------------------------
// fused body measurements
// ------------------------------------------------------------------------------------------
struct Landmarks {
  int face[5];       // HeadTop, HeelLeft, chest, waist, hips
  float bc[5][3];
};

__device__ __forceinline__ float lm_coord(const float *vb, const int32_t *faces, const Landmarks &lm,
                                          int which, int axis) {
  const int f = lm.face[which];
  const float a = vb[(long)faces[f * 3 + 0] * 3 + axis];
  const float b = vb[(long)faces[f * 3 + 1] * 3 + axis];
  const float c = vb[(long)faces[f * 3 + 2] * 3 + axis];
  // (tri * bc.reshape(1,3,1)).sum(dim=1)   (body_measurements.py:130-131,185-195)
  return (a * lm.bc[which][0] + b * lm.bc[which][1]) + c * lm.bc[which][2];
}

__global__ __launch_bounds__(SCAN_THREADS) void measure_scan_kernel(
    const float *__restrict__ v_shaped, const int32_t *__restrict__ faces, int V, int F, int MC,
    Landmarks lm, int *__restrict__ counters, float *__restrict__ vol_partial,
    float4 *__restrict__ points) {
  __shared__ float hs[3];
  __shared__ double red[SCAN_THREADS / 64];
  const int b = blockIdx.y;
  const float *vb = v_shaped + (long)b * V * 3;
  if (threadIdx.x < 3) hs[threadIdx.x] = lm_coord(vb, faces, lm, 2 + threadIdx.x, 1);
  __syncthreads();
  const int f = blockIdx.x * SCAN_THREADS + threadIdx.x;
  double vol = 0.0;
  if (f < F) {
    const int i0 = faces[f * 3], i1 = faces[f * 3 + 1], i2 = faces[f * 3 + 2];
    Tri t;
    t.v0 = v3(vb[(long)i0 * 3], vb[(long)i0 * 3 + 1], vb[(long)i0 * 3 + 2]);
    t.v1 = v3(vb[(long)i1 * 3], vb[(long)i1 * 3 + 1], vb[(long)i1 * 3 + 2]);
    t.v2 = v3(vb[(long)i2 * 3], vb[(long)i2 * 3 + 1], vb[(long)i2 * 3 + 2]);
    // compute_mass (body_measurements.py:201-215), term order as written there
    const float x0 = t.v0.x, y0 = t.v0.y, z0 = t.v0.z, x1 = t.v1.x, y1 = t.v1.y, z1 = t.v1.z,
                x2 = t.v2.x, y2 = t.v2.y, z2 = t.v2.z;
    const float vv = -x2 * y1 * z0 + x1 * y2 * z0 + x2 * y0 * z1 - x0 * y2 * z1 - x1 * y0 * z2 +
                     x0 * y1 * z2;
    vol = (double)vv;
    const float ymin = fminf(y0, fminf(y1, y2)), ymax = fmaxf(y0, fmaxf(y1, y2));
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      const float h = hs[pl];
      if (!(ymin <= h && ymax >= h)) continue;
      // _get_plane_at_heights (body_measurements.py:86-97)
      const V3 c0 = v3(-1.f, h, -1.f), c1 = v3(1.f, h, -1.f), c2 = v3(1.f, h, 1.f),
               c3 = v3(-1.f, h, 1.f);
#pragma unroll
      for (int qi = 0; qi < 2; ++qi) {
        Tri q;
        q.v0 = c0;
        q.v1 = qi == 0 ? c1 : c2;
        q.v2 = qi == 0 ? c2 : c3;
        if (!(aabb_overlap(q, t) && tri_tri_sat(q, t))) continue;
        int *cnt = counters + ((long)b * 3 + pl) * 2 + qi;
        const int slot = atomicAdd(cnt, 1);
        if (slot >= MC) continue;
        V3 bc = v3(0.f, 0.f, 0.f);
        tri_tri_point(q, t, bc);
        // points = sum_k bc_k * tri_k (body_measurements.py:144-147)
        float4 pt;
        pt.x = (t.v0.x * bc.x + t.v1.x * bc.y) + t.v2.x * bc.z;
        pt.y = (t.v0.y * bc.x + t.v1.y * bc.y) + t.v2.y * bc.z;
        pt.z = (t.v0.z * bc.x + t.v1.z * bc.y) + t.v2.z * bc.z;
        pt.w = __int_as_float(f);
        points[(((long)b * 3 + pl) * 2 + qi) * MC + slot] = pt;
      }
    }
  }
  // deterministic block reduction of the signed volume
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) vol += __shfl_xor(vol, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = vol;
  __syncthreads();
  if (threadIdx.x == 0)
    vol_partial[(long)b * gridDim.x + blockIdx.x] = (float)(red[0] + red[1] + red[2] + red[3]);
}

constexpr int HULL_MAX = 1024;

__global__ __launch_bounds__(256) void measure_hull_kernel(
    const float *__restrict__ v_shaped, const int32_t *__restrict__ faces, int V, int MC,
    int n_vol_blocks, Landmarks lm, const int *__restrict__ counters,
    const float *__restrict__ vol_partial, const float4 *__restrict__ points,
    float *__restrict__ out, int *__restrict__ overflow) {
  __shared__ float px[HULL_MAX], py[HULL_MAX], pz[HULL_MAX];
  __shared__ int stack[HULL_MAX + 1];
  const int pl = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int *cnt = counters + ((long)b * 3 + pl) * 2;
  const int c0 = cnt[0], c1 = cnt[1];
  const int n0 = min(c0, MC), n1 = min(c1, MC);
  if (tid == 0 && overflow && (c0 > MC || c1 > MC)) atomicAdd(overflow, (c0 - n0) + (c1 - n1));
  // gather valid points: the reference keeps slots with collision_faces > 0 (:161), i.e. it
  // drops face 0 as well as the empty (-1) slots
  int npow = 1;
  while (npow < n0 + n1) npow <<= 1;
  if (npow < 2) npow = 2;
  for (int i = tid; i < npow; i += 256) {
    float x = INFINITY, y = 0.f, z = INFINITY;
    if (i < n0 + n1) {
      const int qi = i < n0 ? 0 : 1;
      const float4 p = points[(((long)b * 3 + pl) * 2 + qi) * MC + (i < n0 ? i : i - n0)];
      if (__float_as_int(p.w) > 0) { x = p.x; y = p.y; z = p.z; }
    }
    px[i] = x; py[i] = y; pz[i] = z;
  }
  __syncthreads();
  // bitonic sort by (x, z, y): invalid (+inf) entries sink to the end; order is a pure
  // function of the point set, so the atomics above do not make the result nondeterministic
  for (int k = 2; k <= npow; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < npow; i += 256) {
        const int l = i ^ j;
        if (l > i) {
          const bool up = (i & k) == 0;
          const float ax = px[i], az = pz[i], ay = py[i], bx = px[l], bz = pz[l], by = py[l];
          const bool gt = ax > bx || (ax == bx && (az > bz || (az == bz && ay > by)));
          if (gt == up) {
            px[i] = bx; pz[i] = bz; py[i] = by;
            px[l] = ax; pz[l] = az; py[l] = ay;
          }
        }
      }
      __syncthreads();
    }
  if (tid == 0) {
    int n = 0;
    while (n < npow && px[n] != INFINITY) ++n;
    float perim = 0.f;
    if (n >= 2) {
      // Andrew's monotone chain in the (x, z) plane; exact orientation in float64
      auto orient = [&](int o, int a, int c) -> double {
        return ((double)px[a] - (double)px[o]) * ((double)pz[c] - (double)pz[o]) -
               ((double)pz[a] - (double)pz[o]) * ((double)px[c] - (double)px[o]);
      };
      auto edge = [&](int a, int c) -> float {
        const float dx = px[c] - px[a], dy = py[c] - py[a], dz = pz[c] - pz[a];
        return sqrtf(dx * dx + dy * dy + dz * dz);
      };
      int m = 0;
      for (int i = 0; i < n; ++i) {
        while (m >= 2 && orient(stack[m - 2], stack[m - 1], i) <= 0.0) --m;
        stack[m++] = i;
      }
      const int lower = m + 1;
      for (int i = n - 2; i >= 0; --i) {
        while (m >= lower && orient(stack[m - 2], stack[m - 1], i) <= 0.0) --m;
        stack[m++] = i;
      }
      // stack[0..m-1] is the closed hull polygon (stack[m-1] == stack[0])
      for (int i = 0; i + 1 < m; ++i) perim += edge(stack[i], stack[i + 1]);
    }
    out[b * 5 + 2 + pl] = perim;
    if (pl == 0) {
      double vs = 0.0;
      for (int i = 0; i < n_vol_blocks; ++i) vs += (double)vol_partial[(long)b * n_vol_blocks + i];
      out[b * 5 + 0] = (float)(fabs(vs) / 6.0) * 985.0f;   // DENSITY (body_measurements.py:20)
      const float *vb = v_shaped + (long)b * V * 3;
      const float head = lm_coord(vb, faces, lm, 0, 1), heel = lm_coord(vb, faces, lm, 1, 1);
      out[b * 5 + 1] = fabsf(head - heel);                  // compute_height (:182-199)
    }
  }
}

int mesh_to_mesh_bvh(const float *query, const float *target, int B, int Q, int F, int MC,
                     long long *faces_out, float *bcs_out, void *ws, size_t ws_bytes,
                     int *overflow, hipStream_t s);
size_t mesh_to_mesh_bvh_workspace(int B, int Q, int F, int MC);

}  // namespace shapy

using namespace shapy;

static constexpr int SCAN_MAX_Q = 16;

extern "C" size_t shapy_mesh_to_mesh_workspace_bytes(int B, int Q, int F, int max_coll) {
  if (Q <= SCAN_MAX_Q) return 16;
  return mesh_to_mesh_bvh_workspace(B, Q, F, max_coll);
}

extern "C" int shapy_mesh_to_mesh_f32(const float *query, const float *target, int B, int Q, int F,
                                      int max_coll, int64_t *faces_out, float *bcs_out,
                                      void *workspace, size_t workspace_bytes,
                                      int32_t *overflow_out, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (B < 0 || Q < 0 || F < 0 || max_coll <= 0) return SHAPY_EINVAL;
  const size_t nslots = (size_t)B * Q * max_coll;
  if (nslots == 0) return SHAPY_OK;
  SHAPY_HIP_TRY(hipMemsetAsync(faces_out, 0xFF, nslots * sizeof(int64_t), s));   // -1
  SHAPY_HIP_TRY(hipMemsetAsync(bcs_out, 0, nslots * 6 * sizeof(float), s));
  if (overflow_out) SHAPY_HIP_TRY(hipMemsetAsync(overflow_out, 0, sizeof(int32_t), s));
  if (F == 0) return SHAPY_OK;
  if (Q <= SCAN_MAX_Q) {
    hipLaunchKernelGGL(mesh_to_mesh_scan_kernel, dim3(Q, B), dim3(SCAN_THREADS), 0, s, query, target,
                       Q, F, max_coll, (long long *)faces_out, bcs_out, overflow_out);
    return (int)hipGetLastError();
  }
  if (workspace_bytes < mesh_to_mesh_bvh_workspace(B, Q, F, max_coll)) return SHAPY_EWORKSPACE;
  return mesh_to_mesh_bvh(query, target, B, Q, F, max_coll, (long long *)faces_out, bcs_out,
                          workspace, workspace_bytes, overflow_out, s);
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

extern "C" size_t shapy_body_measure_workspace_bytes(int B, int F, int max_coll) {
  const size_t nblk = (F + SCAN_THREADS - 1) / SCAN_THREADS;
  return align_up((size_t)B * 6 * sizeof(int), 256) + align_up((size_t)B * nblk * sizeof(float), 256) +
         (size_t)B * 6 * max_coll * sizeof(float4);
}

extern "C" int shapy_body_measure_f32(const float *v_shaped, const int32_t *faces, int B, int V,
                                      int F, const int32_t *lm_face_host, const float *lm_bary_host,
                                      int max_coll, float *out, void *workspace,
                                      size_t workspace_bytes, int32_t *overflow_out, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (B <= 0) return SHAPY_OK;
  if (max_coll <= 0 || 2 * max_coll > HULL_MAX || F <= 0) return SHAPY_EINVAL;
  if (workspace_bytes < shapy_body_measure_workspace_bytes(B, F, max_coll)) return SHAPY_EWORKSPACE;
  Landmarks lm;
  for (int i = 0; i < 5; ++i) {
    if (lm_face_host[i] < 0 || lm_face_host[i] >= F) return SHAPY_EINVAL;
    lm.face[i] = lm_face_host[i];
    for (int k = 0; k < 3; ++k) lm.bc[i][k] = lm_bary_host[i * 3 + k];
  }
  const int nblk = (F + SCAN_THREADS - 1) / SCAN_THREADS;
  char *w = (char *)workspace;
  int *counters = (int *)w;
  w += align_up((size_t)B * 6 * sizeof(int), 256);
  float *vol = (float *)w;
  w += align_up((size_t)B * nblk * sizeof(float), 256);
  float4 *pts = (float4 *)w;
  SHAPY_HIP_TRY(hipMemsetAsync(counters, 0, (size_t)B * 6 * sizeof(int), s));
  if (overflow_out) SHAPY_HIP_TRY(hipMemsetAsync(overflow_out, 0, sizeof(int32_t), s));
  hipLaunchKernelGGL(measure_scan_kernel, dim3(nblk, B), dim3(SCAN_THREADS), 0, s, v_shaped, faces,
                     V, F, max_coll, lm, counters, vol, pts);
  SHAPY_HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(measure_hull_kernel, dim3(3, B), dim3(256), 0, s, v_shaped, faces, V, max_coll,
                     nblk, lm, counters, vol, pts, out, overflow_out);
  return (int)hipGetLastError();
}
