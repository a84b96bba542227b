// LBVH path of the mesh-mesh intersection operator (query meshes with more than SCAN_MAX_Q
// triangles): the gfx950 counterpart of every kernel of the reference's
// mesh-mesh-intersection/src/mesh_mesh_intersect_cuda_op.cu
//     compute_tri_bboxes :140-149, thrust::reduce(MergeAABB) :863-864,
//     compute_morton_codes :613-668, thrust::sort_by_key :911-912,
//     BuildRadixTree :670-765 (Karras 2012), create_hierarchy :767-821,
//     findPotentialCollisions / traverse_bvh :520-609
// re-designed for MI355X instead of translated:
//   * the reference loops over the batch on the host and synchronises the device 9 times per
//     mesh; here the whole batch is 3 launches on the caller's stream and nothing syncs;
//   * BUILD = one 1024-thread workgroup per mesh does scene bounds -> Morton keys -> sort ->
//     radix tree -> AABB refit back to back (the 20,908-triangle SMPL-X mesh keeps its working
//     set of ~1.5 MB inside one XCD's L2; B meshes give B-way parallelism over 256 CUs).  Keys
//     are 64-bit (morton << 32 | face) and therefore unique: no duplicate-key tie-break
//     (:683-687) is needed and the sort may be an unstable bitonic network;
//   * TRAVERSE is wave-cooperative: each lane of a wave walks the tree for its own query
//     triangle (cheap, divergent AABB tests, per-lane stack in LDS), but the expensive narrow
//     phase -- 11-axis SAT + intersection point (tri_tri.h) -- is deferred: overlapping
//     (query, leaf) pairs are appended to a wave-shared LDS queue and the queue is drained by
//     all 64 lanes together, so the narrow phase always runs on full waves regardless of how
//     unevenly hits are distributed over the queries;
//   * FINISH sorts each query's hits by target face index, which makes the result independent
//     of traversal and atomic order (identical to the scan path in measure.hip and to the CPU
//     oracle as long as no query exceeds max_collisions).
// Compiled with -ffp-contract=off like measure.hip (bit-identical narrow-phase decisions).
#include "tri_tri.h"

namespace shapy {

constexpr int BUILD_THREADS = 1024;
constexpr int STACK_DEPTH = 64;     // reference STACK_SIZE (:45-47)
constexpr int QUEUE_CAP = 512;      // candidate pairs buffered per wave

struct BvhMesh {           // per-mesh slices of the workspace
  unsigned long long *keys;   // [Fpad]
  int *left, *right;          // [F-1]  child: >= 0 internal node, < 0 leaf ~c
  int *parent_i, *parent_l;   // [F-1], [F]
  float *box_i, *box_l;       // [F-1][6], [F][6]   (min xyz, max xyz)
  int *flag;                  // [F-1] refit arrival counters
};

struct BvhLayout {
  size_t keys, left, right, parent_i, parent_l, box_i, box_l, flag, per_mesh, hits_cnt, total;
  int Fpad;
};

static inline size_t up256(size_t x) { return (x + 255) / 256 * 256; }

static BvhLayout bvh_layout(int B, int Q, int F) {
  BvhLayout L;
  int Fpad = 1;
  while (Fpad < F) Fpad <<= 1;
  L.Fpad = Fpad;
  size_t o = 0;
  const size_t ni = F > 1 ? F - 1 : 1;
  L.keys = o; o += up256((size_t)Fpad * 8);
  L.left = o; o += up256(ni * 4);
  L.right = o; o += up256(ni * 4);
  L.parent_i = o; o += up256(ni * 4);
  L.parent_l = o; o += up256((size_t)F * 4);
  L.box_i = o; o += up256(ni * 24);
  L.box_l = o; o += up256((size_t)F * 24);
  L.flag = o; o += up256(ni * 4);
  L.per_mesh = o;
  L.hits_cnt = (size_t)B * L.per_mesh;
  L.total = L.hits_cnt + up256((size_t)B * Q * 4);
  return L;
}

__device__ __forceinline__ BvhMesh mesh_ptrs(char *ws, const BvhLayout &L, int b) {
  char *p = ws + (size_t)b * L.per_mesh;
  BvhMesh m;
  m.keys = (unsigned long long *)(p + L.keys);
  m.left = (int *)(p + L.left);
  m.right = (int *)(p + L.right);
  m.parent_i = (int *)(p + L.parent_i);
  m.parent_l = (int *)(p + L.parent_l);
  m.box_i = (float *)(p + L.box_i);
  m.box_l = (float *)(p + L.box_l);
  m.flag = (int *)(p + L.flag);
  return m;
}

__device__ __forceinline__ unsigned expand_bits(unsigned v) {   // 10 bits -> 30, 2 zeros between
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}

__device__ __forceinline__ int delta(const unsigned long long *keys, int F, int i, int j) {
  if (j < 0 || j >= F) return -1;
  return __clzll(keys[i] ^ keys[j]);
}

__global__ __launch_bounds__(BUILD_THREADS) void bvh_build_kernel(const float *__restrict__ target,
                                                                   int F, BvhLayout L, char *ws) {
  __shared__ float red[6][BUILD_THREADS / 64];
  __shared__ float scene[6];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float *tb = target + (size_t)b * F * 9;
  BvhMesh m = mesh_ptrs(ws, L, b);

  // ---- 1. scene bounds = union of the triangle boxes (:140-149, :863-864) ----
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int f = tid; f < F; f += BUILD_THREADS) {
    const float *t = tb + (size_t)f * 9;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float lo = fminf(t[k], fminf(t[3 + k], t[6 + k]));
      const float hi = fmaxf(t[k], fmaxf(t[3 + k], t[6 + k]));
      mn[k] = fminf(mn[k], lo);
      mx[k] = fmaxf(mx[k], hi);
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    for (int o = 32; o > 0; o >>= 1) {
      mn[k] = fminf(mn[k], __shfl_xor(mn[k], o, 64));
      mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], o, 64));
    }
    if ((tid & 63) == 0) { red[k][tid >> 6] = mn[k]; red[3 + k][tid >> 6] = mx[k]; }
  }
  __syncthreads();
  if (tid < 6) {
    float v = red[tid][0];
    for (int w = 1; w < BUILD_THREADS / 64; ++w)
      v = tid < 3 ? fminf(v, red[tid][w]) : fmaxf(v, red[tid][w]);
    scene[tid] = v;
  }
  __syncthreads();

  // ---- 2. 30-bit Morton code of the centroid (:613-668), 64-bit unique keys ----
  for (int f = tid; f < L.Fpad; f += BUILD_THREADS) {
    unsigned long long key = ~0ull;
    if (f < F) {
      const float *t = tb + (size_t)f * 9;
      unsigned code[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float c = (t[k] + t[3 + k] + t[6 + k]) / 3.0f;
        float x = (c - scene[k]) / (scene[3 + k] - scene[k]);
        x = fminf(fmaxf(x * 1024.0f, 0.0f), 1023.0f);       // NaN (flat scene) -> 0
        code[k] = expand_bits((unsigned)x);
      }
      const unsigned mc = code[0] * 4 + code[1] * 2 + code[2];
      key = ((unsigned long long)mc << 32) | (unsigned)f;
    }
    m.keys[f] = key;
  }
  __syncthreads();

  // ---- 3. bitonic sort of the keys (replaces thrust::sort_by_key :911-912) ----
  for (int k = 2; k <= L.Fpad; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < L.Fpad; i += BUILD_THREADS) {
        const int l = i ^ j;
        if (l > i) {
          const unsigned long long a = m.keys[i], c = m.keys[l];
          if ((a > c) == ((i & k) == 0)) { m.keys[i] = c; m.keys[l] = a; }
        }
      }
      __syncthreads();
    }

  // ---- 4. Karras radix tree over the sorted keys (:696-765) ----
  for (int i = tid; i < F - 1; i += BUILD_THREADS) {
    const int d = (delta(m.keys, F, i, i + 1) - delta(m.keys, F, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(m.keys, F, i, i - d);
    int lmax = 2;
    while (delta(m.keys, F, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
      if (delta(m.keys, F, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(m.keys, F, i, j);
    int s = 0;
    for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
      if (delta(m.keys, F, i, i + (s + t) * d) > dnode) s += t;
      if (t <= 1) break;
    }
    const int gamma = i + s * d + min(d, 0);
    int lc, rc;
    if (min(i, j) == gamma) { lc = ~gamma; m.parent_l[gamma] = i; }
    else { lc = gamma; m.parent_i[gamma] = i; }
    if (max(i, j) == gamma + 1) { rc = ~(gamma + 1); m.parent_l[gamma + 1] = i; }
    else { rc = gamma + 1; m.parent_i[gamma + 1] = i; }
    m.left[i] = lc;
    m.right[i] = rc;
    m.flag[i] = 0;
  }
  if (tid == 0 && F > 1) m.parent_i[0] = -1;
  __syncthreads();

  // ---- 5. leaf boxes in sorted order + bottom-up refit (:767-821) ----
  for (int k = tid; k < F; k += BUILD_THREADS) {
    const int f = (int)(m.keys[k] & 0xffffffffu);
    const float *t = tb + (size_t)f * 9;
    float lo[3], hi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = fminf(t[a], fminf(t[3 + a], t[6 + a]));
      hi[a] = fmaxf(t[a], fmaxf(t[3 + a], t[6 + a]));
    }
    float *bl = m.box_l + (size_t)k * 6;
    bl[0] = lo[0]; bl[1] = lo[1]; bl[2] = lo[2]; bl[3] = hi[0]; bl[4] = hi[1]; bl[5] = hi[2];
  }
  __syncthreads();
  if (F < 2) return;
  for (int k = tid; k < F; k += BUILD_THREADS) {
    int node = m.parent_l[k];
    while (true) {
      __threadfence();                                  // publish this subtree's boxes
      if (atomicAdd(&m.flag[node], 1) == 0) break;      // first child to arrive leaves
      __threadfence();                                  // second: see the sibling's box
      const int lc = m.left[node], rc = m.right[node];
      const volatile float *a = lc < 0 ? m.box_l + (size_t)(~lc) * 6 : m.box_i + (size_t)lc * 6;
      const volatile float *c = rc < 0 ? m.box_l + (size_t)(~rc) * 6 : m.box_i + (size_t)rc * 6;
      float *o = m.box_i + (size_t)node * 6;
      o[0] = fminf(a[0], c[0]); o[1] = fminf(a[1], c[1]); o[2] = fminf(a[2], c[2]);
      o[3] = fmaxf(a[3], c[3]); o[4] = fmaxf(a[4], c[4]); o[5] = fmaxf(a[5], c[5]);
      if (node == 0) break;
      node = m.parent_i[node];
    }
  }
}

__device__ __forceinline__ bool box_overlap(const float *q, const float *n) {
  // checkOverlap (:363-373), closed comparisons
  return q[0] <= n[3] && q[3] >= n[0] && q[1] <= n[4] && q[4] >= n[1] && q[2] <= n[5] && q[5] >= n[2];
}

__global__ __launch_bounds__(64) void bvh_traverse_kernel(const float *__restrict__ query,
                                                          const float *__restrict__ target, int Q,
                                                          int F, int MC, BvhLayout L, char *ws,
                                                          long long *__restrict__ faces_out,
                                                          float *__restrict__ bcs_out,
                                                          int *__restrict__ overflow) {
  __shared__ int stack[STACK_DEPTH][64];       // [level][lane]: conflict-free per-lane stacks
  __shared__ float qtri[64][9];
  __shared__ int queue[QUEUE_CAP];             // (lane << 24) | sorted-leaf index
  __shared__ int q_count;
  const int lane = threadIdx.x, b = blockIdx.y;
  const int q = blockIdx.x * 64 + lane;
  const BvhMesh m = mesh_ptrs(ws, L, b);
  int *hit_cnt = (int *)(ws + L.hits_cnt) + (size_t)b * Q;
  const float *tb = target + (size_t)b * F * 9;

  float qb[6] = {0, 0, 0, 0, 0, 0};
  bool active = q < Q;
  if (active) {
    const float *t = query + ((size_t)b * Q + q) * 9;
#pragma unroll
    for (int k = 0; k < 9; ++k) qtri[lane][k] = t[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      qb[k] = fminf(t[k], fminf(t[3 + k], t[6 + k]));
      qb[3 + k] = fmaxf(t[k], fmaxf(t[3 + k], t[6 + k]));
    }
  }
  if (lane == 0) q_count = 0;
  int sp = 0, node = 0;                         // root = internal node 0
  __syncthreads();

  auto drain = [&]() {
    __syncthreads();
    const int n = q_count;
    for (int i = lane; i < n; i += 64) {
      const int e = queue[i];
      const int owner = (unsigned)e >> 24, leaf = e & 0xffffff;
      const int f = (int)(m.keys[leaf] & 0xffffffffu);
      Tri qt, tt;
      const float *a = qtri[owner];
      qt.v0 = v3(a[0], a[1], a[2]); qt.v1 = v3(a[3], a[4], a[5]); qt.v2 = v3(a[6], a[7], a[8]);
      const float *t = tb + (size_t)f * 9;
      tt.v0 = v3(t[0], t[1], t[2]); tt.v1 = v3(t[3], t[4], t[5]); tt.v2 = v3(t[6], t[7], t[8]);
      if (!tri_tri_sat(qt, tt)) continue;
      const int qq = blockIdx.x * 64 + owner;
      const int slot = atomicAdd(&hit_cnt[qq], 1);
      if (slot >= MC) { if (overflow) atomicAdd(overflow, 1); continue; }
      const size_t o = ((size_t)b * Q + qq) * MC + slot;
      faces_out[o] = f;
      V3 bc;
      if (tri_tri_point(qt, tt, bc)) {
        float *bo = bcs_out + o * 6;
        bo[0] = bc.x; bo[1] = bc.y; bo[2] = bc.z; bo[3] = bc.x; bo[4] = bc.y; bo[5] = bc.z;
      }
    }
    __syncthreads();
    if (lane == 0) q_count = 0;
    __syncthreads();
  };

  while (__any(active)) {
    if (active) {
      const int ch[2] = {m.left[node], m.right[node]};
      int next = -1;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int id = ch[c];
        const float *nb = id < 0 ? m.box_l + (size_t)(~id) * 6 : m.box_i + (size_t)id * 6;
        float nbx[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) nbx[k] = nb[k];
        if (!box_overlap(qb, nbx)) continue;
        if (id < 0) {
          const int pos = atomicAdd(&q_count, 1);      // LDS atomic; capacity kept by the drain rule
          queue[pos] = (lane << 24) | (~id);
        } else if (next < 0) {
          next = id;
        } else if (sp < STACK_DEPTH) {
          stack[sp++][lane] = id;
        }
      }
      if (next >= 0) node = next;
      else if (sp > 0) node = stack[--sp][lane];
      else active = false;
    }
    // every round adds at most 2 entries per lane: drain before the queue could overflow
    __syncthreads();
    if (q_count > QUEUE_CAP - 128) drain();
  }
  drain();
}

// hits of one query sorted by target face (insertion sort; typical n is a few dozen)
__global__ void bvh_sort_hits_kernel(int Q, int MC, BvhLayout L, char *ws,
                                     long long *__restrict__ faces, float *__restrict__ bcs,
                                     long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int *hit_cnt = (const int *)(ws + L.hits_cnt);
  const int n = min(hit_cnt[i], MC);
  long long *f = faces + i * MC;
  float *c = bcs + i * MC * 6;
  for (int a = 1; a < n; ++a) {
    const long long key = f[a];
    float v[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] = c[a * 6 + k];
    int p = a - 1;
    while (p >= 0 && f[p] > key) {
      f[p + 1] = f[p];
#pragma unroll
      for (int k = 0; k < 6; ++k) c[(p + 1) * 6 + k] = c[p * 6 + k];
      --p;
    }
    f[p + 1] = key;
#pragma unroll
    for (int k = 0; k < 6; ++k) c[(p + 1) * 6 + k] = v[k];
  }
}

size_t mesh_to_mesh_bvh_workspace(int B, int Q, int F, int MC) { return bvh_layout(B, Q, F).total; }

int mesh_to_mesh_bvh(const float *query, const float *target, int B, int Q, int F, int MC,
                     long long *faces_out, float *bcs_out, void *ws, size_t ws_bytes,
                     int *overflow, hipStream_t s) {
  if (F < 2 || F >= (1 << 24)) return SHAPY_EINVAL;
  const BvhLayout L = bvh_layout(B, Q, F);
  if (ws_bytes < L.total) return SHAPY_EWORKSPACE;
  char *w = (char *)ws;
  SHAPY_HIP_TRY(hipMemsetAsync(w + L.hits_cnt, 0, (size_t)B * Q * 4, s));
  hipLaunchKernelGGL(bvh_build_kernel, dim3(B), dim3(BUILD_THREADS), 0, s, target, F, L, w);
  SHAPY_HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(bvh_traverse_kernel, dim3((Q + 63) / 64, B), dim3(64), 0, s, query, target, Q,
                     F, MC, L, w, faces_out, bcs_out, overflow);
  SHAPY_HIP_TRY(hipGetLastError());
  const long total = (long)B * Q;
  hipLaunchKernelGGL(bvh_sort_hits_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, Q,
                     MC, L, w, faces_out, bcs_out, total);
  return (int)hipGetLastError();
}

}  // namespace shapy
