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
#include <stdio.h>

#include <mutex>

#include "tri_tri.h"

namespace shapy {

#ifdef SHAPY_BVH_TIMING
__device__ unsigned long long g_bvh_times[16];
#define BVH_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_bvh_times[i] = wall_clock64(); } while (0)
#else
#define BVH_STAMP(i) do {} while (0)
#endif

constexpr int BUILD_THREADS = 1024;
constexpr int BVH_LDS_KEYS = 16384;                       // 64-bit sort keys held in LDS at a time
constexpr size_t BVH_LDS_BYTES = (size_t)BVH_LDS_KEYS * 8;   // 128 KB of dynamic LDS (opt-in)
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

  BVH_STAMP(0);
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

  BVH_STAMP(1);
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

  BVH_STAMP(2);
  // ---- 3. bitonic sort of the keys (replaces thrust::sort_by_key :911-912) ----
  // Blocks of NB keys are sorted / merged inside LDS (128 KB = 16,384 64-bit keys: SMPL-X's 32,768
  // padded keys are two blocks); only the stages whose partner distance reaches NB go through
  // global memory (one stage for two blocks).  Round 2 ran all 120 stages through L2 with 16
  // dependent load-compare-store rounds per thread and stage: ~14 us per mesh after the fence fix.
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
  unsigned long long *lk = reinterpret_cast<unsigned long long *>(dyn);
  const int NB = L.Fpad < BVH_LDS_KEYS ? L.Fpad : BVH_LDS_KEYS;
  auto lds_stages = [&](int base, int k, int jtop) {      // stages j = jtop .. 1 of merge size k
    for (int j = jtop; j > 0; j >>= 1) {
      const int lj = 31 - __clz(j);                        // (j is a power of two: no divisions)
      for (int q = tid; q < NB / 2; q += BUILD_THREADS) {
        const int i = ((q >> lj) << (lj + 1)) | (q & (j - 1)), l = i + j;
        const unsigned long long a = lk[i], c = lk[l];
        if ((a > c) == (((base + i) & k) == 0)) { lk[i] = c; lk[l] = a; }
      }
      __syncthreads();
    }
  };
  for (int base = 0; base < L.Fpad; base += NB) {          // every block: all merge sizes up to NB
    for (int i = tid; i < NB; i += BUILD_THREADS) lk[i] = m.keys[base + i];
    __syncthreads();
    for (int k = 2; k <= NB; k <<= 1) lds_stages(base, k, k >> 1);
    for (int i = tid; i < NB; i += BUILD_THREADS) m.keys[base + i] = lk[i];
    __syncthreads();
  }
  BVH_STAMP(3);
  for (int k = 2 * NB; k <= L.Fpad; k <<= 1) {
    for (int j = k >> 1; j >= NB; j >>= 1) {               // partners in different blocks: global
      for (int i = tid; i < L.Fpad; i += BUILD_THREADS) {
        const int l = i ^ j;
        if (l > i) {
          const unsigned long long a = m.keys[i], c = m.keys[l];
          if ((a > c) == ((i & k) == 0)) { m.keys[i] = c; m.keys[l] = a; }
        }
      }
      __syncthreads();
    }
    for (int base = 0; base < L.Fpad; base += NB) {
      for (int i = tid; i < NB; i += BUILD_THREADS) lk[i] = m.keys[base + i];
      __syncthreads();
      lds_stages(base, k, NB >> 1);
      for (int i = tid; i < NB; i += BUILD_THREADS) m.keys[base + i] = lk[i];
      __syncthreads();
    }
  }

  BVH_STAMP(4);
  // ---- 4. Karras radix tree over the sorted keys (:696-765) ----
  // The ~30 key comparisons per internal node read the sorted keys from LDS when they fit as
  // (30-bit Morton code, 16-bit face) = 6 bytes per key (SMPL-X: 125 KB); otherwise from L2.
  const bool lds_keys = F <= 65535 && (size_t)((F + 3) & ~3) * 4 + (size_t)F * 2 <= BVH_LDS_BYTES;
  unsigned *lmort = reinterpret_cast<unsigned *>(dyn);
  unsigned short *lface = reinterpret_cast<unsigned short *>(dyn + (size_t)((F + 3) & ~3) * 4);
  if (lds_keys) {
    for (int i = tid; i < F; i += BUILD_THREADS) {
      const unsigned long long k = m.keys[i];
      lmort[i] = (unsigned)(k >> 32);
      lface[i] = (unsigned short)(k & 0xffffu);
    }
    __syncthreads();
  }
  auto dlt = [&](int i, int j) -> int {
    if (j < 0 || j >= F) return -1;
    if (!lds_keys) return __clzll(m.keys[i] ^ m.keys[j]);
    const unsigned x = lmort[i] ^ lmort[j];
    return x ? __clz(x) : 32 + __clz((unsigned)(lface[i] ^ lface[j]));
  };
  for (int i = tid; i < F - 1; i += BUILD_THREADS) {
    const int d = (dlt(i, i + 1) - dlt(i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = dlt(i, i - d);
    int lmax = 2;
    while (dlt(i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
      if (dlt(i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = dlt(i, j);
    int s = 0;
    for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
      if (dlt(i, i + (s + t) * d) > dnode) s += t;
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
  }
  if (tid == 0 && F > 1) m.parent_i[0] = -1;
  __syncthreads();

  BVH_STAMP(5);
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
  BVH_STAMP(6);
  if (F < 2) return;
  // Refit in WAVEFRONTS instead of one climbing thread per leaf (the reference's scheme, :767-821,
  // which round 2 kept: a thread climbs until it is the first to arrive at a node -- with 1,024
  // threads for 20,908 leaves that is 21 sequential climbs of up to tree-height dependent L2 round
  // trips each, 0.98 of the build's 1.62 ms per mesh).  Arrival counters (one byte per internal
  // node) and two ready lists live in LDS; round r computes the boxes of all nodes whose two
  // children are done -- every thread takes nodes of the list, nobody waits -- and appends the
  // parents that became ready.  The tree and therefore every box is the same as before.
  const bool lds_refit = (size_t)((F + 2) / 4 * 4) + (size_t)(F / 2 + 1) * 8 + 16 <= BVH_LDS_BYTES;
  if (lds_refit) {
    unsigned *cnt = reinterpret_cast<unsigned *>(dyn);                        // 4 counters per word
    const int cnt_words = (F + 2) / 4;
    int *list[2] = {reinterpret_cast<int *>(dyn) + cnt_words, reinterpret_cast<int *>(dyn) + cnt_words + F / 2 + 1};
    __shared__ int nlist[2];
    for (int i = tid; i < cnt_words; i += BUILD_THREADS) cnt[i] = 0;
    if (tid < 2) nlist[tid] = 0;
    __syncthreads();
    auto arrive = [&](int node, int which) {        // a child of `node` is done; second arrival: ready
      const unsigned sh = 8u * (node & 3);
      const unsigned old = atomicAdd(&cnt[node >> 2], 1u << sh);
      if (((old >> sh) & 0xffu) == 1u) list[which][atomicAdd(&nlist[which], 1)] = node;
    };
    for (int k = tid; k < F; k += BUILD_THREADS) arrive(m.parent_l[k], 0);
    __syncthreads();
    for (int cur = 0;; cur ^= 1) {
      const int n = nlist[cur];
      if (n == 0) break;
      __syncthreads();                               // everybody has read n
      if (tid == 0) nlist[cur ^ 1] = 0;
      __syncthreads();
      for (int q = tid; q < n; q += BUILD_THREADS) {
        const int node = list[cur][q];
        const int lc = m.left[node], rc = m.right[node];
        const float *a = lc < 0 ? m.box_l + (size_t)(~lc) * 6 : m.box_i + (size_t)lc * 6;
        const float *c = rc < 0 ? m.box_l + (size_t)(~rc) * 6 : m.box_i + (size_t)rc * 6;
        float *o = m.box_i + (size_t)node * 6;
        o[0] = fminf(a[0], c[0]); o[1] = fminf(a[1], c[1]); o[2] = fminf(a[2], c[2]);
        o[3] = fmaxf(a[3], c[3]); o[4] = fmaxf(a[4], c[4]); o[5] = fmaxf(a[5], c[5]);
        if (node != 0) arrive(m.parent_i[node], cur ^ 1);
      }
      __syncthreads();                               // (workgroup scope: the boxes of this round are
    }                                                //  visible to the waves of this CU in the next)
  } else {
    for (int i = tid; i < F - 1; i += BUILD_THREADS) m.flag[i] = 0;
    __syncthreads();
    for (int k = tid; k < F; k += BUILD_THREADS) {
      int node = m.parent_l[k];
      while (true) {
        // both children of a node are handled by threads of THIS workgroup, i.e. by waves of one
        // CU that share its L1: workgroup-scope release / acquire is enough.  (Round 2 used
        // __threadfence() here: on a multi-XCD part an agent-scope fence writes back and
        // invalidates the XCD's whole L2 -- two of them per node and wave made the build 187 us
        // PER MESH, 99 % of the LBVH path; profiles/r03n_kernel_stats_bvh_before_fence_fix.csv.)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // publish this subtree's boxes
        if (__hip_atomic_fetch_add(&m.flag[node], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0)
          break;                                                   // first child to arrive leaves
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");     // second: see the sibling's box
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
  __syncthreads();
  BVH_STAMP(7);
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
  {
    // 128 KB of dynamic LDS: opted into per device (the attribute lives in the device's function)
    static std::mutex mu;
    static unsigned long long done = 0;
    int dev = 0;
    SHAPY_HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    if (dev < 0 || dev >= 64 || !(done >> dev & 1ull)) {
      // this library is built for gfx950 (160 KB of LDS per CU); on a device that cannot give one
      // workgroup 128 KB + the kernel's static LDS the build kernel cannot run -- say so instead of
      // failing inside the launch
      int max_lds = 0;
      SHAPY_HIP_TRY(hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
      hipFuncAttributes fa;
      SHAPY_HIP_TRY(hipFuncGetAttributes(&fa, (const void *)bvh_build_kernel));
      if ((size_t)max_lds < BVH_LDS_BYTES + fa.sharedSizeBytes) {
        fprintf(stderr, "shapy: mesh_to_mesh_bvh needs %zu bytes of LDS per workgroup, device %d offers %d "
                        "(library built for gfx950)\n", BVH_LDS_BYTES + fa.sharedSizeBytes, dev, max_lds);
        return SHAPY_EINVAL;
      }
      SHAPY_HIP_TRY(hipFuncSetAttribute((const void *)bvh_build_kernel,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)BVH_LDS_BYTES));
      if (dev >= 0 && dev < 64) done |= 1ull << dev;
    }
  }
  hipLaunchKernelGGL(bvh_build_kernel, dim3(B), dim3(BUILD_THREADS), BVH_LDS_BYTES, s, target, F, L, w);
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

#ifdef SHAPY_BVH_TIMING
extern "C" int shapy_debug_bvh_times(unsigned long long *out_host) {
  return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(shapy::g_bvh_times), sizeof(unsigned long long) * 16);
}
#endif
