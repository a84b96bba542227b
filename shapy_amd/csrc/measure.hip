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
//    one workgroup per mesh stages v_shaped in LDS, scans the face table once (signed-volume
//    sum + y-range candidates of all three planes; no [B,F,3,3] triangle tensor is
//    materialised) and runs the SAT / intersection-point code on the dense candidate queue;
//    then one wave per (mesh, plane) sorts the <= 2*max_coll points in LDS and walks the 2-D
//    convex hull (monotone chain, float64 orientation tests) -- replacing the per-mesh
//    scipy/Qhull call and the D2H copy in front of it.
//
// This translation unit is compiled with -ffp-contract=off so that every float32 decision
// (SAT tolerance tests, barycentric range tests) is bit-identical with the CPU oracle.
#include <stdlib.h>

#include <mutex>

#include "tri_tri.h"

namespace shapy {

constexpr int SCAN_THREADS = 256;

template <typename T>
__device__ __forceinline__ TriT<T> load_tri(const T *p) {
  TriT<T> t;
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

// T = float: every SHAPY call; T = double: the reference's second instantiation (shapy_mesh_to_mesh_f64)
template <typename T>
__global__ __launch_bounds__(SCAN_THREADS) void mesh_to_mesh_scan_kernel(
    const T *__restrict__ query, const T *__restrict__ target, int Q, int F, int MC,
    long long *__restrict__ faces_out, T *__restrict__ bcs_out, int *__restrict__ overflow) {
  __shared__ int wave_cnt[SCAN_THREADS / 64];
  const int q = blockIdx.x, b = blockIdx.y;
  const TriT<T> qt = load_tri(query + ((long)b * Q + q) * 9);
  const T *tb = target + (long)b * F * 9;
  long long *fo = faces_out + ((long)b * Q + q) * MC;
  T *bo = bcs_out + ((long)b * Q + q) * MC * 6;
  int base = 0;
  for (int f0 = 0; f0 < F; f0 += SCAN_THREADS) {
    const int f = f0 + threadIdx.x;
    bool hit = false;
    TriT<T> tt;
    if (f < F) {
      tt = load_tri(tb + (long)f * 9);
      hit = aabb_overlap(qt, tt) && tri_tri_sat(qt, tt);
    }
    int total;
    const int slot = base + block_rank(hit, total, wave_cnt);
    if (hit) {
      if (slot < MC) {
        V3T<T> bc;
        const bool ok = tri_tri_point(qt, tt, bc);
        fo[slot] = f;
        if (ok) {
          T *o = bo + (long)slot * 6;
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

// (tri * bc.reshape(1,3,1)).sum(dim=1)   (body_measurements.py:130-131,185-195)
template <typename VF>
__device__ __forceinline__ float lm_coord(VF vtx, const int32_t *faces, const Landmarks &lm,
                                          int which, int axis) {
  const int f = lm.face[which];
  const float a = vtx(faces[f * 3 + 0], axis), b = vtx(faces[f * 3 + 1], axis),
              c = vtx(faces[f * 3 + 2], axis);
  return (a * lm.bc[which][0] + b * lm.bc[which][1]) + c * lm.bc[which][2];
}

// Candidates of the fused scan: SAT, then first-hit point of (face, plane, plane triangle QI).
// QI is a template parameter: the plane triangle's vertices are then compile-time constants
// (+-1 and the height h), which lets the compiler fold part of the 11 axis tests.
template <int QI>
__device__ __forceinline__ Tri plane_triangle(float h) {
  // _get_plane_at_heights (body_measurements.py:86-97): quad (c0,c1,c2,c3) as 2 triangles
  Tri q;
  q.v0 = v3(-1.f, h, -1.f);
  q.v1 = QI == 0 ? v3(1.f, h, -1.f) : v3(1.f, h, 1.f);
  q.v2 = QI == 0 ? v3(1.f, h, 1.f) : v3(-1.f, h, 1.f);
  return q;
}

template <int QI>
__device__ __forceinline__ bool candidate_hit(const Tri &t, float h) {
  const Tri q = plane_triangle<QI>(h);
  return aabb_overlap(q, t) && tri_tri_sat(q, t);
}

template <int QI>
__device__ __forceinline__ void candidate_point(const Tri &t, float h, int f,
                                                float4 *__restrict__ dst) {
  const Tri q = plane_triangle<QI>(h);
  V3 bc = v3(0.f, 0.f, 0.f);
  tri_tri_point(q, t, bc);
  // points = sum_k bc_k * tri_k (body_measurements.py:144-147)
  float4 pt;
  pt.x = (t.v0.x * bc.x + t.v1.x * bc.y) + t.v2.x * bc.z;
  pt.y = (t.v0.y * bc.x + t.v1.y * bc.y) + t.v2.y * bc.z;
  pt.z = (t.v0.z * bc.x + t.v1.z * bc.y) + t.v2.z * bc.z;
  pt.w = __int_as_float(f);
  *dst = pt;
}

// Phase timing of one workgroup (tuning builds only: SHAPY_HIPCC_FLAGS=-DSHAPY_MEASURE_TIMING,
// read back with shapy_debug_measure_times): wall_clock64 ticks at 100 MHz.
#ifdef SHAPY_MEASURE_TIMING
__device__ unsigned long long g_measure_times[32];
#define M2_STAMP(slot)                                                              \
  do {                                                                              \
    if (blockIdx.x == 0 && blockIdx.y == 7 && threadIdx.x == 0)                     \
      g_measure_times[slot] = wall_clock64();                                       \
  } while (0)
#else
#define M2_STAMP(slot) do {} while (0)
#endif

constexpr int M2_THREADS = 1024;            // 16 waves: one workgroup owns a CU's LDS
constexpr int M2_QCAP = 4096;               // candidate queue entries (face * 4 + plane)
constexpr int M2_MAX_SLICES = 16;
constexpr int M2_STATIC_LDS = M2_QCAP * 4 + 512;
constexpr int M2_LDS_TOTAL = 160 * 1024;

// One workgroup per (mesh slice, mesh).  STAGED: the mesh's v_shaped (V * 12 bytes; 125.7 KB for
// SMPL-X) is copied ONCE, coalesced, into LDS and the 9 coordinate gathers per face are LDS
// reads; the int32 face table streams through in index order (12 contiguous bytes per lane).
// The scan itself only evaluates the signed-volume term and the y-range test against the three
// plane heights; faces that pass (~0.7 % per plane) are queued in LDS and the expensive part
// -- AABB + 11-axis SAT + first-hit point for both plane triangles -- runs afterwards on the
// dense queue (in the scan it would run with 1-2 active lanes in 3 of 4 wave iterations).
// Hit slots come from one global atomic per hit (~300 per mesh); the hull kernel sorts, so the
// order does not matter.  No barrier inside the scan loop.  !STAGED (triangle soups of the reference signature, V = 3 F): same
// code, coordinates gathered from global memory, the faces of a mesh split over several slices.
template <bool STAGED>
__global__ __launch_bounds__(M2_THREADS) void measure_scan2_kernel(
    const float *__restrict__ v_shaped, const int32_t *__restrict__ faces, int V, int F, int Fs,
    int CAP, Landmarks lm, int *__restrict__ counters, float *__restrict__ vol_partial,
    float4 *__restrict__ points, int *__restrict__ overflow, int dbg) {
  extern __shared__ __attribute__((aligned(16))) float sv[];
  __shared__ int queue[M2_QCAP];
  __shared__ int qn;
  __shared__ int lcnt[6];            // hits per (plane, plane triangle) when the mesh is not sliced
  __shared__ float hs[3];
  __shared__ double red[M2_THREADS / 64];
  const int b = blockIdx.y, tid = threadIdx.x;
  const float *vb = v_shaped + (long)b * V * 3;
  M2_STAMP(0);
  int shift = 0;
  // dbg (SHAPY_MEASURE_DBG, tuning only): 1 = no scan loop, 2 = no candidate evaluation,
  // 4 = no staging copy -- wrong results on purpose, to time the phases
  if (STAGED && !(dbg & 4)) {
    // 16-byte copies: mesh b starts at byte b * V * 12, which is only 4-byte aligned; the LDS
    // image is shifted by the same phase so that both sides of the vector copy are aligned
    const int N = V * 3;
    int head = (int)(((16 - ((uintptr_t)vb & 15)) & 15) >> 2);
    if (head > N) head = N;
    shift = (4 - head) & 3;
    if (tid < head) sv[shift + tid] = vb[tid];
    const int n4 = (N - head) >> 2;
    const float4 *g4 = reinterpret_cast<const float4 *>(vb + head);
    float4 *s4 = reinterpret_cast<float4 *>(sv + shift + head);
    // all loads of a thread in flight before the first LDS store (8 x 16 bytes per thread for
    // SMPL-X): the copy runs at the HBM rate instead of one round trip per iteration
    constexpr int LD = 8;
    for (int i0 = tid; i0 < n4; i0 += LD * M2_THREADS) {
      float4 tmp[LD];
#pragma unroll
      for (int u = 0; u < LD; ++u) {
        const int i = i0 + u * M2_THREADS;
        tmp[u] = i < n4 ? g4[i] : float4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < LD; ++u) {
        const int i = i0 + u * M2_THREADS;
        if (i < n4) s4[i] = tmp[u];
      }
    }
    for (int i = head + n4 * 4 + tid; i < N; i += M2_THREADS) sv[shift + i] = vb[i];
  }
  if (tid == 0) qn = 0;
  if (tid < 6) lcnt[tid] = 0;
  __syncthreads();
  M2_STAMP(1);
  auto vtx = [&](int idx, int c) -> float {
    if constexpr (STAGED) return sv[shift + idx * 3 + c];
    else return vb[(long)idx * 3 + c];
  };
  if (tid < 3) hs[tid] = lm_coord(vtx, faces, lm, 2 + tid, 1);
  __syncthreads();
  auto load_face = [&](int f) -> Tri {
    const int i0 = faces[f * 3], i1 = faces[f * 3 + 1], i2 = faces[f * 3 + 2];
    Tri t;
    t.v0 = v3(vtx(i0, 0), vtx(i0, 1), vtx(i0, 2));
    t.v1 = v3(vtx(i1, 0), vtx(i1, 1), vtx(i1, 2));
    t.v2 = v3(vtx(i2, 0), vtx(i2, 1), vtx(i2, 2));
    return t;
  };
  // The scan runs without barriers (every wave streams its faces at its own pace; the face
  // indices of the next iteration are in flight while this one is evaluated).  Candidates
  // beyond the queue capacity (> 4096 per mesh slice: not a body -- a body has ~450) are
  // dropped and reported through *overflow like hits beyond max_collisions.
  const int f_lo = blockIdx.x * Fs, f_hi = min(F, f_lo + Fs);
  double vol = 0.0;
  int dropped = 0;
  const float h0 = hs[0], h1 = hs[1], h2 = hs[2];
  int f = f_lo + tid;
  int i0 = 0, i1 = 0, i2 = 0;
  if (f < f_hi) { i0 = faces[f * 3]; i1 = faces[f * 3 + 1]; i2 = faces[f * 3 + 2]; }
  if (dbg & 1) f = f_hi;
  while (f < f_hi) {
    const int fn = f + M2_THREADS;
    const float x0 = vtx(i0, 0), y0 = vtx(i0, 1), z0 = vtx(i0, 2);
    const float x1 = vtx(i1, 0), y1 = vtx(i1, 1), z1 = vtx(i1, 2);
    const float x2 = vtx(i2, 0), y2 = vtx(i2, 1), z2 = vtx(i2, 2);
    // next face's indices: requested once this face's gathers are on their way, so that the
    // request overlaps the arithmetic below (hipcc waits for ALL outstanding loads before the
    // first use of a loop-carried load result, so it may not be issued any earlier)
    int n0 = 0, n1 = 0, n2 = 0;
    if (fn < f_hi) { n0 = faces[fn * 3]; n1 = faces[fn * 3 + 1]; n2 = faces[fn * 3 + 2]; }
    // compute_mass (body_measurements.py:201-215), term order as written there
    const float vv = -x2 * y1 * z0 + x1 * y2 * z0 + x2 * y0 * z1 - x0 * y2 * z1 - x1 * y0 * z2 +
                     x0 * y1 * z2;
    vol += (double)vv;
    const float ymin = fminf(y0, fminf(y1, y2)), ymax = fmaxf(y0, fmaxf(y1, y2));
    // the y part of the AABB test against the three plane heights
    const int mask = (int)(ymin <= h0 && ymax >= h0) | ((int)(ymin <= h1 && ymax >= h1) << 1) |
                     ((int)(ymin <= h2 && ymax >= h2) << 2);
    if (mask) {
      int slot = atomicAdd(&qn, __popc(mask));
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        if (mask & (1 << pl)) {
          if (slot < M2_QCAP) queue[slot] = f * 4 + pl;
          else ++dropped;
          ++slot;
        }
    }
    f = fn; i0 = n0; i1 = n1; i2 = n2;
  }
  if (dropped && overflow) atomicAdd(overflow, dropped);
  M2_STAMP(2);
  __syncthreads();
  {
    // candidate evaluation: pairs [0, n) test plane triangle 0, [n, 2n) plane triangle 1 (a
    // wave runs ONE specialisation).  Two passes over the SAT would double the cost, so the
    // slot comes from an atomic AFTER the SAT: LDS counters when the mesh belongs to this
    // workgroup alone, global ones when it is sliced.
    const int n = (dbg & 2) ? 0 : (qn < M2_QCAP ? qn : M2_QCAP);
    for (int p0 = 0; p0 < 2 * n; p0 += M2_THREADS) {
      const int p = p0 + tid;
      if (p >= 2 * n) continue;
      const int qi = p >= n, c = queue[qi ? p - n : p], f = c >> 2, pl = c & 3;
      const Tri t = load_face(f);
      const long list = ((long)b * 3 + pl) * 2 + qi;
      const float h = hs[pl];
      if (!(qi == 0 ? candidate_hit<0>(t, h) : candidate_hit<1>(t, h))) continue;
      const int slot = gridDim.x == 1 ? atomicAdd(&lcnt[pl * 2 + qi], 1)
                                      : atomicAdd(counters + list, 1);
      if (slot >= CAP) continue;
      if (qi == 0) candidate_point<0>(t, h, f, points + list * CAP + slot);
      else candidate_point<1>(t, h, f, points + list * CAP + slot);
    }
  }
  __syncthreads();
  M2_STAMP(3);
  if (gridDim.x == 1 && tid < 6) counters[(long)b * 6 + tid] = lcnt[tid];
  // deterministic reduction of the signed volume (fixed lane / wave order)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) vol += __shfl_xor(vol, o, 64);
  if ((tid & 63) == 0) red[tid >> 6] = vol;
  __syncthreads();
  if (tid == 0) {
    double sum = 0.0;
    for (int w = 0; w < M2_THREADS / 64; ++w) sum += red[w];
    vol_partial[(long)b * gridDim.x + blockIdx.x] = (float)sum;
    M2_STAMP(4);
  }
}

// One wave per (mesh, plane).  Everything after the gather is data-parallel over the lanes:
//   1. gather the <= 2 * MC points of the two plane triangles into LDS;
//   2. rank sort by (x, z, slot): every lane counts, for each of its points, the points that
//      precede it (n broadcast LDS reads per point) and scatters it to its rank -- the result is
//      a pure function of the point SET (the atomic order of the scan does not matter).  ~5 us
//      for the ~160 points of a body cross-section; a bitonic network through LDS or through
//      wave shuffles took 25-45 us (one dependent round trip per compare-exchange);
//   3. drop duplicates in (x, z) (every mesh edge that crosses the plane is reported by both
//      triangles sharing it, sometimes one ulp apart in y) by ballot compaction -- coincident
//      points would justify each other's removal in step 4;
//   4. lower and upper hull by ELIMINATION ROUNDS instead of a serial monotone chain (0.55 us per
//      point on one lane: 44 us): every interior point of the x-sorted chain whose turn
//      orient(prev, p, next) has the wrong sign (float64 test, collinear counts as wrong) lies
//      on or inside the segment of two other points of the set, so all of them can be dropped at
//      once; survivors are compacted and tested again until a round removes nothing.  Near-convex
//      cross-sections finish in 2-4 rounds; the fixed point is exactly the strictly convex
//      chain Andrew's scan returns;
//   5. the 3-D perimeter (body_measurements.py:160-179) as a wave reduction over the edges of
//      both chains.
// Overflow (more than MC hits of one plane triangle): the MC LOWEST face indices are kept, the
// rule of the ascending-order CPU oracle -- deterministic as long as the scan could store all
// hits (CAP >= 256 slots per list); the excess is counted in *overflow either way.
__global__ __launch_bounds__(64, 4) void measure_hull2_kernel(
    const float *__restrict__ v_shaped, const int32_t *__restrict__ faces, int V, int MC, int CAP,
    int NP, int n_slices, Landmarks lm, const int *__restrict__ counters,
    const float *__restrict__ vol_partial, const float4 *__restrict__ points,
    float *__restrict__ out, int *__restrict__ overflow) {
  extern __shared__ __attribute__((aligned(16))) float hl[];
  float *px = hl, *py = hl + NP, *pz = hl + 2 * NP;
  // chain index lists: [chain (lower, upper)][ping-pong][NP] as 16-bit indices
  unsigned short *chain = reinterpret_cast<unsigned short *>(hl + 3 * NP);
  const int pl = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  const long list0 = ((long)b * 3 + pl) * 2;
  M2_STAMP(8);
  const int c0 = counters[list0], c1 = counters[list0 + 1];
  const int m0 = min(c0, CAP), m1 = min(c1, CAP);              // stored
  const int n0 = min(c0, MC), n1 = min(c1, MC);                // kept
  if (lane == 0 && overflow && (c0 > MC || c1 > MC)) atomicAdd(overflow, (c0 - n0) + (c1 - n1));
  const int nn = n0 + n1;
  for (int i = lane; i < NP; i += 64) { px[i] = INFINITY; py[i] = 0.f; pz[i] = INFINITY; }
  __syncthreads();
#pragma unroll
  for (int qi = 0; qi < 2; ++qi) {
    const int m = qi ? m1 : m0, n = qi ? n1 : n0, base = qi ? n0 : 0;
    const float4 *pp = points + (list0 + qi) * CAP;
    for (int i = lane; i < m; i += 64) {
      const float4 p = pp[i];
      const int face = __float_as_int(p.w);
      int pos = i;
      if (m > n) {                                   // overflow: rank by face index
        pos = 0;
        for (int k = 0; k < m; ++k) pos += __float_as_int(pp[k].w) < face;
        if (pos >= n) continue;
      }
      // the reference keeps slots with collision_faces > 0 (:161): it drops face 0 as well
      // as the empty (-1) slots
      if (face > 0) { px[base + pos] = p.x; py[base + pos] = p.y; pz[base + pos] = p.z; }
    }
  }
  __syncthreads();
  M2_STAMP(9);
  // ---- rank sort of the slots [0, nn) by (x, z), ties by (y, slot); +inf holes sink to the end ----
  // one order-preserving 64-bit key per point (parked in the chain buffers, which are not
  // needed yet): a comparison is two integer compares instead of a lexicographic float cascade
  constexpr int EPL = 16;                              // slots per lane (NP <= 1024)
  {
    unsigned long long *key = reinterpret_cast<unsigned long long *>(chain);
    auto ord = [](float f) -> unsigned {
      const unsigned u = __float_as_uint(f);
      return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
    };
    for (int i = lane; i < nn + 4; i += 64)
      key[i] = i < nn ? ((unsigned long long)ord(px[i]) << 32) | ord(pz[i]) : ~0ull;
    __syncthreads();
    const int per = (nn + 63) >> 6;                    // slots this wave actually uses per lane
    float ex[EPL], ey[EPL], ez[EPL];
    unsigned long long ke[EPL];
    int rank[EPL];
#pragma unroll
    for (int r = 0; r < EPL; ++r) {
      const int e = lane + 64 * r;
      const bool in = r < per && e < nn;
      ex[r] = in ? px[e] : INFINITY; ey[r] = in ? py[e] : 0.f; ez[r] = in ? pz[e] : INFINITY;
      ke[r] = in ? key[e] : ~0ull;
      rank[r] = 0;
    }
    // ties on (x, z) -- duplicated contour points, whose y may differ by an ulp -- are ordered by
    // (y, slot): the compaction below keeps the FIRST of a run, i.e. the lowest y, so the result
    // is a function of the point SET and not of the order in which the scan kernel's atomics
    // handed out the slots (equal (x, y, z) triples are interchangeable)
    auto before = [&](unsigned long long kj, unsigned long long ke_, int j, int e, float ye) {
      if (kj != ke_) return kj < ke_;
      const unsigned yj = ord(py[j]), yo = ord(ye);
      return yj < yo || (yj == yo && j < e);
    };
    for (int j = 0; j < nn; j += 4) {
      const unsigned long long k0 = key[j], k1 = key[j + 1], k2 = key[j + 2], k3 = key[j + 3];
#pragma unroll
      for (int r = 0; r < EPL; ++r) {
        if (r < per) {
          const int e = lane + 64 * r;
          rank[r] += (int)before(k0, ke[r], j, e, ey[r]) + (int)before(k1, ke[r], j + 1, e, ey[r]) +
                     (int)before(k2, ke[r], j + 2, e, ey[r]) + (int)before(k3, ke[r], j + 3, e, ey[r]);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < EPL; ++r) {
      const int e = lane + 64 * r;
      if (r < per && e < nn) { px[rank[r]] = ex[r]; py[rank[r]] = ey[r]; pz[rank[r]] = ez[r]; }
    }
    __syncthreads();
  }
  M2_STAMP(10);
  // ---- compact: valid entries that differ from their predecessor (in place: the destination
  // of chunk c lies at or below its source and every earlier chunk has been consumed) ----
  int n = 0;
  for (int i0 = 0; i0 < nn; i0 += 64) {
    const int i = i0 + lane;
    const float x = px[i], y = py[i], z = pz[i];
    bool keep = i < nn && x != INFINITY;
    if (keep && i > 0) keep = !(px[i - 1] == x && pz[i - 1] == z);
    const unsigned long long mask = __ballot(keep);
    __syncthreads();
    if (keep) {
      const int d = n + __popcll(mask & ((1ull << lane) - 1ull));
      px[d] = x; py[d] = y; pz[d] = z;
    }
    n += __popcll(mask);
    __syncthreads();
  }
  M2_STAMP(11);
  // ---- lower / upper chain by elimination rounds ----
  int len[2] = {n, n};                                 // current chain lengths (wave-uniform)
  for (int i = lane; i < n; i += 64) {
    chain[0 * 2 * NP + i] = (unsigned short)i;         // lower, buffer 0
    chain[1 * 2 * NP + i] = (unsigned short)i;         // upper, buffer 0
  }
  __syncthreads();
  int cur = 0;
  for (int round = 0; round < 2 * NP; ++round) {
    bool changed = false;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const unsigned short *src = chain + c * 2 * NP + cur * NP;
      unsigned short *dst = chain + c * 2 * NP + (cur ^ 1) * NP;
      const int m = len[c];
      int kept = 0;
      for (int k0 = 0; k0 < m; k0 += 64) {
        const int k = k0 + lane;
        bool keep = k < m;
        unsigned short a = 0;
        if (keep) {
          a = src[k];
          if (k > 0 && k < m - 1) {
            const int o = src[k - 1], q = src[k + 1];
            const double ox = px[o], oz = pz[o];
            const double t = ((double)px[a] - ox) * ((double)pz[q] - oz) -
                             ((double)pz[a] - oz) * ((double)px[q] - ox);
            keep = c == 0 ? t > 0.0 : t < 0.0;         // left turns below, right turns above
          }
        }
        const unsigned long long mask = __ballot(keep);
        if (keep) dst[kept + __popcll(mask & ((1ull << lane) - 1ull))] = a;
        kept += __popcll(mask);
      }
      changed |= kept != m;
      len[c] = kept;
    }
    __syncthreads();
    cur ^= 1;
    if (!changed) break;
  }
  M2_STAMP(12);
  // ---- perimeter: the edges of both chains, one edge per lane and round ----
  // (summed in float64: the result does not depend on how the edges fall onto the lanes)
  double perim_d = 0.0;
  const int ml = len[0], mu = len[1];
  const int ne = (ml > 0 ? ml - 1 : 0) + (mu > 0 ? mu - 1 : 0);
  for (int e0 = 0; e0 < ne; e0 += 64) {
    const int e = e0 + lane;
    float elen = 0.f;
    if (e < ne) {
      const bool lower = e < ml - 1;
      const unsigned short *s2 = chain + (lower ? 0 : 2 * NP) + cur * NP;
      const int k = lower ? e : e - (ml - 1);
      const int a = s2[k], c = s2[k + 1];
      const float dx = px[c] - px[a], dy = py[c] - py[a], dz = pz[c] - pz[a];
      elen = sqrtf(dx * dx + dy * dy + dz * dz);
    }
    double v = (double)elen;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    perim_d += v;
  }
  const float perim = (float)perim_d;
  M2_STAMP(13);
  if (lane == 0) {
    out[b * 5 + 2 + pl] = perim;
    if (pl == 0) {
      double vs = 0.0;
      for (int i = 0; i < n_slices; ++i) vs += (double)vol_partial[(long)b * n_slices + i];
      out[b * 5 + 0] = (float)(fabs(vs) / 6.0) * 985.0f;   // DENSITY (body_measurements.py:20)
      const float *vb = v_shaped + (long)b * V * 3;
      auto vtx = [&](int idx, int c) -> float { return vb[(long)idx * 3 + c]; };
      const float head = lm_coord(vtx, faces, lm, 0, 1), heel = lm_coord(vtx, faces, lm, 1, 1);
      out[b * 5 + 1] = fabsf(head - heel);                  // compute_height (:182-199)
    }
    if (lane == 0) M2_STAMP(14);
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
    hipLaunchKernelGGL(mesh_to_mesh_scan_kernel<float>, dim3(Q, B), dim3(SCAN_THREADS), 0, s, query, target,
                       Q, F, max_coll, (long long *)faces_out, bcs_out, overflow_out);
    return (int)hipGetLastError();
  }
  if (workspace_bytes < mesh_to_mesh_bvh_workspace(B, Q, F, max_coll)) return SHAPY_EWORKSPACE;
  return mesh_to_mesh_bvh(query, target, B, Q, F, max_coll, (long long *)faces_out, bcs_out,
                          workspace, workspace_bytes, overflow_out, s);
}

// The reference's double instantiation (mesh_mesh_intersect_cuda_op.cu:996, AT_DISPATCH_FLOATING_TYPES): same
// observable semantics on float64 triangles -- the predicates keep their float constants (CMP converts to float,
// FLT_EPSILON; the 1e-4 determinant cut), everything else is double arithmetic.  No SHAPY caller passes float64,
// so this is the plain brute-force form for every Q: one workgroup per (mesh, query triangle) scans the targets
// in index order.  Slow for large query meshes (Q x F pair tests per mesh; the float32 operator switches to the
// LBVH there), correct for all.
extern "C" int shapy_mesh_to_mesh_f64(const double *query, const double *target, int B, int Q, int F,
                                      int max_coll, int64_t *faces_out, double *bcs_out,
                                      int32_t *overflow_out, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (B < 0 || Q < 0 || F < 0 || max_coll <= 0) return SHAPY_EINVAL;
  const size_t nslots = (size_t)B * Q * max_coll;
  if (nslots == 0) return SHAPY_OK;
  if (Q > 65535 || B > 65535) return SHAPY_EINVAL;                 // grid (Q, B)
  SHAPY_HIP_TRY(hipMemsetAsync(faces_out, 0xFF, nslots * sizeof(int64_t), s));   // -1
  SHAPY_HIP_TRY(hipMemsetAsync(bcs_out, 0, nslots * 6 * sizeof(double), s));
  if (overflow_out) SHAPY_HIP_TRY(hipMemsetAsync(overflow_out, 0, sizeof(int32_t), s));
  if (F == 0) return SHAPY_OK;
  hipLaunchKernelGGL(mesh_to_mesh_scan_kernel<double>, dim3(Q, B), dim3(SCAN_THREADS), 0, s, query, target,
                     Q, F, max_coll, (long long *)faces_out, bcs_out, overflow_out);
  return (int)hipGetLastError();
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// slots per (mesh, plane, plane triangle) list: twice max_coll, at least 256 -- room to store
// every hit of a list that overflows max_coll, so that the hull kernel can pick the max_coll
// lowest faces deterministically
static int measure_cap(int max_coll) { return 2 * max_coll > 256 ? 2 * max_coll : 256; }

static bool measure_staged(int V) {
  return (size_t)V * 12 + 32 + M2_STATIC_LDS <= (size_t)M2_LDS_TOTAL;
}

extern "C" size_t shapy_body_measure_workspace_bytes(int B, int F, int max_coll) {
  return align_up((size_t)B * 6 * sizeof(int), 256) +
         align_up((size_t)B * M2_MAX_SLICES * sizeof(float), 256) +
         (size_t)B * 6 * measure_cap(max_coll) * sizeof(float4);
}

extern "C" int shapy_body_measure_f32(const float *v_shaped, const int32_t *faces, int B, int V,
                                      int F, const int32_t *lm_face_host, const float *lm_bary_host,
                                      int max_coll, float *out, void *workspace,
                                      size_t workspace_bytes, int32_t *overflow_out, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (B <= 0) return SHAPY_OK;
  if (max_coll <= 0 || max_coll > 512 || F <= 0 || V <= 0 || (long)F * 4 >= 0x7fffffffL)
    return SHAPY_EINVAL;
  if (workspace_bytes < shapy_body_measure_workspace_bytes(B, F, max_coll)) return SHAPY_EWORKSPACE;
  Landmarks lm;
  for (int i = 0; i < 5; ++i) {
    if (lm_face_host[i] < 0 || lm_face_host[i] >= F) return SHAPY_EINVAL;
    lm.face[i] = lm_face_host[i];
    for (int k = 0; k < 3; ++k) lm.bc[i][k] = lm_bary_host[i * 3 + k];
  }
  const int CAP = measure_cap(max_coll);
  char *w = (char *)workspace;
  int *counters = (int *)w;
  w += align_up((size_t)B * 6 * sizeof(int), 256);
  float *vol = (float *)w;
  w += align_up((size_t)B * M2_MAX_SLICES * sizeof(float), 256);
  float4 *pts = (float4 *)w;
  SHAPY_HIP_TRY(hipMemsetAsync(counters, 0, (size_t)B * 6 * sizeof(int), s));
  if (overflow_out) SHAPY_HIP_TRY(hipMemsetAsync(overflow_out, 0, sizeof(int32_t), s));
  static const int dbg = getenv("SHAPY_MEASURE_DBG") ? atoi(getenv("SHAPY_MEASURE_DBG")) : 0;
  int S = 1;
  if (measure_staged(V)) {
    const size_t dyn = (size_t)V * 12 + 32;
    // > 64 KB of dynamic LDS must be opted into -- PER DEVICE (the attribute lives in the
    // device's copy of the function): a bitmask of the devices done, under a mutex
    {
      static std::mutex mu;
      static unsigned long long done = 0;
      int dev = 0;
      SHAPY_HIP_TRY(hipGetDevice(&dev));
      std::lock_guard<std::mutex> lk(mu);
      if (dev < 0 || dev >= 64 || !(done >> dev & 1ull)) {
        SHAPY_HIP_TRY(hipFuncSetAttribute((const void *)measure_scan2_kernel<true>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize,
                                          M2_LDS_TOTAL - M2_STATIC_LDS));
        if (dev >= 0 && dev < 64) done |= 1ull << dev;
      }
    }
    // small batches: several workgroups per mesh (each stages the whole vertex array -- the
    // repeats hit L2 -- and scans a slice of the faces) so that the chip is not left idle
    S = 256 / B;
    if (S > 8) S = 8;
    if (S < 1) S = 1;
    const int Fs = (F + S - 1) / S;
    hipLaunchKernelGGL(measure_scan2_kernel<true>, dim3(S, B), dim3(M2_THREADS), dyn, s, v_shaped,
                       faces, V, F, Fs, CAP, lm, counters, vol, pts, overflow_out, dbg);
  } else {
    S = (F + 4095) / 4096;
    if (S > M2_MAX_SLICES) S = M2_MAX_SLICES;
    const int Fs = (F + S - 1) / S;
    hipLaunchKernelGGL(measure_scan2_kernel<false>, dim3(S, B), dim3(M2_THREADS), 0, s, v_shaped,
                       faces, V, F, Fs, CAP, lm, counters, vol, pts, overflow_out, dbg);
  }
  SHAPY_HIP_TRY(hipGetLastError());
  int NP = 64;                              // >= one wave's worth: the compaction reads whole chunks
  while (NP < 2 * max_coll) NP <<= 1;
  const size_t hull_lds = (size_t)(3 * NP) * 4 + (size_t)(4 * NP) * 2 + 64;   // points + 2 x 2 index lists (+ key pad)
  hipLaunchKernelGGL(measure_hull2_kernel, dim3(3, B), dim3(64), hull_lds, s, v_shaped, faces, V,
                     max_coll, CAP, NP, S, lm, counters, vol, pts, out, overflow_out);
  return (int)hipGetLastError();
}

#ifdef SHAPY_MEASURE_TIMING
extern "C" int shapy_debug_measure_times(unsigned long long *out_host) {
  return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(shapy::g_measure_times),
                                  sizeof(unsigned long long) * 32);
}
#endif
