// Triangle-triangle predicates of the mesh-mesh-intersection operator, templated on the scalar type: float32
// (every SHAPY call) and float64 (the reference's second instantiation, mesh_mesh_intersect_cuda_op.cu:996 --
// which is NOT a double-precision algorithm: CMP converts to float and compares against FLT_EPSILON in both).
//
// Observable semantics of the reference kernels in
//   mesh-mesh-intersection/src/mesh_mesh_intersect_cuda_op.cu
//     CMP                                        :91-92
//     SatCrossEdge                               :151-169
//     point_to_barycentric                       :186-200
//     ray_triangle_intersect                     :202-232
//     isect_interval / TriangleTriangleOverlap   :234-268
//     TriangleTriangleIsectSepAxis               :270-341
//     checkOverlap                               :363-373
//     find_triangle_triangle_intersection_points :375-518
// written from that description for gfx950: including its quirks -- the
// 11-axis SAT with the perpendicular-edge fallback axis, the 1e-4 determinant cut, and the
// fact that both barycentric output slots always receive the FIRST hit point.
#pragma once
#include <float.h>

#include "common.h"

namespace shapy {

template <typename T> struct V3T { T x, y, z; };
using V3 = V3T<float>;

template <typename T> __device__ __forceinline__ V3T<T> v3(T x, T y, T z) { return V3T<T>{x, y, z}; }
template <typename T> __device__ __forceinline__ V3T<T> operator-(V3T<T> a, V3T<T> b) { return v3<T>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <typename T> __device__ __forceinline__ V3T<T> operator+(V3T<T> a, V3T<T> b) { return v3<T>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <typename T> __device__ __forceinline__ V3T<T> operator*(T s, V3T<T> a) { return v3<T>(s * a.x, s * a.y, s * a.z); }
template <typename T> __device__ __forceinline__ T dot3(V3T<T> a, V3T<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T> __device__ __forceinline__ V3T<T> cross3(V3T<T> a, V3T<T> b) {
  return v3<T>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float rmin(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ float rmax(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double rmin(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ double rmax(double a, double b) { return fmax(a, b); }

template <typename T> struct TriT { V3T<T> v0, v1, v2; };
using Tri = TriT<float>;

// CMP (:91-92): fabsf / fmaxf / FLT_EPSILON whatever T is -- the double instantiation converts to float here
template <typename T>
__device__ __forceinline__ bool cmp_eq(T x, T y) {
  return fabsf((float)(x - y)) <= FLT_EPSILON * fmaxf(1.0f, fmaxf(fabsf((float)x), fabsf((float)y)));
}

template <typename T>
__device__ __forceinline__ V3T<T> sat_cross_edge(V3T<T> a, V3T<T> b, V3T<T> c, V3T<T> d) {
  const V3T<T> ab = b - a, cd = d - c;
  V3T<T> r = cross3(ab, cd);
  if (!cmp_eq(dot3(ab, cd), (T)0)) return r;
  const V3T<T> axis = cross3(ab, c - a);
  r = cross3(ab, axis);
  if (!cmp_eq(dot3(r, r), (T)0)) return r;
  return v3<T>(0, 0, 0);
}

// true when `ax` does NOT separate the triangles (closed intervals) or is a null axis
template <typename T>
__device__ __forceinline__ bool axis_keeps(const TriT<T> &q, const TriT<T> &t, V3T<T> ax) {
  T p = dot3(ax, q.v0), qmin = p, qmax = p;
  p = dot3(ax, q.v1); qmin = rmin(qmin, p); qmax = rmax(qmax, p);
  p = dot3(ax, q.v2); qmin = rmin(qmin, p); qmax = rmax(qmax, p);
  p = dot3(ax, t.v0); T tmin = p, tmax = p;
  p = dot3(ax, t.v1); tmin = rmin(tmin, p); tmax = rmax(tmax, p);
  p = dot3(ax, t.v2); tmin = rmin(tmin, p); tmax = rmax(tmax, p);
  if ((qmin <= tmax) && (tmin <= qmax)) return true;
  return cmp_eq(dot3(ax, ax), (T)0);
}

template <typename T>
__device__ __forceinline__ bool tri_tri_sat(const TriT<T> &q, const TriT<T> &t) {
  if (!axis_keeps(q, t, sat_cross_edge(q.v0, q.v1, q.v1, q.v2))) return false;
  if (!axis_keeps(q, t, sat_cross_edge(t.v0, t.v1, t.v1, t.v2))) return false;
  const V3T<T> qa[3] = {q.v0, q.v1, q.v2}, qb[3] = {q.v1, q.v2, q.v0};
  const V3T<T> ta[3] = {t.v0, t.v1, t.v2}, tb[3] = {t.v1, t.v2, t.v0};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      if (!axis_keeps(q, t, sat_cross_edge(qa[i], qb[i], ta[j], tb[j]))) return false;
  return true;
}

template <typename T>
__device__ __forceinline__ bool ray_tri(V3T<T> orig, V3T<T> dir, V3T<T> v0, V3T<T> v1, V3T<T> v2, T &t, V3T<T> &p) {
  const V3T<T> v0v1 = v1 - v0, v0v2 = v2 - v0;
  const V3T<T> pvec = cross3(dir, v0v2);
  const T det = dot3(v0v1, pvec);
  if (fabs((double)det) < 1e-4) return false;
  const T inv = (T)1 / det;
  const V3T<T> tvec = orig - v0;
  const T u = dot3(tvec, pvec) * inv;
  if (u < (T)0 || u > (T)1) return false;
  const V3T<T> qvec = cross3(tvec, v0v1);
  const T v = dot3(dir, qvec) * inv;
  if (v < (T)0 || u + v > (T)1) return false;
  t = dot3(v0v2, qvec) * inv;
  p = t * dir + orig;
  return true;
}

template <typename T>
__device__ __forceinline__ V3T<T> to_bary(V3T<T> p, V3T<T> a, V3T<T> b, V3T<T> c) {
  const V3T<T> e0 = b - a, e1 = c - a, e2 = p - a;
  const T d00 = dot3(e0, e0), d01 = dot3(e0, e1), d11 = dot3(e1, e1);
  const T d20 = dot3(e2, e0), d21 = dot3(e2, e1);
  const T den = d00 * d11 - d01 * d01;
  V3T<T> bc;
  bc.y = (d11 * d20 - d01 * d21) / den;
  bc.z = (d00 * d21 - d01 * d20) / den;
  bc.x = (T)(1.0 - (double)bc.y - (double)bc.z);
  return bc;
}

// First accepted hit among query edges vs target, then target edges vs query (0 <= t <= 1).
// Returns false when none exists (the reference leaves the slot unwritten: zeros).
template <typename T>
__device__ __forceinline__ bool tri_tri_point(const TriT<T> &q, const TriT<T> &tg, V3T<T> &bc) {
  const V3T<T> qo[3] = {q.v0, q.v1, q.v2};
  const V3T<T> qe[3] = {q.v1 - q.v0, q.v2 - q.v1, q.v0 - q.v2};
  const V3T<T> to[3] = {tg.v0, tg.v1, tg.v2};
  const V3T<T> te[3] = {tg.v1 - tg.v0, tg.v2 - tg.v1, tg.v0 - tg.v2};
  T t = 0;
  V3T<T> p, p1 = v3<T>(0, 0, 0), p2;
  bool found = false;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const bool hit = ray_tri(qo[i], qe[i], tg.v0, tg.v1, tg.v2, t, p);
    if (t > (T)1 || t < (T)0) continue;
    if (hit && !found) { p1 = p; found = true; }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const bool hit = ray_tri(to[i], te[i], q.v0, q.v1, q.v2, t, p);
    if (t > (T)1 || t < (T)0) continue;
    if (hit && !found) { p1 = p; found = true; }
    // the reference re-casts from (t + EPS) along the same edge (:481-487): when it succeeds
    // its t is ~ -EPS and the loop continues; only the stale `t` is observable afterwards.
    ray_tri(to[i] + (T)((double)t + 1e-4) * te[i], te[i], q.v0, q.v1, q.v2, t, p2);
  }
  if (!found) return false;
  bc = to_bary(p1, tg.v0, tg.v1, tg.v2);
  return true;
}

template <typename T>
__device__ __forceinline__ bool aabb_overlap(const TriT<T> &a, const TriT<T> &b) {
  const T axn = rmin(a.v0.x, rmin(a.v1.x, a.v2.x)), axx = rmax(a.v0.x, rmax(a.v1.x, a.v2.x));
  const T ayn = rmin(a.v0.y, rmin(a.v1.y, a.v2.y)), ayx = rmax(a.v0.y, rmax(a.v1.y, a.v2.y));
  const T azn = rmin(a.v0.z, rmin(a.v1.z, a.v2.z)), azx = rmax(a.v0.z, rmax(a.v1.z, a.v2.z));
  const T bxn = rmin(b.v0.x, rmin(b.v1.x, b.v2.x)), bxx = rmax(b.v0.x, rmax(b.v1.x, b.v2.x));
  const T byn = rmin(b.v0.y, rmin(b.v1.y, b.v2.y)), byx = rmax(b.v0.y, rmax(b.v1.y, b.v2.y));
  const T bzn = rmin(b.v0.z, rmin(b.v1.z, b.v2.z)), bzx = rmax(b.v0.z, rmax(b.v1.z, b.v2.z));
  return axn <= bxx && axx >= bxn && ayn <= byx && ayx >= byn && azn <= bzx && azx >= bzn;
}

}  // namespace shapy
