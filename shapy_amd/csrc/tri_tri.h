// Triangle-triangle predicates of the mesh-mesh-intersection operator, float32.
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

struct V3 { float x, y, z; };

__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross3(V3 a, V3 b) {
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

struct Tri { V3 v0, v1, v2; };

__device__ __forceinline__ bool cmp_eq(float x, float y) {
  return fabsf(x - y) <= FLT_EPSILON * fmaxf(1.0f, fmaxf(fabsf(x), fabsf(y)));
}

__device__ __forceinline__ V3 sat_cross_edge(V3 a, V3 b, V3 c, V3 d) {
  const V3 ab = b - a, cd = d - c;
  V3 r = cross3(ab, cd);
  if (!cmp_eq(dot3(ab, cd), 0.f)) return r;
  const V3 axis = cross3(ab, c - a);
  r = cross3(ab, axis);
  if (!cmp_eq(dot3(r, r), 0.f)) return r;
  return v3(0.f, 0.f, 0.f);
}

// true when `ax` does NOT separate the triangles (closed intervals) or is a null axis
__device__ __forceinline__ bool axis_keeps(const Tri &q, const Tri &t, V3 ax) {
  float p = dot3(ax, q.v0), qmin = p, qmax = p;
  p = dot3(ax, q.v1); qmin = fminf(qmin, p); qmax = fmaxf(qmax, p);
  p = dot3(ax, q.v2); qmin = fminf(qmin, p); qmax = fmaxf(qmax, p);
  p = dot3(ax, t.v0); float tmin = p, tmax = p;
  p = dot3(ax, t.v1); tmin = fminf(tmin, p); tmax = fmaxf(tmax, p);
  p = dot3(ax, t.v2); tmin = fminf(tmin, p); tmax = fmaxf(tmax, p);
  if ((qmin <= tmax) && (tmin <= qmax)) return true;
  return cmp_eq(dot3(ax, ax), 0.f);
}

__device__ __forceinline__ bool tri_tri_sat(const Tri &q, const Tri &t) {
  if (!axis_keeps(q, t, sat_cross_edge(q.v0, q.v1, q.v1, q.v2))) return false;
  if (!axis_keeps(q, t, sat_cross_edge(t.v0, t.v1, t.v1, t.v2))) return false;
  const V3 qa[3] = {q.v0, q.v1, q.v2}, qb[3] = {q.v1, q.v2, q.v0};
  const V3 ta[3] = {t.v0, t.v1, t.v2}, tb[3] = {t.v1, t.v2, t.v0};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      if (!axis_keeps(q, t, sat_cross_edge(qa[i], qb[i], ta[j], tb[j]))) return false;
  return true;
}

__device__ __forceinline__ bool ray_tri(V3 orig, V3 dir, V3 v0, V3 v1, V3 v2, float &t, V3 &p) {
  const V3 v0v1 = v1 - v0, v0v2 = v2 - v0;
  const V3 pvec = cross3(dir, v0v2);
  const float det = dot3(v0v1, pvec);
  if (fabs((double)det) < 1e-4) return false;
  const float inv = 1.f / det;
  const V3 tvec = orig - v0;
  const float u = dot3(tvec, pvec) * inv;
  if (u < 0.f || u > 1.f) return false;
  const V3 qvec = cross3(tvec, v0v1);
  const float v = dot3(dir, qvec) * inv;
  if (v < 0.f || u + v > 1.f) return false;
  t = dot3(v0v2, qvec) * inv;
  p = t * dir + orig;
  return true;
}

__device__ __forceinline__ V3 to_bary(V3 p, V3 a, V3 b, V3 c) {
  const V3 e0 = b - a, e1 = c - a, e2 = p - a;
  const float d00 = dot3(e0, e0), d01 = dot3(e0, e1), d11 = dot3(e1, e1);
  const float d20 = dot3(e2, e0), d21 = dot3(e2, e1);
  const float den = d00 * d11 - d01 * d01;
  V3 bc;
  bc.y = (d11 * d20 - d01 * d21) / den;
  bc.z = (d00 * d21 - d01 * d20) / den;
  bc.x = (float)(1.0 - (double)bc.y - (double)bc.z);
  return bc;
}

// First accepted hit among query edges vs target, then target edges vs query (0 <= t <= 1).
// Returns false when none exists (the reference leaves the slot unwritten: zeros).
__device__ __forceinline__ bool tri_tri_point(const Tri &q, const Tri &tg, V3 &bc) {
  const V3 qo[3] = {q.v0, q.v1, q.v2};
  const V3 qe[3] = {q.v1 - q.v0, q.v2 - q.v1, q.v0 - q.v2};
  const V3 to[3] = {tg.v0, tg.v1, tg.v2};
  const V3 te[3] = {tg.v1 - tg.v0, tg.v2 - tg.v1, tg.v0 - tg.v2};
  float t = 0.f;
  V3 p, p1 = v3(0.f, 0.f, 0.f), p2;
  bool found = false;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const bool hit = ray_tri(qo[i], qe[i], tg.v0, tg.v1, tg.v2, t, p);
    if (t > 1.f || t < 0.f) continue;
    if (hit && !found) { p1 = p; found = true; }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const bool hit = ray_tri(to[i], te[i], q.v0, q.v1, q.v2, t, p);
    if (t > 1.f || t < 0.f) continue;
    if (hit && !found) { p1 = p; found = true; }
    // the reference re-casts from (t + EPS) along the same edge (:481-487): when it succeeds
    // its t is ~ -EPS and the loop continues; only the stale `t` is observable afterwards.
    ray_tri(to[i] + (float)((double)t + 1e-4) * te[i], te[i], q.v0, q.v1, q.v2, t, p2);
  }
  if (!found) return false;
  bc = to_bary(p1, tg.v0, tg.v1, tg.v2);
  return true;
}

__device__ __forceinline__ bool aabb_overlap(const Tri &a, const Tri &b) {
  const float axn = fminf(a.v0.x, fminf(a.v1.x, a.v2.x)), axx = fmaxf(a.v0.x, fmaxf(a.v1.x, a.v2.x));
  const float ayn = fminf(a.v0.y, fminf(a.v1.y, a.v2.y)), ayx = fmaxf(a.v0.y, fmaxf(a.v1.y, a.v2.y));
  const float azn = fminf(a.v0.z, fminf(a.v1.z, a.v2.z)), azx = fmaxf(a.v0.z, fmaxf(a.v1.z, a.v2.z));
  const float bxn = fminf(b.v0.x, fminf(b.v1.x, b.v2.x)), bxx = fmaxf(b.v0.x, fmaxf(b.v1.x, b.v2.x));
  const float byn = fminf(b.v0.y, fminf(b.v1.y, b.v2.y)), byx = fmaxf(b.v0.y, fmaxf(b.v1.y, b.v2.y));
  const float bzn = fminf(b.v0.z, fminf(b.v1.z, b.v2.z)), bzx = fmaxf(b.v0.z, fmaxf(b.v1.z, b.v2.z));
  return axn <= bxx && axx >= bxn && ayn <= byx && ayx >= byn && azn <= bzx && azx >= bzn;
}

}  // namespace shapy
