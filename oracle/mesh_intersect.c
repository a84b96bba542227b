/* ORACLE (test infrastructure only -- never linked into or loaded by shapy_amd/).
 *
 * Plain-C float32 restatement of the *observable* semantics of the reference's
 * mesh_mesh_intersect_cuda.mesh_to_mesh_forward
 *   (mesh-mesh-intersection/src/mesh_mesh_intersect_cuda_op.cu).
 * The reference's LBVH only prunes candidate pairs by AABB overlap; its result is the set
 * of (query triangle, target triangle) pairs that pass the AABB test and the 11-axis SAT
 * test, with one intersection point per pair.  This file evaluates exactly those rules
 * for every pair (no tree), emitting hits in ascending target-face order.
 *
 * Line references are to mesh_mesh_intersect_cuda_op.cu unless stated otherwise.
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: plain IEEE arithmetic, no FMA).
 *
 * The reference instantiates its kernels for float AND double (AT_DISPATCH_FLOATING_TYPES, :996).  This
 * file is compiled once per scalar type: REAL = float (default; symbols shapy_oracle_*) and, through
 * mesh_intersect_f64.c, REAL = double (symbols shapy_oracle_*_f64).  The double instantiation is NOT a
 * double-precision algorithm: CMP (:91-92) calls fabsf / fmaxf and compares against FLT_EPSILON whatever
 * the type -- the arguments are converted to float there -- and EPSILON is the literal 1e-4 in both.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifndef REAL
#define REAL float
#define SFX(name) name
#define RMIN fminf
#define RMAX fmaxf
#endif

typedef struct { REAL x, y, z; } v3;

static inline v3 mk(REAL x, REAL y, REAL z) { v3 r = {x, y, z}; return r; }
static inline v3 sub(v3 a, v3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 add(v3 a, v3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 scl(REAL s, v3 a) { return mk(s * a.x, s * a.y, s * a.z); }
static inline REAL dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 cross(v3 a, v3 b) {
  return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

/* :91-92  CMP(x, y) */
static inline int CMP(REAL x, REAL y) {
  return fabsf((float)(x - y)) <= FLT_EPSILON * fmaxf(1.0f, fmaxf(fabsf((float)x), fabsf((float)y)));
}

/* :151-169 SatCrossEdge */
static v3 sat_cross_edge(v3 a, v3 b, v3 c, v3 d) {
  v3 ab = sub(b, a), cd = sub(d, c);
  v3 result = cross(ab, cd);
  if (!CMP(dot(ab, cd), (REAL)0)) return result;
  v3 axis = cross(ab, sub(c, a));
  result = cross(ab, axis);
  if (!CMP(dot(result, result), (REAL)0)) return result;
  return mk(0, 0, 0);
}

/* :234-252 isect_interval, :254-268 TriangleTriangleOverlap (closed intervals) */
static int overlap_on_axis(const v3 *q, const v3 *t, v3 ax) {
  REAL p, qmin, qmax, tmin, tmax;
  p = dot(ax, q[0]); qmin = qmax = p;
  p = dot(ax, q[1]); qmin = RMIN(qmin, p); qmax = RMAX(qmax, p);
  p = dot(ax, q[2]); qmin = RMIN(qmin, p); qmax = RMAX(qmax, p);
  p = dot(ax, t[0]); tmin = tmax = p;
  p = dot(ax, t[1]); tmin = RMIN(tmin, p); tmax = RMAX(tmax, p);
  p = dot(ax, t[2]); tmin = RMIN(tmin, p); tmax = RMAX(tmax, p);
  return (qmin <= tmax) && (tmin <= qmax);
}

/* :270-341 TriangleTriangleIsectSepAxis(tri1 = query, tri2 = target) */
int SFX(shapy_oracle_tri_tri_sat)(const REAL *qf, const REAL *tf) {
  const v3 *q = (const v3 *)qf, *t = (const v3 *)tf;
  v3 axes[11];
  axes[0] = sat_cross_edge(q[0], q[1], q[1], q[2]);
  axes[1] = sat_cross_edge(t[0], t[1], t[1], t[2]);
  axes[2] = sat_cross_edge(q[0], q[1], t[0], t[1]);
  axes[3] = sat_cross_edge(q[0], q[1], t[1], t[2]);
  axes[4] = sat_cross_edge(q[0], q[1], t[2], t[0]);
  axes[5] = sat_cross_edge(q[1], q[2], t[0], t[1]);
  axes[6] = sat_cross_edge(q[1], q[2], t[1], t[2]);
  axes[7] = sat_cross_edge(q[1], q[2], t[2], t[0]);
  axes[8] = sat_cross_edge(q[2], q[0], t[0], t[1]);
  axes[9] = sat_cross_edge(q[2], q[0], t[1], t[2]);
  axes[10] = sat_cross_edge(q[2], q[0], t[2], t[0]);
  for (int i = 0; i < 11; ++i) {
    if (!overlap_on_axis(q, t, axes[i])) {
      if (!CMP(dot(axes[i], axes[i]), (REAL)0)) return 0;
    }
  }
  return 1;
}

/* :202-232 ray_triangle_intersect (EPSILON = 1e-4 is a double literal, :53-55) */
static int ray_tri(v3 orig, v3 dir, v3 v0, v3 v1, v3 v2, REAL *t, v3 *p) {
  v3 v0v1 = sub(v1, v0), v0v2 = sub(v2, v0);
  v3 pvec = cross(dir, v0v2);
  REAL det = dot(v0v1, pvec);
  if (fabs((double)det) < 1e-4) return 0;
  REAL invDet = 1 / det;
  v3 tvec = sub(orig, v0);
  REAL u = dot(tvec, pvec) * invDet;
  if (u < 0 || u > 1) return 0;
  v3 qvec = cross(tvec, v0v1);
  REAL v = dot(dir, qvec) * invDet;
  if (v < 0 || u + v > 1) return 0;
  *t = dot(v0v2, qvec) * invDet;
  *p = add(scl(*t, dir), orig);
  return 1;
}

/* :186-200 point_to_barycentric; "1.0 - y - z" is evaluated in double */
static void to_bary(v3 p, v3 a, v3 b, v3 c, REAL *bc) {
  v3 v0 = sub(b, a), v1 = sub(c, a), v2 = sub(p, a);
  REAL d00 = dot(v0, v0), d01 = dot(v0, v1), d11 = dot(v1, v1);
  REAL d20 = dot(v2, v0), d21 = dot(v2, v1);
  REAL denom = d00 * d11 - d01 * d01;
  bc[1] = (d11 * d20 - d01 * d21) / denom;
  bc[2] = (d00 * d21 - d01 * d20) / denom;
  bc[0] = (REAL)(1.0 - (double)bc[1] - (double)bc[2]);
}

/* :375-518 find_triangle_triangle_intersection_points.
 * Observable behaviour (SURVEY.md 2.1 notes): the "second point" search can never
 * succeed -- in the first loop it tests a degenerate triangle (v1,v1,v2) (:433, det == 0),
 * in the second loop the re-cast ray starts (t+EPS) past the hit so its t is -EPS < 0
 * (:481-487) -- hence both output slots receive the barycentrics (w.r.t. the TARGET
 * triangle) of the first accepted hit among: query edges vs target (i=0,1,2), then target
 * edges vs query (i=0,1,2); a hit is accepted iff the ray test succeeds and 0 <= t <= 1.
 * The reference reads `t` uninitialised when the very first test fails (:412,:419; UB with
 * no observable effect because nothing is recorded on that path); we start it at 0.
 * Returns 1 if a point was found (bc[0..2] and bc[3..5] written), else 0 (bc untouched). */
int SFX(shapy_oracle_tri_tri_point)(const REAL *qf, const REAL *tf, REAL *bc) {
  const v3 *q = (const v3 *)qf, *tg = (const v3 *)tf;
  v3 qe[3] = {sub(q[1], q[0]), sub(q[2], q[1]), sub(q[0], q[2])};
  v3 te[3] = {sub(tg[1], tg[0]), sub(tg[2], tg[1]), sub(tg[0], tg[2])};
  REAL t = 0;
  v3 p, p2, p1 = mk(0, 0, 0);
  int found_first = 0;
  for (int i = 0; i < 3; ++i) {
    int hit = ray_tri(q[i], qe[i], tg[0], tg[1], tg[2], &t, &p);
    if (t > 1 || t < 0) continue;
    if (hit && !found_first) { p1 = p; found_first = 1; }
    /* second cast against (v1,v1,v2): det == 0 -> always false, t unchanged */
  }
  for (int i = 0; i < 3; ++i) {
    int hit = ray_tri(tg[i], te[i], q[0], q[1], q[2], &t, &p);
    if (t > 1 || t < 0) continue;
    if (hit && !found_first) { p1 = p; found_first = 1; }
    /* second cast along the same line from (t+EPS) further: if it succeeds its t is
     * -EPS and the loop `continue`s (:487); it still overwrites `t` (:481-486). */
    ray_tri(add(tg[i], scl((REAL)((double)t + 1e-4), te[i])), te[i], q[0], q[1], q[2], &t,
            &p2);
  }
  if (!found_first) return 0;
  to_bary(p1, tg[0], tg[1], tg[2], bc);
  bc[3] = bc[0]; bc[4] = bc[1]; bc[5] = bc[2];
  return 1;
}

/* triangle.hpp:48-52 bbox + :363-373 checkOverlap (closed comparisons) */
static int aabb_overlap(const v3 *a, const v3 *b) {
  REAL amin[3], amax[3], bmin[3], bmax[3];
  const REAL *af = (const REAL *)a, *bf = (const REAL *)b;
  for (int k = 0; k < 3; ++k) {
    amin[k] = RMIN(af[k], RMIN(af[3 + k], af[6 + k]));
    amax[k] = RMAX(af[k], RMAX(af[3 + k], af[6 + k]));
    bmin[k] = RMIN(bf[k], RMIN(bf[3 + k], bf[6 + k]));
    bmax[k] = RMAX(bf[k], RMAX(bf[3 + k], bf[6 + k]));
  }
  return amin[0] <= bmax[0] && amax[0] >= bmin[0] && amin[1] <= bmax[1] &&
         amax[1] >= bmin[1] && amin[2] <= bmax[2] && amax[2] >= bmin[2];
}

/* mesh_mesh_intersect.cpp:36-57 + .cu:969-1079 as observable:
 *   faces_out  int64 [B, Q*max_coll]        (-1 = empty)
 *   bcs_out    REAL [B, Q*max_coll, 2, 3]  (0 = empty)
 * Hits of query triangle q are written to slots q*max_coll + 0.. in ascending target index
 * (the reference's slot order is BVH traversal order, unspecified for consumers).  The
 * reference has no bound check (:551,565); hits beyond max_coll are dropped here and the
 * total number of dropped hits is returned. */
long SFX(shapy_oracle_mesh_to_mesh)(const REAL *query, const REAL *target, int B, int Q, int F,
                               int max_coll, int64_t *faces_out, REAL *bcs_out) {
  long dropped = 0;
  for (long i = 0; i < (long)B * Q * max_coll; ++i) faces_out[i] = -1;
  memset(bcs_out, 0, sizeof(REAL) * (size_t)B * Q * max_coll * 6);
  for (int b = 0; b < B; ++b) {
    for (int qi = 0; qi < Q; ++qi) {
      const REAL *qf = query + ((size_t)b * Q + qi) * 9;
      int n = 0;
      for (int f = 0; f < F; ++f) {
        const REAL *tf = target + ((size_t)b * F + f) * 9;
        if (!aabb_overlap((const v3 *)qf, (const v3 *)tf)) continue;
        if (!SFX(shapy_oracle_tri_tri_sat)(qf, tf)) continue;
        if (n >= max_coll) { ++dropped; continue; }
        size_t slot = ((size_t)b * Q + qi) * max_coll + n;
        faces_out[slot] = f;
        SFX(shapy_oracle_tri_tri_point)(qf, tf, bcs_out + slot * 6);
        ++n;
      }
    }
  }
  return dropped;
}

/* body_measurements.py:201-215 compute_mass (volume only: |sum signed tetra| / 6) and
 * :182-199 compute_height inputs are evaluated in numpy (oracle/measure.py). */
