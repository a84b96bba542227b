// Regressor head + SMPL-X kernels (everything between the backbone features and the posed
// body).  One workgroup per body for the small per-body stages; the two blend-shape GEMMs run
// on the MFMA conv/GEMM kernel (conv_igemm.hip) and are issued by the host.
#include <stdlib.h>

#include "common.h"

namespace shapy {

// ------------------------------------------------------------------------------------------
// IterativeRegression (networks.py:536-592) with the affine-collapsed MLP:
//   t = Wf feat + b ;  p_s = p_{s-1} + t + Wp p_{s-1}
// ------------------------------------------------------------------------------------------
// Phase 1: t[b][r] = Wf[r] . feat[b] + bias[r].  One wave per output row, RB bodies per
// workgroup: a row of Wf (8 KB) is read once per RB bodies and the grid has P/4 x B/RB
// workgroups instead of B (the one-workgroup-per-body version re-read the whole 1.2 MB of Wf for
// every body and ran on 64 of 256 CUs: 196 us at B = 64).  Per (row, body) the summation order
// is unchanged: lane l accumulates k = 4 l + 256 i in order, then a butterfly over the wave.
constexpr int REG_RB = 8;
// PS > 0: the rows are S stacked stages of PS parameters each (fully collapsed regressor,
// shapy_regressor_collapsed_f32) and row r = s * PS + i of body b is written to
// t_out[(s * B + b) * PS + i], the [S,B,P] layout of params_out.
__global__ __launch_bounds__(256) void regressor_feat_kernel(
    const float *__restrict__ feat, const float *__restrict__ Wf, const float *__restrict__ bias,
    float *__restrict__ t_out, int B, int F, int P, int PS) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wave, b0 = blockIdx.y * REG_RB;
  if (r >= P) return;
  const float *w = Wf + (long)r * F;
  float s[REG_RB];
#pragma unroll
  for (int j = 0; j < REG_RB; ++j) s[j] = 0.f;
  for (int k = lane * 4; k < F; k += 256) {
    const float4 wv = *reinterpret_cast<const float4 *>(w + k);
#pragma unroll
    for (int j = 0; j < REG_RB; ++j) {
      const int b = b0 + j < B ? b0 + j : B - 1;
      const float4 xv = *reinterpret_cast<const float4 *>(feat + (long)b * F + k);
      s[j] = fmaf(wv.x, xv.x, s[j]); s[j] = fmaf(wv.y, xv.y, s[j]);
      s[j] = fmaf(wv.z, xv.z, s[j]); s[j] = fmaf(wv.w, xv.w, s[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < REG_RB; ++j) {
    const float v = wave_reduce_sum(s[j]);
    if (lane == 0 && b0 + j < B) {
      if (PS > 0) t_out[((long)(r / PS) * B + b0 + j) * PS + r % PS] = v + bias[r];
      else t_out[(long)(b0 + j) * P + r] = v + bias[r];
    }
  }
}

// Phase 2: the stages, p_s = p_{s-1} + t + Wp p_{s-1}; one workgroup per body.  t arrives in
// the slot of the LAST stage of `out` and is copied to LDS before anything is written.
__global__ __launch_bounds__(256) void regressor_stage_kernel(
    const float *__restrict__ Wp, const float *__restrict__ mean, float *__restrict__ out, int B,
    int P, int S, int mean_stride) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float *t = sm;            // [P]
  float *pa = sm + P;       // [P]
  float *pb = sm + 2 * P;   // [P]
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < P; i += 256) {
    t[i] = out[((long)(S - 1) * B + b) * P + i];
    pa[i] = mean[(long)b * mean_stride + i];
  }
  __syncthreads();
  float *prev = pa, *next = pb;
  const int lane = tid & 63, wave = tid >> 6;
  for (int s = 0; s < S; ++s) {
    // one wave per output row: coalesced reads of the Wp row, butterfly sum (a thread walking
    // its own row touched one cache line per element: 79 us per call at B = 64)
    for (int i = wave; i < P; i += 4) {
      const float *w = Wp + (long)i * P;
      float acc = 0.f;
      for (int j = lane; j < P; j += 64) acc = fmaf(w[j], prev[j], acc);
      acc = wave_reduce_sum(acc);
      if (lane == 0) {
        const float v = prev[i] + (t[i] + acc);
        next[i] = v;
        out[((long)s * B + b) * P + i] = v;
      }
    }
    __syncthreads();
    float *tmp = prev; prev = next; next = tmp;
  }
}

// ------------------------------------------------------------------------------------------
// pose decode + joint regression + kinematic chain, one 64-lane workgroup per body
// ------------------------------------------------------------------------------------------
struct PoseK {
  const int32_t *parents, *neck;
  const float *Jt, *Js, *pose, *coeffs;
  float *rot, *pf, *A, *joints;
  int32_t *dyn_row;
  int J, NB, NBpad, P, Ppad, n_pose, pose_type, n_neck, n_dyn_rows;
};

__device__ __forceinline__ void mat3_mul(const float *a, const float *b, float *c) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      c[i * 3 + j] = a[i * 3 + 0] * b[0 * 3 + j] + a[i * 3 + 1] * b[1 * 3 + j] + a[i * 3 + 2] * b[2 * 3 + j];
}

__global__ __launch_bounds__(64) void smplx_pose_kernel(PoseK k) {
  __shared__ float R[64 * 9];
  __shared__ float Jl[64 * 3];
  __shared__ float G[64 * 12];
  __shared__ int Par[64];
  const int b = blockIdx.x, j = threadIdx.x;
  if (j < k.J) {
    float r[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
    if (j < k.n_pose) {
      if (k.pose_type == SHAPY_POSE_CONT6D) {
        // ContinuousRotReprDecoder.forward (pose_utils.py:138-153): x viewed as (3,2)
        const float *x = k.pose + ((long)b * k.n_pose + j) * 6;
        const float a1x = x[0], a2x = x[1], a1y = x[2], a2y = x[3], a1z = x[4], a2z = x[5];
        float n1 = sqrtf(a1x * a1x + a1y * a1y + a1z * a1z);
        n1 = fmaxf(n1, 1e-12f);
        const float b1x = a1x / n1, b1y = a1y / n1, b1z = a1z / n1;
        const float d = b1x * a2x + b1y * a2y + b1z * a2z;
        const float cx = a2x - d * b1x, cy = a2y - d * b1y, cz = a2z - d * b1z;
        float n2 = sqrtf(cx * cx + cy * cy + cz * cz);
        n2 = fmaxf(n2, 1e-12f);
        const float b2x = cx / n2, b2y = cy / n2, b2z = cz / n2;
        const float b3x = b1y * b2z - b1z * b2y, b3y = b1z * b2x - b1x * b2z,
                    b3z = b1x * b2y - b1y * b2x;
        r[0] = b1x; r[1] = b2x; r[2] = b3x;
        r[3] = b1y; r[4] = b2y; r[5] = b3y;
        r[6] = b1z; r[7] = b2z; r[8] = b3z;
      } else if (k.pose_type == SHAPY_POSE_AXIS_ANGLE) {
        // batch_rodrigues (rotation_utils.py:5-37): eps added to the vector before the norm
        const float *x = k.pose + ((long)b * k.n_pose + j) * 3;
        const float ex = x[0] + 1e-8f, ey = x[1] + 1e-8f, ez = x[2] + 1e-8f;
        const float ang = sqrtf(ex * ex + ey * ey + ez * ez);
        const float rx = x[0] / ang, ry = x[1] / ang, rz = x[2] / ang;
        const float c = cosf(ang), s = sinf(ang), oc = 1.f - c;
        // K = [0 -rz ry; rz 0 -rx; -ry rx 0];  R = I + s K + (1-c) K K
        const float K[9] = {0.f, -rz, ry, rz, 0.f, -rx, -ry, rx, 0.f};
        float KK[9];
        mat3_mul(K, K, KK);
#pragma unroll
        for (int i = 0; i < 9; ++i) r[i] = ((i % 4 == 0) ? 1.f : 0.f) + s * K[i] + oc * KK[i];
      } else {
        const float *x = k.pose + ((long)b * k.n_pose + j) * 9;
#pragma unroll
        for (int i = 0; i < 9; ++i) r[i] = x[i];
      }
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      R[j * 9 + i] = r[i];
      k.rot[((long)b * k.J + j) * 9 + i] = r[i];
    }
    // rest joints: J = J_template + J_shapedirs . coeffs   (== J_regressor (v_template + S c))
    // One fmaf chain per axis in coefficient order, as before -- but the three rows of this joint are
    // walked together and four coefficients at a time, so that 15 loads are in flight instead of one
    // (the loop used to be 3 x NB dependent global loads: most of the kernel's 25 us at B = 64).
    const float *c = k.coeffs + (long)b * k.NBpad;
    const float *js = k.Js + (long)j * 3 * k.NB;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    int l = 0;
    for (; l + 4 <= k.NB; l += 4) {
      float cv[4], a0[4], a1[4], a2[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        cv[q] = c[l + q];
        a0[q] = js[l + q];
        a1[q] = js[k.NB + l + q];
        a2[q] = js[2 * k.NB + l + q];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        s0 = fmaf(a0[q], cv[q], s0);
        s1 = fmaf(a1[q], cv[q], s1);
        s2 = fmaf(a2[q], cv[q], s2);
      }
    }
    for (; l < k.NB; ++l) {
      s0 = fmaf(js[l], c[l], s0);
      s1 = fmaf(js[k.NB + l], c[l], s1);
      s2 = fmaf(js[2 * k.NB + l], c[l], s2);
    }
    Jl[j * 3 + 0] = k.Jt[j * 3 + 0] + s0;
    Jl[j * 3 + 1] = k.Jt[j * 3 + 1] + s1;
    Jl[j * 3 + 2] = k.Jt[j * 3 + 2] + s2;
    // pose feature (lbs.py:176): (R[1:] - I) flattened
    if (j >= 1) {
#pragma unroll
      for (int i = 0; i < 9; ++i)
        k.pf[(long)b * k.Ppad + (j - 1) * 9 + i] = r[i] - ((i % 4 == 0) ? 1.f : 0.f);
    }
  }
  for (int i = k.P + j; i < k.Ppad; i += 64) k.pf[(long)b * k.Ppad + i] = 0.f;
  // parents and tree depth of every joint in LDS: the chain below used to read k.parents[i] from
  // global memory inside a serial 54-step loop on lane 0 (a dependent ~0.5 us load per joint)
  Par[j] = j < k.J ? k.parents[j] : -1;
  __syncthreads();
  int depth = 0;
  if (j < k.J)
    for (int a = j; a > 0 && depth < 64; a = Par[a]) ++depth;
  // batch_rigid_transform (lbs.py:242-295): G_i = G_parent * [R_i | J_i - J_parent], one tree
  // level per step, the joints of a level in parallel (same per-joint arithmetic as the serial
  // walk: every G_i is one 3x4 product of its parent's G and its own local transform)
  if (j == 0) {
#pragma unroll
    for (int i = 0; i < 9; ++i) G[(i / 3) * 4 + (i % 3)] = R[i];
    G[3] = Jl[0]; G[7] = Jl[1]; G[11] = Jl[2];
  }
  __syncthreads();
  for (int d = 1; d < 64; ++d) {
    if (__ballot(depth >= d) == 0ull) break;
    if (depth == d) {
      const int i = j, pa = Par[i];
      const float *Gp = G + pa * 12;
      const float *Ri = R + i * 9;
      const float rel[3] = {Jl[i * 3] - Jl[pa * 3], Jl[i * 3 + 1] - Jl[pa * 3 + 1],
                            Jl[i * 3 + 2] - Jl[pa * 3 + 2]};
      float *Gi = G + i * 12;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
          Gi[r * 4 + c] = Gp[r * 4 + 0] * Ri[0 * 3 + c] + Gp[r * 4 + 1] * Ri[1 * 3 + c] +
                          Gp[r * 4 + 2] * Ri[2 * 3 + c];
        Gi[r * 4 + 3] = Gp[r * 4 + 0] * rel[0] + Gp[r * 4 + 1] * rel[1] + Gp[r * 4 + 2] * rel[2] +
                        Gp[r * 4 + 3];
      }
    }
    __syncthreads();
  }
  if (j == 0) {
    // dynamic-landmark LUT row (lbs.py:28-41): rel = R[c0] (R[c1] (... I)), applied in order
    if (k.dyn_row) {
      float rel[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f}, tmp[9];
      for (int q = 0; q < k.n_neck; ++q) {
        mat3_mul(R + k.neck[q] * 9, rel, tmp);
#pragma unroll
        for (int i = 0; i < 9; ++i) rel[i] = tmp[i];
      }
      const float sy = sqrtf(rel[0] * rel[0] + rel[3] * rel[3]);
      const float ang = -atan2f(-rel[6], sy) * 180.0f / 3.14159265358979323846f;
      int y = (int)rintf(fminf(ang, 39.f));
      if (y < 0) y = (y < -39) ? 78 : 39 - y;
      if (y > k.n_dyn_rows - 1) y = k.n_dyn_rows - 1;
      k.dyn_row[b] = y;
    }
  }
  __syncthreads();
  if (j < k.J) {
    const float *Gj = G + j * 12;
    float *Ao = k.A + ((long)b * k.J + j) * 12;
    const float jx = Jl[j * 3], jy = Jl[j * 3 + 1], jz = Jl[j * 3 + 2];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      Ao[r * 4 + 0] = Gj[r * 4 + 0];
      Ao[r * 4 + 1] = Gj[r * 4 + 1];
      Ao[r * 4 + 2] = Gj[r * 4 + 2];
      Ao[r * 4 + 3] = Gj[r * 4 + 3] - (Gj[r * 4 + 0] * jx + Gj[r * 4 + 1] * jy + Gj[r * 4 + 2] * jz);
      k.joints[((long)b * k.J + j) * 3 + r] = Gj[r * 4 + 3];
    }
  }
}

// ------------------------------------------------------------------------------------------
// skinning (lbs.py:187-190): T[b,v] = sum_j W[v,j] A[b,j] (3x4), vertices = T [v_posed; 1].
// One thread owns NV vertices (256 apart) and NBODY consecutive bodies.  The transforms A[b][j] sit in LDS
// and are read as broadcasts -- which cost the LDS pipe a full 1 KB pass per wave instruction all the
// same, so the register block has to amortise them: 4 vertices x 2 bodies = 6 ds_read_b128 per 48
// v_pk_fma_f32 and joint, LDS and VALU pipes level.  (Round 5, first form: 1 vertex x 4 bodies, 12 reads per
// 24 FMAs: 24.0 us at B = 64, exactly the LDS bound; round 4: 1 x 1, 21.7 us.)  The weights W[j][v]
// (coalesced over v) are requested five joints ahead.  Per output the sum runs over the joints in index order
// -- the same fmaf chain as one thread per (body, vertex).
// ------------------------------------------------------------------------------------------
template <int NV, int NBODY>
__global__ __launch_bounds__(256) void smplx_skin_kernel(const float *__restrict__ Wt,
                                                         const float *__restrict__ A,
                                                         const float *__restrict__ v_posed,
                                                         float *__restrict__ out, int V, int J, int B) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  __shared__ __attribute__((aligned(16))) float As[NBODY][64 * 12];
  const int b0 = blockIdx.y * NBODY;
  for (int i = threadIdx.x; i < NBODY * J * 12; i += 256) {
    const int q = i / (J * 12), r = i % (J * 12);
    const int b = b0 + q < B ? b0 + q : B - 1;            // (tail bodies: a copy of the last one, never stored)
    As[q][r] = A[(long)b * J * 12 + r];
  }
  __syncthreads();
  const int v0 = blockIdx.x * (256 * NV) + threadIdx.x;
  int vi[NV];                                             // (clamped: out-of-range vertices are never stored)
#pragma unroll
  for (int u = 0; u < NV; ++u) vi[u] = v0 + 256 * u < V ? v0 + 256 * u : V - 1;
  f32x2 T[NV][NBODY][6];
#pragma unroll
  for (int u = 0; u < NV; ++u)
#pragma unroll
    for (int q = 0; q < NBODY; ++q)
#pragma unroll
      for (int i = 0; i < 6; ++i) T[u][q][i] = f32x2{0.f, 0.f};
  // weights a whole group of JG joints ahead: there are only ~1.4 waves per SIMD at B = 64, so nothing else
  // hides the ~700-cycle L2 latency -- with one joint of lookahead (192 cycles of FMAs) the kernel took 25.9 us
  // whatever the register block
  constexpr int JG = 5;
  float wn[JG][NV];
  auto wload = [&](int j0) {
#pragma unroll
    for (int g = 0; g < JG; ++g) {
      const int j = j0 + g < J ? j0 + g : J - 1;
#pragma unroll
      for (int u = 0; u < NV; ++u) wn[g][u] = Wt[(long)j * V + vi[u]];
    }
  };
  wload(0);
  for (int j0 = 0; j0 < J; j0 += JG) {
    float w[JG][NV];
#pragma unroll
    for (int g = 0; g < JG; ++g)
#pragma unroll
      for (int u = 0; u < NV; ++u) w[g][u] = wn[g][u];
    if (j0 + JG < J) wload(j0 + JG);
#pragma unroll
    for (int g = 0; g < JG; ++g) {
      if (j0 + g >= J) break;                             // (uniform)
#pragma unroll
      for (int q = 0; q < NBODY; ++q) {
        const f32x4 *a = reinterpret_cast<const f32x4 *>(&As[q][(j0 + g) * 12]);
        const f32x4 a0 = a[0], a1 = a[1], a2 = a[2];
        const f32x2 av[6] = {f32x2{a0[0], a0[1]}, f32x2{a0[2], a0[3]}, f32x2{a1[0], a1[1]},
                             f32x2{a1[2], a1[3]}, f32x2{a2[0], a2[1]}, f32x2{a2[2], a2[3]}};
#pragma unroll
        for (int u = 0; u < NV; ++u) {
          const f32x2 ww = f32x2{w[g][u], w[g][u]};
#pragma unroll
          for (int i = 0; i < 6; ++i) T[u][q][i] = __builtin_elementwise_fma(ww, av[i], T[u][q][i]);
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    const int v = v0 + 256 * u;
    if (v >= V) break;
#pragma unroll
    for (int q = 0; q < NBODY; ++q) {
      const int b = b0 + q;
      if (b >= B) break;
      const float *p = v_posed + ((long)b * V + v) * 3;
      const float x = p[0], y = p[1], z = p[2];
      float *o = out + ((long)b * V + v) * 3;
      o[0] = T[u][q][0][0] * x + T[u][q][0][1] * y + T[u][q][1][0] * z + T[u][q][1][1];
      o[1] = T[u][q][2][0] * x + T[u][q][2][1] * y + T[u][q][3][0] * z + T[u][q][3][1];
      o[2] = T[u][q][4][0] * x + T[u][q][4][1] * y + T[u][q][5][0] * z + T[u][q][5][1];
    }
  }
}

// ------------------------------------------------------------------------------------------
// output keypoints: posed joints ++ landmarks (lbs.py:52-94) and weak-perspective projection
// (camera_projection.py:181-213, iterative_regressor.py:714-733)
// ------------------------------------------------------------------------------------------
struct JointsK {
  const int32_t *faces, *lmk_idx, *dyn_idx, *dyn_row;
  const float *lmk_bc, *dyn_bc, *posed, *verts, *cam;
  float *joints, *proj, *scale;
  int J, V, n_static, n_dyn, n_out;
};

__global__ __launch_bounds__(128) void smplx_joints_kernel(JointsK k) {
  const int b = blockIdx.x;
  float s = 1.f, tx = 0.f, ty = 0.f;
  if (k.cam) {
    const float c0 = k.cam[b * 3];
    s = c0 > 20.f ? c0 : log1pf(expf(c0));   // F.softplus (beta 1, threshold 20)
    tx = k.cam[b * 3 + 1];
    ty = k.cam[b * 3 + 2];
    if (threadIdx.x == 0 && k.scale) k.scale[b] = s;
  }
  for (int i = threadIdx.x; i < k.n_out; i += 128) {
    float x, y, z;
    if (i < k.J) {
      const float *p = k.posed + ((long)b * k.J + i) * 3;
      x = p[0]; y = p[1]; z = p[2];
    } else {
      const int l = i - k.J;
      int f;
      const float *bc;
      if (l < k.n_static) {
        f = k.lmk_idx[l];
        bc = k.lmk_bc + l * 3;
      } else {
        const int row = k.dyn_row[b];
        f = k.dyn_idx[row * k.n_dyn + (l - k.n_static)];
        bc = k.dyn_bc + ((long)row * k.n_dyn + (l - k.n_static)) * 3;
      }
      const float *vb = k.verts + (long)b * k.V * 3;
      const float *p0 = vb + (long)k.faces[f * 3 + 0] * 3;
      const float *p1 = vb + (long)k.faces[f * 3 + 1] * 3;
      const float *p2 = vb + (long)k.faces[f * 3 + 2] * 3;
      x = (p0[0] * bc[0] + p1[0] * bc[1]) + p2[0] * bc[2];
      y = (p0[1] * bc[0] + p1[1] * bc[1]) + p2[1] * bc[2];
      z = (p0[2] * bc[0] + p1[2] * bc[1]) + p2[2] * bc[2];
    }
    float *o = k.joints + ((long)b * k.n_out + i) * 3;
    o[0] = x; o[1] = y; o[2] = z;
    if (k.proj) {
      k.proj[((long)b * k.n_out + i) * 2 + 0] = s * (x + tx);
      k.proj[((long)b * k.n_out + i) * 2 + 1] = s * (y + ty);
    }
  }
}


// ------------------------------------------------------------------------------------------
// stand-alone pose decoders (pose_utils.py:138-153 / rotation_utils.py:5-37) and camera
// ------------------------------------------------------------------------------------------
// one rotation from its parameters: 6-D continuous representation (pose_utils.py:138-153,
// x viewed as (3,2): interleaved [a1x,a2x,a1y,a2y,a1z,a2z]) or axis-angle (rotation_utils.py:5-37)
__device__ __forceinline__ void decode_rotation(const float *p, int type, float (&r)[9]) {
  if (type == SHAPY_POSE_CONT6D) {
    const float a1x = p[0], a2x = p[1], a1y = p[2], a2y = p[3], a1z = p[4], a2z = p[5];
    const float n1 = fmaxf(sqrtf(a1x * a1x + a1y * a1y + a1z * a1z), 1e-12f);
    const float b1x = a1x / n1, b1y = a1y / n1, b1z = a1z / n1;
    const float d = b1x * a2x + b1y * a2y + b1z * a2z;
    const float cx = a2x - d * b1x, cy = a2y - d * b1y, cz = a2z - d * b1z;
    const float n2 = fmaxf(sqrtf(cx * cx + cy * cy + cz * cz), 1e-12f);
    const float b2x = cx / n2, b2y = cy / n2, b2z = cz / n2;
    r[0] = b1x; r[1] = b2x; r[2] = b1y * b2z - b1z * b2y;
    r[3] = b1y; r[4] = b2y; r[5] = b1z * b2x - b1x * b2z;
    r[6] = b1z; r[7] = b2z; r[8] = b1x * b2y - b1y * b2x;
  } else {
    const float ex = p[0] + 1e-8f, ey = p[1] + 1e-8f, ez = p[2] + 1e-8f;
    const float ang = sqrtf(ex * ex + ey * ey + ez * ez);
    const float rx = p[0] / ang, ry = p[1] / ang, rz = p[2] / ang;
    const float c = cosf(ang), s = sinf(ang), oc = 1.f - c;
    const float K[9] = {0.f, -rz, ry, rz, 0.f, -rx, -ry, rx, 0.f};
    float KK[9];
    mat3_mul(K, K, KK);
#pragma unroll
    for (int q = 0; q < 9; ++q) r[q] = ((q % 4 == 0) ? 1.f : 0.f) + s * K[q] + oc * KK[q];
  }
}

__global__ void pose_decode_kernel(const float *__restrict__ x, int type, float *__restrict__ out,
                                   long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float r[9];
  decode_rotation(x + i * (type == SHAPY_POSE_CONT6D ? 6 : 3), type, r);
#pragma unroll
  for (int q = 0; q < 9; ++q) out[i * 9 + q] = r[q];
}

// Everything between the regressor's parameter vectors and the SMPL-X kernels in ONE launch
// (iterative_regressor.py:646-660 + the argument glue of SMPLX.forward, body_models.py:660-700):
// decode the poses of ALL stages, build the zero-padded shape-coefficient rows of the last
// stage and a contiguous copy of its camera parameters.
struct PrepK {
  const float *params;           // [S, B, P]
  float *rot, *coeffs, *cam;     // [S, B, nj, 3, 3], [B, NBpad], [B, 3]
  int S, B, P, pose_off, nj, pose_type, betas_off, n_betas, NBpad, cam_off;
};

__global__ void head_prepare_kernel(PrepK k) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n_rot = (long)k.S * k.B * k.nj, n_co = (long)k.B * k.NBpad;
  const int per = k.pose_type == SHAPY_POSE_CONT6D ? 6 : 3;
  if (i < n_rot) {
    const int j = (int)(i % k.nj);
    const long sb = i / k.nj;
    float r[9];
    decode_rotation(k.params + sb * k.P + k.pose_off + j * per, k.pose_type, r);
#pragma unroll
    for (int q = 0; q < 9; ++q) k.rot[i * 9 + q] = r[q];
  } else if (i < n_rot + n_co) {
    const long q = i - n_rot;
    const int c = (int)(q % k.NBpad);
    const long b = q / k.NBpad;
    k.coeffs[q] = c < k.n_betas ? k.params[((long)(k.S - 1) * k.B + b) * k.P + k.betas_off + c] : 0.f;
  } else if (k.cam && i < n_rot + n_co + (long)k.B * 3) {
    const long q = i - n_rot - n_co;
    k.cam[q] = k.params[((long)(k.S - 1) * k.B + q / 3) * k.P + k.cam_off + q % 3];
  }
}

// Argument glue of SMPLX.forward (body_models.py:660-700: the eye / cat of the pose parts, the zero
// padded coefficient rows) in ONE launch: up to 7 pose parts (rotation matrices [B, n_k, 3, 3], NULL =
// identity) -> pose [B, n_pose, 3, 3]; betas (+ expression) -> coeffs [B, NBpad] and, with an
// expression, coeffs_shape (the same with the expression part zeroed).
struct SmplxPrepK {
  const float *part[7];
  long bs[7];                    // floats between consecutive bodies of part k (0: one row for all)
  int n[7], first[7];            // joints of part k, index of its first joint
  int n_parts, n_pose, B, nb, ne, NBpad;
  const float *betas, *expr;
  long betas_bs, expr_bs;        // floats between consecutive rows (0: one row for all)
  float *pose, *coeffs, *coeffs_shape;
};

__global__ void smplx_prepare_kernel(SmplxPrepK k) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n_rot = (long)k.B * k.n_pose * 9, n_co = (long)k.B * k.NBpad;
  if (i < n_rot) {
    const int q = (int)(i % 9);
    const long bj = i / 9;
    const int j = (int)(bj % k.n_pose);
    const long b = bj / k.n_pose;
    float v = (q == 0 || q == 4 || q == 8) ? 1.f : 0.f;
#pragma unroll
    for (int p = 0; p < 7; ++p)
      if (p < k.n_parts && k.part[p] && j >= k.first[p] && j < k.first[p] + k.n[p])
        v = k.part[p][b * k.bs[p] + (j - k.first[p]) * 9 + q];
    k.pose[i] = v;
  } else if (i < n_rot + n_co) {
    const long r = i - n_rot;
    const int c = (int)(r % k.NBpad);
    const long b = r / k.NBpad;
    const float sh = (c < k.nb && k.betas) ? k.betas[b * k.betas_bs + c] : 0.f;
    const float ex = (c >= k.nb && c < k.nb + k.ne && k.expr) ? k.expr[b * k.expr_bs + (c - k.nb)] : 0.f;
    k.coeffs[r] = c < k.nb ? sh : ex;
    if (k.coeffs_shape) k.coeffs_shape[r] = sh;
  }
}

__global__ void weak_persp_kernel(const float *__restrict__ pts, const float *__restrict__ scale,
                                  const float *__restrict__ transl, float *__restrict__ out, int N,
                                  int scale_first, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long b = i / N;
  const float s = scale[b], tx = transl[b * 2], ty = transl[b * 2 + 1];
  const float x = pts[i * 3], y = pts[i * 3 + 1];
  if (scale_first) {
    out[i * 2] = s * x + tx;
    out[i * 2 + 1] = s * y + ty;
  } else {
    out[i * 2] = s * (x + tx);
    out[i * 2 + 1] = s * (y + ty);
  }
}


// extra joint regressor (body_models.py:738-744): out[b,j,:] = sum_v R[j,v] * vertices[b,v,:]
__global__ __launch_bounds__(256) void joint_regress_kernel(const float *__restrict__ R,
                                                            const float *__restrict__ verts,
                                                            float *__restrict__ out, int V, int Jn) {
  __shared__ float red[4][3];
  const int j = blockIdx.x, b = blockIdx.y;
  const float *r = R + (long)j * V;
  const float *vb = verts + (long)b * V * 3;
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (int v = threadIdx.x; v < V; v += 256) {
    const float w = r[v];
    sx = fmaf(w, vb[v * 3], sx);
    sy = fmaf(w, vb[v * 3 + 1], sy);
    sz = fmaf(w, vb[v * 3 + 2], sz);
  }
  sx = wave_reduce_sum(sx); sy = wave_reduce_sum(sy); sz = wave_reduce_sum(sz);
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6][0] = sx; red[threadIdx.x >> 6][1] = sy; red[threadIdx.x >> 6][2] = sz;
  }
  __syncthreads();
  if (threadIdx.x < 3)
    out[((long)b * Jn + j) * 3 + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}


// B2A: degree-2 polynomial features of the betas -> Linear (attributes/attributes/
// attributes_betas/polynomial.py:61-69,137-140).  Monomial order = itertools
// combinations_with_replacement(range(NB), 1) ++ (..., 2): x_0..x_{NB-1}, x_0x_0, x_0x_1, ...
__global__ void b2a_polynomial_kernel(const float *__restrict__ betas, const float *__restrict__ W,
                                      const float *__restrict__ bias, float *__restrict__ out,
                                      int NB, int NA, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int a = (int)(i % NA);
  const long b = i / NA;
  const float *x = betas + b * NB;
  const float *w = W + (long)a * (NB + NB * (NB + 1) / 2);
  float s = bias[a];
  int m = 0;
  for (int p = 0; p < NB; ++p) s = fmaf(w[m++], x[p], s);
  for (int p = 0; p < NB; ++p)
    for (int q = p; q < NB; ++q) s = fmaf(w[m++], x[p] * x[q], s);
  out[i] = s;
}

}  // namespace shapy

using namespace shapy;

extern "C" int shapy_regressor_affine_f32(const float *features, const float *Wf, const float *Wp,
                                          const float *bias, const float *mean_param,
                                          float *params_out, int B, int F, int P, int num_stages,
                                          int cond_per_body, void *stream) {
  if (B <= 0) return SHAPY_OK;
  if ((F & 3) || P <= 0 || num_stages < 1) return SHAPY_EINVAL;
  float *t = params_out + (long)(num_stages - 1) * B * P;
  hipLaunchKernelGGL(regressor_feat_kernel, dim3((P + 3) / 4, (B + REG_RB - 1) / REG_RB), dim3(256),
                     0, (hipStream_t)stream, features, Wf, bias, t, B, F, P, 0);
  SHAPY_HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(regressor_stage_kernel, dim3(B), dim3(256), 3 * P * sizeof(float),
                     (hipStream_t)stream, Wp, mean_param, params_out, B, P, num_stages,
                     cond_per_body ? P : 0);
  return (int)hipGetLastError();
}

// All stages in one launch: params_out[s] = W_all[s] feat + b_all[s]  (the host collapses
// p_s = p_{s-1} + t + Wp p_{s-1}, p_0 = mean, t = Wf feat + b into W_all [S*P, F], b_all [S*P]
// in float64; valid when every body starts from the same mean, i.e. cond is None).
extern "C" int shapy_regressor_collapsed_f32(const float *features, const float *W_all,
                                             const float *b_all, float *params_out, int B, int F,
                                             int P, int num_stages, void *stream) {
  if (B <= 0) return SHAPY_OK;
  if ((F & 3) || P <= 0 || num_stages < 1) return SHAPY_EINVAL;
  const int R = P * num_stages;
  hipLaunchKernelGGL(regressor_feat_kernel, dim3((R + 3) / 4, (B + REG_RB - 1) / REG_RB), dim3(256),
                     0, (hipStream_t)stream, features, W_all, b_all, params_out, B, F, R, P);
  return (int)hipGetLastError();
}

extern "C" int shapy_smplx_pose_f32(const ShapySmplxModel *m, const float *pose, int pose_type,
                                    int n_pose, const float *coeffs, float *rot_out,
                                    float *pose_feat_out, float *A_out, float *joints_out,
                                    int32_t *dyn_row_out, int B, void *stream) {
  if (B <= 0) return SHAPY_OK;
  if (m->J > 64 || n_pose > m->J || m->P != (m->J - 1) * 9) return SHAPY_EINVAL;
  PoseK k;
  k.parents = m->parents; k.neck = m->neck_kin_chain; k.Jt = m->J_template; k.Js = m->J_shapedirs;
  k.pose = pose; k.coeffs = coeffs; k.rot = rot_out; k.pf = pose_feat_out; k.A = A_out;
  k.joints = joints_out; k.dyn_row = dyn_row_out; k.J = m->J; k.NB = m->NB; k.NBpad = m->NBpad;
  k.P = m->P; k.Ppad = m->Ppad; k.n_pose = n_pose; k.pose_type = pose_type; k.n_neck = m->n_neck;
  k.n_dyn_rows = m->n_dyn_rows;
  hipLaunchKernelGGL(smplx_pose_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, k);
  return (int)hipGetLastError();
}

extern "C" int shapy_smplx_skin_f32(const ShapySmplxModel *m, const float *A, const float *v_posed,
                                    float *vertices_out, int B, void *stream) {
  if (B <= 0) return SHAPY_OK;
  if (m->J > 64) return SHAPY_EINVAL;
  // register block 2 vertices x 2 bodies once there are enough bodies to keep every CU busy with it: 6 LDS
  // reads per 24 packed FMAs (the two pipes level) and 2.6 waves per SIMD at B = 64 -- the 4 x 2 block has the
  // same pipe balance but only 1.4 waves per SIMD, i.e. two rounds of a 10 us wave: 24.1 us
  // (profiles/r05f_kernel_stats_smplx_b64.csv)
  if (B >= 16)
    hipLaunchKernelGGL((smplx_skin_kernel<2, 2>), dim3((m->V + 511) / 512, (B + 1) / 2), dim3(256), 0,
                       (hipStream_t)stream, m->lbs_weights_t, A, v_posed, vertices_out, m->V, m->J, B);
  else
    hipLaunchKernelGGL((smplx_skin_kernel<1, 1>), dim3((m->V + 255) / 256, B), dim3(256), 0,
                       (hipStream_t)stream, m->lbs_weights_t, A, v_posed, vertices_out, m->V, m->J, B);
  return (int)hipGetLastError();
}

extern "C" int shapy_smplx_joints_f32(const ShapySmplxModel *m, const float *posed_joints,
                                      const float *vertices, const int32_t *dyn_row,
                                      const float *camera, float *joints_out, float *proj_out,
                                      float *cam_scale_out, int B, int use_face_contour,
                                      void *stream) {
  if (B <= 0) return SHAPY_OK;
  JointsK k;
  k.faces = m->faces; k.lmk_idx = m->lmk_faces_idx; k.dyn_idx = m->dyn_lmk_faces_idx;
  k.dyn_row = dyn_row; k.lmk_bc = m->lmk_bary; k.dyn_bc = m->dyn_lmk_bary; k.posed = posed_joints;
  k.verts = vertices; k.cam = camera; k.joints = joints_out; k.proj = proj_out;
  k.scale = cam_scale_out; k.J = m->J; k.V = m->V; k.n_static = m->n_static_lmk;
  k.n_dyn = use_face_contour ? m->n_dyn_lmk : 0;
  k.n_out = m->J + k.n_static + k.n_dyn;
  if (k.n_dyn > 0 && !dyn_row) return SHAPY_EINVAL;
  hipLaunchKernelGGL(smplx_joints_kernel, dim3(B), dim3(128), 0, (hipStream_t)stream, k);
  return (int)hipGetLastError();
}

// The whole layer in ONE call: the five to six launches above, enqueued back to back from C (the
// Python host spent ~40 us per call between its ctypes calls -- the layer's kernels take 85 us at
// B = 64 and far less at B = 4, so the gaps were a third to two thirds of its wall time).
static int smplx_gemm(const float *in, int B, int K, const float *wgt, int N, float *out,
                      const float *bias, const float *res, hipStream_t s, int tile = 0) {
  ShapyConv d = {};
  d.in = in; d.wgt = wgt; d.bias = bias; d.res = res; d.out = out;
  d.B = B; d.Hi = d.Wi = d.Ho = d.Wo = 1; d.Cin = K; d.in_ld = K; d.Cout = N;
  d.ksize = 1; d.stride = 1; d.pad = 0; d.out_ld = N; d.out_coff = 0; d.res_ld = N; d.res_coff = 0;
  d.relu = 0; d.ups = 1; d.tile = tile; d.dtype = SHAPY_DTYPE_F32; d.wgt_wino = nullptr;
  return conv2d(d, s);
}

extern "C" int shapy_smplx_forward_f32(const ShapySmplxModel *m, const float *pose, int pose_type,
                                       int n_pose, const float *coeffs, const float *coeffs_shape,
                                       const float *camera, float *v_shaped_full, float *v_shaped,
                                       float *rot, float *pose_feat, float *A, float *posed_joints,
                                       int32_t *dyn_row, float *v_posed, float *vertices,
                                       float *joints_out, float *proj_out, float *cam_scale_out,
                                       int B, int use_face_contour, int shape_only, void *stream) {
  if (B <= 0) return SHAPY_OK;
  hipStream_t s = (hipStream_t)stream;
  const int N = m->V * 3;
  if (coeffs_shape && !v_shaped) return SHAPY_EINVAL;
  int rc;
  if (coeffs_shape && coeffs_shape == coeffs + (size_t)B * m->NBpad && v_shaped == v_shaped_full + (size_t)B * N) {
    // both coefficient sets and both outputs are adjacent (SMPLX.forward lays them out that way): ONE
    // GEMM with M = 2 B -- the shape basis is streamed once and one launch goes away
    rc = smplx_gemm(coeffs, 2 * B, m->NBpad, m->shapedirs_t, N, v_shaped_full, m->v_template, nullptr, s);
    if (rc) return rc;
  } else {
    rc = smplx_gemm(coeffs, B, m->NBpad, m->shapedirs_t, N, v_shaped_full, m->v_template, nullptr, s);
    if (rc) return rc;
    if (coeffs_shape) {
      rc = smplx_gemm(coeffs_shape, B, m->NBpad, m->shapedirs_t, N, v_shaped, m->v_template, nullptr, s);
      if (rc) return rc;
    }
  }
  if (shape_only) return SHAPY_OK;
  rc = shapy_smplx_pose_f32(m, pose, pose_type, n_pose, coeffs, rot, pose_feat, A, posed_joints,
                            dyn_row, B, stream);
  if (rc) return rc;
  // M = batch is skinny and posedirs (61 MB) is streamed exactly once: the 32 x 64 tile with 128-byte K
  // chunks (Ppad % 32 == 0) is the fastest of the sweep at every batch size -- 26.0 us at B = 64, 16.6 at
  // B = 4 against 34.4 / 30.8 for round 4's 64 x 48 tile with three chunks of loads in flight
  // (tools/skinny_gemm_bench.py, profiles/r05e_skinny_gemm_b{4,64}.txt)
  const int pose_tile = (m->Ppad % 32 == 0) ? SHAPY_TILE_32x64 : 0x40000;
  rc = smplx_gemm(pose_feat, B, m->Ppad, m->posedirs_t, N, v_posed, nullptr, v_shaped_full, s, pose_tile);
  if (rc) return rc;
  rc = shapy_smplx_skin_f32(m, A, v_posed, vertices, B, stream);
  if (rc) return rc;
  return shapy_smplx_joints_f32(m, posed_joints, vertices, dyn_row, camera, joints_out, proj_out,
                                cam_scale_out, B, use_face_contour, stream);
}

extern "C" int shapy_pose_decode_f32(const float *pose, int pose_type, float *rot_out, int64_t n,
                                     void *stream) {
  if (n <= 0) return SHAPY_OK;
  if (pose_type != SHAPY_POSE_CONT6D && pose_type != SHAPY_POSE_AXIS_ANGLE) return SHAPY_EINVAL;
  hipLaunchKernelGGL(pose_decode_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, pose, pose_type, rot_out, (long)n);
  return (int)hipGetLastError();
}

extern "C" int shapy_head_prepare_f32(const float *params, int S, int B, int P, int pose_off,
                                      int n_joints, int pose_type, int betas_off, int n_betas,
                                      int NBpad, int cam_off, float *rot_out, float *coeffs_out,
                                      float *cam_out, void *stream) {
  if (S <= 0 || B <= 0) return SHAPY_OK;
  if (pose_type != SHAPY_POSE_CONT6D && pose_type != SHAPY_POSE_AXIS_ANGLE) return SHAPY_EINVAL;
  const int per = pose_type == SHAPY_POSE_CONT6D ? 6 : 3;
  if (pose_off < 0 || pose_off + n_joints * per > P || betas_off < 0 || betas_off + n_betas > P ||
      n_betas > NBpad || (cam_out && (cam_off < 0 || cam_off + 3 > P)))
    return SHAPY_EINVAL;
  PrepK k;
  k.params = params; k.rot = rot_out; k.coeffs = coeffs_out; k.cam = cam_out;
  k.S = S; k.B = B; k.P = P; k.pose_off = pose_off; k.nj = n_joints; k.pose_type = pose_type;
  k.betas_off = betas_off; k.n_betas = n_betas; k.NBpad = NBpad; k.cam_off = cam_off;
  const long total = (long)S * B * n_joints + (long)B * NBpad + (cam_out ? (long)B * 3 : 0);
  hipLaunchKernelGGL(head_prepare_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, k);
  return (int)hipGetLastError();
}

extern "C" int shapy_smplx_prepare_f32(const float *const *parts_host, const int32_t *n_joints_host,
                                       const int64_t *part_bstride_host, int n_parts, const float *betas,
                                       int64_t betas_bstride, int nb, const float *expression,
                                       int64_t expr_bstride, int ne, int NBpad, float *pose_out,
                                       float *coeffs_out, float *coeffs_shape_out, int B, void *stream) {
  if (B <= 0) return SHAPY_OK;
  if (n_parts < 0 || n_parts > 7 || nb < 0 || ne < 0 || nb + ne > NBpad || !coeffs_out ||
      (n_parts > 0 && (!parts_host || !n_joints_host || !pose_out)))
    return SHAPY_EINVAL;
  SmplxPrepK k = {};
  int first = 0;
  for (int p = 0; p < n_parts; ++p) {
    if (n_joints_host[p] < 0) return SHAPY_EINVAL;
    k.part[p] = parts_host[p]; k.n[p] = n_joints_host[p]; k.first[p] = first;
    k.bs[p] = part_bstride_host ? part_bstride_host[p] : (long)n_joints_host[p] * 9;
    if (k.bs[p] < 0) return SHAPY_EINVAL;
    first += n_joints_host[p];
  }
  k.n_parts = n_parts; k.n_pose = first; k.B = B; k.nb = nb; k.ne = ne; k.NBpad = NBpad;
  k.betas = betas; k.expr = expression; k.pose = pose_out; k.coeffs = coeffs_out;
  k.betas_bs = betas_bstride; k.expr_bs = expr_bstride;
  if (betas_bstride < 0 || expr_bstride < 0) return SHAPY_EINVAL;
  k.coeffs_shape = coeffs_shape_out;
  const long total = (long)B * first * 9 + (long)B * NBpad;
  hipLaunchKernelGGL(smplx_prepare_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, k);
  return (int)hipGetLastError();
}

extern "C" int shapy_weak_persp_project_f32(const float *points, const float *scale,
                                            const float *translation, float *out, int B, int N,
                                            int scale_first, void *stream) {
  const long total = (long)B * N;
  if (total <= 0) return SHAPY_OK;
  hipLaunchKernelGGL(weak_persp_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, points, scale, translation, out, N, scale_first, total);
  return (int)hipGetLastError();
}

extern "C" int shapy_joint_regress_f32(const float *regressor, const float *vertices, float *out,
                                       int B, int V, int Jn, void *stream) {
  if (B <= 0 || Jn <= 0) return SHAPY_OK;
  hipLaunchKernelGGL(joint_regress_kernel, dim3(Jn, B), dim3(256), 0, (hipStream_t)stream, regressor,
                     vertices, out, V, Jn);
  return (int)hipGetLastError();
}

extern "C" int shapy_b2a_polynomial_f32(const float *betas, const float *weight, const float *bias,
                                        float *out, int B, int NB, int NA, void *stream) {
  const long total = (long)B * NA;
  if (total <= 0) return SHAPY_OK;
  hipLaunchKernelGGL(b2a_polynomial_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, betas, weight, bias, out, NB, NA, total);
  return (int)hipGetLastError();
}
