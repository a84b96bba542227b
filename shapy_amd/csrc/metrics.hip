// Evaluator metrics on the GPU: aligned point error (v2v, mpjpe) and the point-to-point
// error between meshes of different topology (P2P-20k).
// Reference: regressor/human_shape/utils/metrics.py:31-56 (point_error), :100-277 (alignments),
// :335-365 (PointError), :367-460 (v2vhdError); callers regressor/human_shape/evaluation.py:192-262.
//
// One workgroup per body; all sums are carried in float64 (the reference reduces float32 arrays
// with numpy; we are at least as accurate), every phase re-reads the points from L2.
#include "common.h"

namespace shapy {

constexpr int MT = 1024;            // threads per body
constexpr int MW = MT / 64;

template <int N>
__device__ __forceinline__ void block_sum(double (&v)[N], double *red /* [MW*N + N] */) {
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[i] += __shfl_xor(v[i], o, 64);
  const int w = threadIdx.x >> 6;
  __syncthreads();                  // red may still be read from the previous call
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int i = 0; i < N; ++i) red[w * N + i] = v[i];
  __syncthreads();
  if (threadIdx.x < N) {
    double s = 0.0;
    for (int k = 0; k < MW; ++k) s += red[k * N + threadIdx.x];
    red[MW * N + threadIdx.x] = s;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = red[MW * N + i];
}

// Rotation of the orthogonal Procrustes problem from K = X1 X2^T (metrics.py:137-152):
// R = V Z U^T with K = U S V^T and Z fixing det(R) = +1.  V comes from a Jacobi
// eigen-decomposition of K^T K; u1,u2 = K v / sigma for the two largest singular values and
// u3 = u1 x u2, which turns the reference's Z into diag(1, 1, det V) and never divides by the
// smallest singular value.
__device__ void procrustes_rotation(const double K[9], double R[9]) {
  double A[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s += K[k * 3 + i] * K[k * 3 + j];
      A[i][j] = s;
    }
  for (int sweep = 0; sweep < 24; ++sweep) {
    const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
    if (off < 1e-300 || off < 1e-17 * (fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]))) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (A[p][q] == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) {
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq;
          A[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk;
          A[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
  }
  int o[3] = {0, 1, 2};             // columns by descending eigenvalue
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2 - i; ++j)
      if (A[o[j]][o[j]] < A[o[j + 1]][o[j + 1]]) { const int t = o[j]; o[j] = o[j + 1]; o[j + 1] = t; }
  double v[3][3], u[3][3];          // v[i] = i-th right singular vector
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 3; ++k) v[i][k] = V[k][o[i]];
  for (int i = 0; i < 2; ++i) {
    double n = 0.0;
    for (int r = 0; r < 3; ++r) {
      u[i][r] = K[r * 3 + 0] * v[i][0] + K[r * 3 + 1] * v[i][1] + K[r * 3 + 2] * v[i][2];
      n += u[i][r] * u[i][r];
    }
    if (i == 1) {                   // re-orthogonalise against u1 (guards a tiny sigma_2)
      const double d = u[1][0] * u[0][0] + u[1][1] * u[0][1] + u[1][2] * u[0][2];
      n = 0.0;
      for (int r = 0; r < 3; ++r) { u[1][r] -= d * u[0][r]; n += u[1][r] * u[1][r]; }
    }
    n = n > 0.0 ? 1.0 / sqrt(n) : 0.0;
    for (int r = 0; r < 3; ++r) u[i][r] *= n;
  }
  u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1];
  u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2];
  u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
  const double detV = v[0][0] * (v[1][1] * v[2][2] - v[1][2] * v[2][1]) -
                      v[0][1] * (v[1][0] * v[2][2] - v[1][2] * v[2][0]) +
                      v[0][2] * (v[1][0] * v[2][1] - v[1][1] * v[2][0]);
  const double z[3] = {1.0, 1.0, detV >= 0 ? 1.0 : -1.0};
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c)
      R[r * 3 + c] = z[0] * v[0][r] * u[0][c] + z[1] * v[1][r] * u[1][c] + z[2] * v[2][r] * u[2][c];
}

// mode: 0 none, 1 translation, 2 scale, 3 procrustes (metrics.py:59-277).
__global__ __launch_bounds__(MT) void aligned_point_error_kernel(
    const float *__restrict__ est, const float *__restrict__ gt, int P, int mode,
    float *__restrict__ err, float *__restrict__ err_mean, float *__restrict__ aligned) {
  __shared__ double red[MW * 12 + 12];
  __shared__ double xf[13];         // s*R (9), t (3)
  const int b = blockIdx.x, tid = threadIdx.x;
  const float *e = est + (size_t)b * P * 3;
  const float *g = gt + (size_t)b * P * 3;
  double mu[6] = {0, 0, 0, 0, 0, 0};
  if (mode != 0) {
    for (int p = tid; p < P; p += MT)
      for (int c = 0; c < 3; ++c) { mu[c] += e[p * 3 + c]; mu[3 + c] += g[p * 3 + c]; }
    block_sum<6>(mu, red);
    for (int c = 0; c < 6; ++c) mu[c] /= (double)P;
  }
  double sR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0};
  if (mode == 1) {
    for (int c = 0; c < 3; ++c) t[c] = mu[3 + c] - mu[c];
  } else if (mode >= 2) {
    double m[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};    // K (9), var1, var2
    for (int p = tid; p < P; p += MT) {
      double x1[3], x2[3];
      for (int c = 0; c < 3; ++c) { x1[c] = e[p * 3 + c] - mu[c]; x2[c] = g[p * 3 + c] - mu[3 + c]; }
      for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) m[i * 3 + j] += x1[i] * x2[j];
        m[9] += x1[i] * x1[i];
        m[10] += x2[i] * x2[i];
      }
    }
    block_sum<11>(m, red);
    if (tid == 0) {
      double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, scale;
      if (mode == 2) {
        scale = sqrt(m[10] / m[9]);
      } else {
        procrustes_rotation(m, R);
        double tr = 0.0;            // trace(R K)
        for (int i = 0; i < 3; ++i)
          for (int k = 0; k < 3; ++k) tr += R[i * 3 + k] * m[k * 3 + i];
        scale = tr / m[9];
      }
      for (int i = 0; i < 9; ++i) xf[i] = scale * R[i];
      for (int i = 0; i < 3; ++i)
        xf[9 + i] = mu[3 + i] - (xf[i * 3] * mu[0] + xf[i * 3 + 1] * mu[1] + xf[i * 3 + 2] * mu[2]);
    }
    __syncthreads();
    for (int i = 0; i < 9; ++i) sR[i] = xf[i];
    for (int i = 0; i < 3; ++i) t[i] = xf[9 + i];
  }
  double acc[1] = {0.0};
  for (int p = tid; p < P; p += MT) {
    const double x[3] = {e[p * 3], e[p * 3 + 1], e[p * 3 + 2]};
    double d2 = 0.0;
    for (int i = 0; i < 3; ++i) {
      const double h = sR[i * 3] * x[0] + sR[i * 3 + 1] * x[1] + sR[i * 3 + 2] * x[2] + t[i];
      if (aligned) aligned[((size_t)b * P + p) * 3 + i] = (float)h;
      const double d = h - (double)g[p * 3 + i];
      d2 += d * d;
    }
    const double r = sqrt(d2);
    if (err) err[(size_t)b * P + p] = (float)r;
    acc[0] += r;
  }
  if (err_mean) {
    block_sum<1>(acc, red);
    if (tid == 0) err_mean[b] = (float)(acc[0] / (double)P);
  }
}

// P2P: hd = S @ verts for a CSR point regressor S [P,V] (metrics.py:419-460), float64 like the
// reference (evaluation.py:253-255 casts both meshes to double).
__device__ __forceinline__ void csr_point(const int *__restrict__ rp, const int *__restrict__ ci,
                                          const double *__restrict__ va,
                                          const double *__restrict__ verts, int p, double out[3]) {
  out[0] = out[1] = out[2] = 0.0;
  for (int k = rp[p]; k < rp[p + 1]; ++k) {
    const double w = va[k];
    const double *v = verts + (size_t)ci[k] * 3;
    out[0] += w * v[0];
    out[1] += w * v[1];
    out[2] += w * v[2];
  }
}

__global__ __launch_bounds__(MT) void p2p_error_kernel(
    const int *__restrict__ rp_in, const int *__restrict__ ci_in, const double *__restrict__ va_in,
    const int *__restrict__ rp_tg, const int *__restrict__ ci_tg, const double *__restrict__ va_tg,
    const double *__restrict__ pts_in, const double *__restrict__ pts_tg, int P, int V1, int V2,
    int align, double *__restrict__ err, double *__restrict__ err_mean) {
  __shared__ double red[MW * 3 + 3];
  const int b = blockIdx.x, tid = threadIdx.x;
  const double *vin = pts_in + (size_t)b * V1 * 3;
  const double *vtg = pts_tg + (size_t)b * V2 * 3;
  double t[3] = {0, 0, 0};
  if (align) {
    for (int p = tid; p < P; p += MT) {
      double a[3], c[3];
      csr_point(rp_in, ci_in, va_in, vin, p, a);
      csr_point(rp_tg, ci_tg, va_tg, vtg, p, c);
      for (int i = 0; i < 3; ++i) t[i] += c[i] - a[i];
    }
    block_sum<3>(t, red);
    for (int i = 0; i < 3; ++i) t[i] /= (double)P;
  }
  double acc[1] = {0.0};
  for (int p = tid; p < P; p += MT) {
    double a[3], c[3], d2 = 0.0;
    csr_point(rp_in, ci_in, va_in, vin, p, a);
    csr_point(rp_tg, ci_tg, va_tg, vtg, p, c);
    for (int i = 0; i < 3; ++i) {
      const double d = a[i] + t[i] - c[i];
      d2 += d * d;
    }
    const double r = sqrt(d2);
    err[(size_t)b * P + p] = r;
    acc[0] += r;
  }
  block_sum<1>(acc, red);
  if (tid == 0 && err_mean) err_mean[b] = acc[0] / (double)P;
}

}  // namespace shapy

extern "C" int shapy_aligned_point_error_f32(const float *est, const float *gt, int B, int P,
                                             int alignment, float *err_out, float *err_mean_out,
                                             float *aligned_out, void *stream) {
  if (B < 0 || P <= 0 || alignment < 0 || alignment > 3) return SHAPY_EINVAL;
  if (B == 0) return SHAPY_OK;
  if (!est || !gt) return SHAPY_EINVAL;
  hipLaunchKernelGGL(shapy::aligned_point_error_kernel, dim3(B), dim3(shapy::MT), 0,
                     (hipStream_t)stream, est, gt, P, alignment, err_out, err_mean_out, aligned_out);
  SHAPY_HIP_TRY(hipGetLastError());
  return SHAPY_OK;
}

extern "C" int shapy_p2p_error_f64(const int32_t *in_rowptr, const int32_t *in_col,
                                   const double *in_val, const int32_t *tgt_rowptr,
                                   const int32_t *tgt_col, const double *tgt_val,
                                   const double *input_verts, const double *target_verts, int B,
                                   int P, int V_in, int V_tgt, int align, double *err_out,
                                   double *err_mean_out, void *stream) {
  if (B < 0 || P <= 0 || V_in <= 0 || V_tgt <= 0) return SHAPY_EINVAL;
  if (B == 0) return SHAPY_OK;
  if (!in_rowptr || !in_col || !in_val || !tgt_rowptr || !tgt_col || !tgt_val || !input_verts ||
      !target_verts || !err_out)
    return SHAPY_EINVAL;
  hipLaunchKernelGGL(shapy::p2p_error_kernel, dim3(B), dim3(shapy::MT), 0, (hipStream_t)stream,
                     in_rowptr, in_col, in_val, tgt_rowptr, tgt_col, tgt_val, input_verts,
                     target_verts, P, V_in, V_tgt, align, err_out, err_mean_out);
  SHAPY_HIP_TRY(hipGetLastError());
  return SHAPY_OK;
}
