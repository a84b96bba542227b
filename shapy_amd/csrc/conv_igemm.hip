// Implicit-GEMM convolution on the CDNA4 f32 matrix cores (v_mfma_f32_16x16x4_f32).
//
// Replaces every cuDNN conv + eval BatchNorm + ReLU + residual add + nearest upsample of the
// reference's HighResolutionNet.forward (regressor/human_shape/models/backbone/hrnet.py:426-498)
// and doubles as the GEMM of the SMPL-X blend shapes (lbs.py:171-182, 218-239).
//
// GEMM view: M = B*Ho*Wo output pixels, N = Cout, K = ks*ks*Cin, NHWC activations,
// OHWI weights (K contiguous for both operands).  One 256-thread workgroup (4 waves) owns a
// BM x BN tile; K is consumed in 16-float chunks staged through double-buffered LDS:
// global -> VGPR (float4, issued before the MFMAs of the current chunk) -> LDS -> b128
// fragment reads.  Lane l of a wave reads 4 consecutive k of row (l&15) starting at
// 4*(l>>4) with ONE ds_read_b128 and feeds them to 4 successive MFMAs; A and B use the same
// k permutation so the sum over k is unchanged.  f32 MFMA is exact f32 (an fmaf chain), so
// parity with the CPU reference is at rounding level.
#include "common.h"

namespace shapy {

constexpr int LDS_LD = 20;   // floats per staged row: 16 + 4 pad (keeps b128 alignment)

struct ConvK {
  const float *in, *wgt, *bias, *res;
  float *out;
  int M, Hi, Wi, Cin, in_ld, Ho, Wo, Cout, ks, stride, pad;
  int out_ld, out_coff, res_ld, res_coff, relu, ups;
};

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_f32_kernel(ConvK p) {
  static_assert(WM * WN == 4, "4 waves per workgroup");
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int AR = BM / 64;            // A rows staged per thread
  constexpr int BR = (BN + 63) / 64;     // B rows staged per thread (guarded)
  __shared__ __attribute__((aligned(16))) float lds[2][(BM + BN) * LDS_LD];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int m_blk = blockIdx.y * BM, n_blk = blockIdx.x * BN;
  const int kq = t & 3, lrow = t >> 2;

  // ---- per-thread staging addresses ----
  int a_off[AR], a_h[AR], a_w[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int m = m_blk + lrow + 64 * i;
    const int mm = m < p.M ? m : 0;
    const int wo = mm % p.Wo;
    const int tq = mm / p.Wo;
    const int ho = tq % p.Ho;
    const int b = tq / p.Ho;
    const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
    a_off[i] = ((b * p.Hi + hi0) * p.Wi + wi0) * p.in_ld + kq * 4;
    a_h[i] = m < p.M ? hi0 : -0x40000000;
    a_w[i] = wi0;
  }
  const int Kw = p.ks * p.ks * p.Cin;
  int b_off[BR];
  bool b_ok[BR];
#pragma unroll
  for (int i = 0; i < BR; ++i) {
    const int r = lrow + 64 * i;
    const int n = n_blk + r;
    b_ok[i] = (r < BN) && (n < p.Cout);
    b_off[i] = (b_ok[i] ? n : 0) * Kw + kq * 4;
  }

  float4 a_reg[AR], b_reg[BR];
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

  // chunk iterator state for the NEXT chunk to be fetched
  int kh = 0, kw = 0, c0 = 0;
  const int n_chunks = p.ks * p.ks * (p.Cin >> 4);

  auto gload = [&]() {
    const int tap_in = (kh * p.Wi + kw) * p.in_ld + c0;
    const int tap_w = (kh * p.ks + kw) * p.Cin + c0;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const bool ok = (unsigned)(a_h[i] + kh) < (unsigned)p.Hi &&
                      (unsigned)(a_w[i] + kw) < (unsigned)p.Wi;
      a_reg[i] = ok ? *reinterpret_cast<const float4 *>(p.in + (a_off[i] + tap_in)) : zero4;
    }
#pragma unroll
    for (int i = 0; i < BR; ++i)
      b_reg[i] = b_ok[i] ? *reinterpret_cast<const float4 *>(p.wgt + (b_off[i] + tap_w)) : zero4;
    c0 += 16;
    if (c0 == p.Cin) {
      c0 = 0;
      if (++kw == p.ks) { kw = 0; ++kh; }
    }
  };
  auto lstore = [&](int buf) {
    float *A = lds[buf];
    float *Bt = lds[buf] + BM * LDS_LD;
#pragma unroll
    for (int i = 0; i < AR; ++i)
      *reinterpret_cast<float4 *>(A + (lrow + 64 * i) * LDS_LD + kq * 4) = a_reg[i];
#pragma unroll
    for (int i = 0; i < BR; ++i)
      if (lrow + 64 * i < BN)
        *reinterpret_cast<float4 *>(Bt + (lrow + 64 * i) * LDS_LD + kq * 4) = b_reg[i];
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frag_off = (lane & 15) * LDS_LD + (lane >> 4) * 4;
  const int a_base = wm * (BM / WM) * LDS_LD + frag_off;
  const int b_base = BM * LDS_LD + wn * (BN / WN) * LDS_LD + frag_off;

  gload();
  lstore(0);
  __syncthreads();

  for (int kc = 0; kc < n_chunks; ++kc) {
    const int cur = kc & 1;
    const bool more = kc + 1 < n_chunks;
    if (more) gload();
    const float *L = lds[cur];
    float4 af[TM], bf[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
      af[i] = *reinterpret_cast<const float4 *>(L + a_base + i * 16 * LDS_LD);
#pragma unroll
    for (int j = 0; j < TN; ++j)
      bf[j] = *reinterpret_cast<const float4 *>(L + b_base + j * 16 * LDS_LD);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
      }
    if (more) lstore(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: bias (+ residual) (+ ReLU), plain or upsample-scatter store ----
  const int col_l = lane & 15, row_l = (lane >> 4) * 4;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n_blk + wn * (BN / WN) + j * 16 + col_l;
    if (col >= p.Cout) continue;
    const float bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m_blk + wm * (BM / WM) + i * 16 + row_l + r;
        if (row >= p.M) continue;
        const float v = acc[i][j][r] + bias;
        if (p.ups == 1) {
          float o = v;
          if (p.res) o += p.res[(long)row * p.res_ld + p.res_coff + col];
          if (p.relu) o = fmaxf(o, 0.f);
          p.out[(long)row * p.out_ld + p.out_coff + col] = o;
        } else {
          const int wo = row % p.Wo;
          const int tq = row / p.Wo;
          const int ho = tq % p.Ho;
          const int b = tq / p.Ho;
          const int WoU = p.Wo * p.ups;
          const long pix0 = ((long)(b * p.Ho + ho) * p.ups) * WoU + (long)wo * p.ups;
          for (int dy = 0; dy < p.ups; ++dy)
            for (int dx = 0; dx < p.ups; ++dx) {
              const long pix = pix0 + (long)dy * WoU + dx;
              float o = v;
              if (p.res) o += p.res[pix * p.res_ld + p.res_coff + col];
              if (p.relu) o = fmaxf(o, 0.f);
              p.out[pix * p.out_ld + p.out_coff + col] = o;
            }
        }
      }
    }
  }
}

template <int BM, int BN, int WM, int WN>
static int launch(const ConvK &k, hipStream_t s) {
  dim3 grid((k.Cout + BN - 1) / BN, (k.M + BM - 1) / BM);
  hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WM, WN>), grid, dim3(256), 0, s, k);
  return (int)hipGetLastError();
}

int conv_tile_auto(int M, int Cout) {
  // wave tile is 64x48 / 64x64 for large M, 32x48 / 32x64 when M is small
  const bool n96 = (Cout % 96 == 0) || (Cout % 48 == 0 && Cout > 48) || (Cout % 64 != 0 && Cout > 128);
  if (Cout <= 48) return M >= 8192 ? SHAPY_TILE_256x48 : SHAPY_TILE_64x48;
  if (n96) return M >= 16384 ? SHAPY_TILE_128x96 : SHAPY_TILE_64x96;
  if (Cout <= 64) return M >= 8192 ? SHAPY_TILE_256x64 : SHAPY_TILE_64x64;
  return M >= 16384 ? SHAPY_TILE_128x128 : SHAPY_TILE_64x128;
}

int conv2d_f32(const ShapyConv &d, hipStream_t s) {
  if (d.Cin <= 0 || (d.Cin & 15) || (d.in_ld & 3) || d.ksize < 1 || d.stride < 1 || d.ups < 1)
    return SHAPY_EINVAL;
  if (((uintptr_t)d.in | (uintptr_t)d.wgt) & 15) return SHAPY_EINVAL;
  ConvK k;
  k.in = d.in; k.wgt = d.wgt; k.bias = d.bias; k.res = d.res; k.out = d.out;
  k.M = d.B * d.Ho * d.Wo;
  k.Hi = d.Hi; k.Wi = d.Wi; k.Cin = d.Cin; k.in_ld = d.in_ld; k.Ho = d.Ho; k.Wo = d.Wo;
  k.Cout = d.Cout; k.ks = d.ksize; k.stride = d.stride; k.pad = d.pad;
  k.out_ld = d.out_ld; k.out_coff = d.out_coff; k.res_ld = d.res_ld; k.res_coff = d.res_coff;
  k.relu = d.relu; k.ups = d.ups;
  if (k.M <= 0 || k.Cout <= 0) return SHAPY_OK;
  const int tile = d.tile ? d.tile : conv_tile_auto(k.M, k.Cout);
  switch (tile) {
    case SHAPY_TILE_256x48: return launch<256, 48, 4, 1>(k, s);
    case SHAPY_TILE_128x96: return launch<128, 96, 2, 2>(k, s);
    case SHAPY_TILE_128x128: return launch<128, 128, 2, 2>(k, s);
    case SHAPY_TILE_256x64: return launch<256, 64, 4, 1>(k, s);
    case SHAPY_TILE_64x48: return launch<64, 48, 4, 1>(k, s);
    case SHAPY_TILE_64x96: return launch<64, 96, 2, 2>(k, s);
    case SHAPY_TILE_64x128: return launch<64, 128, 2, 2>(k, s);
    case SHAPY_TILE_64x64: return launch<64, 64, 2, 2>(k, s);
    default: return SHAPY_EINVAL;
  }
}

}  // namespace shapy
