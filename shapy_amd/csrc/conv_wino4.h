// Pieces shared by the two Winograd F(4x4,3x3) kernels (conv_wino4.hip: one launch per layer;
// conv_wino4g.hip: persistent workgroups over a group of layers).
#pragma once
#include "conv_common.h"

namespace shapy {

#ifndef WINO4_RING
#define WINO4_RING 12         // B-fragment positions in flight per multiplying wave (divides 36)
#endif

// one application of B^T (6 x 6) to a 6-vector (of 4 channels)
__device__ __forceinline__ void wino4_bt(const f32x4 (&d)[6], f32x4 (&o)[6]) {
  const f32x4 a = d[4] - 4.f * d[2];
  const f32x4 b = d[3] - 4.f * d[1];
  const f32x4 c = d[4] - d[2];
  const f32x4 e = d[3] - d[1];
  o[0] = 4.f * d[0] - 5.f * d[2] + d[4];
  o[1] = a + b;
  o[2] = a - b;
  o[3] = c + 2.f * e;
  o[4] = c - 2.f * e;
  o[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}

// one application of A^T (4 x 6) to a 6-vector
__device__ __forceinline__ void wino4_at(const float (&m)[6], float (&o)[4]) {
  const float s12 = m[1] + m[2], d12 = m[1] - m[2];
  const float s34 = m[3] + m[4], d34 = m[3] - m[4];
  o[0] = (m[0] + s12) + s34;
  o[1] = fmaf(2.f, d34, d12);
  o[2] = fmaf(4.f, s34, s12);
  o[3] = fmaf(8.f, d34, d12) + m[5];
}

// Workgroup barrier WITHOUT the fence of __syncthreads(): only LDS traffic is ordered across it
// (the staging wave's ds_writes, the multiplying waves' ds_reads).  The fence would also drain
// vmcnt, i.e. make every wave wait for its prefetched global loads (filter ring, next patch) at
// every chunk boundary.
__device__ __forceinline__ void wino4_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}


// Epilogue of a multiplying wave (both F(4x4) kernels): output transform A^T M A in the registers
// of the lane that holds the MFMA results (lane = channel `col`, tiles m_blk + 4 g4 + r, 16 pixels
// each), bias + residual + ReLU, 4-byte stores (16 lanes = 64 contiguous bytes).  Residual loads
// and stores are buffer instructions: per-lane part of the address = the tile's first pixel, pixel
// offset inside the tile = scalar offset; pixels outside the image (partial edge tiles) and dead
// tiles get an out-of-range offset, which drops the access.
// One tile row's residual is in flight ahead of the one being transformed.  (Three rows ahead --
// 48 registers, the dead filter ring -- was measured SLOWER: 55 / 45 / 48 us against 51 / 40 / 41 us
// on the 48 / 96 / 192-channel classes, run H of round 3; the residual loads are not what the
// epilogue waits for, a layer without residual takes the same time.)
struct Wino4Epi {
  int dbg;                     // tuning builds (-DSHAPY_W4G_TIMING): 1 no stores, 2 no residual loads
  void *out;
  const void *res;             // nullptr: none
  const void *in;              // any valid address for the residual resource when res is null
  const float *bias;
  int H, W, tiles, out_ld, out_coff, res_ld, res_coff, relu;
};

__device__ __forceinline__ void wino4_epilogue(const Wino4Epi &e, const f32x4 (&acc)[36], int m_blk,
                                               int col, int g4) {
  constexpr int BAD = 0x40000000;
  const int H = e.H, W = e.W;
  const int TW = (W + 3) >> 2, TH = (H + 3) >> 2;
  const float bias = e.bias ? e.bias[col] : 0.f;
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(e.out, 0, BAD, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void *>(e.res ? e.res : e.in), 0, BAD, 0x00020000);
  const bool has_res = e.res != nullptr;
  const int out_ld = e.out_ld, res_ld = e.res_ld;
  int obase[4], rbase[4], nrow[4], ncol[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int tile = m_blk + 4 * g4 + r;
    const bool live = tile < e.tiles;
    const int tt = live ? tile : 0;
    const int tx = tt % TW;
    const int tq = tt / TW;
    const int ty = tq % TH;
    const int b = tq / TH;
    const int pix0 = (b * H + 4 * ty) * W + 4 * tx;
    obase[r] = live ? (pix0 * out_ld + e.out_coff + col) * 4 : BAD;
    rbase[r] = (live & has_res) ? (pix0 * res_ld + e.res_coff + col) * 4 : BAD;
    nrow[r] = H - 4 * ty;           // >= 4 for a full tile
    ncol[r] = W - 4 * tx;
  }
  float resv[2][16];
  auto rload = [&](int r, float (&rv)[16]) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) {
        bool ok = (a < nrow[r]) & (bb < ncol[r]);
#ifdef SHAPY_W4G_TIMING
        ok &= !(e.dbg & 2);
#endif
        rv[4 * a + bb] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
            rs_res, ok ? rbase[r] : BAD, (a * W + bb) * res_ld * 4, 0));
      }
  };
  rload(0, resv[0]);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (r + 1 < 4) rload(r + 1, resv[(r + 1) & 1]);
    float s[6][4];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const float m[6] = {acc[6 * i + 0][r], acc[6 * i + 1][r], acc[6 * i + 2][r],
                          acc[6 * i + 3][r], acc[6 * i + 4][r], acc[6 * i + 5][r]};
      wino4_at(m, s[i]);                                          // M A   (along x)
    }
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) {
      const float colv[6] = {s[0][bb], s[1][bb], s[2][bb], s[3][bb], s[4][bb], s[5][bb]};
      float y[4];
      wino4_at(colv, y);                                          // A^T (M A)   (along y)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        bool ok = (a < nrow[r]) & (bb < ncol[r]);
        float v = (y[a] + bias) + resv[r & 1][4 * a + bb];
        if (e.relu) v = fmaxf(v, 0.f);
#ifdef SHAPY_W4G_TIMING
        if (e.dbg & 1) ok &= v == 12345.678f;
#endif
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs_out, ok ? obase[r] : BAD,
                                              (a * W + bb) * out_ld * 4, 0);
      }
    }
  }
}

}  // namespace shapy
