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


// one application of A^T (4 x 6) to a 6-vector of 4 channels
__device__ __forceinline__ void wino4_at4(const f32x4 (&m)[6], f32x4 (&o)[4]) {
  const f32x4 s12 = m[1] + m[2], d12 = m[1] - m[2];
  const f32x4 s34 = m[3] + m[4], d34 = m[3] - m[4];
  o[0] = (m[0] + s12) + s34;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    o[1][k] = fmaf(2.f, d34[k], d12[k]);
    o[2][k] = fmaf(4.f, s34[k], s12[k]);
    o[3][k] = fmaf(8.f, d34[k], d12[k]) + m[5][k];
  }
}

// Start stagger of the multiplying waves (both F(4x4) kernels).  Two workgroups share a CU and every
// SIMD's matrix core is shared by one multiplying wave of each; launched together, both multiply at
// the same time (each at half rate) and then both sit in their epilogue / wait for the next staged
// chunk with the matrix cores idle.  An offset d between their multiply phases is PRESERVED from task
// to task (whoever multiplies alone runs at full rate, so the phases neither converge nor drift), so
// one delay at the start -- d ~ epilogue + hand-over time -- puts one workgroup's memory phase under
// the other's MFMA phase for the whole launch.  The second workgroup of a CU is the one whose waves
// got wave slot 1 of their SIMD (HW_ID[0]; speed only: a wrong guess costs nothing but the delay).
__device__ __forceinline__ void wino4_start_stagger(int units) {
  if (units <= 0) return;
  unsigned hw_id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
  if (hw_id & 1u)
    for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(2);       // 2 x 64 clocks
}

// Epilogue of a multiplying wave (both F(4x4) kernels).  The MFMAs run with the FILTER fragment as
// the A operand and the V fragment as B, so the C layout is D[channel 4 g + r][tile l15]: a lane
// holds ONE tile and FOUR CONSECUTIVE output channels of it in every accumulator -- the output
// transform A^T M A runs on float4s in the registers of the lane that holds the MFMA results, and
// the residual loads / stores are 16 bytes per lane (4 lanes = the wave's 64-byte channel segment
// of a pixel): 16 + 16 vector-memory instructions per wave instead of 64 + 64.  (Round 3's layout
// -- tiles along the C rows, one channel per lane, 4-byte accesses -- made the epilogue the
// longest phase of a task: 10-16 us next to 3 x 5 us of multiplies on the 48-channel class,
// profiles/r03k_w4g_phase_stamps.txt; a store tail is bound by the number of store INSTRUCTIONS,
// not by bytes.)  Products and summation order are unchanged: bit-identical outputs.
// Residual loads and stores are buffer instructions: per-lane part of the address = the tile's
// first pixel + the channel group, pixel offset inside the tile = scalar offset; pixels outside the
// image (partial edge tiles) and dead tiles get an out-of-range offset, which drops the access.
// The residual of one pixel column (4 pixels) is in flight ahead of the one being transformed.
struct Wino4Epi {
  int dbg;                     // tuning builds (-DSHAPY_W4G_TIMING): 1 no stores, 2 no residual loads
  void *out;
  const void *res;             // nullptr: none
  const void *in;              // any valid address for the residual resource when res is null
  const float *bias;
  int H, W, tiles, out_ld, out_coff, res_ld, res_coff, relu;
};

// tile: the lane's tile (m_blk + l15); col4: its first output channel (n0 + 4 g4)
__device__ __forceinline__ void wino4_epilogue(const Wino4Epi &e, const f32x4 (&acc)[36], int tile,
                                               int col4) {
  constexpr int BAD = 0x40000000;
  const int H = e.H, W = e.W;
  const int TW = (W + 3) >> 2, TH = (H + 3) >> 2;
  f32x4 bias = f32x4{0.f, 0.f, 0.f, 0.f};
  if (e.bias) bias = f32x4{e.bias[col4], e.bias[col4 + 1], e.bias[col4 + 2], e.bias[col4 + 3]};
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(e.out, 0, BAD, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void *>(e.res ? e.res : e.in), 0, BAD, 0x00020000);
  const bool has_res = e.res != nullptr;
  const int out_ld = e.out_ld, res_ld = e.res_ld;
  const bool live = tile < e.tiles;
  const int tt = live ? tile : 0;
  const int tx = tt % TW;
  const int tq = tt / TW;
  const int ty = tq % TH;
  const int b = tq / TH;
  const int pix0 = (b * H + 4 * ty) * W + 4 * tx;
  const int obase = live ? (pix0 * out_ld + e.out_coff + col4) * 4 : BAD;
  const int rbase = (live & has_res) ? (pix0 * res_ld + e.res_coff + col4) * 4 : BAD;
  const int nrow = H - 4 * ty;      // >= 4 for a full tile
  const int ncol = W - 4 * tx;
  const float relu_lo = e.relu ? 0.f : -__builtin_inff();
  f32x4 resv[2][4];
  auto rload = [&](int bb, f32x4 (&rv)[4]) {       // pixel column bb of the tile: 4 pixels
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      bool ok = (a < nrow) & (bb < ncol);
#if defined(SHAPY_W4G_TIMING) || defined(SHAPY_WINO_TIMING)
      ok &= !(e.dbg & 2);
#endif
      rv[a] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                            rs_res, ok ? rbase : BAD, (a * W + bb) * res_ld * 4, 0));
    }
  };
  rload(0, resv[0]);
  f32x4 s[6][4];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const f32x4 m[6] = {acc[6 * i + 0], acc[6 * i + 1], acc[6 * i + 2],
                        acc[6 * i + 3], acc[6 * i + 4], acc[6 * i + 5]};
    wino4_at4(m, s[i]);                                           // M A   (along x)
  }
#pragma unroll
  for (int bb = 0; bb < 4; ++bb) {
    if (bb + 1 < 4) rload(bb + 1, resv[(bb + 1) & 1]);
    const f32x4 colv[6] = {s[0][bb], s[1][bb], s[2][bb], s[3][bb], s[4][bb], s[5][bb]};
    f32x4 y[4];
    wino4_at4(colv, y);                                           // A^T (M A)   (along y)
    f32x4 v[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      v[a] = (y[a] + bias) + resv[bb & 1][a];
      // ReLU without a branch per store and without hipcc's canonicalising second v_max:
      // max(v, 0) or max(v, -inf)
#pragma unroll
      for (int k = 0; k < 4; ++k) asm("v_max_f32 %0, %1, %2" : "=v"(v[a][k]) : "v"(v[a][k]), "v"(relu_lo));
    }
    // The four pixels of the column are complete (four DIFFERENT register quads) before the first
    // store is issued, and nothing may write a VGPR for two wait states after the last one:
    // gfx950 reads the data of a 16-byte buffer store late, and hipcc (ROCm 7.2) only guards that
    // hazard for MUBUF stores WITHOUT an SGPR offset -- a v_pk_add issued right behind
    // `buffer_store_dwordx4 ..., s54 offen` replaced the second dword of lanes 12-15 of every
    // 16-lane group by the NEXT pixel's value (GPU run D of round 4, profiles/r04d_*).
    int voff[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      bool ok = (a < nrow) & (bb < ncol);
#if defined(SHAPY_W4G_TIMING) || defined(SHAPY_WINO_TIMING)
      if (e.dbg & 1) ok &= v[a][0] == 12345.678f;
#endif
      voff[a] = ok ? obase : BAD;
    }
    // (addresses included: no vector instruction at all between the four stores)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int a = 0; a < 4; ++a)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[a]), rs_out, voff[a],
                                             (a * W + bb) * out_ld * 4, 0);
    asm volatile("s_nop 1");
    __builtin_amdgcn_sched_barrier(0);
  }
}

}  // namespace shapy
