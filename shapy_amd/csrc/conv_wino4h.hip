// EXPERIMENTAL, opt-in (tile flag 0x400000 next to SHAPY_TILE_WINO4; never chosen by default and
// not yet run on a GPU): Winograd F(4x4,3x3) with the 36 positions of a (tile group, 16 channels)
// block split between TWO multiplying waves.  Written at the end of round 2 from the measurements
// of conv_wino4.hip, for the first GPU run of round 3 (tools/gpu_r03_a.sh) to judge.
//
// Why.  conv_wino4.hip is bound by task granularity and registers: one multiplying
// wave = 16 tiles x 16 channels x ALL 36 positions = 144 accumulator registers, so a SIMD holds two
// waves, a CU six multiplying waves on three SIMDs, and a 96-channel layer at B = 64 has only 1,176
// such tasks for 1,024 SIMDs.  The output transform is linear in the rows of M:
//     Y = A^T M A = sum_i A^T[:, i] (M A)[i, :]
// so a wave that owns the positions of rows i = 0..2 (or 3..5) can apply M A and its part of A^T in
// registers and only the 16 partial outputs per (tile, channel) -- not 18 accumulators -- have to
// meet: 72 accumulator registers per wave, a 128-register budget (four waves per SIMD), twice
// as many, half as long tasks.
//
// 512 threads: waves 0..5 multiply (n tile = wave % 3, row half = wave / 3), waves 6..7 stage
// (8 tiles each; lane (tile, channel pair): 36 x buffer_load_dwordx2, B^T d B in packed float32,
// ds_write_b64 into the same swizzled V[p][tile][16 ch] image as conv_wino4.hip).  Two workgroups
// per CU (73.7 KB of LDS each): 12 multiplying + 4 staging waves on 4 SIMDs.
// Epilogue: after the last chunk each multiplying wave turns its 18 accumulators into 16 partial
// outputs for its 4 tiles, hands the partials of two tiles to its partner through LDS (8 KB per
// wave, aliasing the V buffers) and finishes the other two: bias, residual, ReLU, 32 stores per lane.
#include <stdlib.h>

#include "conv_common.h"

namespace shapy {

typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef WINO4H_RING
#define WINO4H_RING 6         // B-fragment positions in flight per multiplying wave (divides 18)
#endif

__device__ __forceinline__ void wino4h_bt(const f32x2 (&d)[6], f32x2 (&o)[6]) {
  const f32x2 a = d[4] - 4.f * d[2];
  const f32x2 b = d[3] - 4.f * d[1];
  const f32x2 c = d[4] - d[2];
  const f32x2 e = d[3] - d[1];
  o[0] = 4.f * d[0] - 5.f * d[2] + d[4];
  o[1] = a + b;
  o[2] = a - b;
  o[3] = c + 2.f * e;
  o[4] = c - 2.f * e;
  o[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}

// one application of A^T (4 x 6) to a 6-vector
__device__ __forceinline__ void wino4h_at(const float (&m)[6], float (&o)[4]) {
  const float s12 = m[1] + m[2], d12 = m[1] - m[2];
  const float s34 = m[3] + m[4], d34 = m[3] - m[4];
  o[0] = (m[0] + s12) + s34;
  o[1] = fmaf(2.f, d34, d12);
  o[2] = fmaf(4.f, s34, s12);
  o[3] = fmaf(8.f, d34, d12) + m[5];
}

__device__ __forceinline__ void wino4h_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int KC>
__global__ __launch_bounds__(512, 4) void conv_wino4h_kernel(ConvK p) {
  constexpr int N = 48;
  constexpr int PSTR = 1024;                          // bytes per position: 16 tiles x 16 ch f32
  constexpr int LDS_V = 36 * PSTR;
  constexpr int R = WINO4H_RING;
  constexpr int BAD = 0x40000000;                     // >= num_records of every buffer used here
  static_assert(18 % R == 0 && R >= 2, "position q lives in ring slot q % R in every chunk");
  __shared__ __attribute__((aligned(16))) char lds[2 * LDS_V];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);        // wave-uniform: scalar branches
  const int wg = conv_tile_index(p);
  const int m_blk = (wg / p.nbx) * 16, n_blk = (wg % p.nbx) * N;
  const int H = p.Hi, W = p.Wi;
  const int TW = (W + 3) >> 2, TH = (H + 3) >> 2;
  const int T = p.wino_tiles;
  const int CC = p.Cin >> 4;

  if (wave >= 6) {
    // =========================== staging waves (8 tiles each) ===========================
    const __amdgpu_buffer_rsrc_t rs_in =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.in), 0, p.in_bytes, 0x00020000);
    const int tile_s = (wave - 6) * 8 + (lane >> 3), c2 = lane & 7;
    unsigned row_off[6], col_off[6];            // (unsigned: two invalid parts sum to 2 GiB)
    {
      const int pix_stride = p.in_ld * 4;
      const int tile = m_blk + tile_s;
      const bool live = tile < T;
      const int tt = live ? tile : 0;
      const int tx = tt % TW;
      const int tq = tt / TW;
      const int ty = tq % TH;
      const int b = tq / TH;
      const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const bool ok = live & ((unsigned)(y0 + i) < (unsigned)H);
        row_off[i] = ok ? (b * H + y0 + i) * W * pix_stride + c2 * 8 : BAD;
      }
#pragma unroll
      for (int j = 0; j < 6; ++j)
        col_off[j] = (unsigned)(x0 + j) < (unsigned)W ? (x0 + j) * pix_stride : BAD;
    }
    // same LDS image as conv_wino4.hip: 16-byte slot (c2 >> 1) of row `tile` at slot ^ f(tile)
    const int st_off =
        tile_s * 64 + ((((c2 >> 1) ^ tile_s ^ (tile_s >> 1)) & 3) << 4) + (c2 & 1) * 8;

    f32x2 raw[6][6];
    auto gload_col = [&](int j, int c0) {
      // (the asm keeps hipcc from hoisting 36 row + column sums out of the loop.  This wave has
      // 128 registers for a 72-register patch: hipcc spills ~11 of them; the staging waves have 2x
      // slack against the three multiplying waves that share a SIMD's matrix core)
      unsigned co = col_off[j] + c0 * 4;
      asm volatile("" : "+v"(co));
#pragma unroll
      for (int i = 0; i < 6; ++i)
        raw[i][j] = __builtin_bit_cast(
            f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_in, (int)(row_off[i] + co), 0, 0));
    };
#pragma unroll
    for (int j = 0; j < 6; ++j) gload_col(j, 0);
    for (int cc = 0; cc < CC; ++cc) {
      const bool more = cc + 1 < CC;
#pragma unroll
      for (int i = 0; i < 6; ++i) {                                // T = d B  (along x)
        f32x2 o[6];
        wino4h_bt(raw[i], o);
#pragma unroll
        for (int j = 0; j < 6; ++j) raw[i][j] = o[j];
        __builtin_amdgcn_sched_barrier(0);
      }
      char *Vb = lds + (cc & 1) * LDS_V + st_off;
#pragma unroll
      for (int j = 0; j < 6; ++j) {                                // V = B^T T  (along y)
        const f32x2 colv[6] = {raw[0][j], raw[1][j], raw[2][j], raw[3][j], raw[4][j], raw[5][j]};
        f32x2 v[6];
        wino4h_bt(colv, v);
#pragma unroll
        for (int i = 0; i < 6; ++i) *reinterpret_cast<f32x2 *>(Vb + (6 * i + j) * PSTR) = v[i];
        __builtin_amdgcn_sched_barrier(0);
        if (more) gload_col(j, (cc + 1) * 16);
        __builtin_amdgcn_sched_barrier(0);
      }
      wino4h_lds_barrier();                  // chunk cc is staged (barrier #cc)
    }
    wino4h_lds_barrier();                    // #CC: every multiply is done (V buffers are free)
    wino4h_lds_barrier();                    // #CC + 1: the partial outputs are in LDS
    return;
  }

  // =========================== multiplying waves ===========================
  // wave w: channels n_blk + 16 (w % 3) .. + 15, positions 18 (w / 3) .. + 17 (rows 3 h .. 3 h + 2)
  const __amdgpu_buffer_rsrc_t rs_u =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.wgt2), 0, p.wgt2_bytes, 0x00020000);
  const int g = lane >> 4, l15 = lane & 15;
  const int jn = wave % 3, h = wave / 3;
  const int frag_off = l15 * 64 + (((g ^ l15 ^ (l15 >> 1)) & 3) << 4) + 18 * h * PSTR;
  const int n0 = n_blk + 16 * jn;
  const int u_lane = ((n0 + l15) * 16 + 4 * g) * 4;
  const int u_pos = CC * p.Cout * 64, u_chunk = p.Cout * 64;
  const int u_half = 18 * h * u_pos;

  u32x4 bring[R];
  auto bload = [&](int slot, int pos, int cc, bool live) {
    bring[slot] = __builtin_amdgcn_raw_buffer_load_b128(rs_u, live ? u_lane : BAD,
                                                        u_half + pos * u_pos + cc * u_chunk, 0);
  };

  f32x4 acc[18];
#pragma unroll
  for (int q = 0; q < 18; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < R; ++q) bload(q, q, 0, true);

  auto chunk = [&](int cc, bool more) {
    wino4h_lds_barrier();                    // chunk cc is staged
    const char *Vb = lds + (cc & 1) * LDS_V + frag_off;
#pragma unroll
    for (int pp = 0; pp < 18; pp += 2) {
      const u32x4 a0 = *reinterpret_cast<const u32x4 *>(Vb + pp * PSTR);
      const u32x4 a1 = *reinterpret_cast<const u32x4 *>(Vb + (pp + 1) * PSTR);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        acc[pp] = __builtin_amdgcn_mfma_f32_16x16x4f32(
            __uint_as_float(a0[kk]), __uint_as_float(bring[pp % R][kk]), acc[pp], 0, 0, 0);
        acc[pp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(
            __uint_as_float(a1[kk]), __uint_as_float(bring[(pp + 1) % R][kk]), acc[pp + 1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int q = pp + e + R;
        if (q < 18) bload((pp + e) % R, q, cc, true);
        else bload((pp + e) % R, q - 18, cc + 1, more);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if constexpr (KC > 0) {
#pragma unroll
    for (int cc = 0; cc < KC; ++cc) chunk(cc, cc + 1 < KC);
  } else {
    for (int cc = 0; cc < CC; ++cc) chunk(cc, cc + 1 < CC);
  }
  wino4h_lds_barrier();                      // #CC: nobody reads V any more

  // ---- partial output transform: rows 3 h .. 3 h + 2 of M, all four tiles of the lane ----
  // y_h[a][b] = sum_{i in rows} A^T[a][i] (M A)[i][b];  h = 0: (s0+s1+s2, s1-s2, s1+s2, s1-s2),
  // h = 1: (s3+s4, 2 (s3-s4), 4 (s3+s4), 8 (s3-s4) + s5)
  float part[4][16];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float s[3][4];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float m[6] = {acc[6 * i + 0][r], acc[6 * i + 1][r], acc[6 * i + 2][r],
                          acc[6 * i + 3][r], acc[6 * i + 4][r], acc[6 * i + 5][r]};
      wino4h_at(m, s[i]);                                         // M A   (along x)
    }
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) {
      const float sum = s[0][bb] + s[1][bb], dif = s[0][bb] - s[1][bb];
      if (h == 0) {                            // rows 0, 1, 2: s[0] = s0, s[1] = s1, s[2] = s2
        const float s12 = s[1][bb] + s[2][bb], d12 = s[1][bb] - s[2][bb];
        part[r][0 + bb] = s[0][bb] + s12;
        part[r][4 + bb] = d12;
        part[r][8 + bb] = s12;
        part[r][12 + bb] = d12;
      } else {                                 // rows 3, 4, 5: s[0] = s3, s[1] = s4, s[2] = s5
        part[r][0 + bb] = sum;
        part[r][4 + bb] = 2.f * dif;
        part[r][8 + bb] = 4.f * sum;
        part[r][12 + bb] = fmaf(8.f, dif, s[2][bb]);
      }
    }
  }
  // ---- exchange: this wave finishes tiles r = 2 h, 2 h + 1 and hands the others to its partner
  // (wave + 3 or wave - 3, same lanes).  Receiver region: 8 KB per wave, value v at v * 256 + 4 lane
  float *xch = reinterpret_cast<float *>(lds);
  const int partner = h == 0 ? wave + 3 : wave - 3;
#pragma unroll
  for (int rl = 0; rl < 2; ++rl)
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const float give = h == 0 ? part[2 + rl][v] : part[rl][v];
      xch[partner * 2048 + (rl * 16 + v) * 64 + lane] = give;
    }
  wino4h_lds_barrier();                      // #CC + 1
  const int col = n0 + l15;
  const float bias = p.bias ? p.bias[col] : 0.f;
  const __amdgpu_buffer_rsrc_t rs_out =
      __builtin_amdgcn_make_buffer_rsrc(p.out, 0, BAD, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void *>(p.res ? p.res : p.in), 0, BAD, 0x00020000);
  const bool has_res = p.res != nullptr;
#pragma unroll
  for (int rl = 0; rl < 2; ++rl) {
    const int tile = m_blk + 4 * g + 2 * h + rl;
    const bool live = tile < T;
    const int tt = live ? tile : 0;
    const int tx = tt % TW;
    const int tq = tt / TW;
    const int ty = tq % TH;
    const int b = tq / TH;
    const int pix0 = (b * H + 4 * ty) * W + 4 * tx;
    const int obase = live ? (pix0 * p.out_ld + p.out_coff + col) * 4 : BAD;
    const int rbase = (live & has_res) ? (pix0 * p.res_ld + p.res_coff + col) * 4 : BAD;
    const int nrow = H - 4 * ty, ncol = W - 4 * tx;
    float resv[16];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) {
        const bool ok = (a < nrow) & (bb < ncol);
        resv[4 * a + bb] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
            rs_res, ok ? rbase : BAD, (a * W + bb) * p.res_ld * 4, 0));
      }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) {
        const bool ok = (a < nrow) & (bb < ncol);
        const float own = h == 0 ? part[rl][4 * a + bb] : part[2 + rl][4 * a + bb];
        const float got = xch[wave * 2048 + (rl * 16 + 4 * a + bb) * 64 + lane];
        float v = ((own + got) + bias) + resv[4 * a + bb];
        if (p.relu) v = fmaxf(v, 0.f);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs_out, ok ? obase : BAD,
                                              (a * W + bb) * p.out_ld * 4, 0);
      }
  }
}

// tile flags 0x100000 | 0x400000 of ShapyConv.tile (EXPERIMENTAL, see the file header)
int conv2d_wino4h(ConvK k, hipStream_t s) {
  if (!conv_wino4_fits(k)) return SHAPY_EINVAL;
  const int B = k.M / (k.Ho * k.Wo);
  k.wino_tiles = B * ((k.Hi + 3) / 4) * ((k.Wi + 3) / 4);
  k.wgt2_bytes = (unsigned)(144ull * k.Cin * k.Cout);          // 36 positions x f32
  k.nbx = k.Cout / 48;
  k.nby = (k.wino_tiles + 15) / 16;
  if (k.swz == 1 && k.nbx % 8 == 0 && k.wgt2_bytes > (2u << 20) && !k.no_nslab) k.swz = 2;
  const dim3 grid(k.nbx * k.nby), blk(512);
  if (k.Cin == 48)
    hipLaunchKernelGGL(conv_wino4h_kernel<3>, grid, blk, 0, s, k);
  else if (k.Cin == 96)
    hipLaunchKernelGGL(conv_wino4h_kernel<6>, grid, blk, 0, s, k);
  else
    hipLaunchKernelGGL(conv_wino4h_kernel<0>, grid, blk, 0, s, k);
  return (int)hipGetLastError();
}

}  // namespace shapy
