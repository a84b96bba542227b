// Winograd F(2x2, 3x3) convolution on the f32 matrix cores (v_mfma_f32_16x16x4_f32).
//
// For the 3x3 / stride 1 / pad 1 layers of HighResolutionNet.forward
// (regressor/human_shape/models/backbone/hrnet.py:175-193: both convs of every BasicBlock, 89 %
// of the network's MACs) the minimal-filtering form
//     Y = A^T [ sum_c (G g_c G^T) (.) (B^T d_c B) ] A
// needs 16 multiplies per 2x2 output tile and channel pair instead of 36: 2.25x fewer MFMAs
// than the implicit GEMM of conv_igemm.hip, which is MFMA-throughput-bound on these layers.
// All tensors stay float32; the transforms only add / subtract (and halve, in the filter
// transform, which the host does in float64), so the result is float32-class: it differs from
// the direct sum by a few ulp of the partial sums (parity tests: 1e-4 on the network output).
//
// GEMM view: 16 independent GEMMs (one per Winograd position p = 4 i + j), each
// [tiles x Cin] x [Cin x Cout].  One 256-thread workgroup owns 16 consecutive tiles (2x2 output
// pixels each, row-major over (b, ty, tx)) x N = 16 NN output channels:
//   * staging: thread (tile, 4-channel group, patch row r) loads the 4 pixels of its patch row
//     (buffer_load_dwordx4, out-of-image taps zeroed by the hardware bounds check), applies the
//     row transform in registers, gets the rows it needs from its quad neighbours by DPP
//     (V[i] = +-T[i] +- T[partner]) and writes V[p = 4 r + j][tile][16 ch] to LDS
//     (16 KB per 16-channel chunk, XOR-swizzled 16-byte slots: conflict-free writes and reads);
//   * wave w multiplies positions 4 w .. 4 w + 3 for all NN channel tiles: A fragments from LDS
//     (one ds_read_b128 = the k-operands of 4 MFMAs, as in conv_igemm.hip), B fragments
//     straight from the transformed filters in global memory (L2-resident, layout
//     [p][Cin/16][Cout][16]: 1 KB contiguous per wave load; they are not shared between waves,
//     so LDS would only add a round trip), prefetched one chunk ahead;
//   * two V buffers, one barrier per chunk (their memory is needed for the epilogue anyway);
//     TM = 2 tile groups per workgroup reuse every B fragment twice (half the L2 traffic of the
//     transformed filters, the dominant stream: 147 KB per workgroup for a 48 -> 48 layer);
//   * epilogue: the 16 positions of a (tile, channel) sit in 4 different waves: accumulators
//     are parked in LDS (aliasing the staging buffer), then thread (tile, 4 channels) applies
//     A^T . A, bias, residual, ReLU and stores 4 pixels x 16 bytes (whole 192-byte rows per
//     pixel across the 12 threads of a tile).
#include <stdlib.h>

#include "conv_common.h"

namespace shapy {

__device__ __forceinline__ float quad_partner(float x) {
  // lanes (0,1,2,3) of every quad read lanes (2,2,1,1): quad_perm = 2 | 2<<2 | 1<<4 | 1<<6
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x5A, 0xf, 0xf, true));
}

// KC > 0: the layer has exactly KC K chunks (Cin = 16 KC) and ALL of them are staged in the
// prologue (KC V buffers, which fit into the memory the accumulator exchange needs anyway): the K
// loop then runs without barriers and without waiting for patch rows.  The generic loop (KC = 0)
// waits ~1.9 us of load latency per chunk for 0.73 us of MFMA issue (prefetch depth 1).
template <int NN, int TM, int KC>
__global__ __launch_bounds__(256, ((TM == 1 && NN <= 3) ? 3 : 2)) void conv_wino_kernel(ConvK p) {
  constexpr int N = 16 * NN, NCP = N + 4, N4 = N / 4, MT = 16 * TM;
  constexpr int PSTR = MT * 64;                       // bytes per position in a V buffer
  constexpr int LDS_V = 16 * PSTR;                    // V[16 pos][MT tiles][16 ch] f32
  constexpr int LDS_X = 16 * 16 * NCP * 4;            // M[16 pos][16 tiles][N + 4] f32
  constexpr int NVB = KC > 2 ? KC : 2;
  static_assert(KC == 0 || TM == 1, "all-K staging is a 16-tile variant");
  static_assert(KC == 0 || KC * LDS_V <= LDS_X, "all-K staging may not cost occupancy");
  // two V buffers (one barrier per chunk: chunk c+1 is staged while chunk c is multiplied); the
  // accumulator exchange of the epilogue reuses the same memory
  __shared__ __attribute__((aligned(16))) char lds[NVB * LDS_V > LDS_X ? NVB * LDS_V : LDS_X];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wg = conv_tile_index(p);
  const int m_blk = (wg / p.nbx) * MT, n_blk = (wg % p.nbx) * N;
  const int m_ld = m_blk;
  const int H = p.Hi, W = p.Wi;
  const int TW = (W + 1) >> 1, TH = (H + 1) >> 1;
  const int T = p.wino_tiles;

  const __amdgpu_buffer_rsrc_t rs_in =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_u =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.wgt2), 0, p.wgt2_bytes, 0x00020000);
  constexpr int OOB = 0x7fffffff;

  // ---- staging role: (patch row r, tile, 4-channel group c4), TM tiles per thread ----
  const int r = t & 3, c4 = (t >> 3) & 3, tile_s = ((t >> 5) << 1) | ((t >> 2) & 1);
  int a_off[TM];
  bool xok[TM][4], rowok[TM];
#pragma unroll
  for (int s = 0; s < TM; ++s) {
    const int tile = m_ld + 16 * s + tile_s;
    const int tt = tile < T ? tile : 0;
    const int tx = tt % TW;
    const int tq = tt / TW;
    const int ty = tq % TH;
    const int b = tq / TH;
    const int y = 2 * ty - 1 + r, x0 = 2 * tx - 1;
    rowok[s] = tile < T && (unsigned)y < (unsigned)H;
    a_off[s] = (((b * H + y) * W + x0) * p.in_ld + c4 * 4) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) xok[s][k] = (unsigned)(x0 + k) < (unsigned)W;
  }
  const int pix_stride = p.in_ld * 4;
  const int fsw = (tile_s ^ (tile_s >> 1)) & 3;
  const int st_off = tile_s * 64 + (((c4 ^ fsw ^ r) & 3) << 4) + 4 * r * PSTR;   // + j * PSTR

  u32x4 raw[TM][4];
  // Every chunk iteration issues the same 4 TM loads into raw[] (no branch around them, so the
  // compiler's vmcnt counts stay exact): the next chunk's patch rows -- or, in the LAST
  // iteration, this thread's residual pixels for the epilogue, whose HBM latency then hides
  // behind the last chunk's MFMAs and the accumulator exchange.
  const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void *>(p.res ? p.res : p.in), 0, 0x7ffffffe, 0x00020000);
  const bool has_res = p.res != nullptr;
  // epilogue item of this thread for tile group mt: (tile tl, channels col .. col + 3); tl / col
  // are recomputed where needed instead of being kept live across the K loop (the 16-tile
  // kernel sits exactly at its 168-register budget: three spilled registers cost 9.6 MB of
  // scratch traffic per launch)
  auto item = [&](int mt, bool (&ok)[4], int (&pixi)[4], int &col) {
    const int c4o = t % N4, tl = t / N4;
    col = n_blk + c4o * 4;
    const int tile = m_blk + 16 * mt + tl;
    const bool active = t < 16 * N4 && tile < T && col < p.Cout;
    const int tt_ = active ? tile : 0;
    const int tx = tt_ % TW;
    const int tq = tt_ / TW;
    const int ty = tq % TH;
    const int b = tq / TH;
    const int oy = 2 * ty, ox = 2 * tx;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int a = k >> 1, bb = k & 1;
      ok[k] = active && oy + a < H && ox + bb < W;
      pixi[k] = (b * H + oy + a) * W + ox + bb;
    }
  };
  auto gload = [&](int c0, bool live) {
    if (live) {
#pragma unroll
      for (int s = 0; s < TM; ++s)
#pragma unroll
        for (int k = 0; k < 4; ++k)
          raw[s][k] = __builtin_amdgcn_raw_buffer_load_b128(
              rs_in, (rowok[s] && xok[s][k]) ? a_off[s] + k * pix_stride + c0 * 4 : OOB, 0, 0);
    } else {
#pragma unroll
      for (int s = 0; s < TM; ++s) {
        bool ok[4];
        int pixi[4], col;
        item(s, ok, pixi, col);
        int off[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {       // plain selects: no divergent branch around a load
          const int o = (pixi[k] * p.res_ld + p.res_coff + col) * 4;
          off[k] = (has_res & ok[k]) ? o : OOB;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
          raw[s][k] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, off[k], 0, 0);
      }
    }
  };
  auto lstore = [&](int buf, u32x4 (&rw)[TM][4]) {
    // signs of B^T for this thread's patch row (recomputed: two registers less across the K loop)
    const int rr = threadIdx.x & 3;
    const float so = rr == 3 ? -1.f : 1.f, sp = (rr & 1) ? 1.f : -1.f;
#pragma unroll
    for (int s = 0; s < TM; ++s) {
      f32x4 d[4], tr[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) d[k] = __builtin_bit_cast(f32x4, rw[s][k]);
      tr[0] = d[0] - d[2];                  // T = d B   (columns of the patch row)
      tr[1] = d[1] + d[2];
      tr[2] = d[2] - d[1];
      tr[3] = d[1] - d[3];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaf(sp, quad_partner(tr[j][e]), so * tr[j][e]);   // B^T T
        *reinterpret_cast<f32x4 *>(lds + buf * LDS_V + st_off + s * 1024 + j * PSTR) = v;
      }
    }
  };

  // ---- MFMA role: wave w owns positions 4 w + pp, all TM tile groups x NN channel tiles ----
  const int kq = lane >> 4, l15 = lane & 15;
  const int frag_off = l15 * 64 + (((kq ^ ((l15 ^ (l15 >> 1)) & 3) ^ wave) & 3) << 4);
  const int CC = p.Cin >> 4;
  const int u_lane = ((n_blk + l15) * 16 + 4 * kq) * 4;
  const int u_pos_stride = CC * p.Cout * 64, u_chunk_stride = p.Cout * 64;

  u32x4 bfr[4][NN];
  auto bload = [&](int pp, int cc, bool live) {
    const int base = live ? u_lane + (4 * wave + pp) * u_pos_stride + cc * u_chunk_stride : OOB - 4096;
#pragma unroll
    for (int n = 0; n < NN; ++n)
      bfr[pp][n] = __builtin_amdgcn_raw_buffer_load_b128(rs_u, base + n * 1024, 0, 0);
  };

  f32x4 acc[4][TM][NN];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
      for (int n = 0; n < NN; ++n) acc[i][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto multiply = [&](const char *Vb, int cc, bool more) {
    // the A fragments of position pp + 1 are requested before the MFMAs of position pp
    u32x4 af[2][TM];
#pragma unroll
    for (int m = 0; m < TM; ++m)
      af[0][m] = *reinterpret_cast<const u32x4 *>(Vb + (4 * wave) * PSTR + m * 1024 + frag_off);
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      if (pp < 3) {
#pragma unroll
        for (int m = 0; m < TM; ++m)
          af[(pp + 1) & 1][m] = *reinterpret_cast<const u32x4 *>(
              Vb + (4 * wave + pp + 1) * PSTR + m * 1024 + frag_off);
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
          for (int n = 0; n < NN; ++n)
            acc[pp][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                __uint_as_float(af[pp & 1][m][kk]), __uint_as_float(bfr[pp][n][kk]),
                acc[pp][m][n], 0, 0, 0);
      // keep the refill of this position's B fragments HERE (hipcc otherwise sinks all 12 loads
      // to the end of the iteration, one LDS store + barrier before their first use)
      __builtin_amdgcn_sched_barrier(0);
      bload(pp, cc + 1, more);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  if constexpr (KC > 0) {
    // ---- all K chunks staged up front (the registers of the accumulators and of the later B
    // refills are still free here), then KC chunks of back-to-back MFMAs ----
    u32x4 rawk[KC][TM][4];
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        rawk[c][0][k] = __builtin_amdgcn_raw_buffer_load_b128(
            rs_in, (rowok[0] && xok[0][k]) ? a_off[0] + k * pix_stride + c * 64 : OOB, 0, 0);
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) bload(pp, 0, true);
#pragma unroll
    for (int c = 0; c < KC; ++c) lstore(c, rawk[c]);
    gload(0, false);                         // residual pixels: land during the K loop
    __syncthreads();
#pragma unroll
    for (int cc = 0; cc < KC; ++cc) {
      multiply(lds + cc * LDS_V, cc, cc + 1 < KC);
    }
    __syncthreads();                         // every wave is done with V: the exchange may overwrite it
  } else {
    gload(0, true);
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) bload(pp, 0, true);
    lstore(0, raw);
    __syncthreads();

    for (int cc = 0; cc < CC; ++cc) {
      const bool more = cc + 1 < CC;
      gload((cc + 1) * 16, more);
      multiply(lds + (cc & 1) * LDS_V, cc, more);
      // the other buffer was last read in iteration cc - 1, which every wave left through the
      // barrier below: it can be overwritten while slower waves still multiply this one
      if (more) lstore((cc + 1) & 1, raw);     // (after the last chunk raw[] holds the residual)
      __syncthreads();
    }
  }

  float *out = reinterpret_cast<float *>(p.out);
  static_assert(16 * N4 <= 256, "one (tile, 4 channels) epilogue item per thread");
#pragma unroll
  for (int mt = 0; mt < TM; ++mt) {
    bool okk[4];
    int pixi[4], col;
    item(mt, okk, pixi, col);
    const int c4o = t % N4, tl = t / N4;
    const bool active = okk[0];              // pixel (0,0) of a live tile is always inside
    bool ok[2][2];
    long pix[2][2];
    f32x4 rv[2][2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      ok[k >> 1][k & 1] = okk[k];
      pix[k >> 1][k & 1] = pixi[k];
      rv[k >> 1][k & 1] = __builtin_bit_cast(f32x4, raw[mt][k]);   // 0 where there is no residual
    }
    f32x4 bias = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias && active) bias = *reinterpret_cast<const f32x4 *>(p.bias + col);

    // ---- park the accumulators of tile group mt: M[p][tile][channel] ----
    if (mt > 0) __syncthreads();             // the previous group's M has been consumed
    {
      float *M = reinterpret_cast<float *>(lds);
#pragma unroll
      for (int pp = 0; pp < 4; ++pp)
#pragma unroll
        for (int n = 0; n < NN; ++n)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg)
            M[((4 * wave + pp) * 16 + 4 * kq + rg) * NCP + n * 16 + l15] = acc[pp][mt][n][rg];
    }
    __syncthreads();

    // ---- output transform + bias + residual + ReLU + store ----
    if (active) {
      const float *Mp = reinterpret_cast<const float *>(lds) + tl * NCP + c4o * 4;
      f32x4 tt[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 m0 = *reinterpret_cast<const f32x4 *>(Mp + (i * 4 + 0) * 16 * NCP);
        const f32x4 m1 = *reinterpret_cast<const f32x4 *>(Mp + (i * 4 + 1) * 16 * NCP);
        const f32x4 m2 = *reinterpret_cast<const f32x4 *>(Mp + (i * 4 + 2) * 16 * NCP);
        const f32x4 m3 = *reinterpret_cast<const f32x4 *>(Mp + (i * 4 + 3) * 16 * NCP);
        tt[i][0] = (m0 + m1) + m2;             // M A
        tt[i][1] = (m1 - m2) - m3;
      }
      f32x4 y[2][2];
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        y[0][bb] = ((tt[0][bb] + tt[1][bb]) + tt[2][bb]) + bias;      // A^T (M A)
        y[1][bb] = ((tt[1][bb] - tt[2][bb]) - tt[3][bb]) + bias;
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
          if (!ok[a][bb]) continue;
          f32x4 v = y[a][bb] + rv[a][bb];
          if (p.relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = relu_keep_nan(v[e]);
          }
          *reinterpret_cast<f32x4 *>(out + pix[a][bb] * p.out_ld + p.out_coff + col) = v;
        }
    }
  }
}

bool conv_wino_eligible(const ConvK &k) {
  return k.wgt2 != nullptr && k.ks == 3 && k.stride == 1 && k.pad == 1 && k.ups == 1 &&
         k.Cin % 16 == 0 && (k.Cout % 48 == 0 || k.Cout % 64 == 0) && k.vec4 && k.Ho == k.Hi &&
         k.Wo == k.Wi &&
         (!k.bias || ((uintptr_t)k.bias & 15) == 0) &&
         // the kernel addresses `res` (and the F(4x4) fallback lands here too) with 32-bit byte
         // offsets against a 2 GiB buffer resource: larger extents take the direct kernel
         4ull * k.M * k.out_ld < 0x7fffffffull && (!k.res || 4ull * k.M * k.res_ld < 0x7fffffffull);
}

// tm: 0 = choose, 1 / 2 = tile groups (16 tiles each) per workgroup.  Two groups halve the
// filter traffic per MFMA (the B fragments are reused for both) at 2 instead of 3 workgroups
// per CU; layers with few tiles keep 1 so that the grid still fills the chip.
int conv2d_wino(ConvK k, int tm, hipStream_t s) {
  const int B = k.M / (k.Ho * k.Wo);
  k.wino_tiles = B * ((k.Hi + 1) / 2) * ((k.Wi + 1) / 2);
  const unsigned long long wb = 64ull * k.Cin * k.Cout;        // 16 positions x f32
  if (wb >= 0x7fffffffull) return SHAPY_EINVAL;
  k.wgt2_bytes = (unsigned)wb;
  const int nn = k.Cout % 48 == 0 ? 3 : 4;            // 48- or 64-channel N tiles
  k.nbx = k.Cout / (16 * nn);
  // measured on MI355X at B = 64 (profiles/conv_bench_r02k_winograd_tm1_tm2.txt): two tile groups
  // (every B fragment used twice) win once the K loop is deep (192 -> 192 @14x14: 65 -> 57 us) and
  // the grid still has >= 1.5 workgroups per CU; 48 -> 48 / 96 -> 96 (3 / 6 chunks: 64 vs 67, 59 vs
  // 61 us) and 384 -> 384 @7x7 (256 workgroups with two groups: 64 vs 73 us) are faster with one
  if (tm == 0) tm = (k.Cin >= 192 && (long)((k.wino_tiles + 31) / 32) * k.nbx >= 384) ? 2 : 1;
  if (nn == 4) tm = 1;                                 // 64 accumulator + 64 B-fragment registers
  k.nby = (k.wino_tiles + 16 * tm - 1) / (16 * tm);
  // transformed filters larger than half an XCD's L2: one N slab per XCD (conv_tile_index)
  if (k.swz == 1 && k.nbx % 8 == 0 && k.wgt2_bytes > (2u << 20) && !k.no_nslab) k.swz = 2;
  // all-K staging (no barriers / patch waits inside the K loop) where the whole K extent fits the
  // exchange buffer: Cin = 48 with 48-wide N tiles, Cin = 64 with 64-wide ones (tile flag
  // 0x20000 keeps the generic loop: A/B benches)
  const dim3 grid(k.nbx * k.nby), blk(256);
  const bool allk = !k.no_allk && tm == 1;
  if (nn == 4 && allk && k.Cin == 64)
    hipLaunchKernelGGL((conv_wino_kernel<4, 1, 4>), grid, blk, 0, s, k);
  else if (nn == 4)
    hipLaunchKernelGGL((conv_wino_kernel<4, 1, 0>), grid, blk, 0, s, k);
  else if (tm == 2)
    hipLaunchKernelGGL((conv_wino_kernel<3, 2, 0>), grid, blk, 0, s, k);
  else if (allk && k.Cin == 48)
    hipLaunchKernelGGL((conv_wino_kernel<3, 1, 3>), grid, blk, 0, s, k);
  else
    hipLaunchKernelGGL((conv_wino_kernel<3, 1, 0>), grid, blk, 0, s, k);
  return (int)hipGetLastError();
}

}  // namespace shapy
