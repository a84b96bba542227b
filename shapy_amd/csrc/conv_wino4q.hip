// Winograd F(4x4, 3x3) convolution on the f32 matrix cores, FOUR multiplying waves per workgroup.
//
// Same layers, same arithmetic and same filter layout as conv_wino4.hip (3x3 / stride 1 / pad 1: both
// convs of every BasicBlock of regressor/human_shape/models/backbone/hrnet.py:175-193); what changes is
// who does what inside the 256-thread workgroup (16 tiles x 48 output channels):
//   * conv_wino4.hip: one wave stages, three waves multiply (each all 36 Winograd positions of 16 output
//     channels, 144 accumulator registers) -- the fourth SIMD of the CU never issues an MFMA.
//   * here every wave multiplies AND stages a quarter of every chunk:
//       - the 36 positions x 3 channel groups of a chunk are 108 position-GEMM items of 4 MFMAs; every wave
//         takes 27 of them: waves 0..2 the positions 0..26 of "their" 16 output channels, wave 3 the
//         positions 27..35 (the lower half of Winograd row 4 and row 5) of ALL three channel groups.  108
//         accumulator registers instead of 144, all four SIMDs multiply.
//       - the output transform Y = A^T M A is linear in M: wave 3 applies the x-direction transform to what it
//         holds (row 5 completely, row 4 for j = 3..5) and hands 8 values per (tile, channel) to the owner
//         wave through LDS (24 KB, in the V buffer the last chunk does not use); the owners add them to
//         their own rows and finish as before (conv_wino4.h: bias, residual, ReLU, split-K, 16-byte stores).
//       - staging: lane (tile, channel) of wave w loads the 6 x 6 patch of ONE channel for tiles 4 w .. + 3
//         (36 buffer_load_dword, zero padding by the bounds check as in conv_wino4.hip), transforms it in
//         scalar registers (B^T d B: 144 VALU) and writes 36 floats to the V image in LDS.  The work is cut
//         into two-instruction steps that sit in the shadows of the chunk's MFMAs (one step per MFMA): rows
//         first, then a column at a time -- a finished column's registers take the same column of the
//         chunk after next.  No staging wave, no idle SIMD, 36 instead of 144 staging registers.
//   One barrier per chunk as before (V is double-buffered: chunk c + 1 is written while chunk c is read).
//   Filter fragments come straight from L2 through a ring of 9 items (36 registers).
#include <type_traits>

#include "conv_common.h"
#include "conv_wino4.h"

// Timing builds only (SHAPY_HIPCC_FLAGS=-DSHAPY_W4Q_DBG=<mask>, results are WRONG): stages of the chunk loop
// removed one at a time -- 1: transform VALU, 2: patch loads, 4: V writes, 8: filter refills, 16: V fragment reads.
#ifndef SHAPY_W4Q_DBG
#define SHAPY_W4Q_DBG 0
#endif

namespace shapy {

namespace {

// B^T (6 x 6) on a 6-vector in six steps of two independent VALU instructions, outputs over the inputs.
// Order chosen for the shortest live ranges: c, e first (kept to the end), o0 / o5 (replace d0 / d5, which
// nothing else reads), then a, b -- the last readers of d1..d4 -- and the four middle outputs.
struct BtTemps {
  float a, b, c, e;
};
template <int STEP>
__device__ __forceinline__ void wino4_bt_step(float &d0, float &d1, float &d2, float &d3, float &d4,
                                              float &d5, BtTemps &t) {
  if constexpr (STEP == 0) {
    t.c = d4 - d2;
    t.e = d3 - d1;
  } else if constexpr (STEP == 1) {
    d0 = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
  } else if constexpr (STEP == 2) {
    d5 = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
  } else if constexpr (STEP == 3) {
    t.a = fmaf(-4.f, d2, d4);
    t.b = fmaf(-4.f, d1, d3);
  } else if constexpr (STEP == 4) {
    d1 = t.a + t.b;
    d2 = t.a - t.b;
  } else {
    d3 = fmaf(2.f, t.e, t.c);
    d4 = fmaf(-2.f, t.e, t.c);
  }
}

template <int I>
using ic = std::integral_constant<int, I>;

// compile-time loop: f(ic<0>{}), f(ic<1>{}), ...
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(ic<I>{});
    static_for<N, I + 1>(f);
  }
}

}  // namespace

// KC > 0: the slice has exactly KC chunks, loop unrolled (exact s_waitcnt bookkeeping, compile-time
// staging schedule); KC == 0: generic loop.  S: split-K slices (conv_wino4.h: Wino4Split; the reduction
// units are the three owner waves).
template <int KC, int S = 1>
__global__ __launch_bounds__(256, 2) void conv_wino4q_kernel(ConvK p) {
  constexpr int PSTR = 1024;                          // bytes per position: 16 tiles x 16 ch f32
  constexpr int LDS_V = 36 * PSTR;
  constexpr int R = 9;                                // filter ring: items in flight (divides 27)
  constexpr int BAD = 0x40000000;                     // >= num_records of every buffer used here
  __shared__ __attribute__((aligned(16))) char lds[2 * LDS_V];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);        // wave-uniform: scalar branches
  const int wg = conv_tile_index(p);
  const int mv = wg / p.nbx, n_i = wg % p.nbx;
  const int m_i = mv / S, slice = mv % S;
  const int m_blk = m_i * 16, n_blk = n_i * 48;
  const int H = p.Hi, W = p.Wi;
  const int TW = (W + 3) >> 2, TH = (H + 3) >> 2;
  const int T = p.wino_tiles;
  const int CC = KC > 0 ? KC : (p.Cin >> 4) / S;      // chunks of this workgroup's slice ...
  const int cbase = slice * CC;                       // ... which starts at chunk cbase of the layer

  // ---------------- staging role: tile 4 wave + (lane >> 4), channel lane & 15 of the chunk ----------------
  const __amdgpu_buffer_rsrc_t rs_in =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.in), 0, p.in_bytes, 0x00020000);
  const int tile_s = 4 * wave + (lane >> 4), ch = lane & 15;
  unsigned row_off[6], col_off[6];              // (unsigned: two invalid parts sum to 2 GiB)
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
      row_off[i] = ok ? (b * H + y0 + i) * W * pix_stride + ch * 4 : BAD;
    }
#pragma unroll
    for (int j = 0; j < 6; ++j)
      col_off[j] = (unsigned)(x0 + j) < (unsigned)W ? (x0 + j) * pix_stride : BAD;
  }
  // LDS image V[p][tile][16 ch]: the 16-byte slot c4 of row `tile` sits at slot c4 ^ f(tile), f(r) =
  // (r ^ r >> 1) & 3 (conv_wino4.hip's layout: conflict-free ds_read_b128 fragments); a wave's
  // ds_write_b32 covers 4 rows x 64 bytes = every bank once
  const int st_off = tile_s * 64 + ((((ch >> 2) ^ tile_s ^ (tile_s >> 1)) & 3) << 4) + (ch & 3) * 4;
  float raw[6][6];
  unsigned co_cur = 0;                         // column part of the load offsets, opaque to the compiler:
  auto set_col = [&](int j) {                  // otherwise hipcc hoists all 36 row + column sums out of the
    co_cur = col_off[j];                       // K loop and keeps them in registers
    asm volatile("" : "+v"(co_cur));
  };
  auto gload = [&](int i, int j, int chunk) {  // patch element (i, j) of chunk `chunk` of the slice
    raw[i][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
        rs_in, (int)(row_off[i] + co_cur), (cbase + chunk) * 64, 0));
  };
  BtTemps bt;

  // ---------------- multiplying role ----------------
  const __amdgpu_buffer_rsrc_t rs_u =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.wgt2), 0, p.wgt2_bytes, 0x00020000);
  const int g = lane >> 4, l15 = lane & 15;
  const int frag_off = l15 * 64 + (((g ^ l15 ^ (l15 >> 1)) & 3) << 4);
  const int u_pos = (p.Cin >> 4) * p.Cout * 64, u_chunk = p.Cout * 64;

  auto run = [&](auto w3tag) {
    constexpr bool W3 = decltype(w3tag)::value;
    constexpr int NA = W3 ? 1 : 3;                    // V fragments per triple of items
    constexpr int AFD = (KC > 0 || W3) ? 2 : 1;       // ... double-buffered (unrolled kernels, wave 3)
    // item q = 0..26 -> (position, channel group): owner waves: (q, wave); wave 3: (27 + q / 3, q % 3)
    const int nw = W3 ? 0 : wave;
    const int u_lane = ((n_blk + 16 * nw + l15) * 16 + 4 * g) * 4;
    u32x4 bring[R];
    auto bload = [&](int slot, int q, int chunk, bool live) {
      const int pos = W3 ? 27 + q / 3 : q;
      bring[slot] = __builtin_amdgcn_raw_buffer_load_b128(
          rs_u, live ? u_lane : BAD, pos * u_pos + (cbase + chunk) * u_chunk + (W3 ? (q % 3) * 1024 : 0), 0);
    };
    f32x4 acc[27];
#pragma unroll
    for (int q = 0; q < 27; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: chunk 0 staged by everybody, chunk 1's patch requested ----
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      set_col(j);
#pragma unroll
      for (int i = 0; i < 6; ++i) gload(i, j, 0);
    }
#pragma unroll
    for (int q = 0; q < R; ++q) bload(q, q, 0, true);
    static_for<6>([&](auto i) {
      static_for<6>([&](auto st) {
        wino4_bt_step<st.value>(raw[i.value][0], raw[i.value][1], raw[i.value][2], raw[i.value][3],
                                raw[i.value][4], raw[i.value][5], bt);
      });
    });
    static_for<6>([&](auto j) {
      static_for<6>([&](auto st) {
        wino4_bt_step<st.value>(raw[0][j.value], raw[1][j.value], raw[2][j.value], raw[3][j.value],
                                raw[4][j.value], raw[5][j.value], bt);
      });
#pragma unroll
      for (int i = 0; i < 6; ++i)
        *reinterpret_cast<float *>(lds + st_off + (6 * i + j.value) * PSTR) = raw[i][j.value];
      if (CC > 1) {
        set_col(j.value);
#pragma unroll
        for (int i = 0; i < 6; ++i) gload(i, j.value, 1);
      }
    });

    Wino4Split sp;
    sp.slab = p.split_ws; sp.slab_bytes = p.split_bytes; sp.slice = slice;
    sp.unit = n_i * 3 + wave; sp.n_units = p.nbx * 3;
    sp.cnt = p.split_cnt + 2 * ((m_i * p.nbx + n_i) * 3 + wave);
    int ticket_v = 0;

    // ---- one chunk: 9 triples of items x 4 MFMAs each, one filler slot behind every MFMA ----
    auto chunk = [&](int cc, bool more, bool more2) {
      wino4_lds_barrier();                   // chunk cc is staged; buffer (cc + 1) & 1 is free
      if constexpr (S > 1 && !W3) {
        if (!more && lane == 0)
          ticket_v = __hip_atomic_fetch_add(sp.cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      const char *Vb = lds + (cc & 1) * LDS_V + frag_off;
      char *Vw = lds + ((cc + 1) & 1) * LDS_V + st_off;
      u32x4 af[AFD][NA];
#pragma unroll
      for (int e = 0; e < NA; ++e)
        af[0][e] = *reinterpret_cast<const u32x4 *>(Vb + (W3 ? 27 : e) * PSTR);
      static_for<9>([&](auto ttag) {
        constexpr int tt = ttag.value, cur = AFD == 2 ? (tt & 1) : 0;
        static_for<12>([&](auto stag) {
          constexpr int s = stag.value, kk = s / 3, e = s % 3, q = 3 * tt + e;
          constexpr int L = 12 * tt + s;       // filler slot index inside the chunk, 0 .. 107
          // A operand = filter fragment, B operand = V fragment: D[channel 4 g + r][tile l15]
          acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(bring[q % R][kk]),
                                                        __uint_as_float(af[cur][W3 ? 0 : e][kk]), acc[q], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          // (a) V fragments of the next triple (generic loop, owner waves: into the registers the item's last
          // MFMA has just read -- the loop-carried state leaves no room for a second set)
          if constexpr (SHAPY_W4Q_DBG & 16) {
          } else if constexpr (AFD == 2) {
            if constexpr (tt < 8 && s < NA)
              af[cur ^ 1][s] = *reinterpret_cast<const u32x4 *>(Vb + (W3 ? 27 + tt + 1 : 3 * (tt + 1) + s) * PSTR);
          } else {
            if constexpr (tt < 8 && kk == 3)
              af[0][e] = *reinterpret_cast<const u32x4 *>(Vb + (3 * (tt + 1) + e) * PSTR);
          }
          // (b) the ring slot this item has just left takes the item nine ahead
          if constexpr (kk == 3 && !(SHAPY_W4Q_DBG & 8)) {
            if constexpr (q + R < 27) bload(q % R, q + R, cc, true);
            else bload(q % R, q + R - 27, cc + 1, more);
          }
          // (c) one staging step: chunk cc + 1 (in `raw`) -> LDS; chunk cc + 2 -> `raw`
          if (more) {
            if constexpr (L >= 30 && L < 66 && !(SHAPY_W4Q_DBG & 1)) {                 // rows: T = d B (along x)
              constexpr int i = (L - 30) / 6, st = (L - 30) % 6;
              wino4_bt_step<st>(raw[i][0], raw[i][1], raw[i][2], raw[i][3], raw[i][4], raw[i][5], bt);
            }
            if constexpr (L >= 66 && L < 102) {                // columns: V = B^T T (along y), to LDS
              constexpr int j = (L - 66) / 6, st = (L - 66) % 6;
              if constexpr (!(SHAPY_W4Q_DBG & 1))
                wino4_bt_step<st>(raw[0][j], raw[1][j], raw[2][j], raw[3][j], raw[4][j], raw[5][j], bt);
              auto stv = [&](int i) {
                if constexpr (!(SHAPY_W4Q_DBG & 4)) *reinterpret_cast<float *>(Vw + (6 * i + j) * PSTR) = raw[i][j];
              };
              if constexpr (st == 1) stv(0);
              if constexpr (st == 2) stv(5);
              if constexpr (st == 4) { stv(1); stv(2); }
              if constexpr (st == 5) { stv(3); stv(4); }
            }
            if constexpr (L >= 72 && !(SHAPY_W4Q_DBG & 2)) {   // the finished column's next-but-one patch
              constexpr int j = (L - 72) / 6, i = (L - 72) % 6;
              if (more2) {
                if constexpr (i == 0) set_col(j);
                gload(i, j, cc + 2);
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        });
      });
    };
    if constexpr (KC > 0) {
      static_for<KC>([&](auto c) { chunk(c.value, c.value + 1 < KC, c.value + 2 < KC); });
    } else {
      for (int cc = 0; cc < CC; ++cc) chunk(cc, cc + 1 < CC, cc + 2 < CC);
    }

    // ---- output transform along x; rows 4 (j = 3..5) and 5 travel from wave 3 to the owners ----
    char *X = lds + (CC & 1) * LDS_V + lane * 16;      // the V buffer the last chunk did not use
    if constexpr (W3) {
#pragma unroll
      for (int n = 0; n < 3; ++n) {
        const f32x4 m3 = acc[0 + n], m4 = acc[3 + n], m5 = acc[6 + n];      // M[4][3..5]
        const f32x4 s34 = m3 + m4, d34 = m3 - m4;
        f32x4 r[8];
        r[0] = s34;
        r[1] = 2.f * d34;
        r[2] = 4.f * s34;
#pragma unroll
        for (int k = 0; k < 4; ++k) r[3][k] = fmaf(8.f, d34[k], m5[k]);
        const f32x4 m[6] = {acc[9 + n], acc[12 + n], acc[15 + n], acc[18 + n], acc[21 + n], acc[24 + n]};
        f32x4 o[4];
        wino4_at4(m, o);                                                     // row 5
#pragma unroll
        for (int k = 0; k < 4; ++k) r[4 + k] = o[k];
#pragma unroll
        for (int k = 0; k < 8; ++k) *reinterpret_cast<f32x4 *>(X + (n * 8 + k) * PSTR) = r[k];
      }
      wino4_lds_barrier();
      return;
    } else {
      Wino4Epi e;
      e.out = p.out; e.res = p.res; e.in = p.in; e.bias = p.bias;
      e.H = H; e.W = W; e.tiles = T; e.out_ld = p.out_ld; e.out_coff = p.out_coff;
      e.res_ld = p.res_ld; e.res_coff = p.res_coff; e.relu = p.relu;
      wino4_epilogue_x<S>(
          e, sp, __builtin_amdgcn_readfirstlane(ticket_v),
          [&](f32x4 (&s)[6][4]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const f32x4 m[6] = {acc[6 * i + 0], acc[6 * i + 1], acc[6 * i + 2],
                                  acc[6 * i + 3], acc[6 * i + 4], acc[6 * i + 5]};
              wino4_at4(m, s[i]);
            }
            const f32x4 s12 = acc[25] + acc[26], d12 = acc[25] - acc[26];
            const f32x4 p0 = acc[24] + s12;
            wino4_lds_barrier();                 // wave 3's part is in LDS
            f32x4 r[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) r[k] = *reinterpret_cast<const f32x4 *>(X + (wave * 8 + k) * PSTR);
            s[4][0] = p0 + r[0];
            s[4][1] = d12 + r[1];
            s[4][2] = s12 + r[2];
            s[4][3] = d12 + r[3];
#pragma unroll
            for (int k = 0; k < 4; ++k) s[5][k] = r[4 + k];
          },
          m_blk + l15, n_blk + 16 * wave + 4 * g, g, lane);
    }
  };
  if (wave == 3) run(std::true_type{});
  else run(std::false_type{});
}

// Launcher of the four-wave kernel: Cout % 48 == 0 (conv2d_wino4 keeps the 64-channel tile of the other
// kernel).  k is prepared by conv2d_wino4 (tiles, nbx / nby, split sizes, swz).
int conv2d_wino4q_launch(const ConvK &k, int S, hipStream_t s) {
  const dim3 grid(k.nbx * k.nby * S), blk(256);
  const int cps = (k.Cin / 16) / S;                              // chunks per slice
#define W4Q_LAUNCH(KC, SV) hipLaunchKernelGGL((conv_wino4q_kernel<KC, SV>), grid, blk, 0, s, k)
  if (S == 1) {
    if (cps == 3) W4Q_LAUNCH(3, 1);
    else if (cps == 6) W4Q_LAUNCH(6, 1);
    else if (cps == 12) W4Q_LAUNCH(12, 1);
    else W4Q_LAUNCH(0, 1);
  } else if (S == 2) {
    if (cps == 12) W4Q_LAUNCH(12, 2);
    else if (cps == 6) W4Q_LAUNCH(6, 2);
    else if (cps == 3) W4Q_LAUNCH(3, 2);
    else W4Q_LAUNCH(0, 2);
  } else if (S == 3 && cps == 8) {
    W4Q_LAUNCH(8, 3);
  } else if (S == 4 && cps == 6) {
    W4Q_LAUNCH(6, 4);
  } else if (S == 4 && cps == 3) {
    W4Q_LAUNCH(3, 4);
  } else {
    return SHAPY_EINVAL;
  }
#undef W4Q_LAUNCH
  return (int)hipGetLastError();
}

}  // namespace shapy
