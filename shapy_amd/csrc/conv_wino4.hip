// Winograd F(4x4, 3x3) convolution on the f32 matrix cores (v_mfma_f32_16x16x4_f32), four multiplying waves
// per workgroup.
//
// Same layers as conv_wino.hip (3x3 / stride 1 / pad 1: both convs of every BasicBlock of
// regressor/human_shape/models/backbone/hrnet.py:175-193), one step further down the
// minimal-filtering ladder:
//     Y = A^T [ sum_c (G g_c G^T) (.) (B^T d_c B) ] A,     d: 6x6 input patch, Y: 4x4 outputs
// 36 multiplies per 16 outputs and channel pair = 2.25 per output (direct: 9, F(2x2,3x3): 4), and a
// patch overlap of 36/16 = 2.25 loads per output pixel instead of 4.  Interpolation points
// {0, +-1, +-2, inf} (Lavin & Gray); filter transform in float64 on the host.  Measured on the CPU
// through the whole HRNet-W48 in float32 (tools/wino4_emulate.py --network): features within
// 1.8e-6 of float64, the same as the direct float32 convolution.
//
// GEMM view: 36 independent GEMMs (Winograd position p = 6 i + j), [tiles x Cin] x [Cin x Cout].
// One 256-thread workgroup = 16 consecutive tiles (4x4 output pixels each, row-major over (b, ty, tx)) x 48
// output channels; K in chunks of 16 input channels through two V images in LDS (36 KB each: V[p][tile][16 ch],
// 16-byte slots XOR-swizzled by the tile so that the ds_read_b128 fragment reads are conflict-free), one
// barrier per chunk; filter fragments straight from L2 (layout [p][Cin/16][Cout][16]: 1 KB contiguous per wave
// load) through a ring of 9 items.  The FILTER fragment is the MFMA's A operand and the V fragment its B
// operand, so the accumulators hold D[channel][tile]: a lane owns one tile and four consecutive output
// channels, the output transform A^T M A runs in registers and the epilogue (conv_wino4.h) moves 16 bytes per
// lane.
//
// Rounds 2-5 ran this layout with ONE staging wave and THREE multiplying waves (each all 36 positions of 16
// output channels, 144 accumulator registers): the fourth SIMD of a CU never issued an MFMA.  Since round 6
// EVERY wave multiplies and stages a quarter of every chunk:
//       - the 36 positions x 3 channel groups of a chunk are 108 position-GEMM items of 4 MFMAs; every wave
//         takes 27 of them: waves 0..2 the positions 0..26 of "their" 16 output channels, wave 3 the
//         positions 27..35 (the lower half of Winograd row 4 and row 5) of ALL three channel groups.  108
//         accumulator registers instead of 144, all four SIMDs multiply.
//       - the output transform Y = A^T M A is linear in M: wave 3 applies the x-direction transform to what it
//         holds (row 5 completely, row 4 for j = 3..5) and hands 8 values per (tile, channel) to the owner
//         wave through LDS (24 KB, in the V buffer the last chunk does not use); the owners add them to
//         their own rows and finish as before (conv_wino4.h: bias, residual, ReLU, split-K, 16-byte stores).
//       - staging: lane (tile, channel) of wave w loads the 6 x 6 patch of ONE channel for tiles 4 w .. + 3
//         (36 buffer_load_dword; zero padding by the buffer bounds check: an out-of-image row / column adds
//         0x40000000 to the byte offset, past num_records -- no select, no branch), transforms it in
//         registers and writes 36 floats to the V image in LDS.  On gfx950 an f32 MFMA and a VALU instruction
//         of the same SIMD do not overlap (tools/mfma_fillers.hip: +6..13 cycles per v_fma between two
//         v_mfma_f32_16x16x4_f32 of a wave, the same for a packed one; LDS / VMEM / SALU fillers are nearly
//         free), so the transform is written for the fewest VALU instructions: the patch lives in row PAIRS
//         (d[2k][j], d[2k+1][j]) -- the x pass is 36 v_pk_* on whole pairs, the y pass 8 per column with
//         op_sel / neg_hi picking the halves (SHAPY_W4Q_PK, below: the shipped form keeps the y pass scalar).  The work
//         is cut into 60 micro-steps behind the chunk's last 60 MFMAs (SHAPY_W4Q_GROUP: one step per MFMA, or
//         blocks of G steps behind every G-th): rows, then a column at a time; a finished column's registers
//         take the same column of the chunk after next.  No staging wave, no idle SIMD, 36 staging registers.
//   One barrier per chunk as before (V is double-buffered: chunk c + 1 is written while chunk c is read).
//   Filter fragments come straight from L2 through a ring of 9 items (36 registers).
#include <type_traits>

#include "conv_common.h"
#include "conv_wino4.h"

// Timing builds only (SHAPY_HIPCC_FLAGS=-DSHAPY_W4Q_DBG=<mask>, results are WRONG): stages of the chunk loop
// removed one at a time -- 1: transform VALU (raw values are written), 2: patch loads, 4: V writes, 8: filter
// refills, 16: V fragment reads.
#ifndef SHAPY_W4Q_DBG
#define SHAPY_W4Q_DBG 0
#endif
// arithmetic of the input transform: 2 = packed in both passes, 1 = packed x pass + scalar y pass, 0 = scalar
// (a v_pk_* costs one issue slot like a scalar VALU instruction but twice its time on the FMA lanes the f32 MFMA
// shares: tools/mfma_fillers.hip).  Same-box A/B at bs 64, images/s pipelined / one at a time
// (profiles/r06f_w4q_transform_arithmetic_ab.txt): 3 + 1-wave kernel 5,190; packed both, spread 5,320 / 5,200;
// scalar, spread 5,290 / 5,160; scalar, blocks of 12 5,365 / 5,230; packed x pass, blocks of 12 5,368 / 5,245.
#ifndef SHAPY_W4Q_PK
#define SHAPY_W4Q_PK 1
#endif
// staging steps per filler slot group (1: one step behind every MFMA; 3 / 6 / 12: blocks behind every 3rd / 6th / 12th)
#ifndef SHAPY_W4Q_GROUP
#define SHAPY_W4Q_GROUP 12
#endif

namespace shapy {

namespace {

using f32x2 = __attribute__((ext_vector_type(2))) float;

// Packed float32 math spelled out: hipcc (ROCm 7.2) scalarises <2 x float> arithmetic on gfx950, and here the
// instruction COUNT is the cost (header).  x * K + y with K an inline constant; k5 = (-5, -5) lives in a
// register pair (5.0 is no inline constant and VOP3P takes no literal).
#define W4Q_PKFMA_CONST(NAME, KSTR, KVAL)                                                             \
  __device__ __forceinline__ f32x2 NAME(f32x2 x, f32x2 y) {                                            \
    if constexpr (SHAPY_W4Q_PK == 0) return f32x2{fmaf(KVAL, x[0], y[0]), fmaf(KVAL, x[1], y[1])};     \
    f32x2 o;                                                                                           \
    asm("v_pk_fma_f32 %0, %1, " KSTR ", %2 op_sel_hi:[1,0,1]" : "=v"(o) : "v"(x), "v"(y));             \
    return o;                                                                                          \
  }
W4Q_PKFMA_CONST(pkfma_p4, "4.0", 4.f)
W4Q_PKFMA_CONST(pkfma_m4, "-4.0", -4.f)
W4Q_PKFMA_CONST(pkfma_p2, "2.0", 2.f)
W4Q_PKFMA_CONST(pkfma_m2, "-2.0", -2.f)
#undef W4Q_PKFMA_CONST
__device__ __forceinline__ f32x2 pkfma_v(f32x2 k, f32x2 x, f32x2 y) {
  if constexpr (SHAPY_W4Q_PK == 0) return f32x2{fmaf(-5.f, x[0], y[0]), fmaf(-5.f, x[1], y[1])};
  f32x2 o;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(o) : "v"(k), "v"(x), "v"(y));
  return o;
}
__device__ __forceinline__ f32x2 pkadd(f32x2 x, f32x2 y) {
  if constexpr (SHAPY_W4Q_PK == 0) return f32x2{x[0] + y[0], x[1] + y[1]};
  f32x2 o;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(o) : "v"(x), "v"(y));
  return o;
}
__device__ __forceinline__ f32x2 pksub(f32x2 x, f32x2 y) {
  if constexpr (SHAPY_W4Q_PK == 0) return f32x2{x[0] - y[0], x[1] - y[1]};
  f32x2 o;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(o) : "v"(x), "v"(y));
  return o;
}

// B^T (6 x 6) along x on a 6-vector of ROW PAIRS (both rows get the same combination: plain packed math),
// in place, six steps of two v_pk_* each.  Order chosen for the shortest live ranges: c, e first (kept to
// the end), o0 / o5 (replace d0 / d5, which nothing else reads), then a, b -- the last readers of d1..d4 --
// and the four middle outputs.
struct BtTemps {
  f32x2 a, b, c, e;
  f32x2 k5;                    // (-5, -5)
};
template <int STEP>
__device__ __forceinline__ void wino4_bt_rows_step(f32x2 &d0, f32x2 &d1, f32x2 &d2, f32x2 &d3, f32x2 &d4,
                                                   f32x2 &d5, BtTemps &t) {
  if constexpr (STEP == 0) {
    t.c = pksub(d4, d2);
    t.e = pksub(d3, d1);
  } else if constexpr (STEP == 1) {
    d0 = pkfma_p4(d0, pkfma_v(t.k5, d2, d4));
  } else if constexpr (STEP == 2) {
    d5 = pkfma_p4(d1, pkfma_v(t.k5, d3, d5));
  } else if constexpr (STEP == 3) {
    t.a = pkfma_m4(d2, d4);
    t.b = pkfma_m4(d1, d3);
  } else if constexpr (STEP == 4) {
    d1 = pkadd(t.a, t.b);
    d2 = pksub(t.a, t.b);
  } else {
    d3 = pkfma_p2(t.e, t.c);
    d4 = pkfma_m2(t.e, t.c);
  }
}

// B^T along y on one column held as three pairs P0 = (d0, d1), P1 = (d2, d3), P2 = (d4, d5); four steps of two
// v_pk_* each; the outputs come back as pairs (o0, o5), (o1, o2), (o3, o4):
//   (o0, o5) = 4 P0 - 5 P1 + P2
//   U = P2 - 4 P1 -> a = U.lo      W = P1 - 4 P0 -> b = W.hi      C = P2 - P1 -> c = C.lo      E = P1 - P0 -> e = E.hi
//   (o1, o2) = (a + b, a - b):   v_pk_add_f32 U, W   with op_sel (lo, hi) for both result halves, neg_hi on W
//   (o3, o4) = (c + 2 e, c - 2 e):   v_pk_fma_f32 E, 2.0, C   with E.hi (negated for the high half), C.lo
__device__ __forceinline__ f32x2 wino4_pk_sum_diff(f32x2 u, f32x2 w) {
  f32x2 o;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(o) : "v"(u), "v"(w));
  return o;
}
__device__ __forceinline__ f32x2 wino4_pk_c_pm_2e(f32x2 e, f32x2 c) {
  f32x2 o;
  asm("v_pk_fma_f32 %0, %1, 2.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0] neg_hi:[1,0,0]" : "=v"(o) : "v"(e), "v"(c));
  return o;
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
__global__ __launch_bounds__(256, 2) void conv_wino4_kernel(ConvK p) {
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
  f32x2 rp[3][6];                              // the patch in row pairs: rp[k][j] = (d[2k][j], d[2k+1][j])
  unsigned co_cur = 0;                         // column part of the load offsets, opaque to the compiler:
  auto set_col = [&](int j) {                  // otherwise hipcc hoists all 36 row + column sums out of the
    co_cur = col_off[j];                       // K loop and keeps them in registers
    asm volatile("" : "+v"(co_cur));
  };
  auto gload = [&](int i, int j, int chunk) {  // patch element (i, j) of chunk `chunk` of the slice
    rp[i >> 1][j][i & 1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
        rs_in, (int)(row_off[i] + co_cur), (cbase + chunk) * 64, 0));
  };
  BtTemps bt;
  bt.k5 = f32x2{-5.f, -5.f};
  asm volatile("" : "+v"(bt.k5));              // (kept in its two registers: not rematerialised per use)
  // ---- staging micro-step m = 0 .. 59: the patch in `rp` -> V image at Vw; chunk `nxt` -> `rp` (if >= 0) ----
  //   0..17   x pass of row pair m / 6, step m % 6
  //   18..21  y pass of column 0;  then for j = 1..5 seven steps: y pass of column j interleaved with the
  //           three load steps of column j - 1 (its registers are free);  57..59 loads of column 5
  auto stage_micro = [&](auto mtag, char *Vw, int nxt) {
    constexpr int m = decltype(mtag)::value;
    auto ypass = [&](auto jtag, auto sttag) {
      constexpr int j = decltype(jtag)::value, st = decltype(sttag)::value;
      auto stv = [&](int i, float v) {
        if constexpr (!(SHAPY_W4Q_DBG & 4)) *reinterpret_cast<float *>(Vw + (6 * i + j) * PSTR) = v;
      };
      if constexpr (SHAPY_W4Q_DBG & 1) {
        if constexpr (st == 0) { stv(0, rp[0][j][0]); stv(5, rp[2][j][1]); }
        if constexpr (st == 3) { stv(1, rp[0][j][1]); stv(2, rp[1][j][0]); stv(3, rp[1][j][1]); stv(4, rp[2][j][0]); }
      } else if constexpr (SHAPY_W4Q_PK < 2 && KC > 0) {     // (generic loop: packed, the scalar form spills there)
        // scalar y pass: d0..d5 = the halves of the column's three pairs
        const float d0 = rp[0][j][0], d1 = rp[0][j][1], d2 = rp[1][j][0], d3 = rp[1][j][1], d4 = rp[2][j][0],
                    d5 = rp[2][j][1];
        if constexpr (st == 0) {
          bt.c[0] = d4 - d2;
          bt.e[0] = d3 - d1;
          stv(0, fmaf(4.f, d0, fmaf(-5.f, d2, d4)));
        } else if constexpr (st == 1) {
          stv(5, fmaf(4.f, d1, fmaf(-5.f, d3, d5)));
          bt.a[0] = fmaf(-4.f, d2, d4);
        } else if constexpr (st == 2) {
          bt.b[0] = fmaf(-4.f, d1, d3);
          stv(3, fmaf(2.f, bt.e[0], bt.c[0]));
          stv(4, fmaf(-2.f, bt.e[0], bt.c[0]));
        } else {
          stv(1, bt.a[0] + bt.b[0]);
          stv(2, bt.a[0] - bt.b[0]);
        }
      } else if constexpr (st == 0) {
        const f32x2 o = pkfma_p4(rp[0][j], pkfma_v(bt.k5, rp[1][j], rp[2][j]));
        stv(0, o[0]); stv(5, o[1]);
      } else if constexpr (st == 1) {
        bt.a = pkfma_m4(rp[1][j], rp[2][j]);
        bt.b = pkfma_m4(rp[0][j], rp[1][j]);
      } else if constexpr (st == 2) {
        bt.c = pksub(rp[2][j], rp[1][j]);
        bt.e = pksub(rp[1][j], rp[0][j]);
      } else {
        const f32x2 o12 = wino4_pk_sum_diff(bt.a, bt.b), o34 = wino4_pk_c_pm_2e(bt.e, bt.c);
        stv(1, o12[0]); stv(2, o12[1]); stv(3, o34[0]); stv(4, o34[1]);
      }
    };
    auto loads = [&](auto jtag, auto sttag) {
      constexpr int j = decltype(jtag)::value, st = decltype(sttag)::value;
      if constexpr (!(SHAPY_W4Q_DBG & 2)) {
        if (nxt >= 0) {
          if constexpr (st == 0) set_col(j);
          gload(2 * st, j, nxt);
          gload(2 * st + 1, j, nxt);
        }
      }
    };
    if constexpr (m < 18) {
      if constexpr (!(SHAPY_W4Q_DBG & 1))
        wino4_bt_rows_step<m % 6>(rp[m / 6][0], rp[m / 6][1], rp[m / 6][2], rp[m / 6][3], rp[m / 6][4],
                                  rp[m / 6][5], bt);
    } else if constexpr (m < 22) {
      ypass(ic<0>{}, ic<m - 18>{});
    } else if constexpr (m < 57) {
      constexpr int j = 1 + (m - 22) / 7, r = (m - 22) % 7;
      if constexpr (r % 2 == 0) ypass(ic<j>{}, ic<r / 2>{});
      else loads(ic<j - 1>{}, ic<r / 2>{});
    } else {
      loads(ic<5>{}, ic<m - 57>{});
    }
  };

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
    static_for<60>([&](auto m) { stage_micro(m, lds + st_off, CC > 1 ? 1 : -1); });

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
          // (c) staging micro-steps: one behind each of the chunk's last 60 MFMAs, or SHAPY_W4Q_GROUP of them
          // together behind every SHAPY_W4Q_GROUP-th MFMA
          if (more) {
            if constexpr ((L + 1) % SHAPY_W4Q_GROUP == 0)
              static_for<SHAPY_W4Q_GROUP>([&](auto k) {
                constexpr int m = L + 1 - SHAPY_W4Q_GROUP + k.value - 48;
                if constexpr (m >= 0) stage_micro(ic<m>{}, Vw, more2 ? cc + 2 : -1);
              });
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
            s[4][0] = acc[24] + s12;
            s[4][1] = d12;
            s[4][2] = s12;
            s[4][3] = d12;
            wino4_lds_barrier();                 // wave 3's part is in LDS
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              s[4][k] = s[4][k] + *reinterpret_cast<const f32x4 *>(X + (wave * 8 + k) * PSTR);
              s[5][k] = *reinterpret_cast<const f32x4 *>(X + (wave * 8 + 4 + k) * PSTR);
            }
          },
          m_blk + l15, n_blk + 16 * wave + 4 * g, g, lane);
    }
  };
  if (wave == 3) run(std::true_type{});
  else run(std::false_type{});
}

// The kernel marks invalid accesses with the byte offset 0x40000000: every tensor it touches has
// to stay below 1 GiB (B <= 334 for HRNet's 256-channel 56x56 map).
bool conv_wino4_fits(const ConvK &k) {
  const unsigned long long lim = 0x40000000ull;
  // 16-byte residual loads / stores of the epilogue: k.vec4 (conv_prepare) = rows and channel
  // offsets in multiples of 4 floats, 16-byte-aligned tensors -- every HRNet tensor; anything else
  // takes another kernel
  const bool al = k.vec4 != 0;
  return al && k.Cout % 48 == 0 && k.in_bytes <= lim && 4ull * k.M * k.out_ld <= lim &&
         (!k.res || 4ull * k.M * k.res_ld <= lim) && 144ull * k.Cin * k.Cout < 0x7fffffffull;
}

// tile flag 0x100000 of ShapyConv.tile: wgt_wino holds F(4x4,3x3) filters [36][Cin/16][Cout][16]
// k.ksplit = S (SHAPY_TILE_W4_KSPLIT): K slices per output tile, 1 = none.  The instantiations below are
// the ones a HRNet plan asks for (chunks per slice unrolled where the class is hot: hipcc's s_waitcnt
// bookkeeping is exact only in straight-line code -- at the header of a real loop it waits for every patch
// load of the chunk after next); any other S / shape combination is SHAPY_EINVAL -- the plan, not the
// kernel, decides where splitting pays.
int conv2d_wino4(ConvK k, hipStream_t s) {
  if (!conv_wino4_fits(k)) return SHAPY_EINVAL;
  const int B = k.M / (k.Ho * k.Wo);
  k.wino_tiles = B * ((k.Hi + 3) / 4) * ((k.Wi + 3) / 4);
  k.wgt2_bytes = (unsigned)(144ull * k.Cin * k.Cout);          // 36 positions x f32
  const int S = k.ksplit < 1 ? 1 : k.ksplit;
  k.nbx = k.Cout / 48;
  k.nby = (k.wino_tiles + 15) / 16;
  const int chunks = k.Cin / 16;
  if (S > 1) {
    // slab: S x (Cout / 16 wave units) x 16 pixels x tiles x 64 bytes; counters: 2 per (m tile, unit)
    const unsigned long long slab = 64ull * 16 * k.wino_tiles * (k.Cout / 16) * S;
    if (S > 4 || chunks % S || !k.split_ws || !k.split_cnt || ((uintptr_t)k.split_ws & 15) ||
        slab > 0x40000000ull || slab > k.split_cap || 2 * k.nby * (k.Cout / 16) > k.split_cnt_cap)
      return SHAPY_EINVAL;
    k.split_bytes = (unsigned)slab;
  }
  // transformed filters larger than half an XCD's L2: one N slab per XCD (conv_tile_index)
  if (k.swz == 1 && k.nbx % 8 == 0 && k.wgt2_bytes > (2u << 20) && !k.no_nslab) k.swz = 2;
  const dim3 grid(k.nbx * k.nby * S), blk(256);
  const int cps = chunks / S;                                    // chunks per slice
#define W4_LAUNCH(KC, SV) hipLaunchKernelGGL((conv_wino4_kernel<KC, SV>), grid, blk, 0, s, k)
  if (S == 1) {
    if (cps == 3) W4_LAUNCH(3, 1);
    else if (cps == 6) W4_LAUNCH(6, 1);
    else if (cps == 12) W4_LAUNCH(12, 1);
    else if (cps == 16) W4_LAUNCH(16, 1);
    else W4_LAUNCH(0, 1);
  } else if (S == 2) {
    if (cps == 12) W4_LAUNCH(12, 2);
    else if (cps == 6) W4_LAUNCH(6, 2);
    else if (cps == 3) W4_LAUNCH(3, 2);
    else W4_LAUNCH(0, 2);
  } else if (S == 3 && cps == 8) {
    W4_LAUNCH(8, 3);
  } else if (S == 4 && cps == 6) {
    W4_LAUNCH(6, 4);
  } else if (S == 4 && cps == 3) {
    W4_LAUNCH(3, 4);
  } else {
    return SHAPY_EINVAL;
  }
#undef W4_LAUNCH
  return (int)hipGetLastError();
}

}  // namespace shapy
