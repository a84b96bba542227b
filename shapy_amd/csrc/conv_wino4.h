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
  void *out;
  const void *res;             // nullptr: none
  const void *in;              // any valid address for the residual resource when res is null
  const float *bias;
  int H, W, tiles, out_ld, out_coff, res_ld, res_coff, relu;
};

// Split-K (conv_wino4.hip, template parameter S > 1): the K loop of one (16 tiles x N channels) output
// tile is cut into S slices that run as S workgroups; every multiplying WAVE is its own reduction unit
// (its 16 tiles x 16 channels), so no workgroup barrier is involved.  Protocol per wave unit, following
// the in-launch split-K recipe of the CDNA programming guide (section 6, guideline 16):
//   * at the start of its LAST chunk a wave draws a ticket (relaxed agent-scope fetch_add; the latency
//     hides behind the chunk's 144 MFMAs).  Tickets 0 .. S-2: output transform, the 16 pixels x 4
//     channels of the lane go to the slice's slab as 16-byte WRITE-THROUGH stores (sc1), the wave drains
//     its stores (s_waitcnt vmcnt(0)) and adds 1 to the unit's `done` counter.  Ticket S-1 = the reducer:
//     polls `done` relaxed until it reads S-1 -- every other slice already holds a ticket, i.e. is
//     resident and past its last barrier, so the wait is bounded --, reads the other slabs with sc1
//     loads, adds the S partial outputs IN SLICE ORDER (own partial from registers: the sum does not
//     depend on who arrived last), then bias + residual + ReLU + store as without a split, and puts both
//     counters back to 0 for the next launch that uses them.
//   * the output transform is linear, so the slices exchange TRANSFORMED partials: 16 values per
//     (tile, channel) instead of 36.
// Slab layout [slice][n tile x wave][pixel 4 bb + a][tile][16 channels] f32: for a fixed pixel the 64
// lanes of a wave (16 tiles x 4 channel quads) write one contiguous 1 KB run.  Counters: two ints per
// (m tile, n tile, wave), zero before the first launch (the caller's job, include/shapy_hip.h).
struct Wino4Split {
  void *slab;
  unsigned slab_bytes;
  int slice;                   // this workgroup's K slice
  int unit;                    // (n tile * waves + wave): slab index inside a slice
  int n_units;                 // n tiles * waves
  int *cnt;                    // this wave unit's {ticket, done}
};

// tile: the lane's tile (m_blk + l15); col4: its first output channel (n0 + 4 g4); g4 = lane >> 4.
// S == 1: no split (sp / ticket unused).
// xf(s): fills s[i][b] = (M A)[i][b], the accumulators transformed along x -- called exactly once on every
// path (the four-wave kernel of conv_wino4.hip completes rows 4 / 5 through an LDS exchange with a
// workgroup barrier inside).
template <int S, typename XF>
__device__ __forceinline__ void wino4_epilogue_x(const Wino4Epi &e, const Wino4Split &sp, int ticket,
                                                 XF &&xf, int tile, int col4, int g4, int lane) {
  constexpr int BAD = 0x40000000;
  const int H = e.H, W = e.W;
  const int TW = (W + 3) >> 2, TH = (H + 3) >> 2;
  const bool live = tile < e.tiles;
  f32x4 s[6][4];
  auto transform_x = [&]() { xf(s); };
  // slab addressing: per-lane part = the tile + channel quad, slice / unit / pixel = scalar offset
  const __amdgpu_buffer_rsrc_t rs_slab =
      __builtin_amdgcn_make_buffer_rsrc(S > 1 ? sp.slab : e.out, 0, S > 1 ? sp.slab_bytes : 0, 0x00020000);
  const int slab_lane = live ? (tile * 16 + 4 * g4) * 4 : BAD;
  const int px_bytes = e.tiles * 64, unit_bytes = 16 * px_bytes;
  if constexpr (S > 1) {
    if (ticket != S - 1) {
      // ---- not the last slice of this unit to arrive: publish the transformed partial and leave ----
      transform_x();
      const int base = (sp.slice * sp.n_units + sp.unit) * unit_bytes;
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) {
        const f32x4 colv[6] = {s[0][bb], s[1][bb], s[2][bb], s[3][bb], s[4][bb], s[5][bb]};
        f32x4 y[4];
        wino4_at4(colv, y);                                       // A^T (M A)   (along y)
        // (the 16-byte store hazard of the output stores below applies here too: four finished register
        // quads, no vector instruction between or right behind the stores)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < 4; ++a)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y[a]), rs_slab, slab_lane,
                                                 base + (4 * bb + a) * px_bytes, /*sc1*/ 16);
        asm volatile("s_nop 1");
        __builtin_amdgcn_sched_barrier(0);
      }
      // every storing wave drains its own write-through stores, then one lane reports
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) (void)__hip_atomic_fetch_add(sp.cnt + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
  }
  f32x4 bias = f32x4{0.f, 0.f, 0.f, 0.f};
  if (e.bias) bias = f32x4{e.bias[col4], e.bias[col4 + 1], e.bias[col4 + 2], e.bias[col4 + 3]};
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(e.out, 0, BAD, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void *>(e.res ? e.res : e.in), 0, BAD, 0x00020000);
  const bool has_res = e.res != nullptr;
  const int out_ld = e.out_ld, res_ld = e.res_ld;
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
      const bool ok = (a < nrow) & (bb < ncol);
      rv[a] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                            rs_res, ok ? rbase : BAD, (a * W + bb) * res_ld * 4, 0));
    }
  };
  // reducer: column bb of the S - 1 OTHER slices' partials, in slice order (all loads unconditional:
  // a "register or load" choice per element would make hipcc branch around every load)
  // (S > 2: one column of partials at a time -- two sets of S - 1 do not fit beside s[][] in 256 registers)
  constexpr int SO = S > 1 ? S - 1 : 1, PB = S > 2 ? 1 : 2;
  f32x4 part[PB][SO][4];
  auto pload = [&](int bb, f32x4 (&pv)[SO][4]) {
#pragma unroll
    for (int k = 0; k < SO; ++k) {
      const int other = k + (k >= sp.slice ? 1 : 0);
      const int base = (other * sp.n_units + sp.unit) * unit_bytes;
#pragma unroll
      for (int a = 0; a < 4; ++a)
        pv[k][a] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                 rs_slab, slab_lane, base + (4 * bb + a) * px_bytes, /*sc1*/ 16));
    }
  };
  rload(0, resv[0]);
  transform_x();
  if constexpr (S > 1) {
    // the other slices' partials are complete when `done` reads S - 1 (each adds 1 behind its drained
    // write-through stores); relaxed polling by one lane, the sc1 loads below bypass this CU's L1
    if (lane == 0)
      while (__hip_atomic_load(sp.cnt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != S - 1)
        __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
    pload(0, part[0]);
  }
#pragma unroll
  for (int bb = 0; bb < 4; ++bb) {
    if (bb + 1 < 4) {
      rload(bb + 1, resv[(bb + 1) & 1]);
      if constexpr (S == 2) pload(bb + 1, part[(bb + 1) & 1]);
    }
    if constexpr (S > 2) {
      if (bb > 0) pload(bb, part[0]);
    }
    const f32x4 colv[6] = {s[0][bb], s[1][bb], s[2][bb], s[3][bb], s[4][bb], s[5][bb]};
    f32x4 y[4];
    wino4_at4(colv, y);                                           // A^T (M A)   (along y)
    if constexpr (S == 2) {
      // (two terms: the sum is the same whichever of them is the register)
#pragma unroll
      for (int a = 0; a < 4; ++a) y[a] = y[a] + part[bb & 1][0][a];
    } else if constexpr (S > 2) {
      // slice order 0 .. S-1, the own partial at position sp.slice (selects on loaded registers only)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        f32x4 sum;
#pragma unroll
        for (int j = 0; j < S; ++j) {
          const f32x4 below = part[0][j < SO ? j : SO - 1][a];            // slice j when j < sp.slice
          const f32x4 above = part[0][j > 0 ? j - 1 : 0][a];              // slice j when j > sp.slice
          const f32x4 t = j == sp.slice ? y[a] : (j < sp.slice ? below : above);
          sum = j == 0 ? t : sum + t;
        }
        y[a] = sum;
      }
    }
    f32x4 v[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      v[a] = (y[a] + bias) + resv[bb & 1][a];
      // ReLU without a branch per store: clamp from below at 0 or at -inf.  Compare + select, not v_max:
      // v_max_f32 returns the OTHER operand for a NaN input, which would turn a NaN activation into 0
      // (or -inf) and hide a numerical blow-up from the Winograd guard; NaN < lo is false, so it passes.
#pragma unroll
      for (int k = 0; k < 4; ++k) v[a][k] = v[a][k] < relu_lo ? relu_lo : v[a][k];
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
      const bool ok = (a < nrow) & (bb < ncol);
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
  if constexpr (S > 1) {
    // every other slice has left the counters (its `done` add was its last access): back to zero for
    // the next launch on them
    if (lane == 0) {
      __hip_atomic_store(sp.cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sp.cnt + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// a wave that holds all 36 positions of its (tile, channel) outputs (conv_wino4g.hip)
template <int S>
__device__ __forceinline__ void wino4_epilogue(const Wino4Epi &e, const Wino4Split &sp, int ticket,
                                               const f32x4 (&acc)[36], int tile, int col4, int g4,
                                               int lane) {
  wino4_epilogue_x<S>(
      e, sp, ticket,
      [&](f32x4 (&s)[6][4]) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const f32x4 m[6] = {acc[6 * i + 0], acc[6 * i + 1], acc[6 * i + 2],
                              acc[6 * i + 3], acc[6 * i + 4], acc[6 * i + 5]};
          wino4_at4(m, s[i]);                                     // M A   (along x)
        }
      },
      tile, col4, g4, lane);
}

}  // namespace shapy
