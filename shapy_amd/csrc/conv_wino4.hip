// Winograd F(4x4, 3x3) convolution on the f32 matrix cores (v_mfma_f32_16x16x4_f32).
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
// One 256-thread workgroup = 16 consecutive tiles (4x4 output pixels each, row-major over
// (b, ty, tx)) x N = 48 (three multiplying waves) or 64 (four) output channels:
//   * staging, ONE wave (wave 3): lane (tile, 4 channels) loads the 6 x 6 patch of its tile as 36
//     16-byte loads (out-of-image taps zeroed by the buffer bounds check), applies B^T d B entirely
//     in registers -- no cross-lane exchange -- and writes V[p][tile][16 ch] to LDS (36 KB per
//     16-channel chunk; 16-byte slots XOR-swizzled by the tile so that the fragment reads are
//     conflict-free).  Two V buffers, one barrier per chunk; the next chunk's patch is in flight
//     in registers while this one is multiplied.
//   * multiply, waves 0..2: ALL 36 positions of ONE 16-channel tile (144 accumulator registers).
//     V fragments from LDS (one ds_read_b128 = the k-operands of 4 MFMAs), filter fragments straight
//     from the transformed filters in global memory, layout [p][Cin/16][Cout][16] (1 KB contiguous
//     per wave load), a ring of WINO4_RING positions ahead.  The FILTER fragment is the MFMA's A
//     operand and the V fragment its B operand, so the accumulators hold D[channel][tile]: a lane
//     owns one tile and four consecutive output channels.  Because a wave owns every position of
//     its (tile, channel) outputs, the output transform A^T M A happens in registers: no
//     accumulator exchange through LDS (conv_wino.hip spends 53 KB of LDS and two barriers on it).
//   * epilogue (conv_wino4.h): per lane 16 pixels x 4 channels: bias, residual, ReLU, 16-byte
//     stores (4 lanes = the wave's 64 contiguous bytes of a pixel; the 3 waves of the workgroup
//     cover the pixel's whole 192-byte row).
// The fourth wave only stages: the register budget (256 per lane) allows two waves per SIMD either
// way, so the slot costs no occupancy.
#include "conv_common.h"
#include "conv_wino4.h"

namespace shapy {

// 256 threads = three multiplying waves (16 output channels each, N = 48 per workgroup) + one
// staging wave.  KC > 0: the layer has exactly KC chunks (Cin = 16 KC) and the multiplying waves'
// chunk loop is unrolled: hipcc's s_waitcnt bookkeeping is exact only in straight-line code -- at
// the header of a real loop it drains vmcnt(0), i.e. waits for the whole filter ring once per
// chunk (an L2 latency of idle matrix cores every 144 MFMAs).
// NW = multiplying waves: 3 (N = 48 output channels per workgroup, 256 threads, two workgroups per CU)
// or 4 (N = 64, 320 threads: layers whose Cout is a multiple of 64 but not of 48 -- layer1's 64 -> 64
// and the head's 512 -> 512; five waves of 256 registers leave room for ONE workgroup per CU, all four
// SIMDs multiply and the staging wave shares one of them).
// S > 1: split-K.  The layers with few output tiles and a deep K loop -- 384 -> 384 on the 7x7 maps: 128
// workgroups of 24 chunks at B = 64, half the CUs idle and every conv of the branch a 60 us serial chain
// (113 us under the contention of a stage-4 module, where this lane ends 300-400 us after the others:
// profiles/r04o_module_tails.txt) -- run S workgroups per output tile, each over Cin / S input channels;
// the wave units exchange their transformed partials through a slab and the last to arrive finishes
// (conv_wino4.h: Wino4Split).  KC = chunks of ONE slice.
template <int KC, int NW = 3, int S = 1>
__global__ __launch_bounds__(64 * (NW + 1), 2) void conv_wino4_kernel(ConvK p) {
  constexpr int N = 16 * NW;
  constexpr int PSTR = 1024;                          // bytes per position: 16 tiles x 16 ch f32
  constexpr int LDS_V = 36 * PSTR;
  constexpr int R = WINO4_RING;
  constexpr int BAD = 0x40000000;                     // >= num_records of every buffer used here
  static_assert(36 % R == 0 && R >= 2, "position q lives in ring slot q % R in every chunk");
  __shared__ __attribute__((aligned(16))) char lds[2 * LDS_V];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);        // wave-uniform: scalar branches
  // workgroup -> (m tile, K slice, n tile): the S slices of an output tile are neighbours in the m-major
  // order of conv_tile_index (same XCD, dispatched together)
  const int wg = conv_tile_index(p);
  const int mv = wg / p.nbx, n_i = wg % p.nbx;
  const int m_i = mv / S, slice = mv % S;
  const int m_blk = m_i * 16, n_blk = n_i * N;
  const int H = p.Hi, W = p.Wi;
  const int TW = (W + 3) >> 2, TH = (H + 3) >> 2;
  const int T = p.wino_tiles;
  const int CC = (p.Cin >> 4) / S;                    // chunks of this workgroup's slice ...
  const int cbase = slice * CC;                       // ... which starts at chunk cbase of the layer

  if (wave == NW) {
    // =========================== staging wave ===========================
    // lane (tile, c4): the 6 x 6 patch of one tile for 4 channels of the current 16-channel
    // chunk: 36 buffer_load_dwordx4 (144 registers -- this wave holds no accumulators).  Zero
    // padding by the buffer bounds check: an out-of-image row / column adds 0x40000000 to the
    // byte offset, so any offset with an invalid part lies in [1 GiB, 4 GiB) >= num_records
    // (the launcher keeps in_bytes <= 1 GiB) -- no select, no branch.
    const __amdgpu_buffer_rsrc_t rs_in =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.in), 0, p.in_bytes, 0x00020000);
    const int tile_s = lane >> 2, c4 = lane & 3;
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
        row_off[i] = ok ? (b * H + y0 + i) * W * pix_stride + c4 * 16 : BAD;
      }
#pragma unroll
      for (int j = 0; j < 6; ++j)
        col_off[j] = (unsigned)(x0 + j) < (unsigned)W ? (x0 + j) * pix_stride : BAD;
    }
    // LDS image V[p][tile][16 ch]: 16-byte slot c4 of row `tile` sits at slot c4 ^ f(tile),
    // f(r) = (r ^ r >> 1) & 3 -- the layout of conv_igemm.hip's staging buffer (conflict-free
    // ds_write_b128 and ds_read_b128, tools/lds_swizzle_check.py)
    const int st_off = tile_s * 64 + (((c4 ^ tile_s ^ (tile_s >> 1)) & 3) << 4);

    f32x4 raw[6][6];
    auto gload_col = [&](int j, int c0) {              // column j of the patch, channels c0 ..
      // (the asm keeps the chunk offset inside the sum: hipcc otherwise hoists the 36
      // loop-invariant row + column sums out of the K loop and spills them to scratch)
      unsigned co = col_off[j] + c0 * 4;
      asm volatile("" : "+v"(co));
#pragma unroll
      for (int i = 0; i < 6; ++i)
        raw[i][j] = __builtin_bit_cast(
            f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)(row_off[i] + co), 0, 0));
    };
#pragma unroll
    for (int j = 0; j < 6; ++j) gload_col(j, cbase * 16);
    for (int cc = 0; cc < CC; ++cc) {
      // chunk cc -> V buffer cc & 1.  That buffer was last read by the multiply of chunk
      // cc - 2, which every wave left through the barrier this wave passed at the end of the
      // previous iteration.
      const bool more = cc + 1 < CC;
#pragma unroll
      for (int i = 0; i < 6; ++i) {                                // T = d B  (along x)
        f32x4 o[6];
        wino4_bt(raw[i], o);
#pragma unroll
        for (int j = 0; j < 6; ++j) raw[i][j] = o[j];
        __builtin_amdgcn_sched_barrier(0);     // (row by row: bounds the live temporaries)
      }
      char *Vb = lds + (cc & 1) * LDS_V + st_off;
#pragma unroll
      for (int j = 0; j < 6; ++j) {                                // V = B^T T  (along y)
        // (each output goes to LDS as soon as it exists: with all six held at once the staging
        // loop needed 10 registers more than the 256 a wave has, and ANY scratch costs ~17 us
        // per LAUNCH on this GPU -- tools/launch_floor.hip: 3.1 us for an empty kernel, 19-22 us
        // for one that touches its private segment)
        {
          auto st = [&](int i, const f32x4 v) { *reinterpret_cast<f32x4 *>(Vb + (6 * i + j) * PSTR) = v; };
          const f32x4 d0 = raw[0][j], d1 = raw[1][j], d2 = raw[2][j], d3 = raw[3][j], d4 = raw[4][j],
                      d5 = raw[5][j];
          st(0, 4.f * d0 - 5.f * d2 + d4);
          __builtin_amdgcn_sched_barrier(0);
          st(5, 4.f * d1 - 5.f * d3 + d5);
          __builtin_amdgcn_sched_barrier(0);
          const f32x4 a = d4 - 4.f * d2, b = d3 - 4.f * d1;
          st(1, a + b);
          st(2, a - b);
          __builtin_amdgcn_sched_barrier(0);
          const f32x4 c = d4 - d2, e = d3 - d1;
          st(3, c + 2.f * e);
          st(4, c - 2.f * e);
        }
        // the column's registers are free again: request the same column of the NEXT chunk, which
        // then has a whole multiply phase to arrive
        __builtin_amdgcn_sched_barrier(0);
        if (more) gload_col(j, (cbase + cc + 1) * 16);
        __builtin_amdgcn_sched_barrier(0);
      }
      wino4_lds_barrier();                   // chunk cc is staged (barrier #cc of CC + 1)
    }
    wino4_lds_barrier();                     // pairs with the multiplying waves' last barrier
    return;
  }

  // =========================== multiplying waves ===========================
  // wave w owns output channels n_blk + 16 w .. + 15 for ALL 36 positions
  const __amdgpu_buffer_rsrc_t rs_u =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.wgt2), 0, p.wgt2_bytes, 0x00020000);
  const int g = lane >> 4, l15 = lane & 15;
  const int frag_off = l15 * 64 + (((g ^ l15 ^ (l15 >> 1)) & 3) << 4);
  const int n0 = n_blk + 16 * wave;
  const int u_lane = ((n0 + l15) * 16 + 4 * g) * 4;
  const int u_pos = (p.Cin >> 4) * p.Cout * 64, u_chunk = p.Cout * 64;

  // B-fragment loads: the per-lane part of the address is ONE register (u_lane); position and
  // chunk go into the scalar offset of the buffer instruction (as a vector offset hipcc keeps 36
  // strength-reduced address registers alive across the K loop)
  u32x4 bring[R];
  auto bload = [&](int slot, int pos, int cc, bool live) {
    bring[slot] = __builtin_amdgcn_raw_buffer_load_b128(rs_u, live ? u_lane : BAD,
                                                        pos * u_pos + (cbase + cc) * u_chunk, 0);
  };

  f32x4 acc[36];
#pragma unroll
  for (int q = 0; q < 36; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < R; ++q) bload(q, q, 0, true);
  // split-K: this wave's reduction unit and its ticket (drawn in the last chunk)
  Wino4Split sp;
  sp.slab = p.split_ws; sp.slab_bytes = p.split_bytes; sp.slice = slice;
  sp.unit = n_i * NW + wave; sp.n_units = p.nbx * NW;
  sp.cnt = p.split_cnt + 2 * ((m_i * p.nbx + n_i) * NW + wave);
  int ticket_v = 0;

  auto chunk = [&](int cc, bool more) {
    wino4_lds_barrier();                     // chunk cc is staged
    if constexpr (S > 1) {
      if (!more && lane == 0)
        ticket_v = __hip_atomic_fetch_add(sp.cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const char *Vb = lds + (cc & 1) * LDS_V + frag_off;
    u32x4 af[2][2];
    af[0][0] = *reinterpret_cast<const u32x4 *>(Vb + 0 * PSTR);
    af[0][1] = *reinterpret_cast<const u32x4 *>(Vb + 1 * PSTR);
#pragma unroll
    for (int pp = 0; pp < 36; pp += 2) {
      const int cur = (pp >> 1) & 1;
      if (pp + 2 < 36) {
        af[cur ^ 1][0] = *reinterpret_cast<const u32x4 *>(Vb + (pp + 2) * PSTR);
        af[cur ^ 1][1] = *reinterpret_cast<const u32x4 *>(Vb + (pp + 3) * PSTR);
      }
      // two positions interleaved: consecutive MFMAs hit different accumulators (40-cycle
      // dependent latency vs a 32-cycle issue interval)
      // A operand = filter fragment, B operand = V fragment: D[channel 4 g + r][tile l15]
      // (conv_wino4.h: the epilogue wants four consecutive channels of one tile per lane).
      // The two filter refills sit BETWEEN the MFMAs of the pair, each right behind an MFMA that has
      // just been issued, so that the VMEM issue overlaps that MFMA's 32 cycles (behind the last MFMA
      // of the pair they cost 5-8 % on the 12- / 24-chunk classes: profiles/r04v_*).  They refill
      // the slots the PREVIOUS pair consumed (lead R - 2 positions); pinned, or the compiler sinks
      // them to the end of the chunk.
      auto refill_prev = [&](int e) {
        if (pp >= 2) {
          const int q = pp - 2 + e + R;
          if (q < 36) bload((pp - 2 + e) % R, q, cc, true);
          else bload((pp - 2 + e) % R, q - 36, cc + 1, more);
        } else if (cc > 0) {
          // positions 34 / 35 of the previous chunk: the same slots for THIS chunk's positions.  (Not in
          // the first chunk: the preloaded ring still holds its positions R - 2, R - 1 there -- even
          // a dropped load would zero them.  cc is a constant in the unrolled instantiations.)
          bload((34 + e) % R, 34 + e + R - 36, cc, true);
        }
      };
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        acc[pp] = __builtin_amdgcn_mfma_f32_16x16x4f32(
            __uint_as_float(bring[pp % R][kk]), __uint_as_float(af[cur][0][kk]), acc[pp], 0, 0, 0);
        if (kk == 0 || kk == 2) {
          __builtin_amdgcn_sched_barrier(0);
          refill_prev(kk >> 1);
          __builtin_amdgcn_sched_barrier(0);
        }
        acc[pp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(
            __uint_as_float(bring[(pp + 1) % R][kk]), __uint_as_float(af[cur][1][kk]), acc[pp + 1],
            0, 0, 0);
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
  wino4_lds_barrier();                       // barrier #CC: the staging wave's closing one

  // ---- epilogue: output transform in registers, (split-K exchange,) bias + residual + ReLU, 16-byte
  // stores (conv_wino4.h; per lane: tile m_blk + l15, channels n0 + 4 g .. + 3)
  Wino4Epi e;
  e.out = p.out; e.res = p.res; e.in = p.in; e.bias = p.bias;
  e.H = H; e.W = W; e.tiles = T; e.out_ld = p.out_ld; e.out_coff = p.out_coff;
  e.res_ld = p.res_ld; e.res_coff = p.res_coff; e.relu = p.relu;
  wino4_epilogue<S>(e, sp, __builtin_amdgcn_readfirstlane(ticket_v), acc, m_blk + l15, n0 + 4 * g, g, lane);
}

// The kernel marks invalid accesses with the byte offset 0x40000000: every tensor it touches has
// to stay below 1 GiB (B <= 334 for HRNet's 256-channel 56x56 map).
bool conv_wino4_fits(const ConvK &k) {
  const unsigned long long lim = 0x40000000ull;
  // 16-byte residual loads / stores of the epilogue: k.vec4 (conv_prepare) = rows and channel
  // offsets in multiples of 4 floats, 16-byte-aligned tensors -- every HRNet tensor; anything else
  // takes another kernel
  const bool al = k.vec4 != 0;
  return al && (k.Cout % 48 == 0 || k.Cout % 64 == 0) && k.in_bytes <= lim && 4ull * k.M * k.out_ld <= lim &&
         (!k.res || 4ull * k.M * k.res_ld <= lim) && 144ull * k.Cin * k.Cout < 0x7fffffffull;
}

// tile flag 0x100000 of ShapyConv.tile: wgt_wino holds F(4x4,3x3) filters [36][Cin/16][Cout][16]
// k.ksplit = S (SHAPY_TILE_W4_KSPLIT): K slices per output tile, 1 = none.  The instantiations below are
// the ones a HRNet plan asks for (chunks per slice unrolled where the class is hot); any other S / shape
// combination is SHAPY_EINVAL -- the plan, not the kernel, decides where splitting pays.
int conv2d_wino4(ConvK k, hipStream_t s) {
  if (!conv_wino4_fits(k)) return SHAPY_EINVAL;
  const int B = k.M / (k.Ho * k.Wo);
  k.wino_tiles = B * ((k.Hi + 3) / 4) * ((k.Wi + 3) / 4);
  k.wgt2_bytes = (unsigned)(144ull * k.Cin * k.Cout);          // 36 positions x f32
  // 64-channel N tile (four multiplying waves, one workgroup per CU): layers whose Cout is no multiple of
  // 48 (conv_wino4_fits: then Cout % 64 == 0)
  const bool n64 = k.Cout % 48 != 0;
  const int NW = n64 ? 4 : 3, S = k.ksplit < 1 ? 1 : k.ksplit;
  k.nbx = k.Cout / (16 * NW);
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
  if (!n64 && !k.w4_legacy) return conv2d_wino4q_launch(k, S, s);     // four multiplying waves
  const dim3 grid(k.nbx * k.nby * S), blk(64 * (NW + 1));
  const int cps = chunks / S;                                    // chunks per slice
#define W4_LAUNCH(KC, NWV, SV) hipLaunchKernelGGL((conv_wino4_kernel<KC, NWV, SV>), grid, blk, 0, s, k)
  if (n64) {
    if (S == 1 && cps == 4) W4_LAUNCH(4, 4, 1);
    else if (S == 1) W4_LAUNCH(0, 4, 1);
    else if (S == 2) W4_LAUNCH(0, 4, 2);
    else if (S == 4) W4_LAUNCH(0, 4, 4);
    else return SHAPY_EINVAL;
  } else if (S == 1) {
    if (cps == 3) W4_LAUNCH(3, 3, 1);
    else if (cps == 6) W4_LAUNCH(6, 3, 1);
    else W4_LAUNCH(0, 3, 1);
  } else if (S == 2) {
    if (cps == 12) W4_LAUNCH(12, 3, 2);
    else if (cps == 6) W4_LAUNCH(6, 3, 2);
    else if (cps == 3) W4_LAUNCH(3, 3, 2);
    else W4_LAUNCH(0, 3, 2);
  } else if (S == 3 && cps == 8) {
    W4_LAUNCH(8, 3, 3);
  } else if (S == 4 && cps == 6) {
    W4_LAUNCH(6, 3, 4);
  } else if (S == 4 && cps == 3) {
    W4_LAUNCH(3, 3, 4);
  } else {
    return SHAPY_EINVAL;
  }
#undef W4_LAUNCH
  return (int)hipGetLastError();
}

}  // namespace shapy
