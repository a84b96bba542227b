// Winograd F(4x4,3x3), PERSISTENT and GROUPED: one launch runs up to four independent
// 3x3 / stride-1 / pad-1 convolutions -- the convs at the same depth of the parallel branches of a
// HighResolutionModule (regressor/human_shape/models/backbone/hrnet.py:175-193, the `for i in
// range(self.num_branches)` loop) -- on workgroups that stay resident and pull tasks.
//
// Why (profiles/r03c_timeline_multistream*.txt, r03b PMC): one launch per layer and branch gives
// the chip 15-20 us of matrix-core work at a time; every launch starts 512 workgroups in lockstep
// (all load their first patch, then all multiply, then all store), and a workgroup's prologue
// (first patch: one HBM latency + the input transform) and epilogue (output transform + stores)
// are as long as its multiply phase on the 48-channel branch.  Measured matrix-core busy time of
// conv_wino4_kernel: 23-38 % of the SIMD cycles; four branches on four streams only interleave
// those phases by accident (105-150 us per depth level against 67-93 us of MFMA time on the three
// SIMDs per CU that multiply).  Here
//   * a TASK is what a workgroup of conv_wino4.hip does: 16 tiles x 48 output channels x all of K
//     of ONE of the group's convolutions; the arithmetic, lane maps, LDS image and filter layout
//     are exactly conv_wino4.hip's, so results equal the per-layer launches' to float32 rounding (that kernel's fourth wave adds
//     Winograd rows 4 / 5 in another association since round 6);
//   * the grid is 2 workgroups per CU = 64 slots per XCD.  XCD x owns a contiguous run of every
//     convolution's n-major task list (a filter slab stays in ITS L2); inside an XCD the tasks are
//     dealt to the slots by a longest-processing-time schedule that the HOST computes per launch
//     (greedy: convolutions in order of decreasing K, every task to the least loaded slot) and
//     passes as a 1 KB table of (count, first index) per convolution and slot in the kernel
//     arguments.  No atomics: a ticket queue was built first and measured -- a device-scope
//     atomicAdd costs 1.3 us in steady state and 8 us in the start-up herd, 48 alone 80 vs 54 us
//     (run F of round 3); dealing the tasks round-robin instead left the slots that start with a
//     24-chunk task of the 7x7 branch with 36 chunk units against a mean of 21;
//   * the staging wave runs ONE TASK AHEAD: while the multiplying waves work on the last chunk of
//     task t and on its epilogue, the patch of task t+1's first chunk is loaded, transformed and
//     staged -- no prologue after the first task.  (The filter ring does drain between tasks:
//     see w4g_task.)
// Barrier protocol (s_barrier, all four waves): one opening barrier (first task id in the mailbox),
// then exactly one per chunk, in the same global chunk order on both sides: the staging wave arrives
// after writing chunk k, a multiplying wave before reading it.  The id of task t+1 is published in
// an LDS mailbox before the barrier of task t's LAST chunk and read right after it.
#include <type_traits>

#include "conv_common.h"
#include "conv_wino4.h"
#include "conv_wino4g_sched.h"

namespace shapy {

struct W4Conv {
  const void *in, *wgt2, *res;
  const float *bias;
  void *out;
  int Hi, Wi, Cin, Cout, in_ld, out_ld, out_coff, res_ld, res_coff, relu;
  int tiles;               // B * ceil(H/4) * ceil(W/4)
  int nbx, nby;            // Cout / 48, ceil(tiles / 16)
  unsigned in_bytes, wgt2_bytes;
};

struct W4Group {
  W4Conv c[4];
  int n;
  // schedule: slot s (= blockIdx.x / 8) of every XCD runs, for g = 0 .. n-1, the tasks
  // [first[g][s], first[g][s] + count[g][s]) of convolution g's per-XCD list (clipped to the
  // list's length on this XCD: lists differ by at most one task between XCDs)
  unsigned short count[4][64], first[4][64];
};

constexpr int W4G_BAD = 0x40000000;           // >= num_records of every buffer used here

// (task order, the packed task id and the host's schedule: conv_wino4g_sched.h)

// the B-fragment addressing of one task (multiplying waves)
struct W4Filt {
  const void *ptr;
  unsigned bytes;
  int u_lane, u_pos, u_chunk;
};

__device__ __forceinline__ W4Filt w4g_filters(const W4Group &G, int task, int wave, int l15, int g4) {
  const W4Conv &c = G.c[task >> 28];
  const int n0 = ((task >> 20) & 0xff) * 48 + 16 * wave;
  W4Filt f;
  f.ptr = c.wgt2;
  f.bytes = c.wgt2_bytes;
  f.u_lane = ((n0 + l15) * 16 + 4 * g4) * 4;
  f.u_chunk = c.Cout * 64;
  f.u_pos = (c.Cin >> 4) * f.u_chunk;
  return f;
}

// One task of a multiplying wave: KC > 0 = exactly KC chunks, chunk loop unrolled (hipcc's
// s_waitcnt bookkeeping is exact only in straight-line code, conv_wino4.hip); KC = 0 = generic loop
// with the last chunk peeled.
// Returns the next task id (read from the mailbox right after the barrier of the last chunk).
// The ring is NOT refilled with the next task's fragments during the last chunk: they would have
// to stay live across the epilogue, next to the 144 accumulators and the epilogue's 70-odd
// temporaries (147 spilled registers when tried) -- the caller requests them after the epilogue.
template <int KC>
__device__ __forceinline__ int w4g_task(const W4Group &G, const int task, int &gc, const int tk,
                                        const char *lds, const volatile int *mbox, const int wave,
                                        const int lane) {
  constexpr int PSTR = 1024, LDS_V = 36 * PSTR, R = WINO4_RING;
  const int g4 = lane >> 4, l15 = lane & 15;
  const int frag_off = l15 * 64 + (((g4 ^ l15 ^ (l15 >> 1)) & 3) << 4);
  const W4Conv &c = G.c[task >> 28];
  const int CC = c.Cin >> 4;
  const W4Filt fc = w4g_filters(G, task, wave, l15, g4);
  const __amdgpu_buffer_rsrc_t rs_c =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(fc.ptr), 0, fc.bytes, 0x00020000);
  // the ring lives and dies inside the task (as a loop-carried array of the caller hipcc kept its
  // stale slots alive across the epilogue: 80 spilled registers)
  u32x4 bring[R];
#pragma unroll
  for (int q = 0; q < R; ++q)
    bring[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_c, fc.u_lane, q * fc.u_pos, 0);

  f32x4 acc[36];
#pragma unroll
  for (int q = 0; q < 36; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  int nxt = -1;

  // LAST = last chunk of the task: no refill beyond position 35; the next task's id arrives with
  // this chunk's barrier
  auto chunk = [&](const int cc, auto last_tag) {
    constexpr bool LAST = decltype(last_tag)::value;
    wino4_lds_barrier();                     // chunk gc is staged
    if constexpr (LAST) nxt = __builtin_amdgcn_readfirstlane(mbox[(tk + 1) & 1]);
    const char *Vb = lds + (gc & 1) * LDS_V + frag_off;
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
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        acc[pp] = __builtin_amdgcn_mfma_f32_16x16x4f32(
            __uint_as_float(bring[pp % R][kk]), __uint_as_float(af[cur][0][kk]), acc[pp], 0, 0, 0);
        acc[pp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(
            __uint_as_float(bring[(pp + 1) % R][kk]), __uint_as_float(af[cur][1][kk]), acc[pp + 1],
            0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int q = pp + e + R;
        if (q < 36) {
          bring[(pp + e) % R] = __builtin_amdgcn_raw_buffer_load_b128(
              rs_c, fc.u_lane, q * fc.u_pos + cc * fc.u_chunk, 0);
        } else if constexpr (!LAST) {
          bring[(pp + e) % R] = __builtin_amdgcn_raw_buffer_load_b128(
              rs_c, fc.u_lane, (q - 36) * fc.u_pos + (cc + 1) * fc.u_chunk, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    ++gc;
  };
  if constexpr (KC > 0) {
#pragma unroll
    for (int cc = 0; cc < KC; ++cc) {
      if (cc + 1 < KC) chunk(cc, std::false_type{});
      else chunk(cc, std::true_type{});
    }
  } else {
    for (int cc = 0; cc + 1 < CC; ++cc) chunk(cc, std::false_type{});
    chunk(CC - 1, std::true_type{});
  }

  // ---- epilogue: output transform in registers, bias + residual + ReLU, store (conv_wino4.hip) ----
  // (the asm hides that the task id is the one known since before the chunks: hipcc otherwise
  // computes the 16 address registers below ahead of the K loop and spills them around it)
  int task_e = __builtin_amdgcn_readfirstlane(task);
  asm volatile("" : "+s"(task_e));
  Wino4Epi e;
  e.out = c.out; e.res = c.res; e.in = c.in; e.bias = c.bias;
  e.H = c.Hi; e.W = c.Wi; e.tiles = c.tiles; e.out_ld = c.out_ld; e.out_coff = c.out_coff;
  e.res_ld = c.res_ld; e.res_coff = c.res_coff; e.relu = c.relu;
  wino4_epilogue<1>(e, Wino4Split{}, 0, acc, (task_e & 0xfffff) * 16 + l15,
                    ((task_e >> 20) & 0xff) * 48 + 16 * wave + 4 * g4, g4, lane);
  return nxt;
}

__global__ __launch_bounds__(256, 2) void conv_wino4g_kernel(W4Group G) {
  constexpr int PSTR = 1024, LDS_V = 36 * PSTR;
  __shared__ __attribute__((aligned(16))) char lds[2 * LDS_V];
  __shared__ volatile int mbox[2];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int xcd = blockIdx.x & 7;

  if (wave == 3) {
    // =========================== staging wave ===========================
    const int tile_s = lane >> 2, c4 = lane & 3;
    const int st_off = tile_s * 64 + (((c4 ^ tile_s ^ (tile_s >> 1)) & 3) << 4);
    unsigned row_off[6], col_off[6];
    const void *in_ptr = nullptr;
    unsigned in_bytes = 0;
    int CCn = 0;
    // patch offsets of a task (conv_wino4.hip: out-of-image rows / columns and dead tiles add
    // 0x40000000, i.e. land beyond num_records and read zero)
    auto setup = [&](int task) {
      const W4Conv &c = G.c[task >> 28];
      in_ptr = c.in;
      in_bytes = c.in_bytes;
      CCn = c.Cin >> 4;
      const int H = c.Hi, W = c.Wi;
      const int TW = (W + 3) >> 2, TH = (H + 3) >> 2;
      const int pix_stride = c.in_ld * 4;
      const int tile = (task & 0xfffff) * 16 + tile_s;
      const bool live = tile < c.tiles;
      const int tt = live ? tile : 0;
      const int tx = tt % TW;
      const int tq = tt / TW;
      const int ty = tq % TH;
      const int b = tq / TH;
      const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const bool ok = live & ((unsigned)(y0 + i) < (unsigned)H);
        row_off[i] = ok ? (b * H + y0 + i) * W * pix_stride + c4 * 16 : W4G_BAD;
      }
#pragma unroll
      for (int j = 0; j < 6; ++j)
        col_off[j] = (unsigned)(x0 + j) < (unsigned)W ? (x0 + j) * pix_stride : W4G_BAD;
    };
    f32x4 raw[6][6];
    auto gload_col = [&](int j, int c0) {
      const __amdgpu_buffer_rsrc_t rs_in =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(in_ptr), 0, in_bytes, 0x00020000);
      unsigned co = col_off[j] + c0 * 4;
      asm volatile("" : "+v"(co));
#pragma unroll
      for (int i = 0; i < 6; ++i)
        raw[i][j] = __builtin_bit_cast(
            f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)(row_off[i] + co), 0, 0));
    };

    const int slot = blockIdx.x >> 3;
    int sg = 0, sk = 0;
    int cur = w4g_next_task(G, xcd, slot, sg, sk);
    if (lane == 0) mbox[0] = cur;
    wino4_lds_barrier();                       // opening barrier: the first task id is published
    int gc = 0, tk = 0;
    if (cur >= 0) {
      setup(cur);
#pragma unroll
      for (int j = 0; j < 6; ++j) gload_col(j, 0);
    }
    while (cur >= 0) {
      const int CC = CCn;
      const int nxt = w4g_next_task(G, xcd, slot, sg, sk);
      for (int cc = 0; cc < CC; ++cc) {
        bool more = cc + 1 < CC;
        int c0n = (cc + 1) * 16;
        if (!more) {
          // last chunk of this task: every patch load of it has been issued, so the offsets may
          // turn to the next task, whose first chunk is requested column by column below
          if (lane == 0) mbox[(tk + 1) & 1] = nxt;
          if (nxt >= 0) {
            setup(nxt);
            more = true;
            c0n = 0;
          }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {                                // T = d B  (along x)
          f32x4 o[6];
          wino4_bt(raw[i], o);
#pragma unroll
          for (int j = 0; j < 6; ++j) raw[i][j] = o[j];
          __builtin_amdgcn_sched_barrier(0);
        }
        char *Vb = lds + (gc & 1) * LDS_V + st_off;
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
          __builtin_amdgcn_sched_barrier(0);
          if (more) gload_col(j, c0n);
          __builtin_amdgcn_sched_barrier(0);
        }
        wino4_lds_barrier();                   // chunk gc is staged
        ++gc;
      }
      cur = nxt;
      ++tk;
    }
    return;
  }

  // =========================== multiplying waves ===========================
  wino4_lds_barrier();                         // opening barrier
  int cur = __builtin_amdgcn_readfirstlane(mbox[0]);
  if (cur < 0) return;
  int gc = 0, tk = 0;
  while (cur >= 0) {
    const int CC = G.c[cur >> 28].Cin >> 4;
    if (CC == 3)
      cur = w4g_task<3>(G, cur, gc, tk, lds, mbox, wave, lane);
    else if (CC == 6)
      cur = w4g_task<6>(G, cur, gc, tk, lds, mbox, wave, lane);
    else
      cur = w4g_task<0>(G, cur, gc, tk, lds, mbox, wave, lane);
    ++tk;
  }
}

// up to 4 convolutions, each one eligible for conv2d_wino4 (conv_wino4_fits, Cin % 16 == 0)
int conv2d_wino4_group(const ConvK *ks, int n, hipStream_t s) {
  if (n < 1 || n > 4) return SHAPY_EINVAL;
  W4Group G;
  G.n = n;
  long tasks = 0;
  for (int i = 0; i < n; ++i) {
    const ConvK &k = ks[i];
    if (!k.wgt2 || !conv_wino_eligible(k) || !conv_wino4_fits(k) || k.Cout % 48 || k.Cout / 48 > 255)
      return SHAPY_EINVAL;                       // (the persistent kernel has the 48-channel tiling only)
    W4Conv &c = G.c[i];
    c.in = k.in; c.wgt2 = k.wgt2; c.res = k.res; c.bias = k.bias; c.out = k.out;
    c.Hi = k.Hi; c.Wi = k.Wi; c.Cin = k.Cin; c.Cout = k.Cout; c.in_ld = k.in_ld;
    c.out_ld = k.out_ld; c.out_coff = k.out_coff; c.res_ld = k.res_ld; c.res_coff = k.res_coff;
    c.relu = k.relu;
    const int B = k.M / (k.Ho * k.Wo);
    c.tiles = B * ((k.Hi + 3) / 4) * ((k.Wi + 3) / 4);
    c.nbx = k.Cout / 48;
    c.nby = (c.tiles + 15) / 16;
    if (c.nby >= (1 << 20)) return SHAPY_EINVAL;
    c.in_bytes = k.in_bytes;
    c.wgt2_bytes = (unsigned)(144ull * k.Cin * k.Cout);
    tasks += (long)c.nbx * c.nby;
  }
  // longest K loop first (stable: equal lengths keep the caller's order)
  for (int i = 1; i < n; ++i)
    for (int j = i; j > 0 && G.c[j].Cin > G.c[j - 1].Cin; --j) {
      const W4Conv tmp = G.c[j];
      G.c[j] = G.c[j - 1];
      G.c[j - 1] = tmp;
    }
  // compute units of the CURRENT device (cached per device: ranks of one process may drive
  // different GPUs)
  static int n_cu_dev[64] = {};
  int dev = 0;
  SHAPY_HIP_TRY(hipGetDevice(&dev));
  int n_cu = (dev >= 0 && dev < 64) ? __atomic_load_n(&n_cu_dev[dev], __ATOMIC_RELAXED) : 0;
  if (!n_cu) {
    SHAPY_HIP_TRY(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    if (n_cu <= 0) n_cu = 256;
    if (dev >= 0 && dev < 64) __atomic_store_n(&n_cu_dev[dev], n_cu, __ATOMIC_RELAXED);
  }
  // two resident workgroups per CU (73.7 KB LDS, 256 VGPRs); slots per XCD: at most 64, and not
  // more than the longest per-XCD list needs
  int slots = 2 * n_cu / 8;
  if (slots > 64) slots = 64;
  if (slots < 1) slots = 1;
  const long per_xcd = (tasks + 7) / 8;
  if (slots > per_xcd) slots = (int)per_xcd;
  if (!w4g_make_schedule(G, slots)) return SHAPY_EINVAL;       // conv_wino4g_sched.h
  hipLaunchKernelGGL(conv_wino4g_kernel, dim3((unsigned)(8 * slots)), dim3(256), 0, s, G);
  return (int)hipGetLastError();
}

}  // namespace shapy
