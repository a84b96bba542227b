// Implicit-GEMM convolution on the CDNA4 matrix cores: float32 (v_mfma_f32_16x16x4_f32, exact
// f32) and bfloat16 storage with f32 accumulation (v_mfma_f32_16x16x32_bf16).
//
// Replaces every cuDNN conv + eval BatchNorm + ReLU + residual add + nearest upsample of the
// reference's HighResolutionNet.forward (regressor/human_shape/models/backbone/hrnet.py:426-498)
// and doubles as the GEMM of the SMPL-X blend shapes (lbs.py:171-182, 218-239).
//
// GEMM view: M = B*Ho*Wo output pixels, N = Cout, K = ks*ks*Cin, NHWC activations,
// OHWI weights (K contiguous for both operands).  One 256-thread workgroup (4 waves) owns a
// BM x BN tile; K is consumed in chunks of KQ 16-byte slots per row staged through one LDS
// buffer: buffer_load_dwordx4 into registers (issued before the MFMAs of the current chunk;
// out-of-image taps are zeroed by the hardware bounds check) -> MFMAs on the chunk in LDS ->
// barrier -> ds_write_b128 of the next chunk -> barrier -> ds_read_b128 fragment reads.
//   f32 : a 16-byte slot = 4 consecutive k.  Lane l reads slot (l >> 4) of row (l & 15) and
//         feeds its 4 floats to 4 successive 16x16x4 MFMAs; A and B use the same k permutation,
//         so the sum over k is unchanged and the result is the exact-f32 fmaf chain.
//   bf16: a slot = 8 consecutive k = exactly the per-lane operand of one 16x16x32 MFMA.
// LDS rows are unpadded; slot s of row r holds k-group s ^ f(r), f(r) = (r ^ (r >> 1)) & (KQ-1),
// which makes both the staging writes and the fragment reads bank-conflict free
// (tools/lds_swizzle_check.py checks the gfx950 lane-group table exhaustively).
#include <stdlib.h>

#include <stdio.h>

#include "conv_common.h"

namespace shapy {

// PD = chunks of global loads in flight per thread (register sets).  PD = 1: the next chunk is
// requested before the MFMAs of the current one, i.e. a chunk may take no less than a load latency
// (~1 us); with bf16 operands a chunk is 3-6 MFMAs per wave (50-100 ns), so the K loop of the
// small-grid layers is one serial chain of load latencies.  PD = 3 keeps three chunks in flight.
// FLAT: K = ks*ks*Cin is consumed as ONE flat index (chunks may straddle filter taps; every 16-byte
// slot still lies inside one tap because Cin is a multiple of the slot width).  Lets bf16 run the
// 48-channel branch without padding its tensors to 64 channels (Cin % 32 != 0).
// SPLIT: split-K.  The K loop of a tile is cut into p.ksplit slices on as many workgroups (neighbours in
// dispatch order); every WAVE is a reduction unit of its own (its TM x TN accumulator tiles): slices exchange
// partial accumulators through a slab (write-through stores, one contiguous 1 KB run per tile and wave) and the
// last wave unit to arrive adds them IN SLICE ORDER and runs the epilogue -- the protocol of the F(4x4) kernel
// (conv_wino4.h: Wino4Split), for the K-deep layers on small maps that stay on this kernel: every 3x3 conv in
// bf16 storage, the head's 1x1 GEMMs and the stride-2 fuse convs at small batches.
// Unless FLAT: the address arithmetic of a staged row -- tap offset, two image-border compares, a select -- is
// done once per filter TAP, not per chunk: per-lane offsets of the current tap in registers, the
// channel chunk in the scalar offset of the buffer load (a 1x1 layer computes them once).  On gfx950 a VALU
// instruction between the f32 MFMAs of a wave is not hidden (tools/mfma_fillers.hip: they share the FMA lanes);
// the per-chunk form spent ~9 of them per 16-MFMA chunk on the 32 x 64 tile: the head's 2048 -> 2048 GEMM 238 ->
// 212 us (124 TFLOP/s), end to end +1.1 % (profiles/r06j_plain_gemm_ab.txt).
template <typename T, int BM, int BN, int WM, int WN, int UPS, int KQ, int PD = 1, bool FLAT = false,
          bool SPLIT = false>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvK p) {
  static_assert(!SPLIT || (UPS == 1 && !FLAT), "split-K: plain epilogue, per-tap K chunks");
  constexpr bool TAPH = !FLAT;                 // per-tap address arithmetic (flat K: a chunk's slots may straddle taps)
  static_assert(WM * WN == 4, "4 waves per workgroup");
  static_assert(KQ == 4 || KQ == 8, "16-byte slots per staged row");
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int ESZ = sizeof(typename T::elem);
  constexpr int BK = KQ * T::EPS;            // K elements per chunk
  constexpr int ROWB = KQ * 16;              // bytes per staged row
  constexpr int RPP = 256 / KQ;              // rows staged per pass of the 256 threads
  constexpr int AR = BM / RPP;               // A rows staged per thread
  constexpr int BR = (BN + RPP - 1) / RPP;   // B rows staged per thread (guarded)
  // ONE staging buffer, two barriers per chunk: the barriers cost nothing measurable, the
  // LDS footprint does (8 workgroups per CU instead of 5 with 32-float chunks: +4.9 % end to end)
  constexpr bool VEC = true;                   // vector epilogue (16-byte rows) available
  constexpr int LDS_STAGE = (BM + BN) * ROWB, LDS_EPI = VEC ? conv_epilogue_vec_bytes(TM, TN) : 0;
  __shared__ __attribute__((aligned(16))) char lds[LDS_STAGE > LDS_EPI ? LDS_STAGE : LDS_EPI];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  // upsample-scatter layers run p.ups_split workgroups per tile (neighbours in dispatch order): each
  // computes the tile's (tiny) GEMM and scatters its share of the UPS rows of every value's block
  int wg = conv_tile_index(p);
  int dy0 = 0, dy1 = UPS;
  if constexpr (UPS > 1) {
    const int G = p.ups_split;
    const int part = wg % G;
    wg /= G;
    dy0 = part * (UPS / G);
    dy1 = dy0 + UPS / G;
  }
  int slice = 0;
  if constexpr (SPLIT) {
    slice = wg % p.ksplit;
    wg /= p.ksplit;
  }
  const int m_blk = (wg / p.nbx) * BM, n_blk = (wg % p.nbx) * BN;
  const int kq = t % KQ, lrow = t / KQ;

  // ---- per-thread staging addresses (byte offsets into buffer descriptors) ----
  // Out-of-image taps and rows beyond M / Cout get an offset past num_records: the buffer
  // load returns 0 for them, so zero padding costs one v_cndmask per load and no data select.
  const __amdgpu_buffer_rsrc_t rs_in =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.wgt), 0, p.wgt_bytes, 0x00020000);
  constexpr int OOB = 0x7fffffff;
  int a_off[AR], a_h[AR], a_w[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int m = m_blk + lrow + RPP * i;
    const int mm = m < p.M ? m : 0;
    const int wo = mm % p.Wo;
    const int tq = mm / p.Wo;
    const int ho = tq % p.Ho;
    const int b = tq / p.Ho;
    const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
    a_off[i] = ((b * p.Hi + hi0) * p.Wi + wi0) * p.in_ld * ESZ + (FLAT ? 0 : kq * 16);
    a_h[i] = m < p.M ? hi0 : -0x40000000;
    a_w[i] = wi0;
  }
  const int Kw = p.ks * p.ks * p.Cin;
  int b_off[BR];
#pragma unroll
  for (int i = 0; i < BR; ++i) {
    const int r = lrow + RPP * i;
    const int n = n_blk + r;
    b_off[i] = ((r < BN) && (n < p.Cout)) ? n * Kw * ESZ + (FLAT ? 0 : kq * 16) : OOB;
  }

  u32x4 a_reg[PD][AR], b_reg[PD][BR];

  // chunk iterator state for the NEXT chunk to be fetched (FLAT: per thread -- the slots of one
  // chunk may belong to different taps; c0 / kflat = channel / flat k of this thread's slot)
  int kh = 0, kw = 0, c0 = FLAT ? kq * T::EPS : 0, kflat = kq * T::EPS;
  int n_chunks = FLAT ? (Kw + BK - 1) / BK : p.ks * p.ks * (p.Cin / BK);
  if constexpr (SPLIT) {
    // this slice's chunks [c_begin, c_begin + n_chunks) of the layer's (the launcher guarantees >= 1 each)
    const int cps = (n_chunks + p.ksplit - 1) / p.ksplit, c_begin = slice * cps;
    n_chunks = n_chunks - c_begin < cps ? n_chunks - c_begin : cps;
    const int cpt = p.Cin / BK, tap = c_begin / cpt;
    c0 = (c_begin % cpt) * BK;
    kh = tap / p.ks;
    kw = tap % p.ks;
  }
  int left = n_chunks;                         // chunks still to be fetched: later requests read zeros

  // TAPH: byte offsets of this thread's staged rows for the CURRENT tap (out-of-image / beyond M or Cout: OOB)
  int a_vo[AR], b_vo[BR];
  auto set_tap = [&]() {
    const int tap_in = (kh * p.Wi + kw) * p.in_ld * ESZ, tap_w = (kh * p.ks + kw) * p.Cin * ESZ;
    const bool tap_ok = kh < p.ks;               // (PD > 1 runs past the last tap)
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const bool ok = (unsigned)(a_h[i] + kh) < (unsigned)p.Hi && (unsigned)(a_w[i] + kw) < (unsigned)p.Wi;
      a_vo[i] = (ok && tap_ok) ? a_off[i] + tap_in : OOB;
    }
#pragma unroll
    for (int i = 0; i < BR; ++i) b_vo[i] = (b_off[i] == OOB || !tap_ok) ? OOB : b_off[i] + tap_w;
  };
  if constexpr (TAPH) set_tap();
  auto gload = [&](int set) {
    if constexpr (TAPH) {
      const int so = c0 * ESZ;
      // PD > 1 keeps requesting chunks past the last one (no branch around a load: exact vmcnt counts); those
      // read zeros: beyond the last tap set_tap() has marked every row out of range, inside a split-K slice's
      // neighbour the uniform `left` decides
      const bool live = PD == 1 || left > 0;
#pragma unroll
      for (int i = 0; i < AR; ++i)
        a_reg[set][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, live ? a_vo[i] : OOB, so, 0);
#pragma unroll
      for (int i = 0; i < BR; ++i)
        b_reg[set][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, live ? b_vo[i] : OOB, so, 0);
      c0 += BK;
      --left;
      if (c0 == p.Cin) {                         // next tap (uniform branch)
        c0 = 0;
        if (++kw == p.ks) { kw = 0; ++kh; }
        set_tap();
      }
      return;
    }
    const int tap_in = ((kh * p.Wi + kw) * p.in_ld + c0) * ESZ;
    const int tap_w = FLAT ? kflat * ESZ : ((kh * p.ks + kw) * p.Cin + c0) * ESZ;
    // (kh < ks: the flat-K kernel's LAST chunk may reach past the K extent in some of its slots)
    const bool live = left > 0 && kh < p.ks;
    --left;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const bool ok = (unsigned)(a_h[i] + kh) < (unsigned)p.Hi &&
                      (unsigned)(a_w[i] + kw) < (unsigned)p.Wi;
      a_reg[set][i] = __builtin_amdgcn_raw_buffer_load_b128(
          rs_in, (ok && live) ? a_off[i] + tap_in : OOB, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < BR; ++i)
      b_reg[set][i] = __builtin_amdgcn_raw_buffer_load_b128(
          rs_w, (b_off[i] == OOB || !live) ? OOB : b_off[i] + tap_w, 0, 0);
    c0 += BK;
    kflat += BK;
    if (FLAT ? c0 >= p.Cin : c0 == p.Cin) {          // FLAT: BK <= Cin, at most one tap per step
      c0 -= p.Cin;
      if (++kw == p.ks) { kw = 0; ++kh; }
    }
  };
  // staging writes: row r, slot kq ^ f(r)
  int st_off[AR > BR ? AR : BR];
#pragma unroll
  for (int i = 0; i < (AR > BR ? AR : BR); ++i) {
    const int r = lrow + RPP * i;
    st_off[i] = r * ROWB + ((kq ^ ((r ^ (r >> 1)) & (KQ - 1))) << 4);
  }
  auto lstore = [&](int set) {
    char *A = lds;
    char *Bt = lds + BM * ROWB;
#pragma unroll
    for (int i = 0; i < AR; ++i) *reinterpret_cast<u32x4 *>(A + st_off[i]) = a_reg[set][i];
#pragma unroll
    for (int i = 0; i < BR; ++i)
      if (lrow + RPP * i < BN) *reinterpret_cast<u32x4 *>(Bt + st_off[i]) = b_reg[set][i];
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment reads: row (lane & 15) of a 16-row tile, slot (lane >> 4) + 4 * sub
  const int fr = ((lane & 15) ^ ((lane & 15) >> 1)) & (KQ - 1);
  int frag_off[KQ / 4];
#pragma unroll
  for (int sub = 0; sub < KQ / 4; ++sub)
    frag_off[sub] = (lane & 15) * ROWB + ((((lane >> 4) + 4 * sub) ^ fr) << 4);
  const int a_base = wm * (BM / WM) * ROWB;
  const int b_base = BM * ROWB + wn * (BN / WN) * ROWB;

  auto compute = [&]() {
    const char *L = lds;
#pragma unroll
    for (int sub = 0; sub < KQ / 4; ++sub) {
      u32x4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[i] = *reinterpret_cast<const u32x4 *>(L + a_base + i * 16 * ROWB + frag_off[sub]);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bf[j] = *reinterpret_cast<const u32x4 *>(L + b_base + j * 16 * ROWB + frag_off[sub]);
      if constexpr (sizeof(typename T::elem) == 4) {
        // k-step outermost: consecutive MFMAs hit different accumulators (the 16x16x4 f32 MFMA
        // has a 40-cycle dependent latency vs a 32-cycle issue interval)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                  __uint_as_float(af[i][kk]), __uint_as_float(bf[j][kk]), acc[i][j], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(bf16x8, af[i]), __builtin_bit_cast(bf16x8, bf[j]), acc[i][j], 0,
                0, 0);
      }
    }
  };

  if constexpr (PD == 1) {
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kc = 0; kc < n_chunks; ++kc) {
      const bool more = kc + 1 < n_chunks;
      if (more) gload(0);
      compute();
      __syncthreads();                         // everybody is done reading the buffer
      if (more) lstore(0);
      __syncthreads();
    }
  } else {
    // PD register sets in flight; set (kc % PD) holds chunk kc.  Loads past the last chunk are
    // issued with out-of-range offsets (they return zeros and are never staged), so that there is
    // no branch around a load and the compiler's vmcnt bookkeeping stays exact.
#pragma unroll
    for (int d = 0; d < PD; ++d) gload(d);
    lstore(0);
    gload(0);                                  // chunk PD
    __syncthreads();
    // (the trip count is rounded up to a multiple of PD: the extra chunks are zeros)
    for (int kc = 0; kc < n_chunks; kc += PD) {
#pragma unroll
      for (int d = 0; d < PD; ++d) {
        compute();
        __syncthreads();                       // everybody is done reading the buffer
        lstore((d + 1) % PD);
        __builtin_amdgcn_sched_barrier(0);
        gload((d + 1) % PD);                   // chunk kc + d + 1 + PD
        __syncthreads();
      }
    }
  }

  if constexpr (SPLIT) {
    const int S = p.ksplit;
    const int unit = wg * 4 + wave, n_units = p.nbx * p.nby * 4;
    int *cnt = p.split_cnt + 2 * unit;
    int tk = 0;
    if (lane == 0) tk = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int ticket = __builtin_amdgcn_readfirstlane(tk);
    const __amdgpu_buffer_rsrc_t rs_slab =
        __builtin_amdgcn_make_buffer_rsrc(p.split_ws, 0, p.split_bytes, 0x00020000);
    auto slab_off = [&](int sl, int tile) {
      return ((sl * n_units + unit) * (TM * TN) + tile) * 1024 + lane * 16;
    };
    if (ticket != S - 1) {
      // publish this slice's partial accumulators (write-through), drain, report, leave
      int off[TM * TN];
#pragma unroll
      for (int q = 0; q < TM * TN; ++q) off[q] = slab_off(slice, q);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rs_slab,
                                                 off[i * TN + j], 0, /*sc1*/ 16);
      asm volatile("s_nop 1\n\ts_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) (void)__hip_atomic_fetch_add(cnt + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    // last to arrive: every other slice holds a ticket, i.e. is resident and past its K loop
    if (lane == 0)
      while (__hip_atomic_load(cnt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != S - 1)
        __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
    f32x4 sum[TM][TN];
    for (int sl = 0; sl < S; ++sl) {
      f32x4 term[TM][TN];
      if (sl == slice) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) term[i][j] = acc[i][j];
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            term[i][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                       rs_slab, slab_off(sl, i * TN + j), 0, /*sc1*/ 16));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) sum[i][j] = sl == 0 ? term[i][j] : sum[i][j] + term[i][j];
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = sum[i][j];
    if (lane == 0) {
      __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(cnt + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }

  if constexpr (VEC) {
    if (p.vec4) {
      conv_epilogue_vec<T, TM, TN, UPS>(p, acc, lds + wave * (LDS_EPI / 4), m_blk + wm * (BM / WM),
                                        n_blk + wn * (BN / WN), lane, dy0, dy1);
      return;
    }
  }
  const int col_l = lane & 15, row_l = (lane >> 4) * 4;
  conv_epilogue<T, TM, TN, UPS>(p, acc, m_blk + wm * (BM / WM) + row_l,
                                n_blk + wn * (BN / WN) + col_l);
}

template <typename T, int BM, int BN, int WM, int WN, int UPS = 1, int KQ = 4>
static int launch(ConvK k, hipStream_t s) {
  k.nbx = (k.Cout + BN - 1) / BN;
  k.nby = (k.M + BM - 1) / BM;
  // The scatter of a conv1x1 + nearest-Upsample(UPS) + add layer re-reads and re-writes UPS x UPS output
  // pixels per computed value from the FEW workgroups of a low-resolution GEMM (25 for the 7x7 source at
  // B = 32: 76 us for the x8 layer in bf16, on the lane that closes a stage-4 module,
  // profiles/r05j_timeline_bf16_b32_verbose.txt): UPS workgroups per tile, one output row of every block
  // each.  The GEMM is recomputed (K <= 384: nothing); vector epilogue only (the scalar form keeps one).
  // Only where the plain launch has few workgroups: the x8 layer at B = 64 / f32 goes 38 -> 26 us (49
  // tiles), the x4 layer with 196 tiles 20 -> 24 (profiles/r05k_scatter_split.txt).
  k.ups_split = (UPS > 1 && k.vec4 && k.nbx * k.nby <= 128) ? UPS : 1;
  const unsigned nwg = (unsigned)(k.nbx * k.nby * k.ups_split);
  // weights larger than half an XCD's L2: one N slab per XCD (conv_tile_index)
  if (k.swz == 1 && k.nbx % 8 == 0 && k.wgt_bytes > (2u << 20) && !k.no_nslab) k.swz = 2;
  if constexpr (sizeof(typename T::elem) == 2 && BM == 64 && BN == 48 && UPS == 1 && KQ == 4) {
    if (k.flat) {
      hipLaunchKernelGGL((conv_igemm_kernel<T, 64, 48, WM, WN, 1, 4, 3, true>), dim3(k.nbx * k.nby),
                         dim3(256), 0, s, k);
      return (int)hipGetLastError();
    }
  }
  if (k.flat) return SHAPY_EINVAL;
  if (k.ksplit > 1) {
    // split-K (the plan's choice, SHAPY_TILE_KSPLIT): plain layers on the small tiles only
    if constexpr (UPS == 1 && BM <= 64 && BN <= 64) {
      const int S = k.ksplit, n_all = k.ks * k.ks * (k.Cin / (KQ * T::EPS)), cps = (n_all + S - 1) / S;
      const unsigned long long slab = 4ull * S * k.nbx * k.nby * BM * BN;
      if ((S - 1) * cps >= n_all || !k.split_ws || !k.split_cnt || ((uintptr_t)k.split_ws & 15) ||
          slab > 0x40000000ull || slab > k.split_cap || 8 * k.nbx * k.nby > k.split_cnt_cap)
        return SHAPY_EINVAL;
      k.split_bytes = (unsigned)slab;
      const dim3 grid((unsigned)(k.nbx * k.nby * S));
      if (k.pd3)
        hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, WM, WN, 1, KQ, 3, false, true>), grid, dim3(256), 0, s, k);
      else
        hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, WM, WN, 1, KQ, 1, false, true>), grid, dim3(256), 0, s, k);
      return (int)hipGetLastError();
    } else {
      return SHAPY_EINVAL;
    }
  }
  if (UPS == 1 && k.pd3)
    hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, WM, WN, 1, KQ, 3>), dim3(nwg), dim3(256), 0, s, k);
  else
    hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, WM, WN, UPS, KQ>), dim3(nwg), dim3(256), 0, s, k);
  return (int)hipGetLastError();
}

// the upsample-scatter epilogue and the long-chunk (KQ = 8) variant exist for the small tiles
template <typename T, int BM, int BN, int WM, int WN>
static int launch_small(const ConvK &k, int kq, hipStream_t s) {
  if (kq == 8 && k.ups == 1) return launch<T, BM, BN, WM, WN, 1, 8>(k, s);
  switch (k.ups) {
    case 1: return launch<T, BM, BN, WM, WN, 1>(k, s);
    case 2: return launch<T, BM, BN, WM, WN, 2>(k, s);
    case 4: return launch<T, BM, BN, WM, WN, 4>(k, s);
    case 8: return launch<T, BM, BN, WM, WN, 8>(k, s);
    default: return SHAPY_EINVAL;
  }
}

int conv_tile_auto(int M, int Cout) {
  // Measured on MI355X at B = 64 (tools/conv_bench.py, profiles/conv_bench_r01*.txt): the
  // kernel is latency-bound, so the smallest tiles (wave tile 16x48 / 32x32, 8 waves per
  // SIMD) beat the large ones on every HRNet class.  Choose the N tile (48 or 64) that pads
  // Cout least; on a tie take 64 unless that leaves too few workgroups to fill 256 CUs.
  const long mt = (M + 63) / 64;
  const int p48 = (Cout + 47) / 48 * 48, p64 = (Cout + 63) / 64 * 64;
  if (p48 < p64) return SHAPY_TILE_64x48;
  if (p64 < p48) return SHAPY_TILE_64x64;
  return mt * (p64 / 64) >= 512 ? SHAPY_TILE_64x64 : SHAPY_TILE_64x48;
}

int conv_tile_auto_bf16(int M, int Cout) {
  // bf16 sweep at B = 32 / 64 (profiles/conv_bench_r02s_bf16_b*.txt): the f32 choice is the best
  // or within 3 % everywhere except the 96-wide layers of the 28x28 branch, where one 96-wide N tile
  // (A staged once instead of twice) wins: 64x96 from M = 25,088 (16 -> 14 us), 128x96 from
  // M = 50,176 (26 -> 21 us)
  // (end to end at bs 64: 8,920-9,040 images/s without the rule, 9,410-9,510 with it)
  if (Cout == 96 && M >= 40000) return SHAPY_TILE_128x96;
  if (Cout == 96 && M >= 20000) return SHAPY_TILE_64x96;
  return conv_tile_auto(M, Cout);
}

int conv_tile_auto_x6(int M, int Cout, int K) {
  // bf16x6 kernel, measured on MI355X at B = 64 (profiles/conv_bench_r01f_f32x6.txt): N tile
  // (48 / 64 / 96) with the least padding; on a tie the 96-wide tile (wave tile 32x48) wins
  // while M keeps >= 196 M-tiles busy, the 48-wide one below that; deep-K wide layers of the
  // head take 128x64.
  // Round 5 (profiles/r05q_head_gemm_x6_vs_f32.txt, M = 3,136): the wide GEMMs of the head take 128x128 (wave
  // tile 64x64: 24 fragment reads per 96 MFMAs) -- 2048 -> 2048 161 us against 227 (128x64) and 253 for the f32
  // kernel, 1536 -> 2048 123 / 179 / 196, 512 -> 2048 56 / 69 / 71.
  const int p48 = (Cout + 47) / 48 * 48, p64 = (Cout + 63) / 64 * 64, p96 = (Cout + 95) / 96 * 96;
  if (Cout >= 1024 && Cout % 128 == 0 && K >= 512 && M >= 2048) return SHAPY_TILE_128x128;
  if (p64 < p48 && p64 <= p96) return (Cout >= 1024 && K >= 1024) ? SHAPY_TILE_128x64 : SHAPY_TILE_64x64;
  if (p96 <= p48 && p96 <= p64 && M >= 8192) return SHAPY_TILE_64x96;
  return p48 <= p64 ? SHAPY_TILE_64x48 : SHAPY_TILE_64x64;
}

template <typename T>
static int dispatch(const ConvK &k, int tile, int kq, hipStream_t s) {
  switch (tile) {
    case SHAPY_TILE_64x48: return launch_small<T, 64, 48, 4, 1>(k, kq, s);
    case SHAPY_TILE_64x64: return launch_small<T, 64, 64, 2, 2>(k, kq, s);
    case SHAPY_TILE_128x48: return launch<T, 128, 48, 4, 1>(k, s);
    case SHAPY_TILE_128x64: return launch<T, 128, 64, 4, 1>(k, s);
    case SHAPY_TILE_64x96: return launch<T, 64, 96, 2, 2>(k, s);
    case SHAPY_TILE_128x96: return launch<T, 128, 96, 2, 2>(k, s);
    case SHAPY_TILE_64x128: return launch<T, 64, 128, 2, 2>(k, s);
    case SHAPY_TILE_128x128: return launch<T, 128, 128, 2, 2>(k, s);
    case SHAPY_TILE_256x48: return launch<T, 256, 48, 4, 1>(k, s);
    case SHAPY_TILE_256x64: return launch<T, 256, 64, 4, 1>(k, s);
    case SHAPY_TILE_32x64:
      if (kq != 8 || k.ups != 1) return SHAPY_EINVAL;
      return launch<T, 32, 64, 2, 2, 1, 8>(k, s);
    default: return SHAPY_EINVAL;
  }
}

// Validates a descriptor and fills the kernel argument block up to (not including) the choice of
// algorithm / tile.  *empty = 1: nothing to compute (M or Cout is 0).
int conv_prepare(const ShapyConv &d, ConvK &k, int *empty) {
  *empty = 0;
  const bool bf16 = d.dtype == SHAPY_DTYPE_BF16, x6 = d.dtype == SHAPY_DTYPE_F32X6;
  if (d.dtype != SHAPY_DTYPE_F32 && !bf16 && !x6) return SHAPY_EINVAL;
  const int esz = bf16 ? 2 : 4, eps = 16 / esz;          // element size, elements per slot
  // bf16: Cin % 32 != 0 (the unpadded 48-channel branch) takes the flat-K kernel: Cin % 8 == 0,
  // Cin >= 32, no upsample epilogue, 64x48 tile
  const bool flat = bf16 && d.Cin % 32 != 0;
  if (d.Cin <= 0 || d.Cin % (x6 ? 4 : flat ? eps : 4 * eps) || d.in_ld % eps || d.ksize < 1 ||
      d.stride < 1 || d.ups < 1 || (flat && (d.Cin < 32 || d.ups != 1)))
    return SHAPY_EINVAL;
  if (((uintptr_t)d.in | (uintptr_t)d.wgt) & 15) return SHAPY_EINVAL;
  k.in = d.in; k.wgt = d.wgt; k.bias = d.bias; k.res = d.res; k.out = d.out;
  k.M = d.B * d.Ho * d.Wo;
  k.Hi = d.Hi; k.Wi = d.Wi; k.Cin = d.Cin; k.in_ld = d.in_ld; k.Ho = d.Ho; k.Wo = d.Wo;
  k.Cout = d.Cout; k.ks = d.ksize; k.stride = d.stride; k.pad = d.pad;
  k.out_ld = d.out_ld; k.out_coff = d.out_coff; k.res_ld = d.res_ld; k.res_coff = d.res_coff;
  k.relu = d.relu; k.ups = d.ups;
  // rows of out / res start on 16-byte boundaries (4 f32 / 8 bf16 channels): the epilogue may use
  // 16-byte accesses
  const int cgm = eps - 1;
  k.vec4 = ((d.out_ld | d.out_coff) & cgm) == 0 && ((uintptr_t)d.out & 15) == 0 &&
           (!d.res || (((d.res_ld | d.res_coff) & cgm) == 0 && ((uintptr_t)d.res & 15) == 0));
  const unsigned long long in_bytes = (unsigned long long)esz * d.B * d.Hi * d.Wi * d.in_ld;
  k.Kp = (d.ksize * d.ksize * d.Cin + 31) / 32 * 32;
  const unsigned long long wgt_bytes =
      x6 ? 6ull * d.Cout * k.Kp : (unsigned long long)esz * d.Cout * d.ksize * d.ksize * d.Cin;
  if (in_bytes >= 0x7fffffffull || wgt_bytes >= 0x7fffffffull) return SHAPY_EINVAL;   // 32-bit offsets
  k.in_bytes = (unsigned)in_bytes; k.wgt_bytes = (unsigned)wgt_bytes;
  if (k.M <= 0 || k.Cout <= 0) *empty = 1;
  // Winograd F(2x2,3x3) when the caller supplies the transformed filters (the host's policy,
  // HighResolutionNet.conv_algo); tile flag 0x2000 forces the direct kernel, 0x4000 / 0x8000
  // one / two tile groups per Winograd workgroup (A/B benches)
  k.wgt2 = d.wgt_wino; k.wgt2_bytes = 0; k.wino_tiles = 0;
  k.swz = (d.tile & 0x400) ? 0 : 1;
  k.no_nslab = (d.tile & 0x10000) ? 1 : 0;
  k.no_allk = (d.tile & 0x20000) ? 1 : 0;
  // F(4x4) split-K (conv_wino4.hip): SHAPY_TILE_W4_KSPLIT(S) in the tile word, slab + counters from
  // the caller
  k.ups_split = 1;
  k.ksplit = ((d.tile >> 21) & 3) + 1;
  k.split_ws = d.split_ws; k.split_bytes = 0; k.split_cnt = d.split_cnt;
  k.split_cap = (unsigned long long)(d.split_kib > 0 ? d.split_kib : 0) << 10;
  k.split_cnt_cap = d.split_cnt_n > 0 ? d.split_cnt_n : 0;
  k.flat = flat ? 1 : 0;
  return SHAPY_OK;
}

int conv2d(const ShapyConv &d, hipStream_t s) {
  ConvK k;
  int empty = 0;
  const int rc0 = conv_prepare(d, k, &empty);
  if (rc0 != SHAPY_OK || empty) return rc0;
  const bool bf16 = d.dtype == SHAPY_DTYPE_BF16, x6 = d.dtype == SHAPY_DTYPE_F32X6;
  const int eps = bf16 ? 8 : 4;
  const bool flat = k.flat != 0;
  // three chunks of global loads in flight: bf16 always (its chunks are 50-100 ns of MFMAs);
  // float32 on request (tile flag 0x40000, A/B benches)
  k.pd3 = (bf16 || (d.tile & 0x40000)) && !(d.tile & 0x80000) ? 1 : 0;
  // tile flag 0x100000: wgt_wino holds F(4x4,3x3) filters [36][Cin/16][Cout][16] (conv_wino4.hip;
  // the host's policy again: HighResolutionNet.conv_algo = 'winograd4') -- there is no other
  // kernel for that layout, so an ineligible layer is an error, not a fallback
  if (d.tile & 0x100000) {
    if (conv_wino4_fits(k)) {      // (size limits first: conv_wino_eligible also refuses > 2 GiB)
      if (d.dtype != SHAPY_DTYPE_F32 || !conv_wino_eligible(k)) return SHAPY_EINVAL;
      return conv2d_wino4(k, s);
    }
    if (d.dtype != SHAPY_DTYPE_F32 || k.Cout % 48) return SHAPY_EINVAL;
    // tensors beyond the kernel's 1 GiB offset scheme (B > 334 at 224x224) or with rows that are
    // not 16-byte aligned: the direct kernel on
    // the untransformed weights, which every layer carries -- slower, same convolution
    k.wgt2 = nullptr;
    static bool warned = false;        // once per process: a 1.7x per-layer perf cliff must not be silent
    if (!warned) {
      warned = true;
      fprintf(stderr, "shapy: F(4x4) Winograd layer %dx%d %d->%d at M=%d is outside the kernel's limits "
                      "(1 GiB addressing, 16-byte-aligned rows); such layers run the direct kernel "
                      "(slower, same result)\n",
              k.Hi, k.Wi, k.Cin, k.Cout, k.M);
    }
  }
  if (d.dtype == SHAPY_DTYPE_F32 && !(d.tile & 0x2000) && conv_wino_eligible(k))
    return conv2d_wino(k, (d.tile & 0x4000) ? 1 : (d.tile & 0x8000) ? 2 : 0, s);
  // d.tile: low byte = SHAPY_TILE_* (0 = auto).  Tuning knobs of tools/conv_bench.py:
  // 0x400 disables the XCD-contiguous workgroup order, 0x200 / 0x800 force the long (8 slots =
  // 128-byte rows) / short (4 slots) K chunk.  Defaults (profiles/conv_bench_r01*.txt):
  // XCD-contiguous always (+2..7 %), long chunks whenever the K loop stays pipelined
  // (K >= 512 elements per slot-width: +6..20 % on the 14x14 / 7x7 layers, slower on K = 64).
  const int Kc = k.ks * k.ks * k.Cin;
  int kq = (Kc >= 128 * eps && k.Cin % (8 * eps) == 0) ? 8 : 4;
  if ((d.tile & 0x200) && k.Cin % (8 * eps) == 0) kq = 8;
  if (d.tile & 0x800) kq = 4;
  int tile = (d.tile & 0xff) ? (d.tile & 0xff)
                             : (x6 ? conv_tile_auto_x6(k.M, k.Cout, Kc)
                                   : bf16 ? conv_tile_auto_bf16(k.M, k.Cout) : conv_tile_auto(k.M, k.Cout));
  // few M tiles (14x14 / 7x7 maps at B = 64): halve BM so that every CU still holds several
  // workgroups (+4..12 %, profiles/conv_bench_r01h_32x64.txt)
  if (!(d.tile & 0xff) && !bf16 && !x6 && kq == 8 && k.ups == 1 && k.Cout % 64 == 0 && k.M <= 12544)
    tile = SHAPY_TILE_32x64;
  // one N tile only (Cout = 64: conv2 of the stem, 3x3 / stride 2 at 112x112): the 32-row tile again --
  // 134 vs 149 us at B = 64 (profiles/r04j_conv_bench_direct_tile_sweep.txt)
  if (!(d.tile & 0xff) && !bf16 && !x6 && kq == 8 && k.ups == 1 && k.Cout == 64 && k.stride == 2)
    tile = SHAPY_TILE_32x64;
  if (k.ups != 1 && tile != SHAPY_TILE_64x48 && tile != SHAPY_TILE_64x64)
    tile = (k.Cout % 64 == 0 && k.Cout % 48 != 0) ? SHAPY_TILE_64x64 : SHAPY_TILE_64x48;
  if (flat) { tile = SHAPY_TILE_64x48; kq = 4; }
  if (x6) return conv2d_x6(k, tile, s);
  return bf16 ? dispatch<BF16>(k, tile, kq, s) : dispatch<F32>(k, tile, kq, s);
}

// Several F(4x4,3x3) layers in ONE persistent launch (conv_wino4g.hip).  Every descriptor must be
// a layer shapy_conv2d would run on the F(4x4) kernel (SHAPY_TILE_WINO4 set, float32, wgt_wino
// given, within its 1 GiB addressing); anything else is SHAPY_EINVAL -- the caller falls back to
// one shapy_conv2d per layer.
int conv2d_group(const ShapyConv *ds, int n, hipStream_t s) {
  if (!ds || n < 1 || n > 4) return SHAPY_EINVAL;
  ConvK ks[4];
  int m = 0;
  for (int i = 0; i < n; ++i) {
    const ShapyConv &d = ds[i];
    // (the persistent kernel has no split-K form: such layers go one by one)
    if (d.dtype != SHAPY_DTYPE_F32 || !(d.tile & 0x100000) || !d.wgt_wino || ((d.tile >> 21) & 3))
      return SHAPY_EINVAL;
    int empty = 0;
    const int rc = conv_prepare(d, ks[m], &empty);
    if (rc != SHAPY_OK) return rc;
    if (empty) continue;
    if (!conv_wino4_fits(ks[m]) || !conv_wino_eligible(ks[m])) return SHAPY_EINVAL;
    ++m;
  }
  if (m == 0) return SHAPY_OK;
  return conv2d_wino4_group(ks, m, s);
}

}  // namespace shapy
