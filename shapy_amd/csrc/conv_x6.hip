// float32 implicit-GEMM convolution on the bf16 matrix cores ("bf16x6" split).
//
// gfx950 multiplies bf16 16x faster than f32 on the matrix cores (v_mfma_f32_16x16x32_bf16:
// 16 cycles for 16x16x32 vs v_mfma_f32_16x16x4_f32: 32 cycles for 16x16x4).  A float32 value
// is EXACTLY the sum of three bf16 values  a = h + m + l  (8 + 8 + 8 significand bits; h =
// truncation of a, m = truncation of a - h, l = a - h - m), each bf16 x bf16 product is exact in
// f32, and the MFMA accumulates in f32.  So
//     a*b = hh + (hm + mh) + (hl + lh + mm)  +  [ml + lm + ll  <= 2^-24 |ab|]
// and six bf16 MFMAs reproduce the f32 product to the last f32 bit or so -- the dropped terms
// are of the size of ONE f32 rounding of the product, the same error class the f32 MFMA chain
// has -- for 6/16 of the matrix-core time.  Storage stays float32 everywhere (activations,
// residuals, outputs): the activations are split in registers between the global load and the
// LDS write; the weights are split ONCE on the host (shapy_amd/utils/split.py) and arrive as
// three bf16 planes  [Cout][3][Kp]  (Kp = K rounded up to 32, zero padded), so their staging
// is a plain 16-byte copy.
//
// Data flow per 32-wide K chunk: buffer_load_dwordx4 (4 consecutive k of one row) -> split
// into 3 x 4 bf16 -> ds_write_b64 into three LDS planes (row = 32 bf16 = 64 B, 16-byte slots
// XOR-swizzled like the f32 kernel) -> ds_read_b128 = the 8 consecutive k a lane feeds to one
// 16x16x32 MFMA.  K = ks*ks*Cin is walked FLAT (every thread carries its own (kh, kw, c)
// iterator), so channel counts that are not a multiple of 32 (48!) need no padding except in
// the last chunk.
#include "conv_common.h"

namespace shapy {

__device__ __forceinline__ unsigned pack_hi16(unsigned x1, unsigned x0) {
  return __builtin_amdgcn_perm(x1, x0, 0x07060302u);     // (x1 & 0xffff0000) | (x0 >> 16)
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// 4 f32 -> three planes of 4 bf16 (8 bytes each); h + m + l == a exactly.  The subtractions
// are written on float2 so that they become v_pk_add_f32 (two lanes of data per instruction).
__device__ __forceinline__ void split3(const u32x4 &a, uint2 &h, uint2 &m, uint2 &l) {
  unsigned r1[4], r2[4];
#pragma unroll
  for (int e = 0; e < 4; e += 2) {
    const f32x2 x = {__uint_as_float(a[e]), __uint_as_float(a[e + 1])};
    const f32x2 xh = {__uint_as_float(a[e] & 0xffff0000u), __uint_as_float(a[e + 1] & 0xffff0000u)};
    const f32x2 d1 = x - xh;
    r1[e] = __float_as_uint(d1[0]);
    r1[e + 1] = __float_as_uint(d1[1]);
    const f32x2 dm = {__uint_as_float(r1[e] & 0xffff0000u), __uint_as_float(r1[e + 1] & 0xffff0000u)};
    const f32x2 d2 = d1 - dm;
    r2[e] = __float_as_uint(d2[0]);
    r2[e + 1] = __float_as_uint(d2[1]);
  }
  h = make_uint2(pack_hi16(a[1], a[0]), pack_hi16(a[3], a[2]));
  m = make_uint2(pack_hi16(r1[1], r1[0]), pack_hi16(r1[3], r1[2]));
  l = make_uint2(pack_hi16(r2[1], r2[0]), pack_hi16(r2[3], r2[2]));
}

template <int BM, int BN, int WM, int WN, int UPS>
__global__ __launch_bounds__(256) void conv_x6_kernel(ConvK p) {
  static_assert(WM * WN == 4, "4 waves per workgroup");
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int ROWB = 64;                       // bytes per plane row: 32 bf16
  constexpr int PS = (BM + BN) * ROWB;           // plane stride
  constexpr int AR = BM / 32;                    // A rows staged per thread
  constexpr int BQ = 3 * BN;                     // B plane-rows per chunk
  constexpr int BP = (BQ + 63) / 64;             // B plane-rows staged per thread (guarded)
  // ONE staging buffer and two barriers per chunk: barriers cost nothing measurable here,
  // LDS footprint does (21.5 KB instead of 43 KB per workgroup doubles the workgroups per CU:
  // +8.5 % end to end, profiles/r01_ablation_probes.txt)
  constexpr int LDS_EPI = UPS == 1 ? conv_epilogue_vec_bytes(TM, TN) : 0;
  __shared__ __attribute__((aligned(16))) char lds[3 * PS > LDS_EPI ? 3 * PS : LDS_EPI];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int wg = conv_tile_index(p);
  const int m_blk = (wg / p.nbx) * BM, n_blk = (wg % p.nbx) * BN;
  const int kq = t & 7, lrow = t >> 3;           // 8 x 16-byte loads cover a row's 32 floats

  const __amdgpu_buffer_rsrc_t rs_in =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.wgt), 0, p.wgt_bytes, 0x00020000);
  constexpr int OOB = 0x7fffffff;
  int a_off[AR], a_h[AR], a_w[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int m = m_blk + lrow + 32 * i;
    const int mm = m < p.M ? m : 0;
    const int wo = mm % p.Wo;
    const int tq = mm / p.Wo;
    const int ho = tq % p.Ho;
    const int b = tq / p.Ho;
    const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
    a_off[i] = ((b * p.Hi + hi0) * p.Wi + wi0) * p.in_ld * 4;
    a_h[i] = m < p.M ? hi0 : -0x40000000;
    a_w[i] = wi0;
  }
  const int Kw = p.ks * p.ks * p.Cin;
  // B: plane-row q = plane * BN + row; 4 threads (16-byte slots) per plane-row of 32 bf16
  const int bslot = t & 3;
  int b_off[BP], b_st[BP];
#pragma unroll
  for (int i = 0; i < BP; ++i) {
    const int q = (t >> 2) + 64 * i;
    const int pl = q / BN, r = q - pl * BN;
    const int n = n_blk + r;
    b_off[i] = (q < BQ && n < p.Cout) ? ((n * 3 + pl) * p.Kp + bslot * 8) * 2 : OOB;
    b_st[i] = pl * PS + (BM + r) * ROWB + (((bslot ^ (r ^ (r >> 1))) & 3) << 4);
  }

  // this thread's position in the flattened K = (kh, kw, c) axis, for the NEXT chunk to fetch
  int kflat = kq * 4, c = kq * 4, kh = 0, kw = 0;
  auto wrap = [&]() {
    while (c >= p.Cin) {
      c -= p.Cin;
      if (++kw == p.ks) { kw = 0; ++kh; }
    }
  };
  wrap();
  const int n_chunks = (Kw + 31) / 32;

  u32x4 a_reg[AR], b_reg[BP];
  int kbyte = 0;                                 // byte offset of the next chunk in a weight row
  auto gload = [&]() {
    const bool valid = kflat < Kw;
    const int tap_in = ((kh * p.Wi + kw) * p.in_ld + c) * 4;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const bool ok = valid && (unsigned)(a_h[i] + kh) < (unsigned)p.Hi &&
                      (unsigned)(a_w[i] + kw) < (unsigned)p.Wi;
      a_reg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, ok ? a_off[i] + tap_in : OOB, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < BP; ++i)
      b_reg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, b_off[i], kbyte, 0);
    kbyte += 64;
    kflat += 32;
    c += 32;
    wrap();
  };

  // staging writes: 8 bytes into slot (kq >> 1) ^ f(r) of row r, half kq & 1
  int st_off[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int r = lrow + 32 * i;
    st_off[i] = r * ROWB + ((((kq >> 1) ^ (r ^ (r >> 1))) & 3) << 4) + ((kq & 1) << 3);
  }
  auto lstore = [&]() {
    char *A = lds;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      uint2 h, m, l;
      split3(a_reg[i], h, m, l);
      *reinterpret_cast<uint2 *>(A + st_off[i]) = h;
      *reinterpret_cast<uint2 *>(A + PS + st_off[i]) = m;
      *reinterpret_cast<uint2 *>(A + 2 * PS + st_off[i]) = l;
    }
#pragma unroll
    for (int i = 0; i < BP; ++i)
      if ((t >> 2) + 64 * i < BQ) *reinterpret_cast<u32x4 *>(lds + b_st[i]) = b_reg[i];
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment reads: lane l feeds k = 8 * (l >> 4) .. + 7 of row (l & 15): one 16-byte slot
  const int fr = (lane & 15) ^ ((lane & 15) >> 1);
  const int frag_off = (lane & 15) * ROWB + ((((lane >> 4) ^ fr) & 3) << 4);
  const int a_base = wm * (BM / WM) * ROWB + frag_off;
  const int b_base = BM * ROWB + wn * (BN / WN) * ROWB + frag_off;

  gload();
  lstore();
  __syncthreads();

  for (int kc = 0; kc < n_chunks; ++kc) {
    const bool more = kc + 1 < n_chunks;
    if (more) gload();
    const char *L = lds;
    // Register budget decides the occupancy of this kernel: only the A fragments (3 planes) stay
    // live; the B fragments are read one N tile at a time and consumed by the 6 x TM MFMAs of
    // that tile (smallest terms first; with TM > 1 consecutive MFMAs alternate accumulators).
    bf16x8 af[3][TM];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[pl][i] = *reinterpret_cast<const bf16x8 *>(L + pl * PS + a_base + i * 16 * ROWB);
    constexpr int PA[6] = {1, 2, 0, 1, 0, 0};   // mm, lh, hl, mh, hm, hh
    constexpr int PB[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      bf16x8 bf[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        bf[pl] = *reinterpret_cast<const bf16x8 *>(L + pl * PS + b_base + j * 16 * ROWB);
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[PA[q]][i], bf[PB[q]], acc[i][j],
                                                              0, 0, 0);
    }
    __syncthreads();                         // everybody is done reading the buffer
    if (more) lstore();
    __syncthreads();
  }

  if constexpr (UPS == 1) {
    if (p.vec4) {
      conv_epilogue_vec<F32, TM, TN, 1>(p, acc, lds + wave * (LDS_EPI / 4), m_blk + wm * (BM / WM),
                                n_blk + wn * (BN / WN), lane);
      return;
    }
  }
  const int col_l = lane & 15, row_l = (lane >> 4) * 4;
  conv_epilogue<F32, TM, TN, UPS>(p, acc, m_blk + wm * (BM / WM) + row_l,
                                  n_blk + wn * (BN / WN) + col_l);
}

template <int BM, int BN, int WM, int WN, int UPS = 1>
static int launch_x6(ConvK k, hipStream_t s) {
  k.nbx = (k.Cout + BN - 1) / BN;
  k.nby = (k.M + BM - 1) / BM;
  hipLaunchKernelGGL((conv_x6_kernel<BM, BN, WM, WN, UPS>), dim3(k.nbx * k.nby), dim3(256), 0, s, k);
  return (int)hipGetLastError();
}

template <int BM, int BN, int WM, int WN>
static int launch_x6_small(const ConvK &k, hipStream_t s) {
  switch (k.ups) {
    case 1: return launch_x6<BM, BN, WM, WN, 1>(k, s);
    case 2: return launch_x6<BM, BN, WM, WN, 2>(k, s);
    case 4: return launch_x6<BM, BN, WM, WN, 4>(k, s);
    case 8: return launch_x6<BM, BN, WM, WN, 8>(k, s);
    default: return SHAPY_EINVAL;
  }
}

int conv2d_x6(const ConvK &k, int tile, hipStream_t s) {
  switch (tile) {
    case SHAPY_TILE_64x48: return launch_x6_small<64, 48, 4, 1>(k, s);
    case SHAPY_TILE_64x64: return launch_x6_small<64, 64, 2, 2>(k, s);
    case SHAPY_TILE_128x48: return launch_x6<128, 48, 4, 1>(k, s);
    case SHAPY_TILE_128x64: return launch_x6<128, 64, 2, 2>(k, s);
    case SHAPY_TILE_64x96: return launch_x6<64, 96, 2, 2>(k, s);
    case SHAPY_TILE_128x96: return launch_x6<128, 96, 2, 2>(k, s);
    case SHAPY_TILE_64x128: return launch_x6<64, 128, 2, 2>(k, s);
    case SHAPY_TILE_128x128: return launch_x6<128, 128, 2, 2>(k, s);
    default: return SHAPY_EINVAL;
  }
}

}  // namespace shapy
