// Shared pieces of the implicit-GEMM convolution kernels (conv_igemm.hip, conv_x6.hip).
#pragma once
#include "common.h"

namespace shapy {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct F32 {
  using elem = float;
  static constexpr int EPS = 4;       // elements per 16-byte slot
  static __device__ __forceinline__ float load(const void *p, long i) {
    return reinterpret_cast<const float *>(p)[i];
  }
  static __device__ __forceinline__ void store(void *p, long i, float v) {
    reinterpret_cast<float *>(p)[i] = v;
  }
};

struct BF16 {
  using elem = unsigned short;
  static constexpr int EPS = 8;
  static __device__ __forceinline__ float load(const void *p, long i) {
    return __uint_as_float((unsigned)reinterpret_cast<const unsigned short *>(p)[i] << 16);
  }
  static __device__ __forceinline__ void store(void *p, long i, float v) {
    unsigned u = __float_as_uint(v);
    u += 0x7fffu + ((u >> 16) & 1u);                    // round to nearest even
    reinterpret_cast<unsigned short *>(p)[i] = (unsigned short)(u >> 16);
  }
};

struct ConvK {
  const void *in, *wgt, *res;
  const float *bias;
  void *out;
  int M, Hi, Wi, Cin, in_ld, Ho, Wo, Cout, ks, stride, pad;
  int out_ld, out_coff, res_ld, res_coff, relu, ups, swz, nbx, nby;
  unsigned in_bytes, wgt_bytes;
  int Kp;                       // bf16x6 weights: K rounded up to 32 (row length of a plane)
  int vec4;                     // float32 out / res rows are 16-byte aligned: vector epilogue
  const void *wgt2;             // Winograd-transformed filters [16][Cin/16][Cout][16] f32, or null
  unsigned wgt2_bytes;
  int wino_tiles;               // B * ceil(H/2) * ceil(W/2)
  int no_nslab;                 // tuning: keep the m-major XCD order for large weights
  int no_allk;                  // tuning: Winograd K loop always chunk by chunk
  int flat;                     // bf16, Cin % 32 != 0: flat-K kernel
  int pd3;                      // implicit GEMM: three chunks of global loads in flight
  int ups_split;                // upsample-scatter layers: workgroups per tile, each scatters ups / ups_split rows
  int ksplit;                   // K slices per output tile (SHAPY_TILE_KSPLIT), 1 = none: F(4x4) and implicit GEMM
  void *split_ws;               // ... their slab (ShapyConv.split_ws), the bytes the launch uses of it ...
  unsigned split_bytes;
  unsigned long long split_cap; // ... and what the caller provides (ShapyConv.split_kib)
  int *split_cnt;               // ... arrival counters (ShapyConv.split_cnt; zero between launches) and how
  int split_cnt_cap;            //     many the caller provides (ShapyConv.split_cnt_n)
};

// Winograd F(2x2,3x3) path of the float32 3x3 / stride-1 layers (conv_wino.hip)
bool conv_wino_eligible(const ConvK &k);
int conv2d_wino(ConvK k, int tm, hipStream_t s);
// Winograd F(4x4,3x3) path (conv_wino4.hip): k.wgt2 holds [36][Cin/16][Cout][16] filters
int conv2d_wino4(ConvK k, hipStream_t s);
bool conv_wino4_fits(const ConvK &k);
// persistent grouped F(4x4) launch (conv_wino4g.hip): up to 4 layers
int conv2d_wino4_group(const ConvK *ks, int n, hipStream_t s);

// float32 storage, bf16x6 split arithmetic on the bf16 matrix cores (conv_x6.hip)
int conv2d_x6(const ConvK &k, int tile, hipStream_t s);

// workgroup -> tile.  Workgroup ids are dealt round-robin to the 8 XCDs (private L2s): with
// swz every XCD gets a CONTIGUOUS run of tiles (n fastest, then m), so the tiles that share
// A rows -- the N tiles of one M tile and the 3x3 halos of neighbouring M tiles -- hit the
// same L2.  Speed only; correctness does not depend on the placement.
__device__ __forceinline__ int conv_tile_index(const ConvK &p) {
  int wg = blockIdx.x;
  if (p.swz == 2) {
    // N-slab per XCD (layers whose weights exceed an XCD's 4 MB L2, e.g. the head's 2048-wide
    // 1x1 convs: 16.8 MB): XCD x owns the n tiles [x nbx/8, (x+1) nbx/8) for ALL m tiles, walked
    // m-major with its own n tiles fastest.  Its weight slab stays L2-resident and every A tile
    // is fetched once per XCD, instead of the whole weight matrix once per group of m tiles
    // (PMC: 1.47 GB of L2 misses per 2048 -> 2048 launch with the m-major order).
    const int npx = p.nbx >> 3, xcd = wg & 7, i = wg >> 3;
    return (i / npx) * p.nbx + xcd * npx + i % npx;
  }
  if (p.swz) {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = wg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
  }
  return wg;
}

// Epilogue of a wave's TM x TN grid of 16x16 accumulator tiles (MFMA C layout: lane l holds
// column l & 15, rows 4 * (l >> 4) .. + 3): bias (+ residual) (+ ReLU), plain or
// upsample-scatter store.  T is the storage type of `res` / `out`.
template <typename T, int TM, int TN, int UPS>
__device__ __forceinline__ void conv_epilogue(const ConvK &p, f32x4 (&acc)[TM][TN], int row0,
                                              int col0) {
  // `res` may alias `out` (in-place accumulation of the fuse layers), so a residual load may
  // not be scheduled across an earlier store by the compiler: issue ALL residual loads of a
  // group first, then all stores.
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = col0 + j * 16;
    const float bias = (p.bias && col < p.Cout) ? p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] += bias;
  }
  if constexpr (UPS == 1) {
    if (p.res) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = col0 + j * 16;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = row0 + i * 16 + r;
            if (col < p.Cout && row < p.M)
              acc[i][j][r] += T::load(p.res, (long)row * p.res_ld + p.res_coff + col);
          }
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = col0 + j * 16;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = row0 + i * 16 + r;
          if (col < p.Cout && row < p.M) {
            const float o = p.relu ? relu_keep_nan(acc[i][j][r]) : acc[i][j][r];
            T::store(p.out, (long)row * p.out_ld + p.out_coff + col, o);
          }
        }
    }
  } else {
    // conv1x1 + BN + nearest Upsample(UPS) + add (+ ReLU): every computed value goes to a
    // UPS x UPS block of output pixels; residual loads are batched RG rows at a time
    constexpr int RG = UPS >= 4 ? 16 / UPS : UPS;     // UPS 2 -> 2, 4 -> 4, 8 -> 2 rows
    const int WoU = p.Wo * UPS;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + i * 16 + r;
        if (row >= p.M) continue;
        const int wo = row % p.Wo;
        const int tq = row / p.Wo;
        const int ho = tq % p.Ho;
        const int b = tq / p.Ho;
        const long pix0 = ((long)(b * p.Ho + ho) * UPS) * WoU + (long)wo * UPS;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = col0 + j * 16;
          if (col >= p.Cout) continue;
          const float v = acc[i][j][r];
          const long rbase = pix0 * p.res_ld + p.res_coff + col;
          const long obase = pix0 * p.out_ld + p.out_coff + col;
#pragma unroll
          for (int dy0 = 0; dy0 < UPS; dy0 += RG) {
            float tmp[RG * UPS];
#pragma unroll
            for (int q = 0; q < RG * UPS; ++q)
              tmp[q] = p.res ? T::load(p.res, rbase + ((long)(dy0 + q / UPS) * WoU + q % UPS) * p.res_ld)
                             : 0.f;
#pragma unroll
            for (int q = 0; q < RG * UPS; ++q) {
              const float o = v + tmp[q];
              T::store(p.out, obase + ((long)(dy0 + q / UPS) * WoU + q % UPS) * p.out_ld,
                       p.relu ? relu_keep_nan(o) : o);
            }
          }
        }
      }
  }
}

// LDS bytes a workgroup needs for conv_epilogue_vec (4 waves, wave tile 16 TM x 16 TN)
constexpr int conv_epilogue_vec_bytes(int TM, int TN) { return 4 * (16 * TM) * (16 * TN + 4) * 4; }

// Epilogue with full-line stores.  The MFMA C layout gives a lane one column and four rows, i.e.
// a wave store instruction writes 64-byte (f32) / 32-byte (bf16) row segments; here every wave
// first parks its (bias-added) tile in a private LDS region and reads it back row-major, so that
// residual loads and stores are 16 bytes per lane (4 f32 / 8 bf16 channels) over whole
// 128..256-byte row segments.  No workgroup barrier: a wave only reads what it wrote itself (the
// staging buffer is free after the K loop).  UPS > 1: conv1x1 + BN + nearest Upsample(UPS) + add
// (+ ReLU) -- the 16-byte channel group goes to a UPS x UPS block of output pixels (the scalar
// scatter wrote 2-byte pieces of 32-byte segments in bf16: 193 us for the 384 -> 64 x8 layer at
// B = 32, profiles/conv_bench_r02q_bf16_b32_pd3_vs_pd1.txt).  `res` may alias `out` (in-place
// accumulation of the fuse layers): every 16-byte group is read and written by the same lane.
template <typename T>
struct VecIO;
template <>
struct VecIO<F32> {
  static constexpr int CG = 4;
  static __device__ __forceinline__ void load(const void *p, long i, float (&v)[4]) {
    const f32x4 x = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const float *>(p) + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = x[e];
  }
  static __device__ __forceinline__ void store(void *p, long i, const float (&v)[4]) {
    *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(p) + i) = f32x4{v[0], v[1], v[2], v[3]};
  }
};
template <>
struct VecIO<BF16> {
  static constexpr int CG = 8;
  static __device__ __forceinline__ void load(const void *p, long i, float (&v)[8]) {
    const u32x4 x = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const unsigned short *>(p) + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[2 * e] = __uint_as_float(x[e] << 16);
      v[2 * e + 1] = __uint_as_float(x[e] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ void store(void *p, long i, const float (&v)[8]) {
    u32x4 x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned lo = __float_as_uint(v[2 * e]), hi = __float_as_uint(v[2 * e + 1]);
      lo += 0x7fffu + ((lo >> 16) & 1u);                // round to nearest even (as BF16::store)
      hi += 0x7fffu + ((hi >> 16) & 1u);
      x[e] = (lo >> 16) | (hi & 0xffff0000u);
    }
    *reinterpret_cast<u32x4 *>(reinterpret_cast<unsigned short *>(p) + i) = x;
  }
};

// dy0 .. dy1: the rows of a value's UPS x UPS block this workgroup scatters (all of them unless the launch
// runs several workgroups per tile, ConvK.ups_split)
template <typename T, int TM, int TN, int UPS>
__device__ __forceinline__ void conv_epilogue_vec(const ConvK &p, f32x4 (&acc)[TM][TN],
                                                  char *lds_wave, int row0, int col0, int lane,
                                                  int dy0 = 0, int dy1 = UPS) {
  constexpr int CG = VecIO<T>::CG;
  constexpr int R = 16 * TM, C = 16 * TN, LDC = C + 4, NG = C / CG;
  static_assert(C % CG == 0, "wave tile width is a multiple of the channel group");
  float *s = reinterpret_cast<float *>(lds_wave);
  const int cl = lane & 15, rl = (lane >> 4) * 4;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = col0 + j * 16 + cl;
    const float bias = (p.bias && col < p.Cout) ? p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) s[(i * 16 + rl + r) * LDC + j * 16 + cl] = acc[i][j][r] + bias;
  }
  // the tile is read back by other lanes of the same wave: LDS executes a wave's instructions
  // in order; the asm keeps the compiler from moving the reads above the writes
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int WoU = p.Wo * UPS;
#pragma unroll
  for (int q0 = 0; q0 < R * NG; q0 += 64) {
    const int q = q0 + lane;
    if ((R * NG) % 64 != 0 && q >= R * NG) break;
    const int rr = q / NG, cg = q % NG;
    const int row = row0 + rr, col = col0 + cg * CG;
    if (row >= p.M || col >= p.Cout) continue;
    float v[CG];
#pragma unroll
    for (int e = 0; e < CG; e += 4) {
      const f32x4 x = *reinterpret_cast<const f32x4 *>(s + rr * LDC + cg * CG + e);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[e + k] = x[k];
    }
    long pix0 = row;
    if constexpr (UPS > 1) {
      const int wo = row % p.Wo;
      const int tq = row / p.Wo;
      const int ho = tq % p.Ho;
      const int b = tq / p.Ho;
      pix0 = ((long)(b * p.Ho + ho) * UPS) * WoU + (long)wo * UPS;
    }
    if (col + CG - 1 < p.Cout) {
      for (int dy = dy0; dy < dy1; ++dy) {
        float rv[UPS][CG];
#pragma unroll
        for (int dx = 0; dx < UPS; ++dx) {
          if (p.res) {
            VecIO<T>::load(p.res, (pix0 + (long)dy * WoU + dx) * p.res_ld + p.res_coff + col, rv[dx]);
          } else {
#pragma unroll
            for (int e = 0; e < CG; ++e) rv[dx][e] = 0.f;
          }
        }
#pragma unroll
        for (int dx = 0; dx < UPS; ++dx) {
          float o[CG];
#pragma unroll
          for (int e = 0; e < CG; ++e) {
            o[e] = v[e] + rv[dx][e];
            if (p.relu) o[e] = relu_keep_nan(o[e]);
          }
          VecIO<T>::store(p.out, (pix0 + (long)dy * WoU + dx) * p.out_ld + p.out_coff + col, o);
        }
      }
    } else {
      for (int dy = dy0; dy < dy1; ++dy)
        for (int dx = 0; dx < UPS; ++dx)
          for (int e = 0; e < CG && col + e < p.Cout; ++e) {
            const long pi = pix0 + (long)dy * WoU + dx;
            const float x = v[e] + (p.res ? T::load(p.res, pi * p.res_ld + p.res_coff + col + e) : 0.f);
            T::store(p.out, pi * p.out_ld + p.out_coff + col + e, p.relu ? relu_keep_nan(x) : x);
          }
    }
  }
}

}  // namespace shapy
