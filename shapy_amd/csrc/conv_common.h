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
  int pd3;                      // implicit GEMM: three chunks of global loads in flight
  int dbg;                      // -DSHAPY_WINO_TIMING builds only: ablation mask (wrong results)
  int stagger_us, stagger_slots;  // Winograd: start delay per resident-workgroup slot (conv_wino.hip)
};

// Winograd F(2x2,3x3) path of the float32 3x3 / stride-1 layers (conv_wino.hip)
bool conv_wino_eligible(const ConvK &k);
int conv2d_wino(ConvK k, int tm, hipStream_t s);

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
            const float o = p.relu ? fmaxf(acc[i][j][r], 0.f) : acc[i][j][r];
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
                       p.relu ? fmaxf(o, 0.f) : o);
            }
          }
        }
      }
  }
}

// LDS bytes a workgroup needs for conv_epilogue_vec (4 waves, wave tile 16 TM x 16 TN)
constexpr int conv_epilogue_vec_bytes(int TM, int TN) { return 4 * (16 * TM) * (16 * TN + 4) * 4; }

// float32 epilogue with full-line stores.  The MFMA C layout gives a lane one column and four
// rows, i.e. a wave store instruction writes 64-byte row segments; here every wave first parks
// its (bias-added) tile in a private LDS region and reads it back row-major, so that residual
// loads and stores are 16 bytes per lane over whole 128..256-byte row segments.  No workgroup
// barrier: a wave only reads what it wrote itself (the staging buffer is free after the K loop).
template <int TM, int TN>
__device__ __forceinline__ void conv_epilogue_vec(const ConvK &p, f32x4 (&acc)[TM][TN],
                                                  char *lds_wave, int row0, int col0, int lane) {
  constexpr int R = 16 * TM, C = 16 * TN, LDC = C + 4, C4 = C / 4;
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
  const float *res = reinterpret_cast<const float *>(p.res);
  float *out = reinterpret_cast<float *>(p.out);
#pragma unroll
  for (int q0 = 0; q0 < R * C4; q0 += 64) {
    const int q = q0 + lane;
    if ((R * C4) % 64 != 0 && q >= R * C4) break;
    const int rr = q / C4, c4 = q % C4;
    const int row = row0 + rr, col = col0 + c4 * 4;
    if (row >= p.M || col >= p.Cout) continue;
    f32x4 v = *reinterpret_cast<const f32x4 *>(s + rr * LDC + c4 * 4);
    float *o = out + (long)row * p.out_ld + p.out_coff + col;
    const float *rp = res ? res + (long)row * p.res_ld + p.res_coff + col : nullptr;
    if (col + 3 < p.Cout) {
      if (rp) v += *reinterpret_cast<const f32x4 *>(rp);
      if (p.relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      *reinterpret_cast<f32x4 *>(o) = v;
    } else {
      for (int e = 0; e < 4 && col + e < p.Cout; ++e) {
        const float x = v[e] + (rp ? rp[e] : 0.f);
        o[e] = p.relu ? fmaxf(x, 0.f) : x;
      }
    }
  }
}

}  // namespace shapy
