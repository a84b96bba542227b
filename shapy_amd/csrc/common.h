// Shared declarations of libshapy_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/shapy_hip.h"

namespace shapy {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

enum {
  SHAPY_TILE_AUTO = 0,
  SHAPY_TILE_256x48 = 1,
  SHAPY_TILE_128x96 = 2,
  SHAPY_TILE_128x128 = 3,
  SHAPY_TILE_256x64 = 4,
  SHAPY_TILE_64x48 = 5,
  SHAPY_TILE_64x96 = 6,
  SHAPY_TILE_64x128 = 7,
  SHAPY_TILE_64x64 = 8,
  SHAPY_TILE_128x48 = 9,
  SHAPY_TILE_128x64 = 10,
  SHAPY_TILE_32x64 = 11,     // long K chunks only (Cin % 32 == 0); for layers with few M tiles
};

int conv2d(const ShapyConv &d, hipStream_t s);
int conv2d_group(const ShapyConv *ds, int n, hipStream_t s);
int conv_tile_auto(int M, int Cout);
int conv_tile_auto_x6(int M, int Cout, int K);

inline int hip_rc(hipError_t e) { return (int)e; }

#define SHAPY_HIP_TRY(expr)                   \
  do {                                        \
    hipError_t _e = (expr);                   \
    if (_e != hipSuccess) return (int)_e;     \
  } while (0)

// ReLU as compare + select: a NaN activation stays NaN, as torch.relu keeps it (a max with 0 would return 0 and
// hide a numerical blow-up from the Winograd guard and from every check downstream).  Same bits for any other input.
__device__ __forceinline__ float relu_keep_nan(float v) { return v < 0.f ? 0.f : v; }

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace shapy
