// Sustained matrix-core rate of the GPU this runs on: long chains of independent MFMAs, no
// memory traffic.  Gives the clock-limited ceiling the conv kernels can be compared with
// (the nominal peaks assume 2.4 GHz; under MFMA load the part runs slower).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak && tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters) {
  const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  if constexpr (MODE == 0) {            // v_mfma_f32_16x16x4_f32, 8 independent accumulators
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    if (s == 12345.f) out[0] = s;
  } else if constexpr (MODE == 1) {     // v_mfma_f32_32x32x2_f32, 4 independent accumulators
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0];
    if (s == 12345.f) out[0] = s;
  } else {                              // v_mfma_f32_16x16x32_bf16
    bf16x8 va, vb;
    for (int j = 0; j < 8; ++j) { va[j] = (__bf16)a; vb[j] = (__bf16)b; }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, vb, acc[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    if (s == 12345.f) out[0] = s;
  }
}

template <int MODE>
static void run(const char *name, double flop_per_mfma, int per_iter, int blocks_per_cu, int iters, float ms_target) {
  float *out;
  hipMalloc(&out, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * blocks_per_cu;
  hipLaunchKernelGGL(mfma_loop<MODE>, dim3(grid), dim3(256), 0, 0, out, iters);   // warm-up
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<MODE>, dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 4 * iters * per_iter * flop_per_mfma;
    printf("%-28s waves/SIMD=%d  %.3f ms  %.1f TFLOP/s\n", name, blocks_per_cu, ms, flop / ms / 1e9);
  }
  hipFree(out);
}

int main() {
  for (int occ : {1, 2, 4, 8}) {
    run<0>("v_mfma_f32_16x16x4_f32", 2.0 * 16 * 16 * 4, 8, occ, 20000 / occ, 0);
    run<1>("v_mfma_f32_32x32x2_f32", 2.0 * 32 * 32 * 2, 4, occ, 20000 / occ, 0);
    run<2>("v_mfma_f32_16x16x32_bf16", 2.0 * 16 * 16 * 32, 8, occ, 40000 / occ, 0);
  }
  return 0;
}
