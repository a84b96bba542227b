// VERDICT r5's bounded side experiment, as a K-loop SKELETON: the position-GEMMs of the F(4x4) kernel on
// v_mfma_f32_16x16x32_bf16 with the exact 3-way split (a = h + m + l: six bf16 products per float32 product, f32
// accumulate) against the same loop on v_mfma_f32_16x16x4_f32 -- only what bounds the K loop: filter fragments from
// an L2-resident buffer through a ring, V fragments from LDS, the MFMAs.  No staging, no transform, no epilogue (the
// split of V into three planes would ride on the bf16 pipe's free VALU; it is not the question here).  The question:
// three bf16 planes are 6 bytes per filter value instead of 4, and a workgroup's 16 tiles use every filter fragment
// exactly once -- does the L1 (64 B/clk/CU) let the 2.7x faster matrix pipe show?
//   hipcc --offload-arch=gfx950 -O3 tools/w4_bf16x3_skeleton.hip -o tools/bin/w4_bf16x3_skeleton && tools/bin/w4_bf16x3_skeleton
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int ITEMS = 27;        // per wave and chunk (four waves x 27 = 36 positions x 3 channel groups)
#ifndef SKEL_RING
#define SKEL_RING 9
#endif
constexpr int R = SKEL_RING;     // filter ring (divides 27): 9 = the kernel's; 27 = a whole chunk ahead (the bound a deeper ring could reach)

// f32: chunk = 16 input channels; one 1 KB filter fragment + one V fragment (ds_read_b128) per item, 4 MFMAs
// bf16x3: chunk = 32 input channels; three 1 KB filter fragments (h, m, l) + three V fragments per item, 6 MFMAs
template <bool X3>
__global__ __launch_bounds__(256, 2) void kloop(const char *__restrict__ U, float *out, int chunks, unsigned u_bytes) {
  constexpr int PL = X3 ? 3 : 1;
  __shared__ __attribute__((aligned(16))) char lds[72 * 1024];      // the real kernel's footprint: two workgroups per CU
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int i = t; i < 72 * 1024 / 4; i += 256) reinterpret_cast<float *>(lds)[i] = 1.0f + (i & 7) * 0.125f;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(U), 0, u_bytes, 0x00020000);
  const int slab = (blockIdx.x & 3) * (chunks * 108 * PL * 1024);    // four N tiles' filters (192-channel class)
  const int u_lane = lane * 16 + wave * ITEMS * PL * 1024 + slab;
  const int frag = lane * 16;
  u32x4 ring[R][PL];
  auto bload = [&](int slot, int q, int c) {
#pragma unroll
    for (int p = 0; p < PL; ++p)
      ring[slot][p] = __builtin_amdgcn_raw_buffer_load_b128(rs, u_lane, (c * 108 * PL + q * PL + p) * 1024, 0);
  };
  f32x4 acc[ITEMS];
#pragma unroll
  for (int q = 0; q < ITEMS; ++q) acc[q] = f32x4{0, 0, 0, 0};
#pragma unroll
  for (int q = 0; q < R; ++q) bload(q, q, 0);
  for (int c = 0; c < chunks; ++c) {
    const char *V = lds + (c & 1) * 36 * 1024 + frag;
#pragma unroll
    for (int tq = 0; tq < ITEMS; tq += 3) {          // triples: consecutive MFMAs hit different accumulators
      u32x4 v[3][PL];
#pragma unroll
      for (int e = 0; e < 3; ++e)
#pragma unroll
        for (int p = 0; p < PL; ++p)
          v[e][p] = *reinterpret_cast<const u32x4 *>(V + (((tq + e) * PL + p) % 36) * 1024);
      if constexpr (X3) {
        // product order: small terms first (ul vh, uh vl, um vm, um vh, uh vm, uh vh)
        constexpr int ua[6] = {2, 0, 1, 1, 0, 0}, va[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int k = 0; k < 6; ++k)
#pragma unroll
          for (int e = 0; e < 3; ++e)
            acc[tq + e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(bf16x8, ring[(tq + e) % R][ua[k]]), __builtin_bit_cast(bf16x8, v[e][va[k]]),
                acc[tq + e], 0, 0, 0);
      } else {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int e = 0; e < 3; ++e)
            acc[tq + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(ring[(tq + e) % R][0][kk]),
                                                              __uint_as_float(v[e][0][kk]), acc[tq + e], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const int q = tq + e;
        if (q + R < ITEMS) bload(q % R, q + R, c);
        else bload(q % R, q + R - ITEMS, c + 1 < chunks ? c + 1 : c);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  float s = 0;
#pragma unroll
  for (int q = 0; q < ITEMS; ++q) s += acc[q][0] + acc[q][3];
  if (s == 12345.f) out[0] = s;
}

template <bool X3>
static void run(const char *name, int chunks, int grid) {
  constexpr int PL = X3 ? 3 : 1;
  const size_t bytes = 4ull * chunks * 108 * PL * 1024;
  char *U;
  float *out;
  hipMalloc(&U, bytes);
  hipMemset(U, 0x3c, bytes);                       // (finite values in both interpretations)
  hipMalloc(&out, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kloop<X3>, dim3(grid), dim3(256), 0, 0, U, out, chunks, (unsigned)bytes);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep)
      printf("%-34s grid %4d  %2d chunks of %2d channels  filters %5.1f MB  %.1f us\n", name, grid, chunks, X3 ? 32 : 16,
             bytes / 1e6, ms * 1e3);
  }
  hipFree(U);
  hipFree(out);
}

int main() {
  // the 192-channel class: 12 chunks of 16 channels = 6 chunks of 32; the 384-channel class: 24 = 12
  for (int grid : {256, 512, 2048}) {
    run<false>("f32  (4 x v_mfma_f32_16x16x4_f32)", 12, grid);
    run<true>("bf16 x 3 (6 x 16x16x32_bf16)", 6, grid);
    run<false>("f32  (4 x v_mfma_f32_16x16x4_f32)", 24, grid);
    run<true>("bf16 x 3 (6 x 16x16x32_bf16)", 12, grid);
  }
  return 0;
}
