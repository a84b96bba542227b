// Does s_setprio arbitrate the matrix pipe between two waves of one SIMD, and do the wave slot ids
// (HW_ID.WAVE_ID) of two co-resident workgroups differ in their low bits?  512 workgroups of 4 waves
// with 72 KB of LDS each (two per CU, one wave of each per SIMD) run the same MFMA loop; mode 1 sets
// the priority 3 - (WAVE_ID & 3).  If priorities arbitrate, the higher one of a SIMD's two waves
// finishes in about half the time of the lower one; if not, both take the same (shared) time.
//   hipcc --offload-arch=gfx950 -O2 tools/prio_probe.hip -o tools/prio_probe && tools/prio_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct Rec {
  unsigned hw, xcc;
  unsigned long long t0, t1;
  int prio, blk;
};
__global__ __launch_bounds__(256) void probe(Rec *out, int iters, int mode) {
  extern __shared__ char lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  int prio = 0;
  if (mode == 1) prio = 3 - (int)(hw & 3u);
  switch (prio) {
    case 3: __builtin_amdgcn_s_setprio(3); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    default: __builtin_amdgcn_s_setprio(0); break;
  }
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  const float x = lane * 0.001f, y = 1.0f + lane;
  if (lane == 0) lds[wave] = 1;
  __syncthreads();
  const unsigned long long t0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, a3, 0, 0, 0);
  }
  const unsigned long long t1 = wall_clock64();
  if (lane == 0) {
    Rec r;
    r.hw = hw; r.xcc = xcc; r.t0 = t0; r.t1 = t1; r.prio = prio; r.blk = blockIdx.x;
    out[blockIdx.x * 4 + wave] = r;
  }
  if (a0[0] + a1[1] + a2[2] + a3[3] == 12345.678f) out[0].hw = 0;      // keep the MFMAs alive
}

int main() {
  const int n_wg = 512, iters = 20000;
  Rec *d = nullptr;
  if (hipMalloc(&d, sizeof(Rec) * n_wg * 4) != hipSuccess) return 1;
  (void)hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
  std::vector<Rec> h(n_wg * 4);
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(probe, dim3(n_wg), dim3(256), 72 * 1024, 0, d, iters, mode);
      if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
    }
    (void)hipMemcpy(h.data(), d, sizeof(Rec) * h.size(), hipMemcpyDeviceToHost);
    // waves per SIMD: key = (xcc, se, sh, cu, simd)
    std::map<unsigned, std::vector<const Rec *>> simd;
    for (const Rec &r : h) {
      const unsigned key = ((r.xcc & 0xf) << 16) | (r.hw & 0xfff0u);       // SE, SH, CU, PIPE, SIMD bits
      simd[key].push_back(&r);
    }
    int pairs = 0, distinct = 0, overlap = 0, other = 0;
    double d_hi = 0, d_lo = 0, d_eq = 0; int n_eq = 0, n_hl = 0;
    int wid_hist[16] = {};
    for (auto &kv : simd) {
      auto &v = kv.second;
      for (const Rec *r : v) wid_hist[r->hw & 15]++;
      if (v.size() != 2) { ++other; continue; }
      ++pairs;
      const Rec *a = v[0], *b = v[1];
      if ((a->hw & 3) != (b->hw & 3)) ++distinct;
      const bool ov = a->t0 < b->t1 && b->t0 < a->t1;
      if (!ov) continue;
      ++overlap;
      const double da = (double)(a->t1 - a->t0) * 0.01, db = (double)(b->t1 - b->t0) * 0.01;   // us (100 MHz)
      if (a->prio != b->prio) {
        ++n_hl;
        d_hi += a->prio > b->prio ? da : db;
        d_lo += a->prio > b->prio ? db : da;
      } else {
        ++n_eq;
        d_eq += 0.5 * (da + db);
      }
    }
    printf("mode %d: %zu SIMDs seen, %d with two waves (%d other), WAVE_ID low bits differ in %d, time-overlapping %d\n",
           mode, simd.size(), pairs, other, distinct, overlap);
    printf("  WAVE_ID histogram:");
    for (int i = 0; i < 16; ++i) printf(" %d", wid_hist[i]);
    printf("\n");
    if (n_eq) printf("  equal priority pairs %d: mean duration %.1f us\n", n_eq, d_eq / n_eq);
    if (n_hl) printf("  unequal priority pairs %d: higher %.1f us, lower %.1f us\n", n_hl, d_hi / n_hl, d_lo / n_hl);
  }
  printf("one wave alone would take %.1f us at 2.1 GHz (4 x %d MFMAs x 32 cycles)\n", 4.0 * iters * 32 / 2100.0, iters);
  return 0;
}
