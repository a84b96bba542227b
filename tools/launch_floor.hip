// Fixed cost of a kernel launch in a dependent chain on this GPU (tuning only):
//     hipcc -O3 --offload-arch=gfx950 tools/launch_floor.hip -o /tmp/launch_floor && /tmp/launch_floor
// Back-to-back launches on one stream, HIP events around 200 of them.  Variants: empty kernel;
// with the F(4x4) kernel's resources (73.7 KB LDS, 256 VGPRs via launch bounds, 4 barriers); with
// one 4-byte store per thread (dirty L2 lines at the kernel boundary); with scratch.
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(256) void k_empty(int *p) {
  if (p == (int *)1) p[0] = 1;
}

__global__ __launch_bounds__(256, 2) void k_lds(int *p, int nbar) {
  __shared__ int lds[73728 / 4];
  lds[threadIdx.x] = threadIdx.x;
  for (int i = 0; i < nbar; ++i) __syncthreads();
  if (lds[(threadIdx.x + 1) & 255] == -1) p[0] = 1;
}

__global__ __launch_bounds__(256) void k_store(int *p, int words) {
  for (int i = 0; i < words; ++i) p[((size_t)i * gridDim.x + blockIdx.x) * 256 + threadIdx.x] = i;
}

__global__ __launch_bounds__(256) void k_scratch(int *p, int n) {
  volatile int a[64];
  for (int i = 0; i < 64; ++i) a[i] = i + n;
  if (a[n & 63] == -1) p[0] = 1;
}

template <typename F>
static float timeit(F f, int iters = 200) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 10; ++i) f();
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) f();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / iters;
}

int main() {
  int *buf;
  hipMalloc(&buf, 512u << 20);
  const int grids[] = {8, 128, 256, 512, 784, 1568, 3136};
  printf("%8s %10s %12s %12s %14s %14s %12s\n", "grid", "empty", "lds+4bar", "lds+24bar", "store 1w/thr", "store 48w/thr", "scratch");
  for (int g : grids) {
    const float a = timeit([&] { hipLaunchKernelGGL(k_empty, dim3(g), dim3(256), 0, 0, buf); });
    const float b = timeit([&] { hipLaunchKernelGGL(k_lds, dim3(g), dim3(256), 0, 0, buf, 4); });
    const float b2 = timeit([&] { hipLaunchKernelGGL(k_lds, dim3(g), dim3(256), 0, 0, buf, 24); });
    const float c = timeit([&] { hipLaunchKernelGGL(k_store, dim3(g), dim3(256), 0, 0, buf, 1); });
    const float c2 = timeit([&] { hipLaunchKernelGGL(k_store, dim3(g), dim3(256), 0, 0, buf, 48); });
    const float d = timeit([&] { hipLaunchKernelGGL(k_scratch, dim3(g), dim3(256), 0, 0, buf, g); });
    printf("%8d %9.2fus %11.2fus %11.2fus %13.2fus %13.2fus %11.2fus\n", g, a, b, b2, c, c2, d);
  }
  return 0;
}
