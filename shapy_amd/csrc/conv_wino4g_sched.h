// Task scheduling of the persistent grouped F(4x4) kernel (conv_wino4g.hip), free of HIP types so that
// the same code is compiled by g++ into a CPU test (tests/native/w4g_sched_test.cpp,
// tests/test_host_cpu.py::test_w4g_schedule_native).
//
// A group holds up to four convolutions g with nbx[g] x nby[g] tasks (n tiles x m tiles) of
// chunks[g] = Cin / 16 K chunks each.  XCD x (workgroup id & 7) owns the flat n-major task ids
// [T x / 8, T (x + 1) / 8) of every convolution (T = nbx nby); inside an XCD, slot s (workgroup id
// >> 3) runs, convolution after convolution, the tasks [first[g][s], first[g][s] + count[g][s]) of the
// XCD's list -- clipped to the list's length, which differs by at most one between XCDs.
#pragma once

#if defined(__HIPCC__)
#define W4G_HD __host__ __device__ __forceinline__
#else
#define W4G_HD inline
#endif

constexpr int W4G_MAX_SLOTS = 64;

// G needs: int n; c[g].nbx, c[g].nby, c[g].Cin; unsigned short count[4][64], first[4][64].

// Host: longest-processing-time schedule for the LONGEST per-XCD list of every convolution
// (ceil(T / 8) tasks), convolutions in the order given (the caller sorts them by decreasing K): every
// task goes to the least loaded slot.  Cost of a task in chunk units: its K loop + 2 for the epilogue
// and the hand-over (measured: epilogue 3-5 us, chunk 2.5-5 us).  Returns false when a count does
// not fit 16 bits.
template <class G>
inline bool w4g_make_schedule(G &g, int slots) {
  if (slots < 1) slots = 1;
  if (slots > W4G_MAX_SLOTS) slots = W4G_MAX_SLOTS;
  long load[W4G_MAX_SLOTS] = {0};
  for (int i = 0; i < 4; ++i)
    for (int q = 0; q < W4G_MAX_SLOTS; ++q) g.count[i][q] = g.first[i][q] = 0;
  for (int i = 0; i < g.n; ++i) {
    const long T = (long)g.c[i].nbx * g.c[i].nby;
    const int cnt = (int)((T + 7) / 8);
    const long cost = g.c[i].Cin / 16 + 2;
    int per[W4G_MAX_SLOTS] = {0};
    for (int t = 0; t < cnt; ++t) {
      int best = 0;
      for (int q = 1; q < slots; ++q)
        if (load[q] < load[best]) best = q;
      load[best] += cost;
      ++per[best];
    }
    int acc = 0;
    for (int q = 0; q < W4G_MAX_SLOTS; ++q) {
      if (per[q] > 65535 || acc > 65535) return false;
      g.count[i][q] = (unsigned short)per[q];
      g.first[i][q] = (unsigned short)acc;
      acc += per[q];
    }
  }
  return true;
}

// Device (and CPU test): next task of (xcd, slot) or -1.  A task id packs (convolution g, n tile,
// m tile) as g << 28 | n << 20 | m.  (sg, sk) = position in the slot's schedule, start at (0, 0).
template <class G>
W4G_HD int w4g_next_task(const G &g, int xcd, int slot, int &sg, int &sk) {
  while (sg < g.n) {
    const int nby = g.c[sg].nby;
    const long T = (long)g.c[sg].nbx * nby;
    const int lo = (int)((T * xcd) >> 3), hi = (int)((T * (xcd + 1)) >> 3);
    const int i = g.first[sg][slot] + sk;
    if (sk < g.count[sg][slot] && i < hi - lo) {
      ++sk;
      const int f = lo + i;
      const int n = f / nby;
      return (sg << 28) | (n << 20) | (f - n * nby);
    }
    ++sg;
    sk = 0;
  }
  return -1;
}
