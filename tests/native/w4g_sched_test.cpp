// CPU test of the task scheduling of the persistent grouped F(4x4) kernel: compiles the SAME header the
// HIP kernel uses (shapy_amd/csrc/conv_wino4g_sched.h) with g++ and replays every (XCD, slot) of a
// launch.  Checks: every task of every convolution is run exactly once; slots beyond `slots` get
// nothing; the schedule is balanced (no slot exceeds the mean by more than one longest task).
//   g++ -O2 -std=c++17 -I shapy_amd/csrc tests/native/w4g_sched_test.cpp -o w4g_sched_test && ./w4g_sched_test
#include <stdio.h>
#include <stdlib.h>

#include <map>
#include <vector>

#include "conv_wino4g_sched.h"

struct Conv { int nbx, nby, Cin; };
struct Group {
  Conv c[4];
  int n;
  unsigned short count[4][64], first[4][64];
};

static int check(const char *name, std::vector<Conv> convs, int slots) {
  Group g;
  g.n = (int)convs.size();
  for (int i = 0; i < g.n; ++i) g.c[i] = convs[i];
  // the launcher lists the convolution with the longest K loop first (stable)
  for (int i = 1; i < g.n; ++i)
    for (int j = i; j > 0 && g.c[j].Cin > g.c[j - 1].Cin; --j) { Conv t = g.c[j]; g.c[j] = g.c[j - 1]; g.c[j - 1] = t; }
  long tasks = 0;
  for (int i = 0; i < g.n; ++i) tasks += (long)g.c[i].nbx * g.c[i].nby;
  const long per_xcd = (tasks + 7) / 8;
  if (slots > per_xcd) slots = (int)per_xcd;
  if (!w4g_make_schedule(g, slots)) { printf("%s: schedule refused\n", name); return 1; }
  std::map<int, int> seen;
  long worst = 0, total_load = 0;
  for (int xcd = 0; xcd < 8; ++xcd)
    for (int slot = 0; slot < 64; ++slot) {
      int sg = 0, sk = 0, last_g = 0;
      long load = 0;
      for (;;) {
        const int t = w4g_next_task(g, xcd, slot, sg, sk);
        if (t < 0) break;
        if (slot >= slots) { printf("%s: slot %d beyond %d got a task\n", name, slot, slots); return 1; }
        const int cg = t >> 28, n = (t >> 20) & 0xff, m = t & 0xfffff;
        if (cg < last_g) { printf("%s: convolutions out of order\n", name); return 1; }
        last_g = cg;
        if (cg >= g.n || n >= g.c[cg].nbx || m >= g.c[cg].nby) { printf("%s: task out of range\n", name); return 1; }
        // XCD affinity: the task lies in the XCD's own run of the n-major list
        const long T = (long)g.c[cg].nbx * g.c[cg].nby, f = (long)n * g.c[cg].nby + m;
        if (f < ((T * xcd) >> 3) || f >= ((T * (xcd + 1)) >> 3)) { printf("%s: task on the wrong XCD\n", name); return 1; }
        ++seen[t];
        load += g.c[cg].Cin / 16 + 2;
      }
      if (load > worst) worst = load;
      total_load += load;
    }
  for (int i = 0; i < g.n; ++i)
    for (int n = 0; n < g.c[i].nbx; ++n)
      for (int m = 0; m < g.c[i].nby; ++m) {
        const int t = (i << 28) | (n << 20) | m;
        if (seen[t] != 1) { printf("%s: task (%d,%d,%d) run %d times\n", name, i, n, m, seen[t]); return 1; }
      }
  if ((long)seen.size() != tasks) { printf("%s: %zu distinct tasks, expected %ld\n", name, seen.size(), tasks); return 1; }
  long longest = 0;
  for (int i = 0; i < g.n; ++i) if (g.c[i].Cin / 16 + 2 > longest) longest = g.c[i].Cin / 16 + 2;
  const double mean = (double)total_load / (8.0 * slots);
  if (worst > mean + longest + 1e-9) { printf("%s: unbalanced: worst slot %ld, mean %.1f, longest task %ld\n", name, worst, mean, longest); return 1; }
  printf("%-34s tasks %6ld  slots %2d  worst slot %4ld  mean %7.1f chunk units\n", name, tasks, slots, worst, mean);
  return 0;
}

static Conv conv(int B, int H, int C, int O) {
  const int tiles = B * ((H + 3) / 4) * ((H + 3) / 4);
  return Conv{O / 48, (tiles + 15) / 16, C};
}

int main() {
  int bad = 0;
  for (int B : {1, 3, 64, 334}) {
    char nm[64];
    snprintf(nm, sizeof nm, "stage4 level B=%d", B);
    bad += check(nm, {conv(B, 56, 48, 48), conv(B, 28, 96, 96), conv(B, 14, 192, 192), conv(B, 7, 384, 384)}, 64);
    snprintf(nm, sizeof nm, "stage3 level B=%d", B);
    bad += check(nm, {conv(B, 56, 48, 48), conv(B, 28, 96, 96), conv(B, 14, 192, 192)}, 64);
    snprintf(nm, sizeof nm, "stage2 level B=%d", B);
    bad += check(nm, {conv(B, 56, 48, 48), conv(B, 28, 96, 96)}, 64);
  }
  bad += check("one tiny layer", {conv(1, 8, 16, 48)}, 64);
  bad += check("Cout 144 (3 n tiles)", {conv(3, 20, 96, 144), conv(5, 9, 48, 96)}, 64);
  bad += check("256->48 transition", {conv(64, 56, 256, 48)}, 64);
  bad += check("few slots (small chip)", {conv(64, 56, 48, 48), conv(64, 7, 384, 384)}, 10);
  bad += check("384 channels, 8 n tiles", {conv(64, 7, 384, 384)}, 64);
  if (bad) { printf("FAILED: %d case(s)\n", bad); return 1; }
  printf("W4G SCHEDULE OK\n");
  return 0;
}
