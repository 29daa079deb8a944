// write_explore.hip -- variations of the store-only model of C4's list stores (see write_order.hip), per allocation:
// which properties of the pattern move its common mode (5.3 TB/s) towards the rare fast one (6.8 TB/s)?
// build: hipcc --offload-arch=gfx950 -O3 -o write_explore write_explore.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <bool NT>
__global__ __launch_bounds__(256) void stores(double *rows, int *act, long stride, int n_nodes, int S, int count, int blocked, int nrows) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long W = (long)gridDim.x * 4;
  const long per = (n_nodes + W - 1) / W;
  for (long it = 0; it < per; it++) {
    const long node = blocked ? wave * per + it : wave + it * W;
    if (node >= n_nodes) break;
    const long base = node * (long)S;
    for (int e0 = 0; e0 < count; e0 += 64) {
      const int e = e0 + lane;
      if (e < ((count + 15) & ~15)) {
        for (int f = 0; f < nrows; f++) {
          if (NT) __builtin_nontemporal_store((double)(node + f), &rows[f * stride + base + e]);
          else rows[f * stride + base + e] = (double)(node + f);
        }
        if (NT) __builtin_nontemporal_store(e, &act[base + e]);
        else act[base + e] = e;
      }
    }
  }
}

int main(int argc, char **argv) {
  const int n_nodes = 65536, count = 311;
  const int allocs = argc > 1 ? atoi(argv[1]) : 5;
  struct V { const char *name; int S, wgs, blocked, nt, nrows; long pad; } v[] = {
      {"base S=736 16w/CU", 736, 4, 0, 1, 16, 0},  {"S=768", 768, 4, 0, 1, 16, 0},       {"S=1024", 1024, 4, 0, 1, 16, 0},
      {"blocked nodes", 736, 4, 1, 1, 16, 0},      {"8 waves/CU", 736, 2, 0, 1, 16, 0},  {"32 waves/CU", 736, 8, 0, 1, 16, 0},
      {"plain stores", 736, 4, 0, 0, 16, 0},       {"8 rows", 736, 4, 0, 1, 8, 0},       {"4 rows", 736, 4, 0, 1, 4, 0},
      {"1 row", 736, 4, 0, 1, 1, 0},               {"row pad 4100", 736, 4, 0, 1, 16, 4100}};
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  for (int al = 0; al < allocs; al++) {
    const long maxstride = (long)n_nodes * 1024 + 8192;
    double *rows;
    int *act;
    if (hipMalloc(&rows, maxstride * 16 * 8) != hipSuccess || hipMalloc(&act, maxstride * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    printf("alloc %d:", al);
    for (auto &x : v) {
      const long stride = (long)n_nodes * x.S + x.pad;
      const double bytes = (double)n_nodes * ((count + 15) & ~15) * (x.nrows * 8 + 4);
      for (int rep = 0; rep < 25; rep++) {
        if (rep == 5) (void)hipEventRecord(a);
        if (x.nt) stores<true><<<256 * x.wgs, 256>>>(rows, act, stride, n_nodes, x.S, count, x.blocked, x.nrows);
        else stores<false><<<256 * x.wgs, 256>>>(rows, act, stride, n_nodes, x.S, count, x.blocked, x.nrows);
      }
      (void)hipEventRecord(b);
      (void)hipEventSynchronize(b);
      float ms;
      (void)hipEventElapsedTime(&ms, a, b);
      printf(" | %s %.2f", x.name, bytes / (ms / 20) / 1e9);
    }
    printf("  (TB/s)\n");
    (void)hipFree(rows);
    (void)hipFree(act);
  }
  return 0;
}
