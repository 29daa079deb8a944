// write_order.hip -- store-only model of C4's list stores (65536 nodes x 311 successors, 16 eight-byte rows + one
// four-byte row, node stride 736, non-temporal stores, one wave per node, 4096 waves resident), to separate the
// placement sensitivity of the real kernel (profiles/r02_c4_placement.txt) from its compute:
//   order 0: as the kernel stores -- per 64-successor step all 17 rows (512 B each)
//   order 1: row-outer -- a wave writes a node's whole segment of one row (2.5 KB contiguous) before the next row
// each on several fresh allocations.   build: hipcc --offload-arch=gfx950 -O3 -o write_order write_order.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void stores(double *rows, int *act, long stride, int n_nodes, int S, int count, int order) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long wstride = (long)gridDim.x * 4;
  for (long node = wave; node < n_nodes; node += wstride) {
    const long base = node * (long)S;
    if (order == 0) {
      for (int e0 = 0; e0 < count; e0 += 64) {
        const int e = e0 + lane;
        if (e < ((count + 15) & ~15)) {
#pragma unroll
          for (int f = 0; f < 16; f++) __builtin_nontemporal_store((double)(node + f), &rows[f * stride + base + e]);
          __builtin_nontemporal_store(e, &act[base + e]);
        }
      }
    } else {
      for (int f = 0; f < 16; f++)
        for (int e0 = 0; e0 < count; e0 += 64) {
          const int e = e0 + lane;
          if (e < ((count + 15) & ~15)) __builtin_nontemporal_store((double)(node + f), &rows[f * stride + base + e]);
        }
      for (int e0 = 0; e0 < count; e0 += 64) {
        const int e = e0 + lane;
        if (e < ((count + 15) & ~15)) __builtin_nontemporal_store(e, &act[base + e]);
      }
    }
  }
}

int main(int argc, char **argv) {
  const int n_nodes = 65536, S = 736, count = 311;
  const int allocs = argc > 1 ? atoi(argv[1]) : 8;
  const long stride = (long)n_nodes * S;
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  const double bytes = (double)n_nodes * ((count + 15) & ~15) * (16 * 8 + 4);
  for (int al = 0; al < allocs; al++) {
    double *rows;
    int *act;
    if (hipMalloc(&rows, stride * 16 * 8) != hipSuccess || hipMalloc(&act, stride * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    printf("alloc %d:", al);
    for (int order = 0; order < 2; order++) {
      for (int rep = 0; rep < 40; rep++) stores<<<256 * 4, 256>>>(rows, act, stride, n_nodes, S, count, order);  // clocks
      (void)hipEventRecord(a);
      for (int rep = 0; rep < 20; rep++) stores<<<256 * 4, 256>>>(rows, act, stride, n_nodes, S, count, order);
      (void)hipEventRecord(b);
      (void)hipEventSynchronize(b);
      float ms;
      (void)hipEventElapsedTime(&ms, a, b);
      printf("  order %d %.4f ms (%.2f TB/s)", order, ms / 20, bytes / (ms / 20) / 1e9);
    }
    printf("\n");
    (void)hipFree(rows);
    (void)hipFree(act);
  }
  return 0;
}
