// write_pattern.hip -- how fast can an MI355X absorb the successor-list stores of C4, as a function of
// the layout?  Pure stores, no compute: one wave per node writes `count` successors (14 state rows + hash
// + cost as f64, action as i32) in 64-successor steps.
//   mode 0: the shipped layout, field-major rows, node k at k*S (S = 736), count = 311 -> gaps
//   mode 1: same rows, nodes packed back to back (CSR-like, start rounded up to 16 entries)
//   mode 2: pure streaming fill of the same number of bytes (upper bound)
//   mode 3: mode 0 with count = S (no gaps, 2.4x the bytes) -- shows the cost of the gaps
// build: hipcc --offload-arch=gfx950 -O3 -o write_pattern write_pattern.hip ; run: ./write_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(256) void pattern(double *rows, int *act, long stride, int n_nodes, int S, int count, int mode,
                                               const int *offs) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long wstride = (long)gridDim.x * 4;
  const int G = (mode >= 5 && mode <= 7) ? (mode - 3) : (mode == 9 ? 2 : mode == 10 ? 4 : mode == 11 ? 8 : 1);  // modes 5, 6, 7: a wave takes 2, 3, 4 consecutive nodes at a time
  for (long it = wave; it * G < n_nodes; it += wstride)
  for (int gi = 0; gi < G; gi++) {
    const long node = it * G + gi;
    if (node >= n_nodes) break;
    const long base = (mode == 1 || mode >= 9) ? offs[node] : node * (long)S;
    for (int e0 = 0; e0 < count; e0 += 64) {
      const int e = e0 + lane;
      if (e < count) {
        const long idx = base + e;
        if (mode == 8) {
#pragma unroll
          for (int f = 0; f < 16; f++) __builtin_nontemporal_store((double)(node + f), &rows[f * stride + idx]);
          __builtin_nontemporal_store(e, &act[idx]);
        } else {
#pragma unroll
          for (int f = 0; f < 16; f++) rows[f * stride + idx] = (double)(node + f);
          act[idx] = e;
        }
      }
    }
  }
}

// mode 4: the layout of mode 0, but a lane owns two consecutive successors and stores 16 bytes at a time
__global__ __launch_bounds__(256) void pattern2(double *rows, int *act, long stride, int n_nodes, int S, int count) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long wstride = (long)gridDim.x * 4;
  for (long node = wave; node < n_nodes; node += wstride) {
    const long base = node * (long)S;
    for (int e0 = 0; e0 < count; e0 += 128) {
      const int e = e0 + 2 * lane;
      if (e + 1 < count) {
        const long idx = base + e;
#pragma unroll
        for (int f = 0; f < 16; f++) *(double2 *)(rows + f * stride + idx) = make_double2((double)(node + f), 1.0);
        *(int2 *)(act + idx) = make_int2(e, e + 1);
      } else if (e < count) {
        const long idx = base + e;
#pragma unroll
        for (int f = 0; f < 16; f++) rows[f * stride + idx] = (double)(node + f);
        act[idx] = e;
      }
    }
  }
}

__global__ void fill(double *p, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = 1.0;
}

int main(int argc, char **argv) {
  const int n_nodes = 65536, S = 736;
  const int count = argc > 3 ? atoi(argv[3]) : 311;
  const long pad = argc > 1 ? atol(argv[1]) : 0;
  const int wgs_per_cu = argc > 2 ? atoi(argv[2]) : 4;  // 4 waves each  // extra doubles between rows (de-aligns the 2^24-byte row stride)
  const long stride = (long)n_nodes * S + pad;
  double *rows;
  int *act, *offs;
  (void)hipMalloc(&rows, stride * 16 * 8);
  printf("row stride %ld doubles (pad %ld)\n", stride, pad);
  (void)hipMalloc(&act, stride * 4);
  (void)hipMalloc(&offs, n_nodes * 4);
  std::vector<int> h(n_nodes);
  long o = 0;
  for (int k = 0; k < n_nodes; k++) { h[k] = (int)o; o += (count + 15) & ~15; }
  (void)hipMemcpy(offs, h.data(), n_nodes * 4, hipMemcpyHostToDevice);
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  for (int mode = 0; mode < 12; mode++) {
    const int cnt = mode == 3 ? S : count;
    const double bytes = (double)n_nodes * cnt * (16 * 8 + 4);
    float best = 1e9;
    for (int rep = 0; rep < 6; rep++) {
      (void)hipEventRecord(a);
      if (mode == 2) fill<<<256 * 8, 256>>>(rows, (long)(bytes / 8));
      else if (mode == 4) pattern2<<<256 * wgs_per_cu, 256>>>(rows, act, stride, n_nodes, S, cnt);
      else pattern<<<256 * wgs_per_cu, 256>>>(rows, act, stride, n_nodes, S, cnt, mode, offs);
      (void)hipEventRecord(b);
      (void)hipEventSynchronize(b);
      float ms;
      (void)hipEventElapsedTime(&ms, a, b);
      if (rep > 0 && ms < best) best = ms;
    }
    printf("mode %d: %.3f ms  %.2f GB  %.2f TB/s\n", mode, best, bytes / 1e9, bytes / best / 1e9);
  }
  return 0;
}
