// write_layout_lib.hip -- the store-only layouts of write_layout.hip as a library, so that profiles/micro/write_layout_vs_kernel.py
// can run them on the SAME state-row allocation the real kernel has just been timed on (is the slow placement mode of the
// kernel visible to pure stores, and does a layout remove it?).  State rows only: 14 doubles per successor.
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o write_layout_lib.so write_layout_lib.hip
#include <hip/hip_runtime.h>

typedef double d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st8(double v, double *p) { asm volatile("global_store_dwordx2 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st16(d2v v, void *p) { asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ int count_of(long node) {
  unsigned h = (unsigned)node * 2654435761u;
  h ^= h >> 15;
  return 150 + (int)(h % 321u);
}

// layout 0: F (14 field-major rows, 8-byte stores)  1: F2 (two successors per lane, 16-byte stores)
//        2: R7t (112-byte records, transposed: 7 stores of 1 KB per 64-successor block)  3: linear fill of the same bytes
__global__ __launch_bounds__(256) void stores(char *buf, long stride, int n_nodes, int S, int layout, int chunk) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long W = (long)gridDim.x * 4;
  double *rows = (double *)buf;
  // nodes dealt in chunks of `chunk` consecutive nodes per wave (the kernel: 4), waves striding over the chunks
  for (long c0 = wave * chunk; c0 < n_nodes; c0 += W * chunk)
  for (long node = c0; node < c0 + chunk && node < n_nodes; node++) {
    const int cpad = (count_of(node) + 15) & ~15;
    const long base = node * (long)S;
    if (layout == 0) {
      for (int e = lane; e < cpad; e += 64) {
#pragma unroll
        for (int f = 0; f < 14; f++) st8((double)(node + f), &rows[f * stride + base + e]);
      }
    } else if (layout == 1) {
      for (int e = 2 * lane; e < cpad; e += 128) {
#pragma unroll
        for (int f = 0; f < 14; f++) st16(d2v{(double)(node + f), 1.0}, &rows[f * stride + base + e]);
      }
    } else if (layout == 2) {
      for (int e0 = 0; e0 < cpad; e0 += 64) {
        const int live = cpad - e0 < 64 ? cpad - e0 : 64;
        char *blk = buf + (base + e0) * 112;
#pragma unroll
        for (int i = 0; i < 7; i++) {
          const int off = i * 1024 + lane * 16;
          if (off < live * 112) st16(d2v{(double)node, (double)i}, blk + off);
        }
      }
    }
  }
}
__global__ __launch_bounds__(256) void fill16(char *buf, long n16) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) st16(d2v{1.0, 2.0}, buf + i * 16);
}

extern "C" float run_layout(void *buf, long stride, int n_nodes, int S, int layout, int chunk, int reps) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  double bytes = 0;
  for (long k = 0; k < n_nodes; k++) {
    unsigned h = (unsigned)k * 2654435761u;
    h ^= h >> 15;
    bytes += (double)(((150 + (int)(h % 321u)) + 15) & ~15) * 112.0;
  }
  auto go = [&]() {
    if (layout == 3) fill16<<<256 * 8, 256>>>((char *)buf, (long)(bytes / 16));
    else stores<<<256 * 4, 256>>>((char *)buf, stride, n_nodes, S, layout, chunk);
  };
  for (int r = 0; r < 30; r++) go();
  (void)hipEventRecord(a);
  for (int r = 0; r < reps; r++) go();
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  return ms / reps;
}
