// write_layout.hip -- round 4: does a RECORD-MAJOR successor list (the reference's own vec_E<Waypoint<Dim>>, one
// 112-byte Waypoint per successor; with hash and cost exactly one 128-byte line) change the slow placement mode of
// C4's list stores?  Store-only model, per allocation (the mode is a property of the allocation): the SAME buffer
// is written under every layout, so a row of the table compares layouts on one placement.
//   F   field-major, the shipped layout: 16 eight-byte rows + one four-byte row, node k at k * S
//   Rn  record-major, lane = successor: eight 16-byte stores per lane at a 128-byte lane stride (naive)
//   Rt  record-major, transposed: store i of a 64-successor block writes bytes [1024 i, 1024 i + 1024) -- every
//       instruction covers eight whole lines (what an LDS transpose in the real kernel would produce)
//   F2  field-major, lane = two consecutive successors: 16-byte stores, 1 KB contiguous per row and instruction
//   fill a linear fill of the same number of bytes with 16-byte stores (the ceiling of this allocation)
//   R7  Rt with 112-byte records (state only; hash, cost, action stay field-major rows): 7 stores of 1 KB per block
// All stores `sc1 nt` like the kernel's; counts per node 150 .. 470 (mean ~311) rounded up to 16; one wave per node,
// 1024 workgroups of 4 waves, nodes strided.
// build: hipcc --offload-arch=gfx950 -O3 -o write_layout write_layout.hip ; run: ./write_layout [allocations]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ int count_of(long node) {
  unsigned h = (unsigned)node * 2654435761u;
  h ^= h >> 15;
  return 150 + (int)(h % 321u);
}
__device__ __forceinline__ void st8(double v, double *p) { asm volatile("global_store_dwordx2 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st4(int v, int *p) { asm volatile("global_store_dword %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory"); }
typedef double d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st16(d2v v, void *p) { asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory"); }

__global__ __launch_bounds__(256) void stores(char *buf, long stride, int n_nodes, int S, int layout) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long W = (long)gridDim.x * 4;
  double *rows = (double *)buf;                  // F: 16 rows of `stride` doubles, then the action row
  int *act = (int *)(buf + stride * 16 * 8);
  for (long node = wave; node < n_nodes; node += W) {
    const int cpad = (count_of(node) + 15) & ~15;
    const long base = node * (long)S;
    for (int e0 = 0; e0 < cpad; e0 += 64) {
      const int e = e0 + lane;
      const int live = cpad - e0 < 64 ? cpad - e0 : 64;  // successors of this block
      if (layout == 0) {
        if (e < cpad) {
#pragma unroll
          for (int f = 0; f < 16; f++) st8((double)(node + f), &rows[f * stride + base + e]);
          st4(e, &act[base + e]);
        }
      } else if (layout == 1) {
        if (e < cpad) {
          char *r = buf + (base + e) * 128;
#pragma unroll
          for (int i = 0; i < 8; i++) st16(d2v{(double)node, (double)i}, r + 16 * i);
          st4(e, &act[base + e]);
        }
      } else if (layout == 2) {
        char *blk = buf + (base + e0) * 128;
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const int off = i * 1024 + lane * 16;
          if (off < live * 128) st16(d2v{(double)node, (double)i}, blk + off);
        }
        if (e < cpad) st4(e, &act[base + e]);
      } else if (layout == 4) {
        // F2: field-major rows, a lane owns TWO consecutive successors of a 128-successor block: one 16-byte store per row
        // and lane = 1 KB contiguous per row-store (and half the store instructions of F)
        if ((e0 & 64) == 0) {
          const int e2 = e0 + 2 * lane;
          if (e2 < cpad) {
#pragma unroll
            for (int f = 0; f < 16; f++) st16(d2v{(double)(node + f), 1.0}, &rows[f * stride + base + e2]);
            asm volatile("global_store_dwordx2 %0, %1, off sc1 nt" ::"v"(&act[base + e2]), "v"(d2v{1.0, 2.0}.x) : "memory");
          }
        }
      } else if (layout == 3) {
        // 112-byte records of the block back to back (block start = (base + e0) * 112: 16-byte aligned since base and e0
        // are multiples of 16), hash + cost as two field-major rows behind the record area
        char *blk = buf + (base + e0) * 112;
        double *hrow = (double *)(buf + stride * 112), *crow = hrow + stride;
#pragma unroll
        for (int i = 0; i < 7; i++) {
          const int off = i * 1024 + lane * 16;
          if (off < live * 112) st16(d2v{(double)node, (double)i}, blk + off);
        }
        if (e < cpad) {
          st8(1.0, &hrow[base + e]);
          st8(2.0, &crow[base + e]);
          st4(e, &act[base + e]);
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void fill16(char *buf, long n16) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) st16(d2v{1.0, 2.0}, buf + i * 16);
}

int main(int argc, char **argv) {
  const int n_nodes = 65536, S = 736;
  const int allocs = argc > 1 ? atoi(argv[1]) : 8;
  const long stride = (long)n_nodes * S;
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  double bytes = 0;
  for (long k = 0; k < n_nodes; k++) {
    unsigned h = (unsigned)k * 2654435761u;
    h ^= h >> 15;
    bytes += (double)(((150 + (int)(h % 321u)) + 15) & ~15) * 132.0;
  }
  const char *name[6] = {"F", "Rn", "Rt", "R7", "F2", "fill"};
  for (int al = 0; al < allocs; al++) {
    char *buf;
    if (hipMalloc(&buf, stride * (16 * 8 + 4)) != hipSuccess) { printf("alloc failed\n"); return 1; }
    printf("alloc %d:", al);
    for (int layout = 0; layout < 6; layout++) {
      if (layout == 1) continue;  // (Rn: 8.4 ms, measured once)
      auto go = [&]() {
        if (layout == 5) fill16<<<256 * 8, 256>>>(buf, (long)(bytes / 16));
        else stores<<<256 * 4, 256>>>(buf, stride, n_nodes, S, layout);
      };
      for (int rep = 0; rep < 40; rep++) go();  // clocks
      (void)hipEventRecord(a);
      for (int rep = 0; rep < 20; rep++) go();
      (void)hipEventRecord(b);
      (void)hipEventSynchronize(b);
      float ms;
      (void)hipEventElapsedTime(&ms, a, b);
      printf("  %s %.4f ms (%.2f TB/s)", name[layout], ms / 20, bytes / (ms / 20) / 1e9);
    }
    printf("\n");
    fflush(stdout);
    (void)hipFree(buf);
  }
  return 0;
}
