// store_model_kernel.hip -- DIAGNOSTIC (mplx_debug_store_model, include/mplx.h): what do the list stores of an expansion
// launch cost on their own, in THIS allocation of the lists?  Round 4 found that the headline launch (C4: 2.75 GB of list
// entries) is exactly as long as writing them takes, and that this time is a property of the memory behind the allocation
// (5.9 against 5.15 TB/s for the same pattern, profiles/r04_store_layouts_vs_kernel.txt).  This kernel writes, for every
// node k, the first count[k] entries of every row present in the lists -- rounded up to whole 128-byte lines like the
// expansion kernels do -- with unspecified values, in the same order (a wave per node, chunks of nodes dealt over the
// waves, `sc1 nt` stores): bench.py reports its time next to the kernel's (`roofline.store_only_ms`).
// It OVERWRITES the successor entries (count[] stays): call it when the results have been consumed.
#include "mplx_internal.h"

namespace mplx {
namespace {

template <typename T>
__device__ __forceinline__ void sm_st(T v, T *p) {
  if constexpr (sizeof(T) == 8) asm volatile("global_store_dwordx2 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dword %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
}

// mode (experiments, profiles/micro/store_model_modes.py): bits 0-7 nodes per chunk (0 = 4); bit 8: ROW-major inside a node
// (a row's whole segment, then the next row) instead of step-major (64 entries of every row, then the next 64)
__global__ __launch_bounds__(256) void store_model_kernel(const int32_t *count, int64_t n_nodes, int64_t S, int32_t *action,
                                                          double *cost, uint64_t *hash, double *state, int64_t state_stride,
                                                          int n_fields, int pad, int mode) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t W = (int64_t)gridDim.x * 4;
  const int kChunk = (mode & 255) ? (mode & 255) : 4;
  for (int64_t c0 = wave * kChunk; c0 < n_nodes; c0 += W * kChunk)
    for (int64_t node = c0; node < c0 + kChunk && node < n_nodes; node++) {
      // (count[] comes from device memory the caller filled: a stale or garbage value must not write past the node's
      // segment of S entries, nor may the line-completing round-up when S is not a multiple of 32)
      int E = count[node];
      E = E < 0 ? 0 : (E > S ? (int)S : E);
      int e16 = pad ? (E + 15) & ~15 : E, e32 = pad ? (E + 31) & ~31 : E;
      e16 = e16 > S ? (int)S : e16;
      e32 = e32 > S ? (int)S : e32;
      const int64_t base = node * S;
      if (mode & 256) {
        if (action) for (int e = lane; e < e32; e += 64) sm_st((int32_t)e, &action[base + e]);
        if (hash) for (int e = lane; e < e16; e += 64) sm_st((uint64_t)node, &hash[base + e]);
        if (state)
          for (int f = 0; f < n_fields; f++)
            for (int e = lane; e < e16; e += 64) sm_st((double)f, &state[(int64_t)f * state_stride + base + e]);
        if (cost) for (int e = lane; e < e16; e += 64) sm_st(1.0, &cost[base + e]);
        continue;
      }
      const bool hybrid = (mode & 512) != 0;  // bit 9: action / hash / cost step-major, then the state rows row-major
      for (int e = lane; e < e32; e += 64) {
        if (action) sm_st((int32_t)e, &action[base + e]);
        if (e < e16) {
          if (hash) sm_st((uint64_t)node, &hash[base + e]);
          if (state && !hybrid)
            for (int f = 0; f < n_fields; f++) sm_st((double)f, &state[(int64_t)f * state_stride + base + e]);
          if (cost) sm_st(1.0, &cost[base + e]);
        }
      }
      if (state && hybrid)
        for (int f = 0; f < n_fields; f++)
          for (int e = lane; e < e16; e += 64) sm_st((double)f, &state[(int64_t)f * state_stride + base + e]);
    }
}

}  // namespace

hipError_t launch_store_model(const int32_t *count, int64_t n_nodes, int64_t S, int32_t *action, double *cost, uint64_t *hash,
                              double *state, int64_t state_stride, int n_fields, int pad, int blocks, int mode, hipStream_t s) {
  if (n_nodes == 0) return hipSuccess;
  hipLaunchKernelGGL(store_model_kernel, dim3((unsigned)blocks), dim3(256), 0, s, count, n_nodes, S, action, cost, hash, state,
                     state_stride, n_fields, pad, mode);
  return hipGetLastError();
}

}  // namespace mplx
