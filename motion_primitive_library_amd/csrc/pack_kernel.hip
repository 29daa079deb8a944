// pack_kernel.hip -- packs the used prefixes of per-node successor lists (mplx_succ_lists layout: node k owns
// entries [k*S, k*S + count[k]) of every row) into one contiguous block per row, for the copy back to host memory
// of mplx_expand_lists (lists_copy_api.cpp).  The lists themselves follow the reference's get_succ outputs
// (include/mpl_planner/env/env_map.h:147-172: succ, succ_cost, action_idx); only count[k] entries per node carry
// information, so only those cross PCIe.  Pure data movement.
#include "mplx_internal.h"

namespace mplx {
namespace {

__global__ __launch_bounds__(256) void pack_rows_kernel(const PackArgs A) {
  const int64_t k = A.node0 + blockIdx.x;
  const int cnt = A.count[k];
  const int64_t src0 = k * A.node_stride;
  const int64_t dst0 = A.offs[k] - A.off0;
  for (int r = 0; r < A.n_rows; r++) {
    if (A.es[r] == 8) {
      const uint64_t *s = (const uint64_t *)A.src[r] + src0;
      uint64_t *d = (uint64_t *)(A.dst + A.dst_off[r]) + dst0;
      for (int e = threadIdx.x; e < cnt; e += 256) d[e] = s[e];
    } else {
      const uint32_t *s = (const uint32_t *)A.src[r] + src0;
      uint32_t *d = (uint32_t *)(A.dst + A.dst_off[r]) + dst0;
      for (int e = threadIdx.x; e < cnt; e += 256) d[e] = s[e];
    }
  }
}

// Exclusive prefix sums of the per-node counts: offs[k] = count[0] + ... + count[k-1], offs[n] = total.  One
// workgroup: each thread sums a contiguous slice, the slice sums are scanned through LDS, then each thread
// writes its slice (n is a frontier size: at most a few million, i.e. microseconds).
__global__ __launch_bounds__(1024) void scan_counts_kernel(const int32_t *count, int64_t n, int64_t *offs) {
  __shared__ int64_t part[1024];
  const int t = threadIdx.x;
  const int64_t per = (n + 1023) / 1024;
  const int64_t a = (int64_t)t * per, b = (a + per < n) ? a + per : n;
  int64_t s = 0;
  for (int64_t k = a; k < b; k++) s += count[k];
  part[t] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int64_t v = t >= d ? part[t - d] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int64_t run = part[t] - s;  // exclusive
  for (int64_t k = a; k < b; k++) {
    offs[k] = run;
    run += count[k];
  }
  if (t == 1023) offs[n] = part[1023];
}

}  // namespace

hipError_t launch_scan_counts(const int32_t *count, int64_t n, int64_t *offs, hipStream_t stream) {
  hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(1024), 0, stream, count, n, offs);
  return hipGetLastError();
}

hipError_t launch_pack_rows(const PackArgs &a, int64_t n_nodes, hipStream_t stream) {
  if (n_nodes <= 0) return hipSuccess;
  hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)n_nodes), dim3(256), 0, stream, a);
  return hipGetLastError();
}

}  // namespace mplx
