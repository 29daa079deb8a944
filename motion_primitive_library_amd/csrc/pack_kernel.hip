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

}  // namespace

hipError_t launch_pack_rows(const PackArgs &a, int64_t n_nodes, hipStream_t stream) {
  if (n_nodes <= 0) return hipSuccess;
  hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)n_nodes), dim3(256), 0, stream, a);
  return hipGetLastError();
}

}  // namespace mplx
