// post_kernel.hip -- what the graph search does with every successor right after
// get_succ, batched on the device, reading the per-node successor
// lists where the expansion kernels left them in HBM:
//   heuristic       env_base<Dim>::get_heur / cal_heur with heur_ignore_dynamics
//                   (reference include/mpl_planner/common/env_base.h:46-64):
//                   0 if the successor is the goal's lattice state, else
//                   w * |pos - goal.pos|_inf / v_max
//   goal tolerance  env_map<Dim>::is_goal, the norm tests
//                   (include/mpl_planner/env/env_map.h:25-37); the ray trace of
//                   :38-43 needs the host's map walk and stays with the caller
//   node identity   the search keys its hash map with the Waypoint, whose ==
//                   compares hash values (waypoint.h:128-135;
//                   graph_search.h:84-88): for every successor the list index of
//                   the FIRST successor of the batch with the same hash, so the
//                   host creates each new state once and only transfers the rest
//                   as edges.  Deterministic: the canonical duplicate is the one
//                   with the smallest list index.
// The identity pass of SMALL batches (below ~256 k list slots: the batches of a search) is an open-addressing table
// in HBM keyed by the 64-bit lattice hash (linear probing, 64-bit CAS for the key, 32-bit atomicMin for the index;
// key and index share one 16-byte slot, so each successor touches one line per pass): two launches, latency bound.
// Large batches (a whole frontier's lists, the gathered lists of all GPUs) take identity_kernel.hip: radix
// partition + per-bucket tables in LDS, the same canon[] at memory bandwidth.
#include "mplx_internal.h"

namespace mplx {
namespace {

constexpr uint64_t kEmpty = ~0ull;

__device__ __forceinline__ uint64_t mix(uint64_t h) {  // table position only; never leaves the device
  h ^= h >> 33;
  h *= 0xff51afd7ed558ccdULL;
  h ^= h >> 33;
  return h;
}

template <int D>
__global__ __launch_bounds__(256) void post_lists_kernel(const PostArgs A) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= A.n_nodes * A.nstride) return;
  const int64_t node = g / A.nstride;
  const int j = (int)(g - node * A.nstride);
  if (j >= A.count[node]) return;
  const uint64_t h = A.hash[g];
  const bool is_goal_state = (h == A.goal_hash);  // `goal_node_ == state` is a hash comparison (env_base.h:47)
  double m = 0;  // lpNorm<Infinity> of pos - goal.pos
  if (A.state) {
#pragma unroll
    for (int i = 0; i < D; i++) {
      const double d = fabs(A.state[(int64_t)i * A.sstride + g] - A.goal[i]);
      m = d > m ? d : m;
    }
  }
  if (A.heur) A.heur[g] = is_goal_state ? 0.0 : (A.v_max > 0 ? A.w * m / A.v_max : A.w * m);
  if (A.flags && !A.state) {
    // lists without state rows: `flags` was written by the expansion launch (mplx_succ_lists::flags, bits 0 - 1);
    // only the first-occurrence bit is added (the table route does the same in post_canon_kernel)
    if (!A.keys && A.canon && A.canon[g] == (int32_t)g) A.flags[g] |= 4;
  } else if (A.flags) {
    bool goaled = m <= A.tol_pos;  // env_map.h:26-28
    if (goaled && A.tol_vel >= 0) {
      double mv = 0;
#pragma unroll
      for (int i = 0; i < D; i++) {
        const double d = fabs(A.state[(int64_t)(D + i) * A.sstride + g] - A.goal[D + i]);
        mv = d > mv ? d : mv;
      }
      goaled = mv <= A.tol_vel;
    }
    if (goaled && A.tol_acc >= 0) {
      double ma = 0;
#pragma unroll
      for (int i = 0; i < D; i++) {
        const double d = fabs(A.state[(int64_t)(2 * D + i) * A.sstride + g] - A.goal[2 * D + i]);
        ma = d > ma ? d : ma;
      }
      goaled = ma <= A.tol_acc;
    }
    if (goaled && A.tol_yaw >= 0) goaled = fabs(A.state[(int64_t)(4 * D) * A.sstride + g] - A.goal[4 * D]) <= A.tol_yaw;
    // bit 2 (first occurrence of the lattice state in the batch): from canon[] when the partitioned identity pass
    // (identity_kernel.hip) ran before this launch; the table route sets it in post_canon_kernel
    const bool first = !A.keys && A.canon && A.canon[g] == (int32_t)g;
    A.flags[g] = (uint8_t)((goaled ? 1 : 0) | (is_goal_state ? 2 : 0) | (first ? 4 : 0));
  }
  if (A.keys) {
    if (h == kEmpty) {  // the one hash the key field cannot hold: a dedicated slot past the table
      atomicMin(&A.keys[A.cap].val, (uint32_t)g);
    } else {
      uint64_t s = mix(h) & (A.cap - 1);
      while (true) {
        const uint64_t old = atomicCAS((unsigned long long *)&A.keys[s].key, (unsigned long long)kEmpty, (unsigned long long)h);
        if (old == kEmpty || old == h) break;
        s = (s + 1) & (A.cap - 1);
      }
      atomicMin(&A.keys[s].val, (uint32_t)g);
    }
  }
}

// Lists whose flags row the expansion launch wrote (mplx_succ_lists::flags), node identity by the partition
// (identity_kernel.hip): all that is left to do is the first-occurrence bit -- count and canon are read, nothing else.
__global__ __launch_bounds__(256) void post_first_flags_kernel(const PostArgs A) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= A.n_nodes * A.nstride) return;
  const int64_t node = g / A.nstride;
  const int j = (int)(g - node * A.nstride);
  if (j >= A.count[node]) return;
  if (A.canon[g] == (int32_t)g) A.flags[g] |= 4;
}

// ... and its undo, for the one case in which the bit was set from a canon[] that turned out wrong (the claimed identity
// pass overflowed a bucket after this launch's first-occurrence pass had been queued behind it).
__global__ __launch_bounds__(256) void post_clear_first_flags_kernel(const PostArgs A) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= A.n_nodes * A.nstride) return;
  const int64_t node = g / A.nstride;
  const int j = (int)(g - node * A.nstride);
  if (j >= A.count[node]) return;
  A.flags[g] &= (uint8_t)~4u;
}

__global__ __launch_bounds__(256) void post_canon_kernel(const PostArgs A) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= A.n_nodes * A.nstride) return;
  const int64_t node = g / A.nstride;
  const int j = (int)(g - node * A.nstride);
  if (j >= A.count[node]) return;
  const uint64_t h = A.hash[g];
  uint64_t s = A.cap;
  if (h != kEmpty) {
    s = mix(h) & (A.cap - 1);
    while (A.keys[s].key != h) s = (s + 1) & (A.cap - 1);
  }
  const int c = (int)A.keys[s].val;
  A.canon[g] = c;
  if (A.flags && c == (int)g) A.flags[g] |= 4;  // first occurrence of this lattice state in the batch
}

}  // namespace

hipError_t launch_post_clear_first_flags(const PostArgs &a, hipStream_t s) {
  const int64_t n = a.n_nodes * a.nstride;
  if (n == 0 || !a.flags) return hipSuccess;
  hipLaunchKernelGGL(post_clear_first_flags_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_post_lists(int dim, const PostArgs &a, hipStream_t s) {
  const int64_t n = a.n_nodes * a.nstride;
  if (n == 0) return hipSuccess;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  if (!a.state && !a.heur && !a.keys) {
    if (a.flags && a.canon) hipLaunchKernelGGL(post_first_flags_kernel, dim3(blocks), dim3(256), 0, s, a);
    return hipGetLastError();
  }
  if (dim == 2) hipLaunchKernelGGL(post_lists_kernel<2>, dim3(blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(post_lists_kernel<3>, dim3(blocks), dim3(256), 0, s, a);
  if (a.keys && a.canon) hipLaunchKernelGGL(post_canon_kernel, dim3(blocks), dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace mplx
