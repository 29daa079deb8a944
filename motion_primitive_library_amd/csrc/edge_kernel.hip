// edge_kernel.hip -- batched re-validation of stored graph edges for incremental
// (LPA*) re-planning.  One lane walks one edge
// (parent waypoint, action id) -> Primitive<Dim>(parent, U[action], dt)
// (env_base::forward_action, reference include/mpl_planner/common/env_base.h:228-231):
//   is_free(Primitive)            env_map.h:60-76: n = ceil(max_v * t / res) uniform
//                                 samples i * (t / n), i = 0 .. n (Primitive::sample,
//                                 primitive.h:415-420); blocked by an occupied cell, a
//                                 cell outside the map or outside the search region
//   calculate_intrinsic_cost      env_base.h:343-345: J + w * dt, reported for free edges
//                                 (StateSpace::decreaseCost, state_space.h:230-253)
//   linked cells                  MapPlanner::getLinkedNodes, map_planner.cpp:125-157:
//                                 the run-length compressed sequence of cell indices the
//                                 same samples fall into (the voxel -> edge table lhm_)
// Same arithmetic rules as expand_kernel.hip (true divisions, no contraction); the
// sample times here are PRODUCTS i * dt, not the accumulated sums of traverse_primitive.
// n == 0 (a primitive that does not move): the reference samples t = 0 * (t / 0) = NaN and
// the resulting cell conversion is undefined behaviour (INT_MIN on x86: "outside", not
// free); the kernel reports such an edge as not free with no linked cells.
#include "mplx_internal.h"
#include "mplx_device_common.h"

namespace mplx {
namespace {

using namespace dev;

template <int D, int K>
__global__ __launch_bounds__(256) void edge_kernel(const EdgeArgs A) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= A.n_edges) return;
  const int act = A.action[e];
  const double T = A.dt;
  Ax<K> ax[D];
  double max_v = 0, J = 0;
  const double org[3] = {A.org0, A.org1, A.org2};
  const int dims[3] = {A.dim0, A.dim1, A.dim2};
#pragma unroll
  for (int i = 0; i < D; i++) {
    const double p = A.parents[(int64_t)(0 * D + i) * A.stride + e];
    const double v = (K >= 2) ? A.parents[(int64_t)(1 * D + i) * A.stride + e] : 0.0;
    const double a = (K >= 3) ? A.parents[(int64_t)(2 * D + i) * A.stride + e] : 0.0;
    const double j = (K >= 4) ? A.parents[(int64_t)(3 * D + i) * A.stride + e] : 0.0;
    const double u = A.U[(int64_t)act * A.udim + i];
    ax[i].init(p, v, a, j, u);
    const double mv = ax[i].max_vel(T);
    if (mv > max_v) max_v = mv;  // env_map.h:62-64
    J += ax[i].effort(T);
  }
  const int n = (int)ceil(max_v * T / A.res);  // env_map.h:65 (no lower bound of 5 here)
  bool is_free = n > 0;
  bool any_outside = false;
  int n_cells = 0;
  if (n > 0) {
    const double sdt = T / n;  // primitive.h:417
    int prev = -1;
    for (int i = 0; i <= n; i++) {
      const double t = i * sdt;
      bool outside = false;
      int cell[D];
#pragma unroll
      for (int k = 0; k < D; k++) {
        cell[k] = (int)round((ax[k].template pos<false>(t) - org[k]) / A.res - 0.5);  // map_util.h:103-108
        outside = outside || cell[k] < 0 || cell[k] >= dims[k];
      }
      any_outside = any_outside || outside;
      int idx = cell[0] + dims[0] * cell[1];  // map_util.h:34-41, in the reference's int arithmetic
      if (D == 3) idx += dims[0] * dims[1] * cell[2];
      if (A.cells) {  // map_planner.cpp:145-152
        if (idx != prev) {
          if (n_cells < A.cell_cap) A.cells[e * A.cell_cap + n_cells] = idx;
          n_cells++;
          prev = idx;
        }
      }
      if (is_free) {
        if (outside) is_free = false;
        else if (A.map[idx] == 100) is_free = false;
        else if (A.region != nullptr && !((A.region[(unsigned)idx >> 5] >> (idx & 31)) & 1u)) is_free = false;
      }
      if (!is_free && !A.cells && !A.outside_out) break;
    }
  }
  if (A.free_out) A.free_out[e] = is_free ? 1 : 0;
  if (A.cost) A.cost[e] = is_free ? J + A.w * A.dt : INFINITY;
  if (A.cell_count) A.cell_count[e] = n_cells;
  if (A.outside_out) A.outside_out[e] = any_outside ? 1 : 0;
}

template <int D, int K>
hipError_t launch_edges_inst(const EdgeArgs &a, hipStream_t s) {
  if (a.n_edges == 0) return hipSuccess;
  hipLaunchKernelGGL((edge_kernel<D, K>), dim3((unsigned)((a.n_edges + 255) / 256)), dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace

hipError_t launch_check_edges(int dim, int control, const EdgeArgs &a, hipStream_t s) {
  const int k = control & 0x0f;
  if (dim == 2) {
    switch (k) {
      case 0x01: return launch_edges_inst<2, 1>(a, s);
      case 0x03: return launch_edges_inst<2, 2>(a, s);
      case 0x07: return launch_edges_inst<2, 3>(a, s);
      case 0x0f: return launch_edges_inst<2, 4>(a, s);
    }
  } else if (dim == 3) {
    switch (k) {
      case 0x01: return launch_edges_inst<3, 1>(a, s);
      case 0x03: return launch_edges_inst<3, 2>(a, s);
      case 0x07: return launch_edges_inst<3, 3>(a, s);
      case 0x0f: return launch_edges_inst<3, 4>(a, s);
    }
  }
  return hipErrorInvalidValue;
}

}  // namespace mplx
