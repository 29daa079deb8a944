// Internal declarations shared by the C-ABI layer (mplx_api.cpp) and the HIP
// kernels (expand_kernel.hip).  Not installed; the public surface is
// include/mplx.h.
#ifndef MPLX_INTERNAL_H
#define MPLX_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mplx {

// Everything one expansion launch needs, passed by value as the kernel
// argument block (lives in SGPRs / constant cache: wave-uniform).
struct ExpandArgs {
  // voxel / occupancy grid, row-major with x fastest (reference
  // include/mpl_collision/map_util.h:34-41)
  const int8_t *map;
  const int8_t *pot;       // potential cells or nullptr (env_map.h:113-118)
  const uint32_t *region;  // search region, 1 bit per cell, or nullptr
  int32_t dim0, dim1, dim2;
  double org0, org1, org2;
  double res;
  // env parameters (env_base.h:368-392, env_map.h:294-296)
  double dt, w, wyaw;
  double v_max, a_max, j_max, yaw_max;
  double pot_w, grad_w;
  // controls [nU][udim]
  const double *U;
  int32_t nU, udim;
  // frontier, field-major [4D+2][node_stride]
  const double *nodes;
  int64_t n_nodes, node_stride;
  // dense successor slots (any may be nullptr)
  uint8_t *status;
  double *cost;
  uint64_t *hash;
  double *state;
  int64_t state_stride;
  int32_t *iters;
};

// Launches the successor-expansion kernel specialised for (dim, control) on
// `stream`.  Returns hipSuccess or the launch error.
hipError_t launch_expand(int dim, int control, const ExpandArgs &args, hipStream_t stream);

// Element-wise math probe (see mplx_selftest_math in mplx.h).
hipError_t launch_math_probe(int op, const double *a, const double *b, double *out, int64_t n,
                             hipStream_t stream);

// Packs a byte-per-cell mask into 1 bit per cell (word = cell >> 5).
hipError_t launch_pack_region(const uint8_t *bytes, uint32_t *bits, int64_t n_cells,
                              hipStream_t stream);

}  // namespace mplx
#endif
