// map_prep_kernel.hip -- map preprocessing on the device: the two
// MapPlanner<Dim> routines that produce the HBM-resident inputs of the successor
// expansion,
//   updatePotentialMap   reference src/mpl_planner/map_planner.cpp:286-391
//                        (mask: createMask :246-283)
//   setSearchRegion      reference src/mpl_planner/map_planner.cpp:46-95
//
// updatePotentialMap.  The reference SCATTERS: every occupied cell stamps a mask
// of (offset, int8 value) pairs around itself with `max` -- O(occupied x |mask|),
// seconds on a CPU at 256^3.  The mask value depends only on the planar distance
// hypot(n0, n1) and on |n2|, and it does not increase with either, so the GATHER
//   out(c) = max over dz of  V[ r2min(c.x, c.y, c.z + dz) ][ |dz| ]
// with r2min = squared planar distance to the nearest source cell of that z-slice
// gives the same cells.  r2min is an exact integer distance transform, separable
// into an x pass and a y pass (both windowed by the mask radius), and V is a tiny
// table the host derives from the reference's own mask arithmetic (hypot / pow on
// the host, exactly as createMask does), so no device transcendental is involved
// and the result is bit-identical to the scatter.  Three streaming passes over
// the grid, each reading a window of 2r+1 cells: HBM / L2 bound.
//
// setSearchRegion.  The cells along the path come from the host (rayTrace is a
// few hundred sequential double operations, map_util.h:117-135); the device ORs
// the (2rn+1)^D box around every path cell into the 1-bit-per-cell region the
// expansion kernels read.
#include "mplx_internal.h"

namespace mplx {
namespace {

constexpr unsigned short kFar = 0xffff;  // "no source within the window"

// x pass: squared distance along x to the nearest source cell (map > 0, inside the
// update box) of the same (y, z) row, or kFar.
__global__ void pot_x_kernel(const int8_t *map, int d0, int d1, int d2, int c1x, int c2x, int c1y, int c2y, int c1z,
                             int c2z, int rn, unsigned short *dx2) {
  const int64_t n = (int64_t)d0 * d1 * d2;
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const int x = (int)(g % d0);
  const int64_t row = g / d0;
  const int y = (int)(row % d1), z = (int)(row / d1);
  unsigned int best = kFar;
  if (y >= c1y && y < c2y && z >= c1z && z < c2z) {
    const int lo = max(x - rn, c1x), hi = min(x + rn, c2x - 1);
    const int8_t *r = map + row * d0;
    for (int s = lo; s <= hi; s++)
      if (r[s] > 0) {
        const unsigned int d = (unsigned)((s - x) * (s - x));
        best = d < best ? d : best;
      }
  }
  dx2[g] = (unsigned short)best;
}

// y pass: r2 = min over dy of dx2(x, y + dy, z) + dy^2, capped at rn^2
__global__ void pot_y_kernel(const unsigned short *dx2, int d0, int d1, int d2, int rn, unsigned short *r2) {
  const int64_t n = (int64_t)d0 * d1 * d2;
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const int x = (int)(g % d0);
  const int64_t row = g / d0;
  const int y = (int)(row % d1), z = (int)(row / d1);
  const unsigned int cap = (unsigned)(rn * rn);
  unsigned int best = kFar;
  const int lo = max(y - rn, 0), hi = min(y + rn, d1 - 1);
  for (int s = lo; s <= hi; s++) {
    const unsigned int a = dx2[((int64_t)z * d1 + s) * d0 + x];
    if (a != kFar) {
      const unsigned int d = a + (unsigned)((s - y) * (s - y));
      if (d <= cap && d < best) best = d;
    }
  }
  r2[g] = (unsigned short)best;
}

// z pass + base value.  lut[r2 * (hn + 1) + |dz|] = int8 mask value or -128 (no mask entry).
__global__ void pot_z_kernel(const int8_t *map, const unsigned short *r2, const int8_t *lut, int d0, int d1, int d2,
                             int c1x, int c2x, int c1y, int c2y, int c1z, int c2z, int hn, int8_t h_max, int8_t *out) {
  const int64_t n = (int64_t)d0 * d1 * d2;
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const int x = (int)(g % d0);
  const int64_t row = g / d0;
  const int y = (int)(row % d1), z = (int)(row / d1);
  const int8_t m = map[g];
  const bool in_box = x >= c1x && x < c2x && y >= c1y && y < c2y && z >= c1z && z < c2z;
  int v = (m > 0 && in_box) ? (int)h_max : (int)m;  // map_planner.cpp:343-345 / :362-364
  const int64_t slice = (int64_t)d0 * d1;
  const int lo = max(z - hn, 0), hi = min(z + hn, d2 - 1);
  for (int s = lo; s <= hi; s++) {
    const unsigned int a = r2[g + (int64_t)(s - z) * slice];
    if (a != kFar) {
      const int dz = s > z ? s - z : z - s;
      const int e = lut[(int64_t)a * (hn + 1) + dz];
      if (e != -128 && e > v) v = e;
    }
  }
  out[g] = (int8_t)v;
}

// OR the box around every path cell into the packed region (bits pre-zeroed).
__global__ void region_box_kernel(const int *cells, int n_cells_path, int dim, int d0, int d1, int d2, int rn0, int rn1,
                                  int rn2, uint32_t *bits) {
  const int bx = 2 * rn0 + 1, by = 2 * rn1 + 1, bz = 2 * rn2 + 1;
  const int64_t per = (int64_t)bx * by * bz;
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= per * n_cells_path) return;
  const int p = (int)(g / per);
  int64_t o = g - (int64_t)p * per;
  const int ox = (int)(o % bx) - rn0;
  o /= bx;
  const int oy = (int)(o % by) - rn1;
  const int oz = (int)(o / by) - rn2;
  const int x = cells[p * 3 + 0] + ox, y = cells[p * 3 + 1] + oy, z = (dim == 3) ? cells[p * 3 + 2] + oz : 0;
  if (x < 0 || x >= d0 || y < 0 || y >= d1 || z < 0 || z >= d2) return;  // isOutside: skipped (map_planner.cpp:85)
  const int64_t idx = x + (int64_t)d0 * y + (int64_t)d0 * d1 * z;
  atomicOr(&bits[idx >> 5], 1u << (idx & 31));
}

__global__ void unpack_region_kernel(const uint32_t *bits, int64_t n_cells, uint8_t *bytes) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_cells) return;
  bytes[g] = (uint8_t)((bits[g >> 5] >> (g & 31)) & 1u);
}

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + 255) / 256); }


// An edit of a few cells of the occupancy map (a sensor update between two plans of an LPA* search, map_planner.cpp:
// 160-185): the int8 cells and -- where the blocked-bit map is current -- its bits are patched in place instead of
// uploading the map again and rebuilding every derived structure (512^3: 128 MiB over PCIe + 1.6 ms of kernels per edit).
// Several edited cells may share a word of the bit map: atomics.
__global__ void edit_map_kernel(const int64_t *idx, const int8_t *val, int64_t n, int64_t n_cells, int8_t *map, uint32_t *blk,
                                const uint32_t *region) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const int64_t c = idx[g];
  if (c < 0 || c >= n_cells) return;  // (checked on the host as well)
  const int8_t v = val[g];
  map[c] = v;
  if (blk) {
    const uint32_t bit = 1u << (c & 31);
    const bool in_region = !region || ((region[c >> 5] >> (c & 31)) & 1u);
    if (v == 100 || !in_region) atomicOr(&blk[c >> 5], bit);  // build_blocked_bits_kernel's rule for occupancy maps
    else atomicAnd(&blk[c >> 5], ~bit);
  }
}

}  // namespace

hipError_t launch_potential_passes(const int8_t *map, const int32_t *d, const int32_t *c1, const int32_t *c2, int rn,
                                   int hn, const int8_t *lut, int8_t h_max, unsigned short *tmp_a,
                                   unsigned short *tmp_b, int8_t *out, hipStream_t s) {
  const int64_t n = (int64_t)d[0] * d[1] * d[2];
  hipLaunchKernelGGL(pot_x_kernel, dim3(blocks_for(n)), dim3(256), 0, s, map, d[0], d[1], d[2], c1[0], c2[0], c1[1],
                     c2[1], c1[2], c2[2], rn, tmp_a);
  hipLaunchKernelGGL(pot_y_kernel, dim3(blocks_for(n)), dim3(256), 0, s, tmp_a, d[0], d[1], d[2], rn, tmp_b);
  hipLaunchKernelGGL(pot_z_kernel, dim3(blocks_for(n)), dim3(256), 0, s, map, tmp_b, lut, d[0], d[1], d[2], c1[0],
                     c2[0], c1[1], c2[1], c1[2], c2[2], hn, h_max, out);
  return hipGetLastError();
}

hipError_t launch_region_boxes(const int *cells, int n_path_cells, int dim, const int32_t *d, const int32_t *rn,
                               uint32_t *bits, hipStream_t s) {
  const int64_t per = (int64_t)(2 * rn[0] + 1) * (2 * rn[1] + 1) * (2 * rn[2] + 1);
  const int64_t total = per * n_path_cells;
  if (total == 0) return hipSuccess;
  hipLaunchKernelGGL(region_box_kernel, dim3(blocks_for(total)), dim3(256), 0, s, cells, n_path_cells, dim, d[0], d[1],
                     d[2], rn[0], rn[1], rn[2], bits);
  return hipGetLastError();
}

hipError_t launch_edit_map(const int64_t *idx, const int8_t *val, int64_t n, int64_t n_cells, int8_t *map, uint32_t *blk,
                           const uint32_t *region, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(edit_map_kernel, dim3(blocks_for(n)), dim3(256), 0, s, idx, val, n, n_cells, map, blk, region);
  return hipGetLastError();
}

// out[i] = cells[idx[i]] (idx checked by the host): a few cells of a map that lives on the device, for host logic
// that needs their values (the prior trajectory's potential cost, env_map.h:189-255)
__global__ void gather_cells_kernel(const int8_t *cells, const int64_t *idx, int64_t n, int8_t *out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = cells[idx[i]];
}

hipError_t launch_gather_cells(const int8_t *cells, const int64_t *idx, int64_t n, int8_t *out, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(gather_cells_kernel, dim3(blocks_for(n)), dim3(256), 0, s, cells, idx, n, out);
  return hipGetLastError();
}

hipError_t launch_unpack_region(const uint32_t *bits, int64_t n_cells, uint8_t *bytes, hipStream_t s) {
  hipLaunchKernelGGL(unpack_region_kernel, dim3(blocks_for(n_cells)), dim3(256), 0, s, bits, n_cells, bytes);
  return hipGetLastError();
}

}  // namespace mplx
