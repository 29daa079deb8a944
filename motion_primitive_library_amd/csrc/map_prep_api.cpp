// map_prep_api.cpp -- C ABI of the map preprocessing routines:
// MapPlanner<Dim>::updatePotentialMap and MapPlanner<Dim>::setSearchRegion with the
// grid work on the device (map_prep_kernel.hip).  The small sequential parts the
// reference does per call -- the potential mask (createMask) and the cells along
// the path (rayTrace) -- stay on the host, in the reference's own arithmetic.
#include "mplx_ctx.h"

#include <array>
#include <cmath>
#include <cstring>
#include <unordered_map>
#include <vector>

using namespace mplx_detail;

namespace {

/* MapUtil::floatToInt, reference include/mpl_collision/map_util.h:103-108 */
inline int float_to_int(double pt, double origin, double res) { return (int)std::round((pt - origin) / res - 0.5); }

}  // namespace

extern "C" {

int mplx_edit_map(mplx_ctx *c, const int64_t *cell_index, const int8_t *values, int64_t n) {
  if (!c) return MPLX_ERR_ARG;
  if (n < 0 || (n > 0 && (!cell_index || !values))) return fail(c, MPLX_ERR_ARG, "mplx_edit_map: bad arguments");
  if (!c->has_map) return fail(c, MPLX_ERR_STATE, "mplx_edit_map: set the map first");
  for (int64_t i = 0; i < n; i++)
    if (cell_index[i] < 0 || cell_index[i] >= c->n_cells)
      return fail(c, MPLX_ERR_ARG, "mplx_edit_map: cell index %lld outside the map of %lld cells", (long long)cell_index[i], (long long)c->n_cells);
  if (n == 0) return MPLX_OK;
  // A cell named twice takes its LAST value (what mplx_set_map with the edited array would hold): the kernel writes the
  // cell and its blocked bit from independent threads, so duplicates with different values must not reach it.
  std::vector<int64_t> u_idx;
  std::vector<int8_t> u_val;
  {
    bool sorted_unique = true;
    for (int64_t i = 1; i < n && sorted_unique; i++) sorted_unique = cell_index[i] > cell_index[i - 1];
    if (!sorted_unique) {
      try {
        std::unordered_map<int64_t, int64_t> last;  // cell -> position of its last mention
        last.reserve((size_t)n * 2);
        for (int64_t i = 0; i < n; i++) last[cell_index[i]] = i;
        if ((int64_t)last.size() < n) {
          u_idx.reserve(last.size());
          u_val.reserve(last.size());
          for (int64_t i = 0; i < n; i++)
            if (last[cell_index[i]] == i) { u_idx.push_back(cell_index[i]); u_val.push_back(values[i]); }
          cell_index = u_idx.data();
          values = u_val.data();
          n = (int64_t)u_idx.size();
        }
      } catch (...) { return fail(c, MPLX_ERR_NOMEM, "mplx_edit_map: out of host memory"); }
    }
  }
  if (int rc = bind_device(c)) return rc;
  if (int rc = resolve_pending(c)) return rc;  // a pending launch read the old cells
  if (int rc = svc_stop(c)) return rc;         // a resident kernel may hold the old cells in its XCD's L2
  const size_t ib = (((size_t)n * 8) + 255) & ~(size_t)255;
  if (int rc = ensure(c, c->edit_buf, ib + (size_t)n)) return rc;
  char *d = (char *)c->edit_buf.p;
  HIP_TRY(c, hipMemcpyAsync(d, cell_index, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(d + ib, values, (size_t)n, hipMemcpyHostToDevice, c->stream));
  c->map_upload_bytes += (uint64_t)n * 9;
  // the blocked bits follow the occupancy map cell by cell; with a potential map installed they are derived from THAT
  // map (env_map.h:113-118 does not consult the occupancy), so the occupancy edit leaves them alone
  uint32_t *blk = (c->blk_ok && !c->has_pot) ? (uint32_t *)c->blk.p : nullptr;
  HIP_TRY(c, mplx::launch_edit_map((const int64_t *)d, (const int8_t *)(d + ib), n, c->n_cells, (int8_t *)c->map.p, blk,
                                   c->has_region ? (const uint32_t *)c->region_bits.p : nullptr, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // the caller may free its arrays on return
  if (blk && c->sat_ok) {
    // the summed-area table (free-box shortcut of the factorised kernels) changes in every entry behind an edited cell:
    // it is switched off and rebuilt by the first launch large enough to be worth it (mplx_api.cpp, lists_device)
    c->sat_ok = false;
    c->sat_stale = true;
  }
  return MPLX_OK;
}

int mplx_read_cells(mplx_ctx *c, int which, const int64_t *cell_index, int64_t n, int8_t *out) {
  if (!c) return MPLX_ERR_ARG;
  if (n < 0 || (n > 0 && (!cell_index || !out)) || (which != 0 && which != 1)) return fail(c, MPLX_ERR_ARG, "mplx_read_cells: bad arguments");
  if (which == 0 ? !c->has_map : !c->has_pot) return fail(c, MPLX_ERR_STATE, which == 0 ? "mplx_read_cells: no map set" : "mplx_read_cells: no potential map set");
  for (int64_t i = 0; i < n; i++)
    if (cell_index[i] < 0 || cell_index[i] >= c->n_cells)
      return fail(c, MPLX_ERR_ARG, "mplx_read_cells: cell index %lld outside the map of %lld cells", (long long)cell_index[i], (long long)c->n_cells);
  if (n == 0) return MPLX_OK;
  if (int rc = bind_device(c)) return rc;
  const size_t ib = (((size_t)n * 8) + 255) & ~(size_t)255;
  if (int rc = ensure(c, c->edit_buf, ib + (size_t)n)) return rc;
  char *d = (char *)c->edit_buf.p;
  HIP_TRY(c, hipMemcpyAsync(d, cell_index, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, mplx::launch_gather_cells((const int8_t *)(which == 0 ? c->map.p : c->pot.p), (const int64_t *)d, n, (int8_t *)(d + ib), c->stream));
  HIP_TRY(c, hipMemcpyAsync(out, d + ib, (size_t)n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return MPLX_OK;
}

int mplx_map_upload_bytes(mplx_ctx *c, uint64_t *bytes) {
  if (!c || !bytes) return MPLX_ERR_ARG;
  *bytes = c->map_upload_bytes;
  return MPLX_OK;
}

int mplx_update_potential_map(mplx_ctx *c, const double *pos, const double *radius, const double *range, double pow_,
                              int8_t *h_map_out) {
  if (!c) return MPLX_ERR_ARG;
  if (!radius || (range && !pos)) return fail(c, MPLX_ERR_ARG, "mplx_update_potential_map: NULL argument");
  if (!c->has_map) return fail(c, MPLX_ERR_STATE, "mplx_update_potential_map: set the map first");
  if (!(pow_ > 0)) return fail(c, MPLX_ERR_ARG, "mplx_update_potential_map: pow must be > 0 (the field must not grow with distance)");
  if (int rc = bind_device(c)) return rc;
  if (int rc = resolve_pending(c)) return rc;
  const int D = c->dim;
  const double res = c->res;
  const int8_t H_MAX = 100;  // map_planner.h:104
  // ---- createMask (map_planner.cpp:246-283) -> value table over (r^2, |n2|)
  const int rn = (int)std::ceil(radius[0] / res);
  const int hn = D == 3 ? (int)std::ceil(radius[2] / res) : 0;
  if (rn < 0 || rn > 255 || hn < 0 || hn > 255)
    return fail(c, MPLX_ERR_ARG, "mplx_update_potential_map: radius of %d x %d cells is out of range", rn, hn);
  const size_t lut_n = (size_t)(rn * rn + 1) * (hn + 1);
  std::vector<int8_t> lut(lut_n, (int8_t)-128);
  const double h_max = H_MAX;
  bool consistent = true;
  for (int n0 = -rn; n0 <= rn; n0++)
    for (int n1 = -rn; n1 <= rn; n1++) {
      if (std::hypot(n0, n1) > rn) continue;
      for (int n2 = -hn; n2 <= hn; n2++) {
        const double h = D == 2 ? h_max * std::pow((1 - (double)std::hypot(n0, n1) / rn), pow_)
                                : h_max * std::pow((1 - (double)std::hypot(n0, n1) / rn) *
                                                       (1 - (double)std::abs(n2) / hn), pow_);
        if (!(h > 1e-3)) continue;
        const size_t k = (size_t)(n0 * n0 + n1 * n1) * (hn + 1) + (size_t)std::abs(n2);
        const int8_t v = (int8_t)h;
        if (lut[k] != -128 && lut[k] != v) consistent = false;
        lut[k] = v;
      }
    }
  if (!consistent)
    return fail(c, MPLX_ERR_STATE, "mplx_update_potential_map: host hypot() is not symmetric; mask is not radial");
  // ---- update box (map_planner.cpp:289-309)
  int32_t c1[3] = {0, 0, 0}, c2[3] = {c->mdim[0], c->mdim[1], c->mdim[2]};
  double rnorm = 0;
  if (range)
    for (int i = 0; i < D; i++) rnorm += range[i] * range[i];
  if (range && std::sqrt(rnorm) > 0) {
    for (int i = 0; i < D; i++) {
      c1[i] = float_to_int(pos[i] - range[i], c->origin[i], res);
      c2[i] = float_to_int(pos[i] + range[i], c->origin[i], res);
      if (c1[i] < 0) c1[i] = 0; else if (c1[i] >= c->mdim[i]) c1[i] = c->mdim[i] - 1;
      if (c2[i] < 0) c2[i] = 0; else if (c2[i] >= c->mdim[i]) c2[i] = c->mdim[i] - 1;
    }
  }
  // ---- device passes; the new map replaces the old one and becomes the potential map
  //      (map_planner.cpp:387-388: setMap(dmap); ENV_->set_potential_map(getMap()))
  const size_t n = (size_t)c->n_cells;
  if (int rc = ensure(c, c->prep_lut, lut_n)) return rc;
  if (int rc = ensure(c, c->prep_a, n * 2)) return rc;
  if (int rc = ensure(c, c->prep_b, n * 2)) return rc;
  if (int rc = ensure(c, c->pot, n)) return rc;
  HIP_TRY(c, hipMemcpyAsync(c->prep_lut.p, lut.data(), lut_n, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, mplx::launch_potential_passes((const int8_t *)c->map.p, c->mdim, c1, c2, rn, hn,
                                           (const int8_t *)c->prep_lut.p, H_MAX, (unsigned short *)c->prep_a.p,
                                           (unsigned short *)c->prep_b.p, (int8_t *)c->pot.p, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->map.p, c->pot.p, n, hipMemcpyDeviceToDevice, c->stream));
  if (h_map_out) HIP_TRY(c, hipMemcpyAsync(h_map_out, c->pot.p, n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // the host-side mask buffer goes out of scope
  c->has_pot = true;
  c->blk_ok = false;
  return MPLX_OK;
}

int mplx_set_search_region_path(mplx_ctx *c, const double *path, int32_t n_points, int32_t dense,
                                const double *search_radius, uint8_t *h_region_out) {
  if (!c) return MPLX_ERR_ARG;
  if ((!path && n_points > 0) || n_points < 0 || !search_radius)
    return fail(c, MPLX_ERR_ARG, "mplx_set_search_region_path: bad arguments");
  if (!c->has_map) return fail(c, MPLX_ERR_STATE, "mplx_set_search_region_path: set the map first");
  if (int rc = bind_device(c)) return rc;
  if (int rc = resolve_pending(c)) return rc;
  const int D = c->dim;
  const double res = c->res;
  auto to_cell = [&](const double *pt) {
    std::array<int, 3> v = {0, 0, 0};
    for (int i = 0; i < D; i++) v[i] = float_to_int(pt[i], c->origin[i], res);
    return v;
  };
  auto outside = [&](const std::array<int, 3> &v) {
    for (int i = 0; i < D; i++)
      if (v[i] < 0 || v[i] >= c->mdim[i]) return true;
    return false;
  };
  // ---- cells along the path (map_planner.cpp:48-58, MapUtil::rayTrace map_util.h:117-135)
  std::vector<int> cells;
  auto push = [&](const std::array<int, 3> &v) { cells.insert(cells.end(), v.begin(), v.end()); };
  if (!dense) {
    for (int i = 1; i < n_points; i++) {
      const double *p1 = path + (size_t)(i - 1) * D, *p2 = path + (size_t)i * D;
      double diff[3] = {0, 0, 0}, linf = 0;
      for (int k = 0; k < D; k++) {
        diff[k] = p2[k] - p1[k];
        linf = std::max(linf, std::fabs(diff[k] / res));
      }
      const double kk = 0.8;
      const int max_diff = (int)(linf / kk);
      const double s = 1.0 / max_diff;
      double step[3];
      for (int k = 0; k < D; k++) step[k] = diff[k] * s;
      std::array<int, 3> prev = {-1, -1, D == 3 ? -1 : 0};
      for (int m = 1; m < max_diff; m++) {
        double pt[3] = {0, 0, 0};
        for (int k = 0; k < D; k++) pt[k] = p1[k] + step[k] * m;
        const std::array<int, 3> v = to_cell(pt);
        if (outside(v)) break;
        if (v != prev) push(v);
        prev = v;
      }
      push(to_cell(p2));
    }
  } else {
    for (int i = 0; i < n_points; i++) push(to_cell(path + (size_t)i * D));
  }
  int32_t rn[3] = {0, 0, 0};
  for (int i = 0; i < D; i++) {
    rn[i] = (int32_t)std::ceil(search_radius[i] / res);
    if (rn[i] < 0) rn[i] = -1;  // an empty offset range, as the reference's loops would have
  }
  // ---- region bits on the device (map_planner.cpp:61-91)
  const size_t words = (size_t)((c->n_cells + 31) >> 5);
  if (int rc = ensure(c, c->region_bits, words * 4)) return rc;
  HIP_TRY(c, hipMemsetAsync(c->region_bits.p, 0, words * 4, c->stream));
  const int n_path_cells = (int)(cells.size() / 3);
  if (n_path_cells > 0 && rn[0] >= 0 && rn[1] >= 0 && rn[2] >= 0) {
    if (int rc = ensure(c, c->prep_a, cells.size() * sizeof(int))) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->prep_a.p, cells.data(), cells.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, mplx::launch_region_boxes((const int *)c->prep_a.p, n_path_cells, D, c->mdim, rn,
                                         (uint32_t *)c->region_bits.p, c->stream));
  }
  if (h_region_out) {
    if (int rc = ensure(c, c->region_bytes, (size_t)c->n_cells)) return rc;
    HIP_TRY(c, mplx::launch_unpack_region((const uint32_t *)c->region_bits.p, c->n_cells, (uint8_t *)c->region_bytes.p,
                                          c->stream));
    HIP_TRY(c, hipMemcpyAsync(h_region_out, c->region_bytes.p, (size_t)c->n_cells, hipMemcpyDeviceToHost, c->stream));
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->has_region = true;
  c->blk_ok = false;
  return MPLX_OK;
}

}  // extern "C"
