// edge_api.cpp -- C ABI of the batched edge re-validation (edge_kernel.hip).
#include "mplx_ctx.h"

using namespace mplx_detail;

extern "C" int mplx_check_edges(mplx_ctx *c, const double *h_parents, const int32_t *h_actions, int64_t n_edges,
                                int64_t stride, const mplx_edges_out *h_out) {
  if (!c) return MPLX_ERR_ARG;
  if (!h_out || n_edges < 0 || stride < n_edges || ((!h_parents || !h_actions) && n_edges > 0))
    return fail(c, MPLX_ERR_ARG, "mplx_check_edges: bad arguments");
  if (h_out->cells && (h_out->cell_cap <= 0 || !h_out->cell_count))
    return fail(c, MPLX_ERR_ARG, "mplx_check_edges: cells need cell_cap > 0 and cell_count");
  if (!c->has_map || !c->has_params || !c->has_U) return fail(c, MPLX_ERR_STATE, "mplx_check_edges: map, params and controls must be set");
  if (n_edges == 0) return MPLX_OK;
  for (int64_t e = 0; e < n_edges; e++)
    if (h_actions[e] < 0 || h_actions[e] >= c->nU)
      return fail(c, MPLX_ERR_ARG, "mplx_check_edges: action %d of edge %lld is outside the control table", h_actions[e], (long long)e);
  if (int rc = bind_device(c)) return rc;
  const int F = 4 * c->dim + 2;
  if (int rc = ensure(c, c->e_parents, (size_t)F * n_edges * 8)) return rc;
  if (int rc = ensure(c, c->e_action, (size_t)n_edges * 4)) return rc;
  HIP_TRY(c, hipMemcpy2DAsync(c->e_parents.p, (size_t)n_edges * 8, h_parents, (size_t)stride * 8, (size_t)n_edges * 8, F,
                              hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->e_action.p, h_actions, (size_t)n_edges * 4, hipMemcpyHostToDevice, c->stream));
  mplx::EdgeArgs a{};
  a.map = (const int8_t *)c->map.p;
  a.region = c->has_region ? (const uint32_t *)c->region_bits.p : nullptr;
  a.dim0 = c->mdim[0]; a.dim1 = c->mdim[1]; a.dim2 = c->mdim[2];
  a.org0 = c->origin[0]; a.org1 = c->origin[1]; a.org2 = c->origin[2];
  a.res = c->res;
  a.dt = c->prm.dt; a.w = c->prm.w;
  a.U = (const double *)c->U.p;
  a.nU = c->nU; a.udim = c->udim;
  a.parents = (const double *)c->e_parents.p;
  a.action = (const int32_t *)c->e_action.p;
  a.n_edges = n_edges; a.stride = n_edges;
  if (h_out->free_flag) { if (int rc = ensure(c, c->e_free, (size_t)n_edges)) return rc; a.free_out = (uint8_t *)c->e_free.p; }
  if (h_out->cost) { if (int rc = ensure(c, c->e_cost, (size_t)n_edges * 8)) return rc; a.cost = (double *)c->e_cost.p; }
  if (h_out->cell_count) { if (int rc = ensure(c, c->e_count, (size_t)n_edges * 4)) return rc; a.cell_count = (int32_t *)c->e_count.p; }
  if (h_out->cells) {
    if (int rc = ensure(c, c->e_cells, (size_t)n_edges * h_out->cell_cap * 4)) return rc;
    a.cells = (int32_t *)c->e_cells.p;
    a.cell_cap = h_out->cell_cap;
  }
  if (h_out->outside) {  // (rides at the end of the free-flag staging buffer)
    if (int rc = ensure(c, c->e_free, (size_t)n_edges * 2)) return rc;
    if (h_out->free_flag) a.free_out = (uint8_t *)c->e_free.p;
    a.outside_out = (uint8_t *)c->e_free.p + n_edges;
  }
  HIP_TRY(c, mplx::launch_check_edges(c->dim, c->prm.control, a, c->stream));
  if (h_out->outside) HIP_TRY(c, hipMemcpyAsync(h_out->outside, a.outside_out, (size_t)n_edges, hipMemcpyDeviceToHost, c->stream));
  if (h_out->free_flag) HIP_TRY(c, hipMemcpyAsync(h_out->free_flag, a.free_out, (size_t)n_edges, hipMemcpyDeviceToHost, c->stream));
  if (h_out->cost) HIP_TRY(c, hipMemcpyAsync(h_out->cost, a.cost, (size_t)n_edges * 8, hipMemcpyDeviceToHost, c->stream));
  if (h_out->cell_count) HIP_TRY(c, hipMemcpyAsync(h_out->cell_count, a.cell_count, (size_t)n_edges * 4, hipMemcpyDeviceToHost, c->stream));
  if (h_out->cells) HIP_TRY(c, hipMemcpyAsync(h_out->cells, a.cells, (size_t)n_edges * h_out->cell_cap * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return MPLX_OK;
}
