// pack_api.cpp -- mplx_pack_lists_device (include/mplx.h): the used prefixes of per-node successor lists
// (the reference's get_succ outputs, include/mpl_planner/env/env_map.h:147-172: node k uses count[k] of the S
// entries reserved for it) packed back to back on the device, with the exclusive prefix sums that index them.
// The packed form is what the multi-GPU all-gather moves (comm_api.cpp) and what an on-device consumer reads.
#include "mplx_ctx.h"

#include <cstdlib>

using namespace mplx_detail;

extern "C" int mplx_pack_lists_device(mplx_ctx *c, const mplx_succ_lists *L, int64_t n_nodes, const mplx_packed_lists *o,
                                      int64_t *h_total) {
  if (!c) return MPLX_ERR_ARG;
  if (!L || !o || n_nodes < 0 || !L->count || !o->offs)
    return fail(c, MPLX_ERR_ARG, "mplx_pack_lists_device: need the lists' count and the output offs");
  if (!c->has_U) return fail(c, MPLX_ERR_STATE, "mplx_pack_lists_device: controls not set");
  if ((o->action && !L->action) || (o->cost && !L->cost) || (o->hash && !L->hash) || (o->state && !L->state))
    return fail(c, MPLX_ERR_ARG, "mplx_pack_lists_device: an output row is requested that the lists do not have");
  if (o->state && o->state_stride < o->capacity)
    return fail(c, MPLX_ERR_ARG, "mplx_pack_lists_device: state_stride < capacity");
  if (int rc = bind_device(c)) return rc;
  if (int rc = resolve_pending(c)) return rc;  // yaw pinning: the lists must be final
  const int F = 4 * c->dim + 2;
  const int64_t S = L->node_stride ? L->node_stride : c->nU;
  HIP_TRY(c, mplx::launch_scan_counts(L->count, n_nodes, o->offs, c->stream));
  if (o->count && n_nodes > 0)
    HIP_TRY(c, hipMemcpyAsync(o->count, L->count, (size_t)n_nodes * 4, hipMemcpyDeviceToDevice, c->stream));
  int64_t total = -1;
  if (h_total || o->capacity < n_nodes * (int64_t)c->nU) {
    HIP_TRY(c, hipMemcpyAsync(&total, o->offs + n_nodes, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (h_total) *h_total = total;
    if (total > o->capacity)
      return fail(c, MPLX_ERR_ARG, "mplx_pack_lists_device: %lld entries do not fit the capacity %lld", (long long)total,
                  (long long)o->capacity);
  }
  if (n_nodes == 0 || total == 0) return MPLX_OK;
  mplx::PackArgs a{};
  int r = 0;
  auto row = [&](const void *src, void *dst, int es) {
    a.src[r] = src;
    a.dst_off[r] = (int64_t)(uintptr_t)dst;  // a.dst stays null: the rows are independent allocations (flat addresses)
    a.es[r] = es;
    r++;
  };
  if (o->cost) row(L->cost, o->cost, 8);
  if (o->hash) row(L->hash, o->hash, 8);
  if (o->state)
    for (int f = 0; f < F; f++) row(L->state + (size_t)f * L->state_stride, o->state + (size_t)f * o->state_stride, 8);
  if (o->action) row(L->action, o->action, 4);
  if (r == 0) return MPLX_OK;
  a.n_rows = r;
  a.node_stride = S;
  a.count = L->count;
  a.offs = o->offs;
  a.node0 = 0;
  a.off0 = 0;
  a.dst = nullptr;
  HIP_TRY(c, mplx::launch_pack_rows(a, n_nodes, c->stream));
  return MPLX_OK;
}

// Diagnostic: the list stores of an expansion launch on their own (store_model_kernel.hip).  Overwrites the entries.
extern "C" int mplx_debug_store_model(mplx_ctx *c, const mplx_succ_lists *L, int64_t n_nodes) {
  if (!c) return MPLX_ERR_ARG;
  if (!L || n_nodes < 0 || !L->count) return fail(c, MPLX_ERR_ARG, "mplx_debug_store_model: need the lists' count");
  if (!c->has_U) return fail(c, MPLX_ERR_STATE, "mplx_debug_store_model: controls not set");
  if (int rc = bind_device(c)) return rc;
  if (int rc = resolve_pending(c)) return rc;
  const int64_t S = L->node_stride ? L->node_stride : c->nU;
  const int pad = (S % 32 == 0 && !c->tune.no_line_pad && c->nU >= mplx::kLinePadMinControls) ? 1 : 0;  // (the expansion's own rule)
  // (experiments: MPLX_STORE_MODEL_MODE / _WGS vary the order inside a node, the nodes per chunk and the workgroups per CU)
  const char *em = getenv("MPLX_STORE_MODEL_MODE"), *ew = getenv("MPLX_STORE_MODEL_WGS");
  const int mode = em ? atoi(em) : 0, wgs = ew && atoi(ew) > 0 ? atoi(ew) : 5;
  HIP_TRY(c, mplx::launch_store_model(L->count, n_nodes, S, L->action, L->cost, L->hash, L->state, L->state_stride,
                                      4 * c->dim + 2, pad, c->n_cus * wgs, mode, c->stream));
  return MPLX_OK;
}
