// post_api.cpp -- C ABI of the successor post-processing: heuristic,
// goal tolerances and node identity of a whole batch of successor lists, on the
// device (post_kernel.hip).
#include "mplx_ctx.h"
#include "host_planner.hpp"

#include <cstring>

using namespace mplx_detail;

// max_entries: an upper bound on the emitted successors (sizes the identity table)
static int post_lists_impl(mplx_ctx *c, const mplx_succ_lists *d_lists, int64_t n_nodes, const mplx_goal_spec *goal,
                           const mplx_post *d_out, uint64_t max_entries) {
  if (!c) return MPLX_ERR_ARG;
  if (!d_lists || !goal || !d_out || n_nodes < 0 || !goal->goal)
    return fail(c, MPLX_ERR_ARG, "mplx_post_lists_device: NULL argument");
  if (!d_lists->count || !d_lists->hash || !d_lists->state)
    return fail(c, MPLX_ERR_ARG, "mplx_post_lists_device: the lists need count, hash and state");
  if (!c->has_U) return fail(c, MPLX_ERR_STATE, "mplx_post_lists_device: controls not set");
  if (n_nodes == 0) return MPLX_OK;
  if (int rc = bind_device(c)) return rc;
  if (int rc = resolve_pending(c)) return rc;  // yaw pinning: the lists must be final
  const int D = c->dim, F = 4 * D + 2;
  const int64_t S = d_lists->node_stride ? d_lists->node_stride : c->nU;
  const int64_t n = n_nodes * S;
  if (n >= 0x7f7f7f7fLL) return fail(c, MPLX_ERR_ARG, "mplx_post_lists_device: %lld list entries exceed the int32 index", (long long)n);
  mplx::PostArgs a{};
  a.count = d_lists->count;
  a.hash = d_lists->hash;
  a.state = d_lists->state;
  a.n_nodes = n_nodes;
  a.nstride = S;
  a.sstride = d_lists->state_stride;
  std::memcpy(a.goal, goal->goal, sizeof(double) * F);
  a.goal_hash = mplx::host::lattice_hash(D, goal->goal_control ? goal->goal_control : goal->control, goal->goal);
  a.w = goal->w;
  a.v_max = goal->v_max;
  a.tol_pos = goal->tol_pos;
  a.tol_vel = goal->tol_vel;
  a.tol_acc = goal->tol_acc;
  a.tol_yaw = goal->tol_yaw;
  a.heur = d_out->heur;
  a.flags = d_out->flags;
  a.canon = d_out->canon;
  if (d_out->canon) {
    uint64_t cap = 1024;
    while (cap < max_entries) cap <<= 1;  // >= the emitted successors whatever the frontier
    if (cap < 2 * max_entries && cap < (1ull << 27)) cap <<= 1;
    if (int rc = ensure(c, c->post_keys, (cap + 1) * sizeof(mplx::PostArgs::Slot))) return rc;
    // all bytes 0xff: key = ~0 (empty), val = 0xffffffff (above every list index, unsigned atomicMin)
    HIP_TRY(c, hipMemsetAsync(c->post_keys.p, 0xff, (cap + 1) * sizeof(mplx::PostArgs::Slot), c->stream));
    a.keys = (mplx::PostArgs::Slot *)c->post_keys.p;
    a.cap = cap;
  }
  HIP_TRY(c, mplx::launch_post_lists(D, a, c->stream));
  return MPLX_OK;
}

extern "C" int mplx_post_lists_device(mplx_ctx *c, const mplx_succ_lists *d_lists, int64_t n_nodes,
                                      const mplx_goal_spec *goal, const mplx_post *d_out) {
  if (!c) return MPLX_ERR_ARG;
  return post_lists_impl(c, d_lists, n_nodes, goal, d_out, (uint64_t)(n_nodes > 0 ? n_nodes : 0) * (uint64_t)c->nU);
}

// Packed lists are one long list: entry i of every row, offs[n_nodes] entries in all.  The strided kernel reads them as
// a single "node" whose count is that total -- the low word of the last prefix sum, where it lies on the device.
extern "C" int mplx_post_packed_device(mplx_ctx *c, const mplx_packed_lists *p, int64_t n_nodes, const mplx_goal_spec *goal,
                                       const mplx_post *d_out) {
  if (!c) return MPLX_ERR_ARG;
  if (!p || !goal || !d_out || n_nodes < 0 || !p->offs || !p->hash || !p->state)
    return fail(c, MPLX_ERR_ARG, "mplx_post_packed_device: the packed lists need offs, hash and state");
  if (p->capacity <= 0 || p->capacity >= 0x7f7f7f7fLL)
    return fail(c, MPLX_ERR_ARG, "mplx_post_packed_device: capacity %lld outside the int32 index", (long long)p->capacity);
  if (!c->has_U) return fail(c, MPLX_ERR_STATE, "mplx_post_packed_device: controls not set");
  mplx_succ_lists one{};
  one.count = (int32_t *)(p->offs + n_nodes);  // little-endian low word of offs[n_nodes] (< 2^31, checked above)
  one.hash = p->hash;
  one.state = p->state;
  one.state_stride = p->state_stride;
  one.node_stride = p->capacity;
  // the identity table is sized by the caller-visible bound on the entries: capacity
  return post_lists_impl(c, &one, 1, goal, d_out, (uint64_t)p->capacity);
}
