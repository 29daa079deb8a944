// post_api.cpp -- C ABI of the successor post-processing: heuristic,
// goal tolerances and node identity of a whole batch of successor lists, on the
// device (post_kernel.hip).
#include "mplx_ctx.h"
#include "host_planner.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>

using namespace mplx_detail;

// max_entries: an upper bound on the emitted successors (sizes the identity table)
static int post_lists_impl(mplx_ctx *c, const mplx_succ_lists *d_lists, int64_t n_nodes, const mplx_goal_spec *goal,
                           const mplx_post *d_out, uint64_t max_entries) {
  if (!c) return MPLX_ERR_ARG;
  if (!d_lists || !goal || !d_out || n_nodes < 0 || !goal->goal)
    return fail(c, MPLX_ERR_ARG, "mplx_post_lists_device: NULL argument");
  if (!d_lists->count || !d_lists->hash)
    return fail(c, MPLX_ERR_ARG, "mplx_post_lists_device: the lists need count and hash");
  // Lists without state rows: the heuristic needs positions, so only the identity (canon) can be asked for; a `flags`
  // row is then taken as the one the expansion launch wrote (mplx_succ_lists::flags) and gets its bit 2 added.
  if (!d_lists->state && (d_out->heur || !d_out->canon))
    return fail(c, MPLX_ERR_ARG, "mplx_post_lists_device: lists without state rows can only be asked for canon (+ bit 2 of an existing flags row)");
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
  // node identity: batches above ~256 k slots go through the radix partition + LDS tables (identity_kernel.hip), small
  // ones through the table in HBM (two launches).  MPLX_POST_PARTITION_MIN moves the switch (tests: 0 = always).
  const char *e_min = getenv("MPLX_POST_PARTITION_MIN"), *e_fill = getenv("MPLX_POST_FILL"), *e_bits = getenv("MPLX_POST_BITS");
  const int64_t partition_min = e_min ? (int64_t)atoll(e_min) : (int64_t)1 << 18;
  if (d_out->canon && n >= partition_min && (n_nodes == 1 || S + 4096 < (1 << 24))) {  // (the slot -> node arithmetic of the partition is f32-exact below 2^24)
    mplx::IdentityArgs ia{};
    ia.count = d_lists->count;
    ia.hash = d_lists->hash;
    ia.n_nodes = n_nodes;
    ia.nstride = S;
    ia.canon = d_out->canon;
    ia.n_slots = n;
    mplx::identity_plan(n, &ia.b1, &ia.b2);
    if (e_bits) {  // diagnostic: "b1,b2" (b1 <= 6 with two levels, <= 8 with one; b2 <= 8)
      int x = 0, y = 0;
      if (sscanf(e_bits, "%d,%d", &x, &y) == 2 && x >= 0 && y >= 0 && y <= 8 && x <= (y ? 6 : 8)) { ia.b1 = x; ia.b2 = y; }
    }
    ia.fill = mplx::identity_default_fill();
    if (e_fill && atoi(e_fill) >= 1 && atoi(e_fill) < ia.fill) ia.fill = atoi(e_fill);
    int64_t ctr1 = 0, ctr2 = 0;
    mplx::identity_sizes(n, ia.b1, ia.b2, &ia.tiles1, &ia.tiles2_cap, &ctr1, &ctr2);
    const int levels = ia.b2 ? 2 : 1;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    // workspace of the exact form: pairs of both levels, counters, block totals, segments
    const size_t sz_h = up((size_t)n * 8), sz_g = up((size_t)n * 4);
    const size_t sz_c1 = up((size_t)ctr1 * 4), sz_c2 = up((size_t)ctr2 * 4);
    const size_t sz_t1 = up((size_t)(ctr1 / 4096 + 1) * 4), sz_t2 = up((size_t)(ctr2 / 4096 + 1) * 4);
    const size_t sz_r = up(((size_t)1 << (ia.b1 + ia.b2)) * 4 + 4);
    const size_t total = levels * (sz_h + sz_g) + sz_c1 + sz_c2 + sz_t1 + sz_t2 + sz_r + 1024;
    // ... and of the claimed form (two levels): buckets of fixed capacity, cursors, the tile table of level 2
    const char *e_cl = getenv("MPLX_POST_CLAIMED"), *e_cap = getenv("MPLX_POST_CAP");
    // (sticky fall-back: a frontier that overflowed a bucket -- heavy duplication of few lattice states -- does so call
    // after call; the next calls of this context take the capacity-free form straight away instead of paying claimed +
    // synchronisation + exact each time, and the claimed form gets another try after kIdentityBackoff calls)
    constexpr int kIdentityBackoff = 8;
    const bool backoff = c->id_backoff > 0 && !(e_cl && atoi(e_cl) == 1);
    if (backoff) c->id_backoff--;
    const bool claimed = levels == 2 && !(e_cl && atoi(e_cl) == 0) && !backoff;
    int64_t subcap1 = 0, cap2 = 0, pairs1 = 0, pairs2 = 0, tiles2_max = 0, cur_words = 0;
    size_t total_cl = 0, cl_h0 = 0, cl_h1 = 0, cl_g0 = 0, cl_g1 = 0, cl_cur = 0, cl_ts = 0;
    if (claimed) {
      mplx::identity_claimed_sizes(n, ia.b1, ia.b2, &subcap1, &cap2, &pairs1, &pairs2, &tiles2_max, &cur_words);
      if (e_cap) {  // diagnostic: "subcap1,cap2" (smaller capacities: the overflow path)
        long long x = 0, y = 0;
        if (sscanf(e_cap, "%lld,%lld", &x, &y) == 2 && x >= 16 && y >= 16 && x <= subcap1 && y <= cap2) { subcap1 = x; cap2 = y; }
      }
      cl_h0 = up((size_t)pairs1 * 8); cl_g0 = up((size_t)pairs1 * 4);
      cl_h1 = up((size_t)pairs2 * 8); cl_g1 = up((size_t)pairs2 * 4);
      cl_cur = up((size_t)cur_words * 4); cl_ts = up((size_t)(1 + 512 + tiles2_max) * 4);
      total_cl = cl_h0 + cl_g0 + cl_h1 + cl_g1 + cl_cur + cl_ts;
    }
    if (int rc = ensure(c, c->post_ws, total > total_cl ? total : total_cl)) return rc;
    bool exact = !claimed;
    c->last_identity_form = claimed ? 1 : 2;
    if (claimed) {
      if (!c->id_ovf_host) {
        HIP_TRY(c, hipHostMalloc((void **)&c->id_ovf_host, 64, hipHostMallocCoherent));
        *c->id_ovf_host = 0;
      }
      mplx::IdentityArgs ca = ia;
      char *w = (char *)c->post_ws.p;
      ca.hk[0] = (uint64_t *)w; w += cl_h0;
      ca.hk[1] = (uint64_t *)w; w += cl_h1;
      ca.gi[0] = (uint32_t *)w; w += cl_g0;
      ca.gi[1] = (uint32_t *)w; w += cl_g1;
      ca.cur1 = (uint32_t *)w;
      ca.cur2 = ca.cur1 + 512 * 16;
      ca.ovf = ca.cur2 + ((size_t)1 << (ia.b1 + ia.b2));
      w += cl_cur;
      ca.tile_seg = (uint32_t *)w;
      ca.claimed = 1;
      ca.ovf_host = c->id_ovf_host;
      ca.subcap1 = subcap1;
      ca.cap2 = cap2;
      ca.tiles2_max = tiles2_max;
      HIP_TRY(c, mplx::launch_identity_claimed(ca, cur_words, c->stream));
      // heuristic and flags behind it at once (no idle GPU while the host learns the outcome), then the one host round
      // trip of the pass: did every run fit its bucket?  Heavy duplication of few lattice states can fill one bucket
      // beyond any fixed capacity; the exact form below has no capacities, and the (idempotent) heuristic / flags kernel
      // runs again on its canon[].
      // (lists without state rows: the pass only ORs the first-occurrence bit into the flags row the expansion launch
      // wrote, which is not idempotent on a canon[] that turns out wrong -- queued all the same, and undone below in
      // the one case that needs it)
      HIP_TRY(c, mplx::launch_post_lists(D, a, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      if (*(volatile int32_t *)c->id_ovf_host == 0) return MPLX_OK;
      if (!a.state && a.flags) HIP_TRY(c, mplx::launch_post_clear_first_flags(a, c->stream));
      *c->id_ovf_host = 0;
      exact = true;
      c->last_identity_form = 3;
      c->id_backoff = kIdentityBackoff;
    }
    if (exact) {
      char *w = (char *)c->post_ws.p;
      for (int l = 0; l < levels; l++) { ia.hk[l] = (uint64_t *)w; w += sz_h; }
      for (int l = 0; l < levels; l++) { ia.gi[l] = (uint32_t *)w; w += sz_g; }
      ia.cnt[0] = (uint32_t *)w; w += sz_c1;
      ia.cnt[1] = (uint32_t *)w; w += sz_c2;
      ia.tot[0] = (uint32_t *)w; w += sz_t1;
      ia.tot[1] = (uint32_t *)w; w += sz_t2;
      ia.range = (uint32_t *)w; w += sz_r;
      ia.seg = (uint32_t *)w;
      HIP_TRY(c, mplx::launch_identity(ia, ctr1, ctr2, c->stream));
    }
  } else if (d_out->canon) {
    c->last_identity_form = 0;
    uint64_t cap = 1024;
    while (cap < max_entries) cap <<= 1;  // >= the emitted successors whatever the frontier
    if (cap < 2 * max_entries && cap < (1ull << 27)) cap <<= 1;
    if (int rc = ensure(c, c->post_keys, (cap + 1) * sizeof(mplx::PostArgs::Slot))) return rc;
    // all bytes 0xff: key = ~0 (empty), val = 0xffffffff (above every list index, unsigned atomicMin)
    HIP_TRY(c, hipMemsetAsync(c->post_keys.p, 0xff, (cap + 1) * sizeof(mplx::PostArgs::Slot), c->stream));
    a.keys = (mplx::PostArgs::Slot *)c->post_keys.p;
    a.cap = cap;
  }
  if (a.state || a.flags || a.keys) HIP_TRY(c, mplx::launch_post_lists(D, a, c->stream));
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
