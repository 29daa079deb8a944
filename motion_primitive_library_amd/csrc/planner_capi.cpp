// planner_capi.cpp -- C entry points of the host search (include/mplx.h,
// "host search" section) on top of host_planner.hpp.
#include "../../include/mplx.h"
#include "../../include/mplx_debug.h"
#include "host_planner.hpp"
#include "host_lpastar.hpp"
#include "mplx_ctx.h"

#include <new>
#include <string>

struct mplx_planner {
  mplx::host::Planner pl;
  mplx::host::LpaPlanner lpa;  // PlannerBase::setLPAstar(true): the state space that outlives plan()
  bool use_lpa = false;
  const mplx::host::PlanResult &result() const { return use_lpa ? lpa.last : pl.last; }
  mplx_ctx *ctx = nullptr;
  bool device_heur = false;  // the expansion launches deliver the default heuristic of every successor (mplx_set_goal)
  bool want_device_heur = getenv("MPLX_PLAN_DEVICE_HEUR") != nullptr && atoi(getenv("MPLX_PLAN_DEVICE_HEUR")) != 0;
  std::string err;
};

namespace {

// engine-backed providers
int engine_single(void *user, const double *node, double *succ, double *cost, int32_t *action, int32_t *n) {
  mplx_planner *p = (mplx_planner *)user;
  return mplx_get_succ(p->ctx, node, succ, cost, action, n);
}
int engine_batch(void *user, const double *nodes, int64_t n, uint8_t *status, double *cost, double *state) {
  mplx_planner *p = (mplx_planner *)user;
  mplx_succ o{};
  o.status = status;
  o.cost = cost;
  o.state = state;
  o.state_stride = n * p->pl.nU;
  return mplx_expand(p->ctx, nodes, n, n, &o);
}

int engine_lists(void *user, const double *nodes, int64_t n, int32_t *count, int32_t *action, double *cost,
                 uint64_t *hash, double *state) {
  mplx_planner *p = (mplx_planner *)user;
  mplx_succ_lists o{};
  o.count = count;
  o.action = action;
  o.cost = cost;
  o.hash = hash;
  o.state = state;
  o.state_stride = n * p->pl.nU;
  return mplx_expand_lists(p->ctx, nodes, n, n, &o);
}

int engine_packed(void *user, const double *nodes, int64_t n, mplx::host::PackedView *out) {
  mplx_planner *p = (mplx_planner *)user;
  mplx_detail::PackedLists pl;
  if (int rc = mplx_detail::expand_lists_packed(p->ctx, nodes, n, n, !p->pl.edges_only || p->pl.check_states, &pl, p->device_heur)) return rc;
  out->total = pl.total;
  out->count = pl.count;
  out->offs = pl.offs;
  out->cost = pl.cost;
  out->hash = pl.hash;
  out->action = pl.action;
  out->state = pl.state;
  out->heur = pl.heur;
  return 0;
}

int engine_edges(void *user, const double *parents, const int32_t *actions, int64_t n, uint8_t *free_flag, double *cost,
                 int32_t *cells, int32_t *cell_count, int32_t cell_cap) {
  mplx_planner *p = (mplx_planner *)user;
  mplx_edges_out o{};
  o.free_flag = free_flag;
  o.cost = cost;
  o.cells = cells;
  o.cell_count = cell_count;
  o.cell_cap = cell_cap;
  return mplx_check_edges(p->ctx, parents, actions, n, n, &o);
}

int fail(mplx_planner *p, int code, const char *msg) {
  if (p) p->err = msg;
  return code;
}

}  // namespace

extern "C" {

int mplx_planner_create(int dim, mplx_planner **out) {
  if (!out || (dim != 2 && dim != 3)) return MPLX_ERR_ARG;
  mplx_planner *p = new (std::nothrow) mplx_planner();
  if (!p) return MPLX_ERR_ARG;
  p->pl.dim = dim;
  p->pl.grid.dim = dim;
  p->lpa.cfg = &p->pl;
  *out = p;
  return MPLX_OK;
}

void mplx_planner_destroy(mplx_planner *p) { delete p; }

const char *mplx_planner_last_error(const mplx_planner *p) { return p ? p->err.c_str() : ""; }

int mplx_planner_attach_ctx(mplx_planner *p, mplx_ctx *ctx) {
  if (!p || !ctx) return MPLX_ERR_ARG;
  p->ctx = ctx;
  p->pl.single = engine_single;
  p->pl.batched = engine_batch;
  p->pl.lists = engine_lists;
  p->pl.packed = getenv("MPLX_PLAN_NO_PACKED") ? nullptr : engine_packed;  // the knob: A/B only
  // edges only: the device delivers (action, cost, hash); the states of new nodes are evaluated on the host
  // (host_planner.hpp::forward_state).  MPLX_PLAN_FULL_STATES=1 moves the states as well; MPLX_PLAN_CHECK_STATES=1
  // moves them and counts disagreements with the host evaluation (tests).
  p->pl.edges_only = getenv("MPLX_PLAN_FULL_STATES") == nullptr;
  p->pl.check_states = getenv("MPLX_PLAN_CHECK_STATES") != nullptr;
  p->pl.check_perturb = getenv("MPLX_PLAN_CHECK_PERTURB") ? atoi(getenv("MPLX_PLAN_CHECK_PERTURB")) : -1;
  p->pl.user = p;
  p->lpa.edges = engine_edges;
  p->lpa.edges_user = p;
  return MPLX_OK;
}

int mplx_planner_set_provider(mplx_planner *p, mplx_succ_fn single, mplx_batch_fn batched, void *user) {
  if (!p || (!single && !batched)) return MPLX_ERR_ARG;
  p->ctx = nullptr;
  p->pl.single = single;
  p->pl.batched = batched;
  p->pl.lists = nullptr;
  p->pl.packed = nullptr;
  p->pl.user = user;
  return MPLX_OK;
}

int mplx_planner_set_map(mplx_planner *p, const int8_t *cells, const int32_t *dim, const double *origin,
                         double res) {
  if (!p || !cells || !dim || !origin || !(res > 0)) return fail(p, MPLX_ERR_ARG, "mplx_planner_set_map: bad arguments");
  size_t n = 1;
  for (int i = 0; i < p->pl.dim; i++) {
    if (dim[i] <= 0) return fail(p, MPLX_ERR_ARG, "mplx_planner_set_map: bad dim");
    p->pl.grid.n[i] = dim[i];
    p->pl.grid.origin[i] = origin[i];
    n *= (size_t)dim[i];
  }
  p->pl.grid.res = res;
  try { p->pl.grid.cells.assign(cells, cells + n); } catch (...) { return fail(p, MPLX_ERR_NOMEM, "mplx_planner_set_map: out of host memory"); }
  return MPLX_OK;
}

int mplx_planner_edit_map(mplx_planner *p, const int64_t *cell_index, const int8_t *values, int64_t n) {
  if (!p) return MPLX_ERR_ARG;
  if (n < 0 || (n > 0 && (!cell_index || !values))) return fail(p, MPLX_ERR_ARG, "mplx_planner_edit_map: bad arguments");
  const int64_t n_cells = (int64_t)p->pl.grid.cells.size();
  if (n_cells == 0 && n > 0) return fail(p, MPLX_ERR_STATE, "mplx_planner_edit_map: set the map first");
  for (int64_t i = 0; i < n; i++)
    if (cell_index[i] < 0 || cell_index[i] >= n_cells) return fail(p, MPLX_ERR_ARG, "mplx_planner_edit_map: cell index outside the map");
  for (int64_t i = 0; i < n; i++) p->pl.grid.cells[(size_t)cell_index[i]] = values[i];  // in order: the last value of a repeated cell stays
  return MPLX_OK;
}

int mplx_planner_set_controls(mplx_planner *p, const double *U, int32_t nU, int32_t udim) {
  if (!p || !U || nU <= 0 || udim < p->pl.dim) return fail(p, MPLX_ERR_ARG, "mplx_planner_set_controls: bad arguments");
  try { p->pl.U.assign(U, U + (size_t)nU * udim); } catch (...) { return fail(p, MPLX_ERR_NOMEM, "mplx_planner_set_controls: out of host memory"); }
  p->pl.nU = nU;
  p->pl.udim = udim;
  return MPLX_OK;
}

int mplx_planner_configure(mplx_planner *p, const mplx_planner_config *c) {
  if (!p || !c) return MPLX_ERR_ARG;
  p->pl.control = c->control;
  p->pl.goal_control = c->goal_control ? c->goal_control : c->control;
  p->pl.max_expand = c->max_expand;
  p->pl.batch = c->batch < 1 ? 1 : c->batch;
  p->pl.dt = c->dt;
  p->pl.w = c->w;
  p->pl.v_max = c->v_max;
  p->pl.eps = c->epsilon;
  p->pl.tol_pos = c->tol_pos;
  p->pl.tol_vel = c->tol_vel;
  p->pl.tol_acc = c->tol_acc;
  p->pl.tol_yaw = c->tol_yaw;
  return MPLX_OK;
}

int mplx_planner_plan(mplx_planner *p, const double *start, const double *goal, mplx_plan_summary *out) {
  if (!p || !start || !goal || !out) return MPLX_ERR_ARG;
  if (p->pl.grid.cells.empty()) return fail(p, MPLX_ERR_STATE, "mplx_planner_plan: map not set");
  if (p->pl.nU <= 0) return fail(p, MPLX_ERR_STATE, "mplx_planner_plan: controls not set");
  if (!p->pl.single && !p->pl.batched && !p->pl.packed) return fail(p, MPLX_ERR_STATE, "mplx_planner_plan: no successor provider attached");
  p->device_heur = false;
  if (p->ctx && p->pl.packed && p->pl.eps != 0 && p->want_device_heur && !p->pl.has_prior()) {
    // env_base::set_goal for the device (planner_base.h:301): the kernels compute what graph_search.h:84-88 asks of
    // the env for every new successor -- the default heuristic -- while the successor is in registers.
    // OFF by default (mplx_planner_use_device_heuristic / MPLX_PLAN_DEVICE_HEUR=1): one successor in fifteen creates a
    // node, so the search needs 7 % of the row it pays 8 bytes per successor for on the PCIe link -- measured on the
    // 3D problems 262.6 against 254.3 ms (160^3) and 13.2 against 13.0 ms (120^3) with the row
    // (profiles/r05_device_heur_ab.txt); evaluating |pos - goal| of the few new successors costs the host less.
    // The row pays where the consumer is on the device (mplx_post_*: no second pass over hash and position rows).
    mplx_goal_spec g{};
    g.goal = goal;
    g.control = p->pl.control;
    g.goal_control = p->pl.goal_control;
    g.w = p->pl.w;
    g.v_max = p->pl.v_max;
    g.tol_pos = p->pl.tol_pos; g.tol_vel = p->pl.tol_vel; g.tol_acc = p->pl.tol_acc; g.tol_yaw = p->pl.tol_yaw;
    if (mplx_set_goal(p->ctx, &g) == MPLX_OK) p->device_heur = true;
  }
  int rc;
  try {
    rc = p->use_lpa ? p->lpa.plan(start, goal) : p->pl.plan(start, goal);
  } catch (const std::exception &e) {  // no exception crosses the C ABI
    p->err = std::string("mplx_planner_plan: ") + e.what();
    return MPLX_ERR_NOMEM;
  } catch (...) {
    p->err = "mplx_planner_plan: unexpected exception";
    return MPLX_ERR_NOMEM;
  }
  // the search is over: its resident expansion kernel (if one served the batches) leaves now instead of polling until its
  // idle time-out -- anything else in the process that synchronises the device would wait for it
  if (p->ctx) (void)mplx_detail::svc_stop(p->ctx);
  if (rc != 0) {
    p->err = "successor provider failed";
    if (p->ctx) p->err += std::string(": ") + mplx_last_error(p->ctx);
    return rc < 0 ? rc : MPLX_ERR_HIP;
  }
  const mplx::host::PlanResult &r = p->result();
  out->ok = r.ok ? 1 : 0;
  out->expansions = r.expansions;
  out->closed = r.closed;
  out->opened = r.opened;
  out->nodes = r.nodes;
  out->device_launches = r.device_launches;
  out->pairs = r.pairs;
  out->state_mismatches = (int32_t)(r.state_mismatches > 0x7fffffff ? 0x7fffffff : r.state_mismatches);
  out->cost = r.cost;
  out->total_time = r.total_time;
  for (int i = 0; i < 4; i++) out->J[i] = r.J[i];
  out->segments = (int32_t)r.traj_actions.size();
  return MPLX_OK;
}

int mplx_planner_use_device_heuristic(mplx_planner *p, int on) {
  if (!p) return MPLX_ERR_ARG;
  p->want_device_heur = on != 0;
  return MPLX_OK;
}

int mplx_planner_set_prior_trajectory(mplx_planner *p, const mplx_planner *from) {
  if (!p) return MPLX_ERR_ARG;
  if (!from) { p->pl.clear_prior(); return MPLX_OK; }
  const mplx::host::PlanResult &r = from->result();
  if (!r.ok || r.traj_actions.empty()) return fail(p, MPLX_ERR_STATE, "mplx_planner_set_prior_trajectory: the other planner holds no trajectory");
  if (from->pl.dim != p->pl.dim) return fail(p, MPLX_ERR_ARG, "mplx_planner_set_prior_trajectory: dimensions differ");
  if (p->pl.grid.cells.empty()) return fail(p, MPLX_ERR_STATE, "mplx_planner_set_prior_trajectory: set the map first");
  try {
    if (p->pl.set_prior_trajectory(r.traj_nodes.data(), r.traj_actions.data(), (int)r.traj_actions.size(), from->pl.control,
                                   from->pl.U.data(), from->pl.udim, from->pl.dt) != 0)
      return fail(p, MPLX_ERR_STATE, "mplx_planner_set_prior_trajectory: configure dt (> 0) and the map of this planner first");
  } catch (...) {
    return fail(p, MPLX_ERR_NOMEM, "mplx_planner_set_prior_trajectory: out of host memory");
  }
  return MPLX_OK;
}

namespace {
struct PotSource { const int8_t *host; mplx_ctx *ctx; };
int pot_lookup(void *user, const int64_t *idx, int64_t n, int8_t *out) {
  const PotSource *s = (const PotSource *)user;
  if (s->host) {
    for (int64_t i = 0; i < n; i++) out[i] = s->host[idx[i]];
    return 0;
  }
  return mplx_read_cells(s->ctx, 1, idx, n, out);
}
}  // namespace

int mplx_planner_set_prior_trajectory_potential(mplx_planner *p, const mplx_planner *from, const int8_t *potential,
                                                double potential_weight, double gradient_weight) {
  if (!p) return MPLX_ERR_ARG;
  if (!from) { p->pl.clear_prior(); return MPLX_OK; }
  const mplx::host::PlanResult &r = from->result();
  if (!r.ok || r.traj_actions.empty()) return fail(p, MPLX_ERR_STATE, "mplx_planner_set_prior_trajectory_potential: the other planner holds no trajectory");
  if (from->pl.dim != p->pl.dim) return fail(p, MPLX_ERR_ARG, "mplx_planner_set_prior_trajectory_potential: dimensions differ");
  if (p->pl.grid.cells.empty()) return fail(p, MPLX_ERR_STATE, "mplx_planner_set_prior_trajectory_potential: set the map first");
  if (!potential && !p->ctx)
    return fail(p, MPLX_ERR_STATE, "mplx_planner_set_prior_trajectory_potential: no potential map given and no context attached to read it from");
  PotSource src{potential, p->ctx};
  try {
    const int rc = p->pl.set_prior_trajectory(r.traj_nodes.data(), r.traj_actions.data(), (int)r.traj_actions.size(), from->pl.control,
                                              from->pl.U.data(), from->pl.udim, from->pl.dt, pot_lookup, &src, potential_weight,
                                              gradient_weight);
    if (rc == -2) {
      const std::string why = std::string("mplx_planner_set_prior_trajectory_potential: the potential map could not be read: ") +
                              (p->ctx ? mplx_last_error(p->ctx) : "no context");
      return fail(p, MPLX_ERR_STATE, why.c_str());
    }
    if (rc != 0) return fail(p, MPLX_ERR_STATE, "mplx_planner_set_prior_trajectory_potential: configure dt (> 0) and the map of this planner first");
  } catch (...) {
    return fail(p, MPLX_ERR_NOMEM, "mplx_planner_set_prior_trajectory_potential: out of host memory");
  }
  return MPLX_OK;
}

int mplx_planner_timing(const mplx_planner *p, mplx_plan_timing *out) {
  if (!p || !out) return MPLX_ERR_ARG;
  const mplx::host::PlanResult &r = p->result();
  out->total_ms = r.t_total;
  out->provider_ms = r.t_provider;
  out->fill_ms = r.t_fill;
  out->pick_ms = r.t_pick;
  out->relax_ms = r.t_relax;
  out->recover_ms = r.t_recover;
  out->relaxed = r.relaxed;
  out->improved = r.improved;
  out->pushes = r.pushes;
  out->materialised = r.materialised;
  out->heur_from_device = r.heur_from_provider;
  return MPLX_OK;
}

int mplx_planner_trajectory(mplx_planner *p, double *nodes, int32_t *actions, int32_t cap) {
  if (!p || !nodes || !actions) return MPLX_ERR_ARG;
  const mplx::host::PlanResult &r = p->result();
  const int f = p->pl.F();
  const int32_t n = (int32_t)r.traj_actions.size();
  if (cap < n) return fail(p, MPLX_ERR_ARG, "mplx_planner_trajectory: capacity too small");
  for (int32_t s = 0; s < n; s++) {
    for (int k = 0; k < f; k++) nodes[(size_t)s * f + k] = r.traj_nodes[(size_t)s * f + k];
    actions[s] = r.traj_actions[(size_t)s];
  }
  return MPLX_OK;
}

int mplx_planner_trajectory_end(mplx_planner *p, double *node) {
  if (!p || !node) return MPLX_ERR_ARG;
  const mplx::host::PlanResult &r = p->result();
  if (!r.ok || (int)r.traj_end.size() != p->pl.F()) return fail(p, MPLX_ERR_STATE, "mplx_planner_trajectory_end: no trajectory");
  for (int k = 0; k < p->pl.F(); k++) node[k] = r.traj_end[(size_t)k];
  return MPLX_OK;
}

int mplx_selftest_forward_state(int32_t dim, int32_t control, const double *node, const double *u, double dt,
                                double *out) {
  if ((dim != 2 && dim != 3) || !node || !u || !out) return MPLX_ERR_ARG;
  mplx::host::forward_state(dim, control, node, u, dt, out);
  return MPLX_OK;
}

int mplx_planner_set_lpastar(mplx_planner *p, int on) {
  if (!p) return MPLX_ERR_ARG;
  p->use_lpa = on != 0;
  return MPLX_OK;
}

int mplx_planner_reset(mplx_planner *p) {
  if (!p) return MPLX_ERR_ARG;
  p->lpa.reset();
  return MPLX_OK;
}

int mplx_planner_set_edge_provider(mplx_planner *p, mplx_edges_fn fn, void *user) {
  if (!p || !fn) return MPLX_ERR_ARG;
  p->lpa.edges = fn;
  p->lpa.edges_user = user;
  return MPLX_OK;
}

// LPA* bookkeeping calls: same error discipline as plan() (nothing throws across the ABI; the resident kernel of a
// search leaves before another call synchronises the device)
#define MPLX_LPA_CALL(expr, what)                                                                   \
  if (!p) return MPLX_ERR_ARG;                                                                      \
  if (!p->use_lpa) return fail(p, MPLX_ERR_STATE, what ": mplx_planner_set_lpastar(p, 1) first");   \
  if (p->ctx) (void)mplx_detail::svc_stop(p->ctx);                                                  \
  int rc;                                                                                           \
  try { rc = (expr); } catch (...) { return fail(p, MPLX_ERR_NOMEM, what ": out of host memory"); } \
  if (rc != 0) {                                                                                    \
    p->err = what " failed";                                                                        \
    if (p->ctx) p->err += std::string(": ") + mplx_last_error(p->ctx);                              \
    return rc < 0 ? rc : MPLX_ERR_HIP;                                                              \
  }

int mplx_planner_linked_nodes(mplx_planner *p, double *points, int64_t cap_points, int64_t *n_points, int64_t *n_cells,
                              int64_t *n_entries) {
  std::vector<double> pts;
  int64_t np = 0;
  MPLX_LPA_CALL(p->lpa.linked_nodes(points ? &pts : nullptr, &np), "mplx_planner_linked_nodes")
  if (points) {
    const int64_t m = np < cap_points ? np : cap_points;
    for (int64_t i = 0; i < m * p->pl.dim; i++) points[i] = pts[(size_t)i];
  }
  if (n_points) *n_points = np;
  if (n_cells) *n_cells = (int64_t)p->lpa.linked_cells();
  if (n_entries) *n_entries = p->lpa.linked_entries();
  return MPLX_OK;
}

int mplx_planner_update_blocked_nodes(mplx_planner *p, const int32_t *cells, int64_t n) {
  if (p && (n < 0 || (n > 0 && !cells))) return fail(p, MPLX_ERR_ARG, "mplx_planner_update_blocked_nodes: bad arguments");
  MPLX_LPA_CALL(p->lpa.update_blocked(cells, n), "mplx_planner_update_blocked_nodes")
  return MPLX_OK;
}

int mplx_planner_update_cleared_nodes(mplx_planner *p, const int32_t *cells, int64_t n) {
  if (p && (n < 0 || (n > 0 && !cells))) return fail(p, MPLX_ERR_ARG, "mplx_planner_update_cleared_nodes: bad arguments");
  MPLX_LPA_CALL(p->lpa.update_cleared(cells, n), "mplx_planner_update_cleared_nodes")
  return MPLX_OK;
}

int mplx_planner_sub_state_space(mplx_planner *p, int32_t time_step) {
  MPLX_LPA_CALL(p->lpa.sub_state_space(time_step), "mplx_planner_sub_state_space")
  return MPLX_OK;
}

int mplx_planner_closed_set(mplx_planner *p, double *pos, int32_t cap, int32_t *n) {
  if (!p || !n) return MPLX_ERR_ARG;
  *n = p->use_lpa ? p->lpa.closed_positions(pos, cap) : p->pl.closed_positions(pos, cap);
  return MPLX_OK;
}

int mplx_planner_open_set(mplx_planner *p, double *states, int32_t cap, int32_t *n) {
  if (!p || !n) return MPLX_ERR_ARG;
  *n = p->use_lpa ? p->lpa.open_states(states, cap) : p->pl.open_states(states, cap);  // the heap, as planner_base.h:77-81 walks it
  return MPLX_OK;
}

}  // extern "C"
