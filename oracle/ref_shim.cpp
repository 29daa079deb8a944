/*
 * ref_shim.cpp -- drives the REFERENCE'S OWN get_succ through the C interface
 * of mpl_oracle.h, to validate the restatement in mpl_oracle.cpp.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Built only where /root/reference exists
 * (oracle/Makefile target `ref`), output oracle/_ref/libmpl_ref.so (git-ignored,
 * travels to the GPU box as a prebuilt file).  No reference source is copied:
 * the headers are included from where they lie,
 *     <mpl_planner/env/env_map.h>   (get_succ, traverse_primitive)
 *     <mpl_collision/map_util.h>    (MapUtil)
 * and everything they include (env_base.h, trajectory.h, primitive.h,
 * waypoint.h, math.h, ...).  Eigen and Boost are not installed here, so the
 * include path puts oracle/stub_include first: a minimal stand-in for the
 * handful of Eigen vector operations and boost::hash_combine (classic form)
 * those headers use.  All polynomial / limit / sampling / cost arithmetic that
 * is executed is therefore the reference's own code.
 */
#include <mpl_collision/map_util.h>
#include <mpl_planner/env/env_map.h>

#include <chrono>
#include <cmath>
#include <limits>
#include <memory>
#include <atomic>
#include <thread>
#include <vector>

#include "mpl_oracle.h"

namespace {

template <int D>
struct Rig {
  std::shared_ptr<MPL::MapUtil<D>> map_util;
  std::unique_ptr<MPL::env_map<D>> env;

  static std::shared_ptr<MPL::MapUtil<D>> make_map(const mpl_oracle_env *e) {
    auto mu = std::make_shared<MPL::MapUtil<D>>();
    Vecf<D> ori;
    Veci<D> dim;
    size_t n = 1;
    for (int i = 0; i < D; i++) { ori(i) = e->origin[i]; dim(i) = e->map_dim[i]; n *= (size_t)e->map_dim[i]; }
    MPL::Tmap cells(e->map, e->map + n);
    mu->setMap(ori, dim, cells, e->res);
    return mu;
  }

  /* `shared`: a MapUtil built once for all threads of a run (env_map only reads it, env_map.h:288) */
  explicit Rig(const mpl_oracle_env *e, std::shared_ptr<MPL::MapUtil<D>> shared = nullptr) {
    map_util = shared ? shared : make_map(e);
    size_t n = 1;
    for (int i = 0; i < D; i++) n *= (size_t)e->map_dim[i];
    env.reset(new MPL::env_map<D>(map_util));
    vec_E<VecDf> U;
    for (int i = 0; i < e->nU; i++) {
      VecDf u(e->udim);
      for (int k = 0; k < e->udim; k++) u(k) = e->U[(size_t)i * e->udim + k];
      U.push_back(u);
    }
    env->set_u(U);
    env->set_v_max(e->v_max);
    env->set_a_max(e->a_max);
    env->set_j_max(e->j_max);
    env->set_yaw_max(e->yaw_max);
    env->set_dt(e->dt);
    env->set_w(e->w);
    env->set_wyaw(e->wyaw);
    env->set_potential_weight(e->potential_weight);
    env->set_gradient_weight(e->gradient_weight);
    if (e->potential) env->set_potential_map(std::vector<int8_t>(e->potential, e->potential + n));
    if (e->region) {
      std::vector<bool> reg(n);
      for (size_t i = 0; i < n; i++) reg[i] = e->region[i] != 0;
      env->set_search_region(reg);
    }
  }
};

template <int D>
Waypoint<D> load_wp(const double *nodes, int64_t stride, int64_t k, int control) {
  Waypoint<D> w((Control::Control)control);
  for (int i = 0; i < D; i++) {
    w.pos(i) = nodes[(0 * D + i) * stride + k];
    w.vel(i) = nodes[(1 * D + i) * stride + k];
    w.acc(i) = nodes[(2 * D + i) * stride + k];
    w.jrk(i) = nodes[(3 * D + i) * stride + k];
  }
  w.yaw = nodes[(4 * D) * stride + k];
  w.t = nodes[(4 * D + 1) * stride + k];
  return w;
}

template <int D>
void store_wp(double *out, int64_t stride, int64_t slot, const Waypoint<D> &w) {
  for (int i = 0; i < D; i++) {
    out[(0 * D + i) * stride + slot] = w.pos(i);
    out[(1 * D + i) * stride + slot] = w.vel(i);
    out[(2 * D + i) * stride + slot] = w.acc(i);
    out[(3 * D + i) * stride + slot] = w.jrk(i);
  }
  out[(4 * D) * stride + slot] = w.yaw;
  out[(4 * D + 1) * stride + slot] = w.t;
}

/* The reference does not report how many samples traverse_primitive executed;
 * recount them with the reference's own loop header (env_map.h:91-99) and its
 * own evaluate / floatToInt / isOutside / isOccupied calls. */
template <int D>
int count_iters(const mpl_oracle_env *e, const MPL::MapUtil<D> &mu_c, const Primitive<D> &pr) {
  MPL::MapUtil<D> &mu = const_cast<MPL::MapUtil<D> &>(mu_c);
  decimal_t max_v = 0;
  for (int i = 0; i < D; i++)
    if (pr.max_vel(i) > max_v) max_v = pr.max_vel(i);
  int n = std::max(5, (int)std::ceil(max_v * pr.t() / mu.getRes()));
  decimal_t dt = pr.t() / n;
  int it = 0;
  for (decimal_t t = 0; t < pr.t(); t += dt) {
    it++;
    const auto pt = pr.evaluate(t);
    const Veci<D> pn = mu.floatToInt(pt.pos);
    if (mu.isOutside(pn)) return it;
    const int idx = mu.getIndex(pn);
    if (e->region && !e->region[idx]) return it;
    if (e->potential) {
      if (e->potential[idx] >= 100) return it;
    } else if (mu.isOccupied(pn))
      return it;
  }
  return it;
}

/* start gate of a timed run: the threads set their env up, then all start together */
struct Gate {
  std::atomic<int> ready{0};
  std::atomic<bool> go{false};
  void arrive_and_wait() {
    ready.fetch_add(1);
    while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
  }
};

template <int D>
void expand_range(const mpl_oracle_env *e, const double *nodes, int64_t n_nodes, int64_t lo, int64_t hi,
                  mpl_oracle_out *out, mpl_oracle_stats *st, bool dense,
                  std::shared_ptr<MPL::MapUtil<D>> shared_map = nullptr, Gate *gate = nullptr,
                  std::atomic<int64_t> *next = nullptr, int64_t chunk = 1) {
  Rig<D> rig(e, shared_map);
  if (gate) gate->arrive_and_wait();
  const int64_t n_slots = n_nodes * e->nU;
  vec_E<Waypoint<D>> succ;
  std::vector<decimal_t> cost;
  std::vector<int> act;
  mpl_oracle_stats s = {0, 0, 0, 0, 0, 0, 0.0};
  /* `next`: the threads claim chunks of nodes from a shared counter instead of one fixed range each -- a timed run on
   * a busy many-core host then ends when the work does, not when the unluckiest thread's range does */
  for (bool more = true; more;) {
  if (next) {
    lo = next->fetch_add(chunk, std::memory_order_relaxed);
    hi = std::min<int64_t>(n_nodes, lo + chunk);
    if (lo >= n_nodes) break;
  } else {
    more = false;
  }
  for (int64_t k = lo; k < hi; k++) {
    const Waypoint<D> curr = load_wp<D>(nodes, n_nodes, k, e->control);
    rig.env->expanded_nodes_.clear();
    rig.env->expanded_edges_.clear();
    rig.env->get_succ(curr, succ, cost, act);  /* <- the reference's hot path */
    s.pairs += e->nU;
    s.emitted += (int64_t)succ.size();
    for (size_t m = 0; m < cost.size(); m++)
      if (!std::isinf(cost[m])) { s.finite++; s.sum_finite_cost += cost[m]; }
    if (!dense) continue;
    size_t m = 0;
    for (int i = 0; i < e->nU; i++) {
      const int64_t slot = k * e->nU + i;
      Primitive<D> pr(curr, rig.env->U_[i], e->dt);
      if (m < act.size() && act[m] == i) {
        const bool fin = !std::isinf(cost[m]);
        int iters = 0;
        if (!(curr.pos == succ[m].pos)) iters = count_iters<D>(e, *rig.map_util, pr);
        s.samples += iters;
        if (out) {
          if (out->status) out->status[slot] = fin ? MPL_SLOT_FINITE : MPL_SLOT_BLOCKED;
          if (out->cost) out->cost[slot] = cost[m];
          if (out->hash) out->hash[slot] = hash_value(succ[m]);
          if (out->state) store_wp<D>(out->state, n_slots, slot, succ[m]);
          if (out->iters) out->iters[slot] = iters;
        }
        m++;
      } else {
        Waypoint<D> tn = pr.evaluate(e->dt);
        const bool same = (tn == curr);
        if (same) s.skip_same++; else s.skip_dyn++;
        tn.t = curr.t + e->dt;
        if (out) {
          if (out->status) out->status[slot] = same ? MPL_SLOT_SKIP_SAME : MPL_SLOT_SKIP_DYN;
          if (out->cost) out->cost[slot] = std::numeric_limits<double>::infinity();
          if (out->hash) out->hash[slot] = hash_value(tn);
          if (out->state) store_wp<D>(out->state, n_slots, slot, tn);
          if (out->iters) out->iters[slot] = 0;
        }
      }
    }
  }
  }
  *st = s;
}

void add_stats(mpl_oracle_stats &a, const mpl_oracle_stats &b) {
  a.pairs += b.pairs; a.emitted += b.emitted; a.finite += b.finite;
  a.skip_same += b.skip_same; a.skip_dyn += b.skip_dyn; a.samples += b.samples;
  a.sum_finite_cost += b.sum_finite_cost;
}

template <typename F>
void run_threads(int64_t n, int threads, F f) {
  if (threads < 1) threads = 1;
  if ((int64_t)threads > n) threads = (int)std::max<int64_t>(1, n);
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; t++) pool.emplace_back(f, t, n * t / threads, n * (t + 1) / threads);
  for (auto &th : pool) th.join();
}

/* `seconds` (optional): wall time of the expansion alone -- one MapUtil shared by all threads, per-thread
 * env_map objects set up before the clock starts */
int run(const mpl_oracle_env *env, const double *nodes, int64_t n_nodes, mpl_oracle_out *out, int threads,
        mpl_oracle_stats *stats, bool dense, double *seconds = nullptr) {
  if (!env || (env->dim != 2 && env->dim != 3) || !env->map || !env->U || env->nU <= 0 || n_nodes < 0) return -1;
  if (threads < 1) threads = 1;
  if ((int64_t)threads > n_nodes) threads = (int)std::max<int64_t>(1, n_nodes);
  std::vector<mpl_oracle_stats> per((size_t)threads, mpl_oracle_stats{0, 0, 0, 0, 0, 0, 0.0});
  std::shared_ptr<MPL::MapUtil<2>> m2;
  std::shared_ptr<MPL::MapUtil<3>> m3;
  if (env->dim == 2) m2 = Rig<2>::make_map(env); else m3 = Rig<3>::make_map(env);
  Gate gate;
  /* timed runs: chunks of nodes from a shared counter, about 32 per thread (at least 16 nodes, at most 1024) */
  std::atomic<int64_t> next{0};
  const int64_t chunk = std::min<int64_t>(1024, std::max<int64_t>(16, n_nodes / ((int64_t)threads * 32)));
  std::atomic<int64_t> *claim = (seconds && threads > 1) ? &next : nullptr;
  std::chrono::steady_clock::time_point t0;
  std::thread starter([&] {
    while (gate.ready.load() < threads) std::this_thread::yield();
    t0 = std::chrono::steady_clock::now();
    gate.go.store(true, std::memory_order_release);
  });
  run_threads(n_nodes, threads, [&](int t, int64_t lo, int64_t hi) {
    if (env->dim == 2) expand_range<2>(env, nodes, n_nodes, lo, hi, out, &per[(size_t)t], dense, m2, &gate, claim, chunk);
    else expand_range<3>(env, nodes, n_nodes, lo, hi, out, &per[(size_t)t], dense, m3, &gate, claim, chunk);
  });
  starter.join();
  if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (stats) {
    *stats = {0, 0, 0, 0, 0, 0, 0.0};
    for (auto &p : per) add_stats(*stats, p);
  }
  return 0;
}

}  // namespace

extern "C" {

int mpl_oracle_expand(const mpl_oracle_env *env, const double *nodes, int64_t n_nodes, mpl_oracle_out *out,
                      int threads, mpl_oracle_stats *stats) {
  return run(env, nodes, n_nodes, out, threads, stats, true);
}

double mpl_oracle_time_expand(const mpl_oracle_env *env, const double *nodes, int64_t n_nodes, int threads,
                              int reps, mpl_oracle_stats *stats) {
  double best = 1e300;
  for (int r = 0; r < (reps < 1 ? 1 : reps); r++) {
    double sec = 0;
    if (run(env, nodes, n_nodes, nullptr, threads, stats, false, &sec) != 0) return -1.0;
    best = std::min(best, sec);
  }
  return best;
}

}  // extern "C"

template <int D>
static int get_succ_lists(const mpl_oracle_env *e, const double *node, double *succ, double *cost,
                          int32_t *action, int32_t *n_succ) {
  Rig<D> rig(e);
  vec_E<Waypoint<D>> s;
  std::vector<decimal_t> c;
  std::vector<int> a;
  rig.env->get_succ(load_wp<D>(node, 1, 0, e->control), s, c, a);
  const int F = 4 * D + 2;
  for (size_t m = 0; m < s.size(); m++) {
    double *o = succ + m * F;
    for (int i = 0; i < D; i++) {
      o[0 * D + i] = s[m].pos(i); o[1 * D + i] = s[m].vel(i);
      o[2 * D + i] = s[m].acc(i); o[3 * D + i] = s[m].jrk(i);
    }
    o[4 * D] = s[m].yaw;
    o[4 * D + 1] = s[m].t;
    cost[m] = c[m];
    action[m] = a[m];
  }
  *n_succ = (int32_t)s.size();
  return 0;
}

/* env_map<Dim>::is_goal on an all-free map (so that the ray trace of env_map.h:38-43 cannot fail) */
template <int D>
static int32_t goal_tol(const double *wp, const double *goal, double tol_pos, double tol_vel, double tol_acc,
                        double tol_yaw) {
  auto mu = std::make_shared<MPL::MapUtil<D>>();
  Vecf<D> ori;
  Veci<D> dim;
  for (int i = 0; i < D; i++) { ori(i) = -1e6; dim(i) = 4; }
  mu->setMap(ori, dim, MPL::Tmap((size_t)(D == 2 ? 16 : 64), 0), 5e5);
  MPL::env_map<D> env(mu);
  env.set_tol_pos(tol_pos);
  env.set_tol_vel(tol_vel);
  env.set_tol_acc(tol_acc);
  env.set_tol_yaw(tol_yaw);
  env.set_goal(load_wp<D>(goal, 1, 0, 0x1f));
  return env.is_goal(load_wp<D>(wp, 1, 0, 0x1f)) ? 1 : 0;
}

/* Batched edge re-validation through the reference's own env_map (forward_action, is_free(Primitive),
 * calculate_intrinsic_cost) and the loop body of MapPlanner::getLinkedNodes (map_planner.cpp:136-153). */
template <int D>
static void ref_check_edges(const mpl_oracle_env *e, const double *parents, const int32_t *actions, int64_t n,
                            uint8_t *free_out, double *cost_out, int32_t *cells, int32_t *cell_count, int32_t cell_cap) {
  Rig<D> rig(e);
  for (int64_t k = 0; k < n; k++) {
    const Waypoint<D> par = load_wp<D>(parents, n, k, e->control);
    Primitive<D> pr;
    rig.env->forward_action(par, actions[k], pr);
    decimal_t max_v = 0;
    for (int i = 0; i < D; i++)
      if (pr.max_vel(i) > max_v) max_v = pr.max_vel(i);
    const int N = 1.0 * std::ceil(max_v * pr.t() / rig.map_util->getRes());
    bool is_free = false;
    int n_cells = 0;
    if (N > 0) {  /* N == 0 is undefined behaviour in the reference (sample time NaN) */
      is_free = rig.env->is_free(pr);
      if (cells) {
        int prev_id = -1;
        vec_E<Waypoint<D>> ws = pr.sample(N);
        for (const auto &w : ws) {
          int id = rig.map_util->getIndex(rig.map_util->floatToInt(w.pos));
          if (id != prev_id) {
            if (n_cells < cell_cap) cells[k * cell_cap + n_cells] = id;
            n_cells++;
            prev_id = id;
          }
        }
      }
    }
    if (free_out) free_out[k] = is_free ? 1 : 0;
    if (cost_out) cost_out[k] = is_free ? rig.env->calculate_intrinsic_cost(pr) : std::numeric_limits<decimal_t>::infinity();
    if (cell_count) cell_count[k] = n_cells;
  }
}

extern "C" int mpl_oracle_check_edges(const mpl_oracle_env *env, const double *parents, const int32_t *actions,
                                      int64_t n, uint8_t *free_out, double *cost_out, int32_t *cells,
                                      int32_t *cell_count, int32_t cell_cap) {
  if (!env || (n > 0 && (!parents || !actions))) return -1;
  if (env->dim == 2) ref_check_edges<2>(env, parents, actions, n, free_out, cost_out, cells, cell_count, cell_cap);
  else if (env->dim == 3) ref_check_edges<3>(env, parents, actions, n, free_out, cost_out, cells, cell_count, cell_cap);
  else return -1;
  return 0;
}

extern "C" {

int mpl_oracle_get_succ(void *user, const double *node, double *succ, double *cost, int32_t *action,
                        int32_t *n_succ) {
  const mpl_oracle_env *e = (const mpl_oracle_env *)user;
  if (!e || !node || !succ || !cost || !action || !n_succ) return -1;
  return e->dim == 2 ? get_succ_lists<2>(e, node, succ, cost, action, n_succ)
                     : get_succ_lists<3>(e, node, succ, cost, action, n_succ);
}

int mpl_oracle_batch(void *user, const double *nodes, int64_t n, uint8_t *status, double *cost,
                     double *state) {
  mpl_oracle_out o = {status, cost, nullptr, state, nullptr};
  return mpl_oracle_expand((const mpl_oracle_env *)user, nodes, n, &o, 1, nullptr);
}

uint64_t mpl_oracle_hash(int32_t dim, int32_t control, const double *wp) {
  if (dim == 2) return hash_value(load_wp<2>(wp, 1, 0, control));
  return hash_value(load_wp<3>(wp, 1, 0, control));
}

/* env_base::get_heur with the default heur_ignore_dynamics_ = true */
double mpl_oracle_heur(int32_t dim, int32_t control, double w, double v_max, const double *wp,
                       const double *goal) {
  if (dim == 2) {
    MPL::env_base<2> env;
    env.set_w(w);
    env.set_v_max(v_max);
    env.set_goal(load_wp<2>(goal, 1, 0, control));
    return env.get_heur(load_wp<2>(wp, 1, 0, control));
  }
  MPL::env_base<3> env;
  env.set_w(w);
  env.set_v_max(v_max);
  env.set_goal(load_wp<3>(goal, 1, 0, control));
  return env.get_heur(load_wp<3>(wp, 1, 0, control));
}

int32_t mpl_oracle_goal_tol(int32_t dim, const double *wp, const double *goal, double tol_pos, double tol_vel,
                            double tol_acc, double tol_yaw) {
  if (dim == 2) return goal_tol<2>(wp, goal, tol_pos, tol_vel, tol_acc, tol_yaw);
  return goal_tol<3>(wp, goal, tol_pos, tol_vel, tol_acc, tol_yaw);
}

int32_t mpl_oracle_loop_count(double T, int32_t n) {
  decimal_t dt = T / n;
  int32_t it = 0;
  for (decimal_t t = 0; t < T; t += dt) it++;
  return it;
}

}  // extern "C"
