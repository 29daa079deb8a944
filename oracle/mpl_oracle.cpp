/*
 * mpl_oracle.cpp -- CPU parity oracle for the successor-expansion hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT (see mpl_oracle.h).
 *
 * What it restates (all paths relative to /root/reference, MPL v1.2):
 *   Poly1                 include/mpl_basis/primitive.h:21-198   (Primitive1D)
 *   Prim<D>               include/mpl_basis/primitive.h:205-431  (Primitive<Dim>)
 *   validate_*            include/mpl_basis/primitive.h:450-525
 *   solve_roots/quad      include/mpl_basis/math.h:22-32,117-131
 *   ipow / wrap_angle     include/mpl_basis/math.h:197-205, 15-19
 *   State<D> / lattice_hash  include/mpl_basis/waypoint.h:23-57, 93-135
 *   Grid<D>               include/mpl_collision/map_util.h:34-69,103-108
 *   traverse / expand_node   include/mpl_planner/env/env_map.h:90-132,147-172
 *   intrinsic cost / heuristic  include/mpl_planner/common/env_base.h:343-345,46-64
 *
 * The arithmetic is IEEE binary64, evaluated in exactly the expression order
 * of the reference sources (C++ left-to-right at equal precedence), built with
 * -ffp-contract=off and without -march flags, like the reference's own CMake
 * build (CMakeLists.txt:5-8).  The data structures mirror the reference's
 * (per-call std::vector temporaries in the root finder, max_vel evaluated in
 * validate and again in traverse, a full p/v/a/j evaluation per sample) so the
 * oracle is also a fair "port" CPU baseline.
 *
 * Third-party arithmetic that is NOT under /root/reference:
 *   boost::hash_combine (waypoint.h:98-121) -- Boost version unpinned by the
 *   reference; restated here as the classic (< 1.81) formula
 *       seed ^= (size_t)v + 0x9e3779b9 + (seed << 6) + (seed >> 2)
 *   Eigen (unpinned): only 2-/3-vector norm, normalized(), dot on the path.
 *
 * PARITY PINNING
 *   (1) README.md:199-202 transcript of test_planner_2d: closed set 615,
 *       T = 35.0, J(VEL) = 36.75, J(ACC) = 1.5 -- reproduced by
 *       tests/test_plan_known_answer.py through this oracle.
 *   (2) oracle/_ref: the reference's own headers (primitive.h, waypoint.h,
 *       math.h, map_util.h, env_base.h, env_map.h) compiled where they lie
 *       against a minimal Eigen/Boost stand-in (oracle/stub_include) and
 *       compared slot-for-slot with this file (tests/test_oracle_vs_ref.py,
 *       fixtures in tests/golden/).
 *   The reference's own tests hold no assertions.
 */
#include "mpl_oracle.h"

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

namespace {

constexpr double kInf = std::numeric_limits<double>::infinity();

/* control.h:10-20 */
enum : int {
  C_VEL = 0x01, C_ACC = 0x03, C_JRK = 0x07, C_SNP = 0x0f,
  C_VELxYAW = 0x11, C_ACCxYAW = 0x13, C_JRKxYAW = 0x17, C_SNPxYAW = 0x1f
};

/* math.h:197-205 -- repeated multiplication starting from 1 */
inline double ipow(double t, int n) {
  double r = 1;
  while (n > 0) { r *= t; n--; }
  return r;
}

/* math.h:15-19 */
inline double wrap_angle(double a) {
  while (a > M_PI) a -= 2.0 * M_PI;
  while (a < -M_PI) a += 2.0 * M_PI;
  return a;
}

/* math.h:22-32 : b t^2 + c t + d = 0 */
std::vector<double> quad_roots(double b, double c, double d) {
  std::vector<double> r;
  double p = c * c - 4 * b * d;
  if (p < 0) return r;
  r.push_back((-c - sqrt(p)) / (2 * b));
  r.push_back((-c + sqrt(p)) / (2 * b));
  return r;
}

/* math.h:117-131.  The quartic / cubic branches (math.h:35-110) cannot be
 * reached by the forward primitives of get_succ: their leading coefficient
 * c[0] is structurally zero (primitive.h:34-50), so a == b == 0 always. */
std::vector<double> solve_roots(double a, double b, double c, double d, double e) {
  std::vector<double> r;
  if (a != 0 || b != 0) {
    fprintf(stderr, "mpl_oracle: cubic/quartic extrema are outside the hot path\n");
    abort();
  } else if (c != 0) {
    return quad_roots(c, d, e);
  } else if (d != 0) {
    r.push_back(-e / d);
    return r;
  }
  return r;
}

/* primitive.h:21-198 -- c[0] is the highest-order coefficient */
struct Poly1 {
  double c[6] = {0, 0, 0, 0, 0, 0};

  /* primitive.h:128-131 */
  double p(double t) const {
    return c[0] / 120 * ipow(t, 5) + c[1] / 24 * ipow(t, 4) + c[2] / 6 * ipow(t, 3) +
           c[3] / 2 * t * t + c[4] * t + c[5];
  }
  /* primitive.h:134-137 */
  double v(double t) const {
    return c[0] / 24 * ipow(t, 4) + c[1] / 6 * ipow(t, 3) + c[2] / 2 * t * t + c[3] * t + c[4];
  }
  /* primitive.h:140-142 */
  double a(double t) const {
    return c[0] / 6 * ipow(t, 3) + c[1] / 2 * t * t + c[2] * t + c[3];
  }
  /* primitive.h:145 */
  double j(double t) const { return c[0] / 2 * t * t + c[1] * t + c[2]; }

  /* primitive.h:92-122 */
  double effort(double t, int control) const {
    if (control == C_VEL || control == C_VELxYAW)
      return c[0] * c[0] / 5184 * ipow(t, 9) + c[0] * c[1] / 576 * ipow(t, 8) +
             (c[1] * c[1] / 252 + c[0] * c[2] / 168) * ipow(t, 7) +
             (c[0] * c[3] / 72 + c[1] * c[2] / 36) * ipow(t, 6) +
             (c[2] * c[2] / 20 + c[0] * c[4] / 60 + c[1] * c[3] / 15) * ipow(t, 5) +
             (c[2] * c[3] / 4 + c[1] * c[4] / 12) * ipow(t, 4) +
             (c[3] * c[3] / 3 + c[2] * c[4] / 3) * ipow(t, 3) + c[3] * c[4] * t * t +
             c[4] * c[4] * t;
    else if (control == C_ACC || control == C_ACCxYAW)
      return c[0] * c[0] / 252 * ipow(t, 7) + c[0] * c[1] / 36 * ipow(t, 6) +
             (c[1] * c[1] / 20 + c[0] * c[2] / 15) * ipow(t, 5) +
             (c[0] * c[3] / 12 + c[1] * c[2] / 4) * ipow(t, 4) +
             (c[2] * c[2] / 3 + c[1] * c[3] / 3) * ipow(t, 3) + c[2] * c[3] * t * t +
             c[3] * c[3] * t;
    else if (control == C_JRK || control == C_JRKxYAW)
      return c[0] * c[0] / 20 * ipow(t, 5) + c[0] * c[1] / 4 * ipow(t, 4) +
             (c[1] * c[1] + c[0] * c[2]) / 3 * ipow(t, 3) + c[1] * c[2] * t * t + c[2] * c[2] * t;
    else if (control == C_SNP || control == C_SNPxYAW)
      return c[0] * c[0] / 3 * ipow(t, 3) + c[0] * c[1] * t * t + c[1] * c[1] * t;
    return 0;
  }

  /* primitive.h:152-162 */
  std::vector<double> extrema_v(double t) const {
    std::vector<double> roots = solve_roots(0, c[0] / 6, c[1] / 2, c[2], c[3]);
    std::vector<double> ts;
    for (double it : roots) {
      if (it > 0 && it < t) ts.push_back(it);
      else if (it >= t) break;
    }
    return ts;
  }
  /* primitive.h:169-179 */
  std::vector<double> extrema_a(double t) const {
    std::vector<double> roots = solve_roots(0, 0, c[0] / 2, c[1], c[2]);
    std::vector<double> ts;
    for (double it : roots) {
      if (it > 0 && it < t) ts.push_back(it);
      else if (it >= t) break;
    }
    return ts;
  }
  /* primitive.h:186-193 */
  std::vector<double> extrema_j(double t) const {
    std::vector<double> ts;
    if (c[0] != 0) {
      double s = -c[1] * 2 / c[0];
      if (s > 0 && s < t) ts.push_back(s);
    }
    return ts;
  }
};

/* waypoint.h:23-57 */
template <int D>
struct State {
  double pos[D], vel[D], acc[D], jrk[D];
  double yaw = 0, t = 0;
  int control = 0;
  bool enable_t = false;
  explicit State(int c = 0) : control(c) {
    for (int i = 0; i < D; i++) pos[i] = vel[i] = acc[i] = jrk[i] = 0;
  }
  bool use_pos() const { return control & 1; }
  bool use_vel() const { return control & 2; }
  bool use_acc() const { return control & 4; }
  bool use_jrk() const { return control & 8; }
  bool use_yaw() const { return control & 16; }
};

/* boost::hash_combine, classic form; hash_value(int) is the sign-extending
 * conversion to size_t. */
inline void fold(uint64_t &seed, int id) {
  seed ^= (uint64_t)(int64_t)id + 0x9e3779b9ULL + (seed << 6) + (seed >> 2);
}

/* waypoint.h:93-125.  `int id = std::round(x / q)` is a value conversion of an
 * in-range double. */
template <int D>
uint64_t lattice_hash(const State<D> &k) {
  uint64_t val = 0;
  for (int i = 0; i < D; i++) {
    if (k.use_pos()) { int id = std::round(k.pos[i] / 0.01); fold(val, id); }
    if (k.use_vel()) { int id = std::round(k.vel[i] / 0.1); fold(val, id); }
    if (k.use_acc()) { int id = std::round(k.acc[i] / 0.1); fold(val, id); }
    if (k.use_jrk()) { int id = std::round(k.jrk[i] / 0.1); fold(val, id); }
  }
  if (k.use_yaw()) { int id = std::round(k.yaw / 0.1); fold(val, id); }
  if (k.enable_t) { int id = std::round(k.t / 0.1); fold(val, id); }
  return val;
}

/* primitive.h:205-431 */
template <int D>
struct Prim {
  double T = 0;
  int control = 0;
  std::array<Poly1, D> ax;
  Poly1 yaw;

  Prim() {}
  /* primitive.h:220-256 with the 1-D constructors of :34-50 */
  Prim(const State<D> &s, const double *u, double t) : T(t), control(s.control) {
    const int base = control & 0x0f;
    for (int i = 0; i < D; i++) {
      double *c = ax[i].c;
      if (base == C_SNP) { c[1] = u[i]; c[2] = s.jrk[i]; c[3] = s.acc[i]; c[4] = s.vel[i]; c[5] = s.pos[i]; }
      else if (base == C_JRK) { c[2] = u[i]; c[3] = s.acc[i]; c[4] = s.vel[i]; c[5] = s.pos[i]; }
      else if (base == C_ACC) { c[3] = u[i]; c[4] = s.vel[i]; c[5] = s.pos[i]; }
      else if (base == C_VEL) { c[4] = u[i]; c[5] = s.pos[i]; }
    }
    if (control & 0x10) { yaw.c[4] = u[D]; yaw.c[5] = s.yaw; }
  }

  /* primitive.h:321-331 */
  State<D> evaluate(double t) const {
    State<D> w(control);
    for (int k = 0; k < D; k++) {
      w.pos[k] = ax[k].p(t);
      w.vel[k] = ax[k].v(t);
      w.acc[k] = ax[k].a(t);
      w.jrk[k] = ax[k].j(t);
      if (w.use_yaw()) w.yaw = wrap_angle(yaw.p(t));
    }
    return w;
  }

  /* primitive.h:353-363 */
  double max_vel(int k) const {
    std::vector<double> ts = ax[k].extrema_v(T);
    double m = std::max(std::abs(ax[k].v(0)), std::abs(ax[k].v(T)));
    for (double it : ts)
      if (it > 0 && it < T) { double v = std::abs(ax[k].v(it)); m = v > m ? v : m; }
    return m;
  }
  /* primitive.h:369-379 */
  double max_acc(int k) const {
    std::vector<double> ts = ax[k].extrema_a(T);
    double m = std::max(std::abs(ax[k].a(0)), std::abs(ax[k].a(T)));
    for (double it : ts)
      if (it > 0 && it < T) { double a = std::abs(ax[k].a(it)); m = a > m ? a : m; }
    return m;
  }
  /* primitive.h:384-394 */
  double max_jrk(int k) const {
    std::vector<double> ts = ax[k].extrema_j(T);
    double m = std::max(std::abs(ax[k].j(0)), std::abs(ax[k].j(T)));
    for (double it : ts)
      if (it > 0 && it < T) { double j = std::abs(ax[k].j(it)); m = j > m ? j : m; }
    return m;
  }
  /* primitive.h:403-407 */
  double effort() const {
    double j = 0;
    for (const auto &a : ax) j += a.effort(T, control);
    return j;
  }
};

enum Which { W_VEL, W_ACC, W_JRK };

/* primitive.h:483-496 */
template <int D>
bool within(const Prim<D> &pr, double lim, Which w) {
  if (lim <= 0) return true;
  for (int i = 0; i < D; i++) {
    if (w == W_VEL && pr.max_vel(i) > lim) return false;
    else if (w == W_ACC && pr.max_acc(i) > lim) return false;
    else if (w == W_JRK && pr.max_jrk(i) > lim) return false;
  }
  return true;
}

/* primitive.h:504-525 */
template <int D>
bool yaw_ok(const Prim<D> &pr, double my) {
  if (my <= 0) return true;
  std::vector<State<D>> ws(2);
  ws[0] = pr.evaluate(0);
  ws[1] = pr.evaluate(pr.T);
  for (const auto &w : ws) {
    const double vx = w.vel[0], vy = w.vel[1];
    if (vx != 0 || vy != 0) {
      /* Eigen normalized(): v / sqrt(squaredNorm) component-wise, then dot */
      const double s = sqrt(vx * vx + vy * vy);
      double d = vx / s * cos(w.yaw) + vy / s * sin(w.yaw);
      if (d < cos(my)) return false;
    }
  }
  return true;
}

/* primitive.h:450-475 */
template <int D>
bool valid_dynamics(const Prim<D> &pr, double mv, double ma, double mj, double myaw) {
  switch (pr.control) {
    case C_ACC: return within(pr, mv, W_VEL);
    case C_JRK: return within(pr, mv, W_VEL) && within(pr, ma, W_ACC);
    case C_SNP: return within(pr, mv, W_VEL) && within(pr, ma, W_ACC) && within(pr, mj, W_JRK);
    case C_VELxYAW: return yaw_ok(pr, myaw);
    case C_ACCxYAW: return yaw_ok(pr, myaw) && within(pr, mv, W_VEL);
    case C_JRKxYAW: return yaw_ok(pr, myaw) && within(pr, mv, W_VEL) && within(pr, ma, W_ACC);
    case C_SNPxYAW:
      return yaw_ok(pr, myaw) && within(pr, mv, W_VEL) && within(pr, ma, W_ACC) &&
             within(pr, mj, W_JRK);
    default: return true; /* incl. plain VEL: no dynamic limits are checked */
  }
}

/* map_util.h:34-69,103-108 + the env parameters of env_base.h:368-404 */
template <int D>
struct Env {
  const mpl_oracle_env *e;
  /* debug side effects of env_map.h:154,166 (cleared per node here so memory
   * stays bounded; the reference clears them per plan()) */
  mutable std::vector<std::array<double, D>> expanded_nodes;
  mutable std::vector<Prim<D>> expanded_edges;

  void to_cell(const double *pt, int *pn) const {      /* map_util.h:103-108 */
    for (int i = 0; i < D; i++) pn[i] = std::round((pt[i] - e->origin[i]) / e->res - 0.5);
  }
  int index(const int *pn) const {                      /* map_util.h:34-41 */
    if (D == 2) return pn[0] + e->map_dim[0] * pn[1];
    return pn[0] + e->map_dim[0] * pn[1] + e->map_dim[0] * e->map_dim[1] * pn[2];
  }
  bool outside(const int *pn) const {                   /* map_util.h:51-55 */
    for (int i = 0; i < D; i++)
      if (pn[i] < 0 || pn[i] >= e->map_dim[i]) return true;
    return false;
  }
  bool occupied(const int *pn) const {                  /* map_util.h:48,64-69 */
    if (outside(pn)) return false;
    return e->map[index(pn)] == 100;
  }

  /* env_map.h:90-132 */
  double traverse(const Prim<D> &pr, int *iters) const {
    double max_v = 0;
    for (int i = 0; i < D; i++)
      if (pr.max_vel(i) > max_v) max_v = pr.max_vel(i);
    int n = std::max(5, (int)std::ceil(max_v * pr.T / e->res));
    double c = 0;
    double dt = pr.T / n;
    int it = 0;
    for (double t = 0; t < pr.T; t += dt) {
      it++;
      *iters = it;
      const State<D> pt = pr.evaluate(t);
      int pn[3];
      to_cell(pt.pos, pn);
      /* the reference forms idx before the bounds test but reads it only
       * after (env_map.h:102-106); compute it after the test here */
      if (outside(pn)) return kInf;
      const int idx = index(pn);
      if (e->region && !e->region[idx]) return kInf;
      if (e->potential) {
        if (e->potential[idx] < 100 && e->potential[idx] > 0) {
          double vn = 0; /* Eigen norm(): sqrt of the sum of squares */
          for (int k = 0; k < D; k++) vn += pt.vel[k] * pt.vel[k];
          c += dt * (e->potential_weight * e->potential[idx] + e->gradient_weight * sqrt(vn));
        } else if (e->potential[idx] >= 100)
          return kInf;
      } else if (occupied(pn))
        return kInf;
      if (e->wyaw > 0 && pt.use_yaw()) {
        const double vx = pt.vel[0], vy = pt.vel[1];
        const double s = sqrt(vx * vx + vy * vy);
        if (s > 1e-5) {
          double v_value = 1 - (vx / s * cos(pt.yaw) + vy / s * sin(pt.yaw));
          c += e->wyaw * v_value * dt;
        }
      }
    }
    return c;
  }

  /* env_map.h:147-172 */
  void expand_node(const State<D> &curr, std::vector<State<D>> &succ, std::vector<double> &cost,
                   std::vector<int> &act, std::vector<int> *iters_out) const {
    succ.clear();
    cost.clear();
    act.clear();
    if (iters_out) iters_out->clear();
    std::array<double, D> cp;
    for (int i = 0; i < D; i++) cp[i] = curr.pos[i];
    expanded_nodes.push_back(cp);
    for (int i = 0; i < e->nU; i++) {
      Prim<D> pr(curr, e->U + (size_t)i * e->udim, e->dt);
      State<D> tn = pr.evaluate(e->dt);
      if (lattice_hash(tn) == lattice_hash(curr) ||
          !valid_dynamics(pr, e->v_max, e->a_max, e->j_max, e->yaw_max))
        continue;
      tn.t = curr.t + e->dt;
      succ.push_back(tn);
      bool same_pos = true;
      for (int k = 0; k < D; k++) same_pos = same_pos && (curr.pos[k] == tn.pos[k]);
      int iters = 0;
      double c = same_pos ? 0 : traverse(pr, &iters);
      if (!std::isinf(c)) {
        c += pr.effort() + e->w * e->dt; /* env_base.h:343-345 */
        expanded_edges.push_back(pr);
      }
      cost.push_back(c);
      act.push_back(i);
      if (iters_out) iters_out->push_back(iters);
    }
  }
};

template <int D>
State<D> load_state(const double *nodes, int64_t stride, int64_t k, int control) {
  State<D> s(control);
  for (int i = 0; i < D; i++) {
    s.pos[i] = nodes[(0 * D + i) * stride + k];
    s.vel[i] = nodes[(1 * D + i) * stride + k];
    s.acc[i] = nodes[(2 * D + i) * stride + k];
    s.jrk[i] = nodes[(3 * D + i) * stride + k];
  }
  s.yaw = nodes[(4 * D) * stride + k];
  s.t = nodes[(4 * D + 1) * stride + k];
  return s;
}

template <int D>
void store_state(double *out, int64_t stride, int64_t slot, const State<D> &s) {
  for (int i = 0; i < D; i++) {
    out[(0 * D + i) * stride + slot] = s.pos[i];
    out[(1 * D + i) * stride + slot] = s.vel[i];
    out[(2 * D + i) * stride + slot] = s.acc[i];
    out[(3 * D + i) * stride + slot] = s.jrk[i];
  }
  out[(4 * D) * stride + slot] = s.yaw;
  out[(4 * D + 1) * stride + slot] = s.t;
}

void add_stats(mpl_oracle_stats &a, const mpl_oracle_stats &b) {
  a.pairs += b.pairs; a.emitted += b.emitted; a.finite += b.finite;
  a.skip_same += b.skip_same; a.skip_dyn += b.skip_dyn; a.samples += b.samples;
  a.sum_finite_cost += b.sum_finite_cost;
}

/* Dense expansion of nodes [lo, hi). Slots that the reference does not emit
 * are still described (state, hash, reason) so a dense device result can be
 * compared slot-for-slot. */
template <int D>
void expand_range(const mpl_oracle_env *e, const double *nodes, int64_t n_nodes, int64_t lo,
                  int64_t hi, mpl_oracle_out *out, mpl_oracle_stats *st) {
  Env<D> env{e, {}, {}};
  const int64_t n_slots = n_nodes * e->nU;
  std::vector<State<D>> succ;
  std::vector<double> cost;
  std::vector<int> act, iters;
  mpl_oracle_stats s = {0, 0, 0, 0, 0, 0, 0.0};
  for (int64_t k = lo; k < hi; k++) {
    const State<D> curr = load_state<D>(nodes, n_nodes, k, e->control);
    env.expanded_nodes.clear();
    env.expanded_edges.clear();
    env.expand_node(curr, succ, cost, act, &iters);
    /* default-fill every slot of this node as "not emitted" */
    const uint64_t hcur = lattice_hash(curr);
    size_t m = 0;
    for (int i = 0; i < e->nU; i++) {
      const int64_t slot = k * e->nU + i;
      s.pairs++;
      if (m < act.size() && act[m] == i) {
        const bool fin = !std::isinf(cost[m]);
        if (out) {
          if (out->status) out->status[slot] = fin ? MPL_SLOT_FINITE : MPL_SLOT_BLOCKED;
          if (out->cost) out->cost[slot] = cost[m];
          if (out->hash) out->hash[slot] = lattice_hash(succ[m]);
          if (out->state) store_state<D>(out->state, n_slots, slot, succ[m]);
          if (out->iters) out->iters[slot] = iters[m];
        }
        s.emitted++;
        s.samples += iters[m];
        if (fin) { s.finite++; s.sum_finite_cost += cost[m]; }
        m++;
      } else {
        /* recompute why the reference skipped it (env_map.h:158) */
        Prim<D> pr(curr, e->U + (size_t)i * e->udim, e->dt);
        State<D> tn = pr.evaluate(e->dt);
        const uint64_t h = lattice_hash(tn);
        const bool same = (h == hcur);
        if (same) s.skip_same++; else s.skip_dyn++;
        tn.t = curr.t + e->dt;
        if (out) {
          if (out->status) out->status[slot] = same ? MPL_SLOT_SKIP_SAME : MPL_SLOT_SKIP_DYN;
          if (out->cost) out->cost[slot] = kInf;
          if (out->hash) out->hash[slot] = h;
          if (out->state) store_state<D>(out->state, n_slots, slot, tn);
          if (out->iters) out->iters[slot] = 0;
        }
      }
    }
  }
  *st = s;
}

template <int D>
void time_range(const mpl_oracle_env *e, const double *nodes, int64_t n_nodes, int64_t lo,
                int64_t hi, mpl_oracle_stats *st) {
  Env<D> env{e, {}, {}};
  std::vector<State<D>> succ;
  std::vector<double> cost;
  std::vector<int> act, iters;
  mpl_oracle_stats s = {0, 0, 0, 0, 0, 0, 0.0};
  for (int64_t k = lo; k < hi; k++) {
    const State<D> curr = load_state<D>(nodes, n_nodes, k, e->control);
    env.expanded_nodes.clear();
    env.expanded_edges.clear();
    env.expand_node(curr, succ, cost, act, &iters);
    s.pairs += e->nU;
    s.emitted += (int64_t)succ.size();
    for (size_t m = 0; m < cost.size(); m++) {
      s.samples += iters[m];
      if (!std::isinf(cost[m])) { s.finite++; s.sum_finite_cost += cost[m]; }
    }
  }
  *st = s;
}

template <typename F>
void run_threads(int64_t n, int threads, F f) {
  if (threads < 1) threads = 1;
  if ((int64_t)threads > n) threads = (int)std::max<int64_t>(1, n);
  if (threads == 1) { f(0, (int64_t)0, n); return; }
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; t++) {
    int64_t lo = n * t / threads, hi = n * (t + 1) / threads;
    pool.emplace_back(f, t, lo, hi);
  }
  for (auto &th : pool) th.join();
}

bool env_ok(const mpl_oracle_env *e) {
  if (!e || (e->dim != 2 && e->dim != 3) || !e->map || !e->U || e->nU <= 0) return false;
  const int need = e->dim + ((e->control & 0x10) ? 1 : 0);
  return e->udim >= need;
}

}  // namespace

extern "C" {

int mpl_oracle_expand(const mpl_oracle_env *env, const double *nodes, int64_t n_nodes,
                      mpl_oracle_out *out, int threads, mpl_oracle_stats *stats) {
  if (!env_ok(env) || (!nodes && n_nodes > 0) || n_nodes < 0) return -1;
  if (threads < 1) threads = 1;
  std::vector<mpl_oracle_stats> per((size_t)threads);
  for (auto &p : per) p = {0, 0, 0, 0, 0, 0, 0.0};
  run_threads(n_nodes, threads, [&](int t, int64_t lo, int64_t hi) {
    if (env->dim == 2) expand_range<2>(env, nodes, n_nodes, lo, hi, out, &per[t]);
    else expand_range<3>(env, nodes, n_nodes, lo, hi, out, &per[t]);
  });
  if (stats) {
    *stats = {0, 0, 0, 0, 0, 0, 0.0};
    for (auto &p : per) add_stats(*stats, p);
  }
  return 0;
}

double mpl_oracle_time_expand(const mpl_oracle_env *env, const double *nodes, int64_t n_nodes,
                              int threads, int reps, mpl_oracle_stats *stats) {
  if (!env_ok(env) || !nodes || n_nodes <= 0) return -1.0;
  if (threads < 1) threads = 1;
  if (reps < 1) reps = 1;
  double best = 1e300;
  std::vector<mpl_oracle_stats> per((size_t)threads);
  for (int r = 0; r < reps; r++) {
    for (auto &p : per) p = {0, 0, 0, 0, 0, 0, 0.0};
    auto t0 = std::chrono::steady_clock::now();
    run_threads(n_nodes, threads, [&](int t, int64_t lo, int64_t hi) {
      if (env->dim == 2) time_range<2>(env, nodes, n_nodes, lo, hi, &per[t]);
      else time_range<3>(env, nodes, n_nodes, lo, hi, &per[t]);
    });
    auto t1 = std::chrono::steady_clock::now();
    best = std::min(best, std::chrono::duration<double>(t1 - t0).count());
  }
  if (stats) {
    *stats = {0, 0, 0, 0, 0, 0, 0.0};
    for (auto &p : per) add_stats(*stats, p);
  }
  return best;
}

}  // extern "C"

template <int D>
static int get_succ_lists(const mpl_oracle_env *e, const double *node, double *succ, double *cost,
                          int32_t *action, int32_t *n_succ) {
  Env<D> env{e, {}, {}};
  std::vector<State<D>> s;
  std::vector<double> c;
  std::vector<int> a;
  env.expand_node(load_state<D>(node, 1, 0, e->control), s, c, a, nullptr);
  const int F = 4 * D + 2;
  for (size_t m = 0; m < s.size(); m++) {
    double *o = succ + m * F;
    for (int i = 0; i < D; i++) {
      o[0 * D + i] = s[m].pos[i]; o[1 * D + i] = s[m].vel[i];
      o[2 * D + i] = s[m].acc[i]; o[3 * D + i] = s[m].jrk[i];
    }
    o[4 * D] = s[m].yaw;
    o[4 * D + 1] = s[m].t;
    cost[m] = c[m];
    action[m] = a[m];
  }
  *n_succ = (int32_t)s.size();
  return 0;
}

extern "C" {

int mpl_oracle_get_succ(void *user, const double *node, double *succ, double *cost, int32_t *action,
                        int32_t *n_succ) {
  const mpl_oracle_env *e = (const mpl_oracle_env *)user;
  if (!env_ok(e) || !node || !succ || !cost || !action || !n_succ) return -1;
  return e->dim == 2 ? get_succ_lists<2>(e, node, succ, cost, action, n_succ)
                     : get_succ_lists<3>(e, node, succ, cost, action, n_succ);
}

int mpl_oracle_batch(void *user, const double *nodes, int64_t n, uint8_t *status, double *cost,
                     double *state) {
  mpl_oracle_out o = {status, cost, nullptr, state, nullptr};
  return mpl_oracle_expand((const mpl_oracle_env *)user, nodes, n, &o, 1, nullptr);
}

uint64_t mpl_oracle_hash(int32_t dim, int32_t control, const double *wp) {
  if (dim == 2) return lattice_hash(load_state<2>(wp, 1, 0, control));
  return lattice_hash(load_state<3>(wp, 1, 0, control));
}

/* env_base.h:46-64, heur_ignore_dynamics_ branch, no prior trajectory */
double mpl_oracle_heur(int32_t dim, int32_t control, double w, double v_max, const double *wp,
                       const double *goal) {
  if (mpl_oracle_hash(dim, control, wp) == mpl_oracle_hash(dim, control, goal)) return 0;
  double m = 0; /* lpNorm<Infinity> */
  for (int i = 0; i < dim; i++) m = std::max(m, std::abs(wp[i] - goal[i]));
  if (v_max > 0) return w * m / v_max;
  return w * m;
}

/* env_map<Dim>::is_goal, the norm tests (env_map.h:25-37); the ray trace of
 * :38-43 is not part of this (it cannot fail on a free map). */
int32_t mpl_oracle_goal_tol(int32_t dim, const double *wp, const double *goal, double tol_pos, double tol_vel,
                            double tol_acc, double tol_yaw) {
  auto linf = [&](int row) {
    double m = 0;
    for (int i = 0; i < dim; i++) m = std::max(m, std::abs(wp[row * dim + i] - goal[row * dim + i]));
    return m;
  };
  bool goaled = linf(0) <= tol_pos;
  if (goaled && tol_vel >= 0) goaled = linf(1) <= tol_vel;
  if (goaled && tol_acc >= 0) goaled = linf(2) <= tol_acc;
  if (goaled && tol_yaw >= 0) goaled = std::abs(wp[4 * dim] - goal[4 * dim]) <= tol_yaw;
  return goaled ? 1 : 0;
}

int32_t mpl_oracle_loop_count(double T, int32_t n) {
  double dt = T / n;
  int32_t it = 0;
  for (double t = 0; t < T; t += dt) it++;
  return it;
}

}  // extern "C"

/* ---- batched edge re-validation ------------------------- */
namespace {
template <int D>
void check_edges(const mpl_oracle_env *e, const double *parents, const int32_t *actions, int64_t n, uint8_t *free_out,
                 double *cost_out, int32_t *cells, int32_t *cell_count, int32_t cell_cap) {
  Env<D> env{e};
  for (int64_t k = 0; k < n; k++) {
    const State<D> par = load_state<D>(parents, n, k, e->control);
    /* env_base::forward_action, env_base.h:228-231 */
    Prim<D> pr(par, e->U + (size_t)actions[k] * e->udim, e->dt);
    /* env_map::is_free(Primitive), env_map.h:60-76, and MapPlanner::getLinkedNodes, map_planner.cpp:136-153 */
    double max_v = 0;
    for (int i = 0; i < D; i++)
      if (pr.max_vel(i) > max_v) max_v = pr.max_vel(i);
    const int N = std::ceil(max_v * pr.T / e->res);
    bool is_free = N > 0; /* N == 0: the reference samples t = NaN (undefined cell); reported not free */
    int n_cells = 0, prev_id = -1;
    if (N > 0) {
      const double dt = pr.T / N; /* Primitive::sample, primitive.h:415-420 */
      for (int i = 0; i <= N; i++) {
        const State<D> pt = pr.evaluate(i * dt);
        int pn[3];
        env.to_cell(pt.pos, pn);
        const int id = env.index(pn);
        if (cells && id != prev_id) {
          if (n_cells < cell_cap) cells[k * cell_cap + n_cells] = id;
          n_cells++;
          prev_id = id;
        }
        if (is_free) {
          if (env.occupied(pn) || env.outside(pn)) is_free = false;
          else if (e->region && !e->region[id]) is_free = false;
        }
      }
    }
    if (free_out) free_out[k] = is_free ? 1 : 0;
    if (cost_out) cost_out[k] = is_free ? pr.effort() + e->w * e->dt : kInf; /* env_base.h:343-345 */
    if (cell_count) cell_count[k] = n_cells;
  }
}
}  // namespace

extern "C" int mpl_oracle_check_edges(const mpl_oracle_env *env, const double *parents, const int32_t *actions,
                                      int64_t n, uint8_t *free_out, double *cost_out, int32_t *cells,
                                      int32_t *cell_count, int32_t cell_cap) {
  if (!env || (n > 0 && (!parents || !actions))) return -1;
  if (env->dim == 2) check_edges<2>(env, parents, actions, n, free_out, cost_out, cells, cell_count, cell_cap);
  else if (env->dim == 3) check_edges<3>(env, parents, actions, n, free_out, cost_out, cells, cell_count, cell_cap);
  else return -1;
  return 0;
}

/* ===================================================================== *
 *  Map preprocessing: CPU restatements, same structure  *
 *  as the reference's loops.  TEST INFRASTRUCTURE, like everything here. *
 * ===================================================================== */
namespace {

/* MapUtil::floatToInt, include/mpl_collision/map_util.h:103-108 */
inline int prep_float_to_int(double pt, double origin, double res) { return (int)std::round((pt - origin) / res - 0.5); }

struct PrepGrid {
  int dim;
  int d[3];
  double origin[3];
  double res;
  bool outside(const int *pn) const { /* map_util.h:43-57 */
    for (int i = 0; i < dim; i++)
      if (pn[i] < 0 || pn[i] >= d[i]) return true;
    return false;
  }
  int64_t index(const int *pn) const { /* map_util.h:34-41 */
    return dim == 2 ? pn[0] + (int64_t)d[0] * pn[1] : pn[0] + (int64_t)d[0] * pn[1] + (int64_t)d[0] * d[1] * pn[2];
  }
};

}  // namespace

extern "C" {

/* MapPlanner<Dim>::createMask + updatePotentialMap, src/mpl_planner/map_planner.cpp:246-283, 286-391.
 * map_in / map_out: int8 cells, x fastest.  radius, range, pos: `dim` doubles (range all zero = global).
 * H_MAX = 100 (map_planner.h:104).                                                                     */
int mpl_oracle_update_potential_map(int32_t dim, const int8_t *map_in, const int32_t *map_dim, const double *origin,
                                    double res, const double *pos, const double *radius, const double *range,
                                    double pow_, int8_t *map_out) {
  if (dim != 2 && dim != 3) return -1;
  PrepGrid g;
  g.dim = dim;
  g.res = res;
  int64_t n_cells = 1;
  for (int i = 0; i < 3; i++) {
    g.d[i] = i < dim ? map_dim[i] : 1;
    g.origin[i] = i < dim ? origin[i] : 0;
    n_cells *= g.d[i];
  }
  const int8_t H_MAX = 100;
  /* createMask :246-283 */
  struct Ent { int n[3]; int8_t v; };
  std::vector<Ent> mask;
  const double h_max = H_MAX;
  const int rn = (int)std::ceil(radius[0] / res);
  if (dim == 2) {
    for (int n0 = -rn; n0 <= rn; n0++)
      for (int n1 = -rn; n1 <= rn; n1++) {
        if (std::hypot(n0, n1) > rn) continue;
        const double h = h_max * std::pow((1 - (double)std::hypot(n0, n1) / rn), pow_);
        if (h > 1e-3) mask.push_back(Ent{{n0, n1, 0}, (int8_t)h});
      }
  } else {
    const int hn = (int)std::ceil(radius[2] / res);
    for (int n0 = -rn; n0 <= rn; n0++)
      for (int n1 = -rn; n1 <= rn; n1++)
        for (int n2 = -hn; n2 <= hn; n2++) {
          if (std::hypot(n0, n1) > rn) continue;
          const double h = h_max * std::pow((1 - (double)std::hypot(n0, n1) / rn) * (1 - (double)std::abs(n2) / hn), pow_);
          if (h > 1e-3) mask.push_back(Ent{{n0, n1, n2}, (int8_t)h});
        }
  }
  /* updatePotentialMap :286-391 */
  int c1[3] = {0, 0, 0}, c2[3] = {g.d[0], g.d[1], g.d[2]};
  double rnorm = 0;
  for (int i = 0; i < dim; i++) rnorm += range[i] * range[i];
  if (std::sqrt(rnorm) > 0) {
    for (int i = 0; i < dim; i++) {
      c1[i] = prep_float_to_int(pos[i] - range[i], g.origin[i], res);
      c2[i] = prep_float_to_int(pos[i] + range[i], g.origin[i], res);
      if (c1[i] < 0) c1[i] = 0; else if (c1[i] >= g.d[i]) c1[i] = g.d[i] - 1;
      if (c2[i] < 0) c2[i] = 0; else if (c2[i] >= g.d[i]) c2[i] = g.d[i] - 1;
    }
  }
  std::vector<int8_t> dmap(map_in, map_in + n_cells);
  int n[3] = {0, 0, 0};
  const int z1 = dim == 3 ? c1[2] : 0, z2 = dim == 3 ? c2[2] : 1;
  for (n[0] = c1[0]; n[0] < c2[0]; n[0]++)
    for (n[1] = c1[1]; n[1] < c2[1]; n[1]++)
      for (n[2] = z1; n[2] < z2; n[2]++) {
        const int64_t idx = g.index(n);
        if (map_in[idx] > 0) {
          dmap[idx] = H_MAX;
          for (const Ent &it : mask) {
            const int nn[3] = {n[0] + it.n[0], n[1] + it.n[1], n[2] + it.n[2]};
            if (!g.outside(nn)) {
              const int64_t ni = g.index(nn);
              dmap[ni] = std::max(dmap[ni], it.v);
            }
          }
        }
      }
  std::memcpy(map_out, dmap.data(), (size_t)n_cells);
  return 0;
}

/* MapPlanner<Dim>::setSearchRegion with MapUtil::rayTrace, map_planner.cpp:46-95, map_util.h:117-135.
 * path: [n_points][dim]; region_out: one byte per cell (1 = inside).                                  */
int mpl_oracle_search_region(int32_t dim, const int32_t *map_dim, const double *origin, double res, const double *path,
                             int32_t n_points, int32_t dense, const double *search_radius, uint8_t *region_out) {
  if (dim != 2 && dim != 3) return -1;
  PrepGrid g;
  g.dim = dim;
  g.res = res;
  int64_t n_cells = 1;
  for (int i = 0; i < 3; i++) {
    g.d[i] = i < dim ? map_dim[i] : 1;
    g.origin[i] = i < dim ? origin[i] : 0;
    n_cells *= g.d[i];
  }
  std::vector<std::array<int, 3>> ps;
  auto to_cell = [&](const double *pt) {
    std::array<int, 3> c = {0, 0, 0};
    for (int i = 0; i < dim; i++) c[i] = prep_float_to_int(pt[i], g.origin[i], res);
    return c;
  };
  if (!dense) {
    for (int i = 1; i < n_points; i++) {
      const double *p1 = path + (size_t)(i - 1) * dim, *p2 = path + (size_t)i * dim;
      /* rayTrace :117-135 */
      double diff[3] = {0, 0, 0}, linf = 0;
      for (int k = 0; k < dim; k++) {
        diff[k] = p2[k] - p1[k];
        linf = std::max(linf, std::fabs(diff[k] / res));
      }
      const double kk = 0.8;
      const int max_diff = (int)(linf / kk);
      const double s = 1.0 / max_diff;
      double step[3];
      for (int k = 0; k < dim; k++) step[k] = diff[k] * s;
      std::array<int, 3> prev = {-1, -1, dim == 3 ? -1 : 0};
      for (int m = 1; m < max_diff; m++) {
        double pt[3] = {0, 0, 0};
        for (int k = 0; k < dim; k++) pt[k] = p1[k] + step[k] * m;
        std::array<int, 3> c = to_cell(pt);
        if (g.outside(c.data())) break;
        if (c != prev) ps.push_back(c);
        prev = c;
      }
      ps.push_back(to_cell(p2));
    }
  } else {
    for (int i = 0; i < n_points; i++) ps.push_back(to_cell(path + (size_t)i * dim));
  }
  int rn[3] = {0, 0, 0};
  for (int i = 0; i < dim; i++) rn[i] = (int)std::ceil(search_radius[i] / res);
  std::memset(region_out, 0, (size_t)n_cells);
  for (const auto &it : ps)
    for (int a = -rn[0]; a <= rn[0]; a++)
      for (int b = -rn[1]; b <= rn[1]; b++)
        for (int c = -rn[2]; c <= rn[2]; c++) {
          const int pn[3] = {it[0] + a, it[1] + b, it[2] + c};
          if (g.outside(pn)) continue;
          region_out[g.index(pn)] = 1;
        }
  return 0;
}

}  // extern "C"
