// host_planner.hpp -- the CALLER of the hot path, kept on the host.
//
// A compact restatement of the search side of MPL on plain C++ types (no Eigen,
// no Boost), so that MapPlanner::plan() exists end to end around the device
// get_succ:
//   GraphSearch::Astar / recoverTraj   reference include/mpl_planner/common/graph_search.h:39-182, 369-455
//   State / StateSpace / compare_pair  reference include/mpl_planner/common/state_space.h:16-104
//   PlannerBase::plan                  reference include/mpl_planner/common/planner_base.h:275-325
//   env_map::is_goal / is_free(pt)     reference include/mpl_planner/env/env_map.h:25-51
//   env_base::get_heur (default)       reference include/mpl_planner/common/env_base.h:46-64
//   MapUtil::rayTrace / floatToInt     reference include/mpl_collision/map_util.h:103-134
//   Primitive1D::J (any order)         reference include/mpl_basis/primitive.h:92-122
// Successors come from a provider with the shape of env_base::get_succ
// (env_base.h:358-362) -- in the product that is mplx_get_succ / mplx_expand on
// the GPU; the planner itself never evaluates a primitive against the map.
//
// Batched expansion: get_succ is a pure function of the node,
// so the planner may expand the popped node together with the best not yet
// expanded OPEN nodes in one device launch and serve later pops from that
// cache.  The search order, and therefore the plan, is unchanged.
#ifndef MPLX_HOST_PLANNER_HPP
#define MPLX_HOST_PLANNER_HPP

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <limits>
#include <memory>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <queue>
#include <unordered_map>
#include <vector>

namespace mplx {
namespace host {

constexpr double kInf = std::numeric_limits<double>::infinity();

// ---- lattice hash, identical to the device's (waypoint.h:93-125)
inline void fold(uint64_t &seed, int id) {
  seed ^= (uint64_t)(int64_t)id + 0x9e3779b9ULL + (seed << 6) + (seed >> 2);
}
inline uint64_t lattice_hash(int dim, int control, const double *w) {
  uint64_t h = 0;
  for (int i = 0; i < dim; i++) {
    if (control & 1) fold(h, (int)std::round(w[0 * dim + i] / 0.01));
    if (control & 2) fold(h, (int)std::round(w[1 * dim + i] / 0.1));
    if (control & 4) fold(h, (int)std::round(w[2 * dim + i] / 0.1));
    if (control & 8) fold(h, (int)std::round(w[3 * dim + i] / 0.1));
  }
  if (control & 16) fold(h, (int)std::round(w[4 * dim] / 0.1));
  return h;
}

// ---- occupancy grid on the host (start / goal tests only)
struct Grid {
  int dim = 0;
  int n[3] = {1, 1, 1};
  double origin[3] = {0, 0, 0};
  double res = 0;
  std::vector<int8_t> cells;

  void to_cell(const double *pt, int *pn) const {  // map_util.h:103-108
    for (int i = 0; i < dim; i++) pn[i] = (int)std::round((pt[i] - origin[i]) / res - 0.5);
  }
  bool outside(const int *pn) const {
    for (int i = 0; i < dim; i++)
      if (pn[i] < 0 || pn[i] >= n[i]) return true;
    return false;
  }
  int64_t index(const int *pn) const {
    int64_t idx = pn[0] + (int64_t)n[0] * pn[1];
    if (dim == 3) idx += (int64_t)n[0] * n[1] * pn[2];
    return idx;
  }
  bool is_free(const int *pn) const {  // map_util.h:44,57-62
    if (outside(pn)) return false;
    const int8_t v = cells[(size_t)index(pn)];
    return v < 100 && v >= 0;
  }
  bool is_occupied(const int *pn) const {  // map_util.h:48,64-69
    if (outside(pn)) return false;
    return cells[(size_t)index(pn)] == 100;
  }
  // map_util.h:117-134; returns false as soon as a traced cell is occupied
  bool ray_clear(const double *p1, const double *p2) const {
    double diff[3], m = 0;
    for (int i = 0; i < dim; i++) {
      diff[i] = p2[i] - p1[i];
      m = std::max(m, std::abs(diff[i] / res));
    }
    const double k = 0.8;
    const int max_diff = (int)(m / k);
    const double s = 1.0 / max_diff;
    double step[3];
    for (int i = 0; i < dim; i++) step[i] = diff[i] * s;
    int prev[3] = {-1, -1, -1};
    for (int q = 1; q < max_diff; q++) {
      double pt[3];
      int pn[3];
      for (int i = 0; i < dim; i++) pt[i] = p1[i] + step[i] * q;
      to_cell(pt, pn);
      if (outside(pn)) break;
      bool differs = false;
      for (int i = 0; i < dim; i++) differs = differs || pn[i] != prev[i];
      if (differs && is_occupied(pn)) return false;
      for (int i = 0; i < dim; i++) prev[i] = pn[i];
    }
    return true;
  }
};

// ---- provider with the shape of env_base<Dim>::get_succ
//      node: 4D+2 doubles; succ: [nU][4D+2]; returns 0 on success
typedef int (*succ_fn)(void *user, const double *node, double *succ, double *cost, int32_t *action,
                       int32_t *n_succ);
// batched form: nodes field-major [4D+2][n]; outputs dense slots like mplx_expand
typedef int (*batch_fn)(void *user, const double *nodes, int64_t n, uint8_t *status, double *cost,
                        double *state /*[4D+2][n*nU]*/);

// batched form producing per-node lists (the layout of mplx_succ_lists with node stride nU and
// state row stride n*nU), including the lattice hash of every successor
typedef int (*lists_fn)(void *user, const double *nodes, int64_t n, int32_t *count, int32_t *action, double *cost,
                        uint64_t *hash, double *state /*[4D+2][n*nU]*/);

// batched form handing out the engine's own landing buffer: per-node lists packed back to back (node k owns
// entries [offs[k], offs[k+1]) of every row), valid until the provider's next call -- no copy into caller arrays
struct PackedView {
  int64_t total = 0;
  const int32_t *count = nullptr;   // [n]
  const int64_t *offs = nullptr;    // [n + 1]
  const double *cost = nullptr;     // [total]
  const uint64_t *hash = nullptr;   // [total]
  const int32_t *action = nullptr;  // [total]
  const double *state = nullptr;    // [4D+2][total]
};
typedef int (*packed_fn)(void *user, const double *nodes, int64_t n, PackedView *out);

// What one expansion hands to the relaxation loop: m successors, fields of successor s at state[r * fs + s * es].
struct SuccView {
  int32_t m = 0;
  const double *cost = nullptr;
  const uint64_t *keys = nullptr;  // lattice hashes when the provider supplies them
  const int32_t *act = nullptr;
  const double *state = nullptr;
  int64_t fs = 1, es = 1;
};

struct Node;
typedef Node *NodePtr;  // nodes live in Planner::pool (a deque: stable addresses, one allocation per block)

// state_space.h:37-70 (A* fields)
struct Node {
  // the fields every relaxation touches share the first cache line
  double g = kInf, rhs = kInf, h = kInf;
  // pred_coord / pred_action_cost / pred_action_id of state_space.h:49-53: a list threaded through the
  // planner's one pool of records (Planner::preds), in insertion order -- no allocation per node
  int32_t pred_head = -1, pred_tail = -1;
  int heap_pos = -1;
  bool opened = false, closed = false;
  bool cached = false;  // successor cache (batched expansion)
  uint64_t key = 0;
  double coord[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // 4D+2 used
  std::vector<double> c_succ;
  std::vector<double> c_cost;
  std::vector<int32_t> c_act;
  std::vector<uint64_t> c_key;  // lattice hashes of the cached successors (when the provider supplies them)
  // packed provider: one recycled buffer [cost m][hash m][state (4D+2) x m][action m] (Planner::take_blob)
  char *c_blob = nullptr;
  int32_t c_m = 0;
  // ... or, until the next launch, the node's index in the provider's own landing buffer (Planner::cur_view)
  int32_t c_slot = -1;
  uint32_t c_batch = 0;
  uint32_t pick_stamp = 0;  // launch for which the node was last picked (a re-opened node sits in the heap twice)
  bool c_has_state = true;
};

// hm_ of the reference's StateSpace (state_space.h:78): lattice hash -> node.  Open addressing with linear
// probing, key and pointer side by side (one cache line per look-up at millions of nodes, where a node-based
// std::unordered_map takes three); prefetch() lets the relaxation loop hide that one miss.
class NodeMap {
 public:
  // The slot of `key`, created empty (nullptr) when absent; the caller fills a new slot at once.
  NodePtr &operator[](uint64_t key) {
    if ((n_ + 1) * 10 > cap_ * 6) grow();
    size_t i = slot(key);
    while (slots_[i].val) {
      if (slots_[i].key == key) return slots_[i].val;
      i = (i + 1) & (cap_ - 1);
    }
    slots_[i].key = key;
    n_++;
    return slots_[i].val;
  }
  // The node of `key` or nullptr, without inserting.
  NodePtr peek(uint64_t key) const {
    if (!cap_) return nullptr;
    size_t i = slot(key);
    while (slots_[i].val) {
      if (slots_[i].key == key) return slots_[i].val;
      i = (i + 1) & (cap_ - 1);
    }
    return nullptr;
  }
  void prefetch(uint64_t key) const {
    if (cap_) __builtin_prefetch(&slots_[slot(key)]);
  }
  size_t size() const { return n_; }
  void clear() {
    std::fill(slots_.begin(), slots_.end(), Slot{0, nullptr});
    n_ = 0;
  }

 private:
  struct Slot { uint64_t key; NodePtr val; };
  size_t slot(uint64_t k) const {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    return (size_t)k & (cap_ - 1);
  }
  void grow() {
    std::vector<Slot> old;
    old.swap(slots_);
    cap_ = cap_ ? cap_ * 2 : 1024;
    slots_.assign(cap_, Slot{0, nullptr});
    for (const Slot &o : old)
      if (o.val) {
        size_t i = slot(o.key);
        while (slots_[i].val) i = (i + 1) & (cap_ - 1);
        slots_[i] = o;
      }
  }
  std::vector<Slot> slots_;
  size_t cap_ = 0, n_ = 0;
};

// Mutable binary max-heap on compare_pair (state_space.h:16-34): the top is the
// smallest f, ties go to the smaller min(g, rhs).  Sift rules follow a 2-ary
// boost::heap::d_ary_heap: sift-up stops at equality, sift-down swaps at
// equality and prefers the first maximal child.
class OpenList {
 public:
  struct Item { double f; NodePtr n; };
  bool empty() const { return q_.empty(); }
  size_t size() const { return q_.size(); }
  const Item &top() const { return q_.front(); }
  const std::vector<Item> &items() const { return q_; }
  void push(double f, const NodePtr &n) {
    q_.push_back({f, n});
    n->heap_pos = (int)q_.size() - 1;
    up((int)q_.size() - 1);
  }
  void pop() {
    q_.front().n->heap_pos = -1;
    if (q_.size() > 1) {
      std::swap(q_.front(), q_.back());
      q_.pop_back();
      q_.front().n->heap_pos = 0;
      down(0);
    } else {
      q_.pop_back();
    }
  }
  // key became better (smaller f): state_space `increase` (graph_search.h:133)
  void increase(const NodePtr &n, double f) {
    q_[(size_t)n->heap_pos].f = f;
    up(n->heap_pos);
  }

 private:
  static bool less(const Item &a, const Item &b) {
    if (a.f == b.f) return std::min(a.n->g, a.n->rhs) > std::min(b.n->g, b.n->rhs);
    return a.f > b.f;
  }
  void swap_at(int i, int j) {
    std::swap(q_[(size_t)i], q_[(size_t)j]);
    q_[(size_t)i].n->heap_pos = i;
    q_[(size_t)j].n->heap_pos = j;
  }
  void up(int i) {
    while (i > 0) {
      const int p = (i - 1) / 2;
      if (less(q_[(size_t)p], q_[(size_t)i])) { swap_at(p, i); i = p; } else return;
    }
  }
  void down(int i) {
    const int n = (int)q_.size();
    for (;;) {
      const int l = 2 * i + 1;
      if (l >= n) return;
      int c = l;
      if (l + 1 < n && less(q_[(size_t)l], q_[(size_t)l + 1])) c = l + 1;
      if (!less(q_[(size_t)c], q_[(size_t)i])) { swap_at(c, i); i = c; } else return;
    }
  }
  std::vector<Item> q_;
};

// Primitive1D::J for an arbitrary effort order (primitive.h:92-122), used only
// to report the trajectory's J(VEL..SNP) like the reference's tests do.
inline double ipow(double t, int n) { double r = 1; while (n-- > 0) r *= t; return r; }
inline double effort_1d(const double c[6], double t, int order) {
  if (order == 1)
    return c[0] * c[0] / 5184 * ipow(t, 9) + c[0] * c[1] / 576 * ipow(t, 8) +
           (c[1] * c[1] / 252 + c[0] * c[2] / 168) * ipow(t, 7) + (c[0] * c[3] / 72 + c[1] * c[2] / 36) * ipow(t, 6) +
           (c[2] * c[2] / 20 + c[0] * c[4] / 60 + c[1] * c[3] / 15) * ipow(t, 5) +
           (c[2] * c[3] / 4 + c[1] * c[4] / 12) * ipow(t, 4) + (c[3] * c[3] / 3 + c[2] * c[4] / 3) * ipow(t, 3) +
           c[3] * c[4] * t * t + c[4] * c[4] * t;
  if (order == 2)
    return c[0] * c[0] / 252 * ipow(t, 7) + c[0] * c[1] / 36 * ipow(t, 6) +
           (c[1] * c[1] / 20 + c[0] * c[2] / 15) * ipow(t, 5) + (c[0] * c[3] / 12 + c[1] * c[2] / 4) * ipow(t, 4) +
           (c[2] * c[2] / 3 + c[1] * c[3] / 3) * ipow(t, 3) + c[2] * c[3] * t * t + c[3] * c[3] * t;
  if (order == 3)
    return c[0] * c[0] / 20 * ipow(t, 5) + c[0] * c[1] / 4 * ipow(t, 4) + (c[1] * c[1] + c[0] * c[2]) / 3 * ipow(t, 3) +
           c[1] * c[2] * t * t + c[2] * c[2] * t;
  if (order == 4) return c[0] * c[0] / 3 * ipow(t, 3) + c[0] * c[1] * t * t + c[1] * c[1] * t;
  return 0;
}

// End state of the forward primitive (node, u, T): Primitive<Dim>::evaluate(T) of primitive.h:321-331 with the 1-D
// polynomials of :128-145 for the coefficient vectors of :34-50, written out per control order exactly as the
// kernels evaluate them (csrc/mplx_device_common.h, Ax<K>::pos/vel/acc/jrk<true>) -- IEEE doubles, no contraction
// (the library is built with -ffp-contract=off), so the host value equals the device's bit for bit.  Lets the search
// ask the device for (action, cost, hash) only and build the 112-byte state of the few successors that are new.
inline double wrap_pi(double a) {  // mpl_basis/math.h:15-19
  while (a > M_PI) a -= 2.0 * M_PI;
  while (a < -M_PI) a += 2.0 * M_PI;
  return a;
}
inline void forward_state(int dim, int control, const double *nd, const double *u, double T, double *out) {
  const int K = (control & 8) ? 4 : (control & 4) ? 3 : (control & 2) ? 2 : 1;
  const double t3 = (T * T) * T;
  for (int i = 0; i < dim; i++) {
    const double p = nd[i], v = nd[dim + i], a = nd[2 * dim + i], j = nd[3 * dim + i], ui = u[i];
    double np, nv = 0.0, na = 0.0, nj = 0.0;
    if (K == 1) {
      np = (0.0 + ui * T) + p;
      nv = 0.0 + ui;
    } else if (K == 2) {
      np = ((0.0 + ((ui / 2) * T) * T) + v * T) + p;
      nv = (0.0 + ui * T) + v;
      na = 0.0 + ui;
    } else if (K == 3) {
      np = (((0.0 + (ui / 6) * t3) + ((a / 2) * T) * T) + v * T) + p;
      nv = ((0.0 + ((ui / 2) * T) * T) + a * T) + v;
      na = (0.0 + ui * T) + a;
      nj = 0.0 + ui;
    } else {
      np = ((((0.0 + (ui / 24) * (t3 * T)) + (j / 6) * t3) + ((a / 2) * T) * T) + v * T) + p;
      nv = (((0.0 + (ui / 6) * t3) + ((j / 2) * T) * T) + a * T) + v;
      na = ((0.0 + ((ui / 2) * T) * T) + j * T) + a;
      nj = (0.0 + ui * T) + j;
    }
    out[i] = np;
    out[dim + i] = nv;
    out[2 * dim + i] = na;
    out[3 * dim + i] = nj;
  }
  out[4 * dim] = (control & 16) ? wrap_pi((0.0 + u[dim] * T) + nd[4 * dim]) : 0.0;
  out[4 * dim + 1] = nd[4 * dim + 1] + T;  // env_map.h:161
}

struct PlanResult {
  bool ok = false;
  double cost = kInf;
  int expansions = 0;       // expand_iteration (graph_search.h:64)
  int closed = 0, opened = 0, nodes = 0;
  int device_launches = 0;  // provider calls actually made
  int spec_hits = 0;        // expansions served from lists that rode along in an earlier launch (speculated children)
  int64_t pairs = 0;        // node x control pairs evaluated by the provider
  int64_t state_mismatches = 0;  // check_states: successors whose host-evaluated state differs from the device's
  double total_time = 0;
  double J[4] = {0, 0, 0, 0};  // J(VEL), J(ACC), J(JRK), J(SNP) of the trajectory
  std::vector<double> traj_nodes;  // [segments][4D+2] start state of each primitive
  std::vector<int32_t> traj_actions;
  std::vector<double> traj_end;    // [4D+2] the state the last primitive reaches (the last of Trajectory::getWaypoints)
};

class Planner {
 public:
  int dim = 2;
  int control = 0x03;
  int goal_control = 0;  // control flag of the goal waypoint (0 = the search's): env_base.h:47 compares the goal with a
                         // state by hash, and each side is hashed with its own flags (waypoint.h:93-125)
  double dt = 1.0, w = 10.0, v_max = -1.0, eps = 1.0;
  double tol_pos = 0.5, tol_vel = -1, tol_acc = -1, tol_yaw = -1;
  int max_expand = -1;
  int batch = 1;  // nodes per provider launch (1 = the reference's one-at-a-time loop)
  std::vector<double> U;
  int nU = 0, udim = 0;
  Grid grid;
  succ_fn single = nullptr;
  batch_fn batched = nullptr;
  lists_fn lists = nullptr;  // preferred over `batched` when set: compact lists + device-side hashes
  packed_fn packed = nullptr;  // preferred over `lists`: the same lists without the copy into caller arrays
  bool edges_only = false;     // packed provider delivers no states: new nodes are built with forward_state()
  bool check_states = false;   // test hook (needs states): count host / device state mismatches
  int check_perturb = -1;      // test hook of the test hook: the n-th checked state gets one bit flipped, so the
                               // counter must come out as exactly 1
  void *user = nullptr;

  std::deque<Node> pool;
  struct PredRec { uint64_t key; double cost; int32_t action; int32_t next; };
  std::vector<PredRec> preds;
  NodeMap hm;
  OpenList pq;
  PlanResult last;

  int F() const { return 4 * dim + 2; }

  double heur(const double *s, const double *goal) const {  // env_base.h:46-64
    return heur_keyed(s, lattice_hash(dim, control, s), goal, lattice_hash(dim, goal_control ? goal_control : control, goal));
  }
  double heur_keyed(const double *s, uint64_t s_key, const double *goal, uint64_t goal_key) const {
    if (s_key == goal_key) return 0;
    double m = 0;
    for (int i = 0; i < dim; i++) m = std::max(m, std::abs(s[i] - goal[i]));
    return v_max > 0 ? w * m / v_max : w * m;
  }

  bool is_goal(const double *s, const double *goal) const {  // env_map.h:25-45
    auto linf = [&](int off) {
      double m = 0;
      for (int i = 0; i < dim; i++) m = std::max(m, std::abs(s[off * dim + i] - goal[off * dim + i]));
      return m;
    };
    bool goaled = linf(0) <= tol_pos;
    if (goaled && tol_vel >= 0) goaled = linf(1) <= tol_vel;
    if (goaled && tol_acc >= 0) goaled = linf(2) <= tol_acc;
    if (goaled && tol_yaw >= 0) goaled = std::abs(s[4 * dim] - goal[4 * dim]) <= tol_yaw;
    if (goaled && !grid.ray_clear(s, goal)) return false;
    return goaled;
  }

  // PlannerBase::plan (A*), planner_base.h:275-325 + GraphSearch::Astar
  int plan(const double *start, const double *goal) {
    last = PlanResult();
    t_succ = t_provider = t_fill = t_pick = 0;
    checked_states = 0;
    hm.clear();
    pool.clear();
    if (spec.cap) spec.clear();  // (the map may have changed since the last plan)
    spec_now = spec_cfg();
    spec_keys.clear();  // (states a plan with batch > 1 assembled must not ride along in this plan's launches)
    spec_states.clear();
    preds.clear();
    all_blobs.clear();
    free_blobs.clear();
    cur_blob = nullptr;
    cur_group.clear();
    cur_view = PackedView();
    pq = OpenList();
    if (!single && !batched) return -1;
    int pn[3];
    grid.to_cell(start, pn);
    if (!grid.is_free(pn)) return 0;  // "start is not free": plan() == false
    const int f = F();
    if (is_goal(start, goal)) { last.ok = true; last.cost = 0; return 0; }

    pool.emplace_back();
    NodePtr curr = &pool.back();
    std::copy(start, start + f, curr->coord);
    curr->key = lattice_hash(dim, control, start);
    curr->g = 0;
    curr->h = eps == 0 ? 0 : heur(start, goal);
    curr->opened = true;
    pq.push(curr->g + eps * curr->h, curr);
    hm[curr->key] = curr;

    v_succ.resize((size_t)nU * f);
    v_cost.resize((size_t)nU);
    v_act.resize((size_t)nU);
    v_keys.resize((size_t)nU);
    const uint64_t goal_key = lattice_hash(dim, goal_control ? goal_control : control, goal);
    int expand_iteration = 0;
    bool reached = false;
    double sc[14];
    for (;;) {
      expand_iteration++;
      curr = pq.top().n;
      pq.pop();
      curr->closed = true;
      const auto t_s0 = std::chrono::steady_clock::now();
      SuccView sv;
      if (int rc = successors(curr, &sv)) return rc;
      t_succ += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_s0).count();
      const int n_succ = sv.m;
      const bool have_keys = sv.keys != nullptr;
      // The relaxation is bound by cache misses of the node map and of the nodes (1 M look-ups on the 3D
      // problems): with the device's hashes at hand the slots are prefetched kAhead successors ahead and the
      // nodes half that far.
      constexpr int kAhead = 16;
      if (have_keys)
        for (int s = 0; s < n_succ && s < kAhead; s++) hm.prefetch(sv.keys[s]);
      for (int s = 0; s < n_succ; s++) {
        if (have_keys) {
          if (s + kAhead < n_succ) hm.prefetch(sv.keys[s + kAhead]);
          if (s + kAhead / 2 < n_succ && !std::isinf(sv.cost[s + kAhead / 2])) {
            const int q = s + kAhead / 2;
            if (const Node *nx = hm.peek(sv.keys[q])) {
              __builtin_prefetch(nx);
            } else if (sv.state && sv.fs > 64) {
              // a state not seen before: its fields will be gathered from rows far apart (lists read in place)
              for (int r = 0; r < f; r++) __builtin_prefetch(&sv.state[(int64_t)r * sv.fs + (int64_t)q * sv.es]);
            }
          }
        }
        const double c_s = sv.cost[s];
        if (std::isinf(c_s)) continue;  // graph_search.h:81
        uint64_t key;
        bool have_sc = false;
        auto gather = [&] {
          if (!have_sc) {
            if (sv.state) {
              for (int r = 0; r < f; r++) sc[r] = sv.state[(int64_t)r * sv.fs + (int64_t)s * sv.es];
              if (check_states) {  // test hook: the host evaluation against the device's states
                double hs[14];
                forward_state(dim, control, curr->coord, &U[(size_t)sv.act[s] * udim], dt, hs);
                if (check_perturb >= 0 && checked_states++ == check_perturb) {
                  uint64_t b;
                  std::memcpy(&b, &hs[0], 8);
                  b ^= 1ull;
                  std::memcpy(&hs[0], &b, 8);
                }
                if (std::memcmp(hs, sc, sizeof(double) * (size_t)f) != 0) last.state_mismatches++;
              }
            } else {
              forward_state(dim, control, curr->coord, &U[(size_t)sv.act[s] * udim], dt, sc);
            }
          }
          have_sc = true;
        };
        if (have_keys) key = sv.keys[s];
        else { gather(); key = lattice_hash(dim, control, sc); }
        NodePtr &child = hm[key];
        if (!child) {
          gather();
          pool.emplace_back();
          child = &pool.back();
          std::copy(sc, sc + f, child->coord);
          child->key = key;
          child->h = eps == 0 ? 0 : heur_keyed(sc, key, goal, goal_key);
        }
        {
          const int32_t rec = (int32_t)preds.size();
          preds.push_back({curr->key, c_s, sv.act[s], -1});
          if (child->pred_tail >= 0) preds[(size_t)child->pred_tail].next = rec;
          else child->pred_head = rec;
          child->pred_tail = rec;
        }
        const double tentative = curr->g + c_s;
        if (tentative < child->g) {
          child->g = tentative;
          const double fval = child->g + eps * child->h;
          if (child->opened && !child->closed) {
            pq.increase(child, fval);
          } else {
            pq.push(fval, child);
            child->opened = true;
          }
        }
      }
      if (is_goal(curr->coord, goal)) { reached = true; break; }
      if (max_expand > 0 && expand_iteration >= max_expand) break;
      if (pq.empty()) break;
    }
    if (getenv("MPLX_PLAN_TIMING"))
      fprintf(stderr, "[host_planner] successors() %.1f ms (provider %.1f ms, cache fill %.1f ms, candidate pick %.1f ms)\n",
              t_succ, t_provider, t_fill, t_pick);
    last.expansions = expand_iteration;
    last.nodes = (int)hm.size();
    for (const Node &nd : pool)
      if (nd.closed) last.closed++;
    // PlannerBase::getOpenSet walks the heap (planner_base.h:77-81): a closed node that was pushed again counts
    last.opened = (int)pq.size();
    if (!reached) return 0;
    if (recover(curr, start)) { last.ok = true; last.cost = curr->g; }
    return 0;
  }

 private:
  double t_succ = 0, t_provider = 0, t_fill = 0, t_pick = 0;  // MPLX_PLAN_TIMING diagnostics
  int64_t checked_states = 0;
  std::vector<int32_t> b_cnt, b_act;  // staging of one batched launch (lists provider)
  std::vector<double> b_cost, b_state;
  std::vector<uint64_t> b_hash;
  std::vector<double> v_succ, v_cost;  // the current expansion's successors (providers that fill caller arrays)
  std::vector<int32_t> v_act;
  std::vector<uint64_t> v_keys;
  char *cur_blob = nullptr;            // ... or the packed lists of the node being expanded
  // The provider's landing buffer of the latest launch stays valid until the next one: nodes expanded before
  // that (9 in 10) are read in place; the others are moved to buffers of their own just before the next launch.
  PackedView cur_view;
  uint32_t cur_batch = 0, pick_counter = 0;
  std::vector<NodePtr> cur_group;
  std::vector<int> aux_buf;  // storage of the candidate walk's position heap, recycled between launches

  // ---- speculation on states that are not nodes yet.  The open list can only offer nodes that exist; in a goal-directed
  // descent the next node to be popped is usually a CHILD of one that is being expanded in this very launch, created
  // only after the launch returns -- a miss and another launch (C1: 75 launches for 615 expansions whatever the batch
  // size).  The children's states are a pure function of parent and control (forward_state, bit-identical to the
  // device's), so they ride along in the parent's launch; their lists wait here under their lattice hash, together with
  // the exact state they were computed from (a node of that hash may have been created from another parent with another
  // state: only a bitwise equal state is served).  get_succ is pure: the search cannot tell where a list came from.
  // Measured on C1 (profiles/micro/plan_spec_sweep.sh): children of the first 1 / 2 / 4 / 8 nodes of a launch: 59 / 55 /
  // 49 / 46 launches and 1.11 / 1.14 / 1.09 / 1.21 ms against 75 launches and 1.25 - 1.33 ms without; a second level
  // (children of the popped node's most promising children) removed no launch at all -- every remaining miss is a node
  // created since the last launch, a child of a node served from the cache, not a descent two levels deep.
  int spec_parents = -1;  // nodes of a launch whose children ride along (0: none; -1: automatic -- env MPLX_PLAN_SPEC)
  struct SpecStore {
    int nU = 0, F = 0;
    size_t cap = 0, count = 0, mask = 0;
    bool has_keys = false;
    std::vector<uint64_t> tab_key;
    std::vector<int32_t> tab_val;  // entry + 1, 0 = empty
    std::vector<double> coord, cost;
    std::vector<uint64_t> keys;
    std::vector<int32_t> act, m;
    void reset(int nU_, int F_, size_t cap_) {
      nU = nU_; F = F_; cap = cap_;
      size_t t = 1;
      while (t < 4 * cap) t <<= 1;
      mask = t - 1;
      tab_key.assign(t, 0); tab_val.assign(t, 0);
      coord.resize(cap * (size_t)F); cost.resize(cap * (size_t)nU); keys.resize(cap * (size_t)nU);
      act.resize(cap * (size_t)nU); m.assign(cap, 0);
      count = 0;
    }
    void clear() { std::fill(tab_val.begin(), tab_val.end(), 0); count = 0; }
    static size_t mixk(uint64_t k) { k ^= k >> 31; k *= 0x9e3779b97f4a7c15ull; k ^= k >> 29; return (size_t)k; }
    int find(uint64_t key) const {
      if (!cap) return -1;
      for (size_t i = mixk(key) & mask;; i = (i + 1) & mask) {
        if (!tab_val[i]) return -1;
        if (tab_key[i] == key) return tab_val[i] - 1;
      }
    }
    int insert(uint64_t key) {  // a new entry (the caller has checked that the key is absent)
      if (count == cap) clear();  // full: forget everything (entries are only ever hints)
      size_t i = mixk(key) & mask;
      while (tab_val[i]) i = (i + 1) & mask;
      tab_key[i] = key;
      tab_val[i] = (int32_t)++count;
      return (int)count - 1;
    }
  } spec;
  std::vector<double> nodes_buf;     // the launch's node rows (recycled)
  std::vector<double> spec_states;   // [n][F] of the launch being assembled
  std::vector<uint64_t> spec_keys;   // their lattice hashes
  int spec_now = 0;  // spec_cfg() of the plan() under way
  int spec_cfg() {
    if (spec_parents >= 0) return spec_parents;
    if (const char *e = getenv("MPLX_PLAN_SPEC")) return atoi(e);
    return nU <= 32 ? 4 : 0;  // (a child per control per parent: small control tables only)
  }
  void keep(Node &nd) {  // lists of `nd` out of the landing buffer into a recycled buffer
    const int f = F();
    const size_t m = (size_t)cur_view.count[nd.c_slot], o = (size_t)cur_view.offs[nd.c_slot];
    nd.c_m = (int32_t)m;
    if (!nd.c_blob) nd.c_blob = take_blob();
    char *b = nd.c_blob;
    std::memcpy(b, cur_view.cost + o, m * 8);
    std::memcpy(b + m * 8, cur_view.hash + o, m * 8);
    if (cur_view.state)
      for (int r = 0; r < f; r++) std::memcpy(b + m * 8 * (size_t)(2 + r), cur_view.state + (size_t)r * cur_view.total + o, m * 8);
    std::memcpy(b + m * 8 * (size_t)(2 + f), cur_view.action + o, m * 4);
    nd.c_has_state = cur_view.state != nullptr;
    nd.c_slot = -1;
  }
  // Buffers of the packed lists: fixed capacity (a full list), recycled when their node has been expanded, so
  // that steady state touches no fresh pages (a fresh 40-KB allocation per node cost more than the copy itself).
  std::vector<std::unique_ptr<char[]>> all_blobs;
  std::vector<char *> free_blobs;
  char *take_blob() {
    if (!free_blobs.empty()) { char *b = free_blobs.back(); free_blobs.pop_back(); return b; }
    all_blobs.emplace_back(new char[(size_t)nU * (size_t)(8 + 8 + 8 * F() + 4)]);
    return all_blobs.back().get();
  }
  // One get_succ, possibly served from / filling the batch cache.
  int successors(const NodePtr &curr, SuccView *v) {
    const int f = F();
    if (batch <= 1 || (!batched && !lists && !packed)) {
      last.device_launches++;
      last.pairs += nU;
      int32_t m = 0;
      if (single) {
        if (int rc = single(user, curr->coord, v_succ.data(), v_cost.data(), v_act.data(), &m)) return rc;
        *v = SuccView{m, v_cost.data(), nullptr, v_act.data(), v_succ.data(), 1, f};
        return 0;
      }
      if (int rc = run_batch({curr})) return rc;
      return fetch(curr, v);
    }
    const int n_spec_parents = spec_now;
    if (!curr->cached && n_spec_parents > 0 && spec.cap) {
      const int e = spec.find(curr->key);
      if (e >= 0 && std::memcmp(&spec.coord[(size_t)e * f], curr->coord, sizeof(double) * (size_t)f) == 0) {
        const int32_t m = spec.m[(size_t)e];
        std::copy(&spec.cost[(size_t)e * nU], &spec.cost[(size_t)e * nU] + m, v_cost.begin());
        std::copy(&spec.act[(size_t)e * nU], &spec.act[(size_t)e * nU] + m, v_act.begin());
        if (spec.has_keys) std::copy(&spec.keys[(size_t)e * nU], &spec.keys[(size_t)e * nU] + m, v_keys.begin());
        *v = SuccView{m, v_cost.data(), spec.has_keys ? v_keys.data() : nullptr, v_act.data(), nullptr, 1, f};
        last.spec_hits++;
        return 0;
      }
    }
    if (!curr->cached) {
      const auto t_p0 = std::chrono::steady_clock::now();
      // the popped node plus the best open nodes that have no list yet: best-first walk of the heap
      // array (k smallest of a binary heap with an auxiliary heap of positions), O(k log k) whatever
      // the size of the open list
      std::vector<NodePtr> group{curr};
      curr->pick_stamp = ++pick_counter;
      const size_t want = (size_t)batch - 1;
      const std::vector<OpenList::Item> &h = pq.items();
      auto worse = [&](int a, int b) {  // max-heap on "better", so top() is the best position
        const OpenList::Item &x = h[(size_t)a], &y = h[(size_t)b];
        if (x.f != y.f) return x.f > y.f;
        return std::min(x.n->g, x.n->rhs) > std::min(y.n->g, y.n->rhs);
      };
      // (the walk stops after ~2 x batch heap positions: what lies deeper is not popped soon enough to be worth a
      // slot, and on a small problem most of the top of the heap already holds its lists -- walking 16 x batch
      // positions per launch, the first version, was 3.0 of C1's 5.3 ms; the position heap lives in a member vector)
      std::vector<int> &aux = aux_buf;
      aux.clear();
      if (!h.empty()) aux.push_back(0);
      auto push = [&](int pos) { aux.push_back(pos); std::push_heap(aux.begin(), aux.end(), worse); };
      size_t visited = 0;
      while (!aux.empty() && group.size() - 1 < want && visited < 2 * want + 16) {
        std::pop_heap(aux.begin(), aux.end(), worse);
        const int i = aux.back();
        aux.pop_back();
        visited++;
        Node *cand = h[(size_t)i].n;
        if (!cand->cached && cand->pick_stamp != pick_counter) {
          cand->pick_stamp = pick_counter;
          group.push_back(cand);
        }
        if (2 * i + 1 < (int)h.size()) push(2 * i + 1);
        if (2 * i + 2 < (int)h.size()) push(2 * i + 2);
      }
      // children of the first nodes of the launch (the popped one, then the best of the open list)
      spec_states.clear();
      spec_keys.clear();
      if (n_spec_parents > 0) {
        if (spec.cap == 0 || spec.nU != nU || spec.F != f) spec.reset(nU, f, 2048);
        double cs[14];
        // children of `par`: into the launch -- as nodes when they exist and wait for lists, as bare states otherwise
        auto children = [&](const double *par) {
          for (int i = 0; i < nU; i++) {
            forward_state(dim, control, par, &U[(size_t)i * udim], dt, cs);
            const uint64_t key = lattice_hash(dim, control, cs);
            if (Node *ex = hm.peek(key)) {
              // a node already: closed or served -> nothing to do; open without lists -> an ordinary member of the launch
              if (!ex->closed && !ex->cached && ex->pick_stamp != pick_counter && group.size() < (size_t)batch + 64) {
                ex->pick_stamp = pick_counter;
                group.push_back(ex);
              }
              continue;
            }
            if (spec.find(key) >= 0) continue;  // waiting already (or another state of that hash is: first come, first kept)
            bool dup = false;
            for (uint64_t k2 : spec_keys) dup = dup || k2 == key;
            if (dup) continue;
            spec_keys.push_back(key);
            spec_states.insert(spec_states.end(), cs, cs + f);
          }
        };
        // (the group grows while its first members' children are looked at: only members picked from the heap count)
        const size_t n_par = std::min(group.size(), (size_t)n_spec_parents);
        for (size_t gi = 0; gi < n_par; gi++) children(group[gi]->coord);
      }
      t_pick += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_p0).count();
      if (int rc = run_batch(group)) return rc;
    }
    return fetch(curr, v);
  }

  int run_batch(const std::vector<NodePtr> &group) {
    const int f = F();
    // (speculated states only ever ride along into a store of their own shape: a control table or state size that
    // changed since they were assembled would index the store's arrays with the wrong strides)
    if (!spec_keys.empty() && (spec.cap == 0 || spec.nU != nU || spec.F != f)) { spec_keys.clear(); spec_states.clear(); }
    const int64_t ng = (int64_t)group.size(), ns = (int64_t)spec_keys.size();
    const int64_t n = ng + ns;  // the launch: the nodes of the group, then the speculated states
    std::vector<double> &nodes = nodes_buf;
    nodes.resize((size_t)f * n);
    for (int64_t k = 0; k < ng; k++)
      for (int r = 0; r < f; r++) nodes[(size_t)r * n + k] = group[(size_t)k]->coord[(size_t)r];
    for (int64_t k = 0; k < ns; k++)
      for (int r = 0; r < f; r++) nodes[(size_t)r * n + ng + k] = spec_states[(size_t)k * f + r];
    const int64_t slots = n * nU;
    // lists of the speculated states into the store (entry e <- column ng + k of the launch)
    auto harvest = [&](int64_t k, int32_t m, const double *cost, const uint64_t *keys, const int32_t *act) {
      const int e = spec.insert(spec_keys[(size_t)k]);
      std::copy(&spec_states[(size_t)k * f], &spec_states[(size_t)k * f] + f, &spec.coord[(size_t)e * f]);
      spec.m[(size_t)e] = m;
      std::copy(cost, cost + m, &spec.cost[(size_t)e * nU]);
      std::copy(act, act + m, &spec.act[(size_t)e * nU]);
      if (keys) std::copy(keys, keys + m, &spec.keys[(size_t)e * nU]);
    };
    last.device_launches++;
    last.pairs += slots;
    if (packed) {
      const auto t_f0 = std::chrono::steady_clock::now();
      for (NodePtr p : cur_group)  // what the previous launch delivered and the search has not consumed yet
        if (p->cached && p->c_slot >= 0 && p->c_batch == cur_batch) keep(*p);  // (a re-opened node is closed AND waiting)
      const auto t_l0 = std::chrono::steady_clock::now();
      t_fill += std::chrono::duration<double, std::milli>(t_l0 - t_f0).count();
      cur_group.clear();
      cur_batch++;
      if (int rc = packed(user, nodes.data(), n, &cur_view)) return rc;
      t_provider += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_l0).count();
      for (int64_t k = 0; k < ng; k++) {
        Node &nd = *group[(size_t)k];
        nd.c_slot = (int32_t)k;
        nd.c_batch = cur_batch;
        nd.cached = true;
      }
      if (ns) spec.has_keys = true;
      for (int64_t k = 0; k < ns; k++) {
        const size_t o = (size_t)cur_view.offs[ng + k];
        harvest(k, cur_view.count[ng + k], cur_view.cost + o, cur_view.hash + o, cur_view.action + o);
      }
      cur_group.assign(group.begin(), group.end());
      return 0;
    }
    if (lists) {
      // compact per-node lists with the device's lattice hashes: no scan of skipped slots, no host hashing
      b_cnt.resize((size_t)n);
      b_act.resize((size_t)slots);
      b_cost.resize((size_t)slots);
      b_hash.resize((size_t)slots);
      b_state.resize((size_t)f * slots);
      const auto t_l0 = std::chrono::steady_clock::now();
      if (int rc = lists(user, nodes.data(), n, b_cnt.data(), b_act.data(), b_cost.data(), b_hash.data(), b_state.data()))
        return rc;
      const auto t_l1 = std::chrono::steady_clock::now();
      t_provider += std::chrono::duration<double, std::milli>(t_l1 - t_l0).count();
      if (ns) spec.has_keys = true;
      for (int64_t k = 0; k < ns; k++) {
        const int64_t o = (ng + k) * nU;
        harvest(k, b_cnt[(size_t)(ng + k)], b_cost.data() + o, b_hash.data() + o, b_act.data() + o);
      }
      for (int64_t k = 0; k < ng; k++) {
        Node &nd = *group[(size_t)k];
        const int32_t m = b_cnt[(size_t)k];
        const int64_t o = k * nU;
        nd.c_succ.resize((size_t)m * f);
        for (int r = 0; r < f; r++) {
          const double *row = b_state.data() + (size_t)r * slots + o;
          for (int32_t j = 0; j < m; j++) nd.c_succ[(size_t)j * f + r] = row[j];
        }
        nd.c_cost.assign(b_cost.begin() + o, b_cost.begin() + o + m);
        nd.c_act.assign(b_act.begin() + o, b_act.begin() + o + m);
        nd.c_key.assign(b_hash.begin() + o, b_hash.begin() + o + m);
        nd.cached = true;
      }
      t_fill += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_l1).count();
      return 0;
    }
    std::vector<uint8_t> st((size_t)slots);
    std::vector<double> cs((size_t)slots), state((size_t)f * slots);
    const auto t_b0 = std::chrono::steady_clock::now();
    if (int rc = batched(user, nodes.data(), n, st.data(), cs.data(), state.data())) return rc;
    const auto t_b1 = std::chrono::steady_clock::now();
    t_provider += std::chrono::duration<double, std::milli>(t_b1 - t_b0).count();
    if (ns) spec.has_keys = false;
    for (int64_t k = 0; k < ns; k++) {  // dense slots: the emitted ones, in control order
      int32_t m = 0;
      for (int i = 0; i < nU; i++) {
        const int64_t sl = (ng + k) * nU + i;
        if (st[(size_t)sl] != 1 && st[(size_t)sl] != 2) continue;
        v_cost[(size_t)m] = cs[(size_t)sl];
        v_act[(size_t)m] = i;
        m++;
      }
      harvest(k, m, v_cost.data(), nullptr, v_act.data());
    }
    for (int64_t k = 0; k < ng; k++) {
      Node &nd = *group[(size_t)k];
      nd.c_succ.clear(); nd.c_cost.clear(); nd.c_act.clear();
      for (int i = 0; i < nU; i++) {
        const int64_t s = k * nU + i;
        if (st[(size_t)s] != 1 && st[(size_t)s] != 2) continue;  // emitted: finite or blocked
        for (int r = 0; r < f; r++) nd.c_succ.push_back(state[(size_t)r * slots + s]);
        nd.c_cost.push_back(cs[(size_t)s]);
        nd.c_act.push_back(i);
      }
      nd.cached = true;
    }
    t_fill += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_b1).count();
    return 0;
  }

  // Hands the cached lists of `n` to the relaxation loop and forgets them: the reference expands a node again
  // when a cheaper path re-opens it after it was closed (graph_search.h:108-141 pushes it a second time; that
  // happens with an inconsistent heuristic, i.e. eps > 1 or the default v_max <= 0), so the cache entry must not
  // outlive its one use -- a re-popped node goes through run_batch again (get_succ is pure: same lists).
  int fetch(const NodePtr &n, SuccView *v) {
    if (!n->cached) return -1;
    const int f = F();
    n->cached = false;
    if (packed && n->c_slot >= 0 && n->c_batch == cur_batch) {
      // still in the landing buffer of the latest launch: read in place
      const size_t o = (size_t)cur_view.offs[n->c_slot];
      *v = SuccView{cur_view.count[n->c_slot], cur_view.cost + o, cur_view.hash + o, cur_view.action + o,
                    cur_view.state ? cur_view.state + o : nullptr, cur_view.total, 1};
      n->c_slot = -1;
      return 0;
    }
    if (packed) {
      if (cur_blob) free_blobs.push_back(cur_blob);  // the previous expansion's lists are done with
      cur_blob = n->c_blob;  // the planner takes the node's lists over for the duration of this expansion
      n->c_blob = nullptr;
      n->c_slot = -1;
      const size_t m = (size_t)n->c_m;
      n->c_m = 0;
      const char *b = cur_blob;
      *v = SuccView{(int32_t)m, (const double *)b, (const uint64_t *)(b + m * 8), (const int32_t *)(b + m * 8 * (size_t)(2 + f)),
                    n->c_has_state ? (const double *)(b + m * 16) : nullptr, (int64_t)m, 1};
      return 0;
    }
    std::copy(n->c_succ.begin(), n->c_succ.end(), v_succ.begin());
    std::copy(n->c_cost.begin(), n->c_cost.end(), v_cost.begin());
    std::copy(n->c_act.begin(), n->c_act.end(), v_act.begin());
    const bool have_keys = n->c_key.size() == n->c_act.size() && !n->c_act.empty();
    if (have_keys) std::copy(n->c_key.begin(), n->c_key.end(), v_keys.begin());
    *v = SuccView{(int32_t)n->c_act.size(), v_cost.data(), have_keys ? v_keys.data() : nullptr, v_act.data(), v_succ.data(), 1, f};
    n->c_key.clear(); n->c_key.shrink_to_fit();
    n->c_succ.clear(); n->c_succ.shrink_to_fit();
    n->c_cost.clear(); n->c_act.clear();
    return 0;
  }

  // GraphSearch::recoverTraj, graph_search.h:369-455
  bool recover(NodePtr curr, const double *start) {
    const int f = F();
    const uint64_t start_key = lattice_hash(dim, control, start);
    last.traj_end.assign(curr->coord, curr->coord + f);
    std::vector<std::vector<double>> from;
    std::vector<int32_t> acts;
    bool found = false;
    while (curr->pred_head >= 0) {
      int min_id = -1;
      double min_rhs = kInf, min_g = kInf;
      for (int32_t i = curr->pred_head; i >= 0; i = preds[(size_t)i].next) {
        const PredRec &pr = preds[(size_t)i];
        const NodePtr &p = hm[pr.key];
        const double v = p->g + pr.cost;
        if (min_rhs > v) { min_rhs = v; min_g = p->g; min_id = (int)i; }
        else if (!std::isinf(pr.cost) && min_rhs == v) {
          if (min_g < p->g) { min_g = p->g; min_id = (int)i; }
        }
      }
      if (min_id < 0) break;
      const int a = preds[(size_t)min_id].action;
      curr = hm[preds[(size_t)min_id].key];
      from.push_back(std::vector<double>(curr->coord, curr->coord + f));
      acts.push_back(a);
      if (curr->key == start_key) { found = true; break; }
    }
    if (!found) return false;
    std::reverse(from.begin(), from.end());
    std::reverse(acts.begin(), acts.end());
    // Trajectory(prs): total time and efforts (trajectory.h:52-57, 250-254)
    const int K = (control & 8) ? 4 : (control & 4) ? 3 : (control & 2) ? 2 : 1;
    for (size_t s = 0; s < from.size(); s++) {
      const double *nd = from[s].data();
      const double *u = &U[(size_t)acts[s] * udim];
      last.total_time += dt;
      for (int order = 1; order <= 4; order++) {
        double j = 0;
        for (int i = 0; i < dim; i++) {
          double c[6] = {0, 0, 0, 0, 0, 0};
          c[5] = nd[i];
          if (K == 1) c[4] = u[i];
          if (K == 2) { c[4] = nd[dim + i]; c[3] = u[i]; }
          if (K == 3) { c[4] = nd[dim + i]; c[3] = nd[2 * dim + i]; c[2] = u[i]; }
          if (K == 4) { c[4] = nd[dim + i]; c[3] = nd[2 * dim + i]; c[2] = nd[3 * dim + i]; c[1] = u[i]; }
          j += effort_1d(c, dt, order);
        }
        last.J[order - 1] += j;
      }
      last.traj_nodes.insert(last.traj_nodes.end(), nd, nd + f);
      last.traj_actions.push_back(acts[s]);
    }
    return true;
  }
};

}  // namespace host
}  // namespace mplx
#endif
