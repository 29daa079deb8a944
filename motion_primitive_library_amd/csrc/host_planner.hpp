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
#include <new>
#include <vector>

#include <sys/mman.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

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
  const double *heur = nullptr;     // [total] default heuristic of every successor (env_base.h:46-64), when the provider computes it
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
  const double *heur = nullptr;    // the provider's heuristic of every successor, or null (then the search evaluates it)
};

// ---- the search's bookkeeping (StateSpace of the reference, state_space.h:37-104), laid out for the relaxation loop.
// A 3D search with hundreds of controls relaxes 300+ edges per expansion and eight in ten of them neither create a
// node nor improve one: all such a relaxation needs is the child's g and the head of its predecessor list.  Both live
// INSIDE the hash table's slot, so that relaxation is one cache line (the slot, prefetched ahead through the device's
// lattice hashes) plus a sequential append to the predecessor pool.  Everything else about a node -- heuristic, heap
// handle, flags, where its state comes from -- is in `Cold`, touched only when a node is created, improved, picked
// for a launch or popped.  Nodes are 32-bit indices in creation order; their 4D+2 coordinates are not stored at
// creation: a node remembers the (parent, control) that created it (the reference keeps the state of the FIRST
// parent that reached a lattice hash, graph_search.h:83-85) and forward_state() rebuilds exactly that state if and
// when the node is picked for expansion -- one node in fifteen on the 3D problems.
struct Slot {          // hm_ of the reference (state_space.h:78): lattice hash -> node, plus the node's hot fields
  uint64_t key;
  double g;            // State::g (state_space.h:58)
  uint32_t idx;        // node index (creation order)
  int32_t pred_head;   // newest record of the node's predecessor list (Planner::preds), -1 = none
  uint32_t epoch;      // slot is live iff epoch == table epoch (clear() is O(1): planners are re-used plan after plan)
  uint32_t pad;
};
static_assert(sizeof(Slot) == 32, "two slots per cache line");

class NodeTable {
 public:
  ~NodeTable() { std::free(s_); }
  NodeTable() = default;
  NodeTable(const NodeTable &) = delete;
  NodeTable &operator=(const NodeTable &) = delete;
  static constexpr uint32_t kNoIndex = 0xffffffffu;
  // Room for `more` insertions without a rehash: slot pointers stay valid that long.
  void reserve(size_t more) {
    while ((n_ + more) * 10 > cap_ * 6) grow();
  }
  // The slot of `key`; `fresh` says it was created by this call (g = inf, no predecessors, idx = kNoIndex until the
  // caller numbers the node).  Call reserve() first; the pointer is valid until the next reserve().
  Slot *insert(uint64_t key, bool *fresh) {
    size_t i = home(key);
    for (;; i = (i + 1) & (cap_ - 1)) {
      Slot &s = s_[i];
      if (s.epoch != epoch_) {
        s.key = key;
        s.g = kInf;
        s.idx = kNoIndex;
        s.pred_head = -1;
        s.epoch = epoch_;
        n_++;
        *fresh = true;
        return &s;
      }
      if (s.key == key) { *fresh = false; return &s; }
    }
  }
  Slot *find(uint64_t key) const {
    if (!cap_) return nullptr;
    for (size_t i = home(key);; i = (i + 1) & (cap_ - 1)) {
      Slot &s = s_[i];
      if (s.epoch != epoch_) return nullptr;
      if (s.key == key) return &s;
    }
  }
  void prefetch(uint64_t key) const {
    if (cap_) __builtin_prefetch(&s_[home(key)]);
  }
  size_t size() const { return n_; }
  void clear() {
    n_ = 0;
    if (++epoch_ == 0) {  // (4 G plans later)
      std::memset((void *)s_, 0, cap_ * sizeof(Slot));
      epoch_ = 1;
    }
  }

 private:
  size_t home(uint64_t k) const {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    return (size_t)k & (cap_ - 1);
  }
  static Slot *alloc(size_t cap) {
    const size_t bytes = cap * sizeof(Slot), huge = (size_t)2 << 20;
    void *p = nullptr;
    if (posix_memalign(&p, bytes >= huge ? huge : 64, bytes) != 0) throw std::bad_alloc();
#ifdef MADV_HUGEPAGE
    if (bytes >= huge) (void)madvise(p, bytes, MADV_HUGEPAGE);  // 100+ MB of random probes: 4-KB pages would miss the TLB every time
#endif
    std::memset(p, 0, bytes);
    return (Slot *)p;
  }
  void grow() {
    Slot *old = s_;
    const size_t old_cap = cap_;
    cap_ = cap_ ? cap_ * 2 : (size_t)1 << 16;
    s_ = alloc(cap_);
    for (size_t j = 0; j < old_cap; j++)
      if (old[j].epoch == epoch_) {
        size_t i = home(old[j].key);
        while (s_[i].epoch == epoch_) i = (i + 1) & (cap_ - 1);
        s_[i] = old[j];
      }
    std::free(old);
  }
  Slot *s_ = nullptr;
  size_t cap_ = 0, n_ = 0;
  uint32_t epoch_ = 1;
};

// Allocator of the search's big random-access arrays (nodes, heap, handles): 2-MB aligned and advised as huge pages
// from 4 MB on -- a 50-MB node array on 4-KB pages misses the TLB on nearly every relaxation that improves a node.
template <class T>
struct HugeAlloc {
  using value_type = T;
  HugeAlloc() = default;
  template <class O> HugeAlloc(const HugeAlloc<O> &) {}
  T *allocate(size_t n) {
    const size_t bytes = n * sizeof(T), huge = (size_t)2 << 20;
    void *p = nullptr;
    if (posix_memalign(&p, bytes >= 2 * huge ? huge : 64, bytes ? bytes : 64) != 0) throw std::bad_alloc();
#ifdef MADV_HUGEPAGE
    if (bytes >= 2 * huge) (void)madvise(p, bytes, MADV_HUGEPAGE);
#endif
    return (T *)p;
  }
  void deallocate(T *p, size_t) { std::free(p); }
  template <class O> bool operator==(const HugeAlloc<O> &) const { return true; }
  template <class O> bool operator!=(const HugeAlloc<O> &) const { return false; }
};

// An append-only array whose new elements are NOT initialised (std::vector::resize would write every element once
// before the search writes it again): the predecessor pool grows by 300 entries per expansion, 14.5 M per 160^3 plan.
template <class T>
class RawVec {
 public:
  RawVec() = default;
  RawVec(const RawVec &) = delete;
  RawVec &operator=(const RawVec &) = delete;
  ~RawVec() { std::free(p_); }
  size_t size() const { return n_; }
  void clear() { n_ = 0; }
  T *data() { return p_; }
  const T &operator[](size_t i) const { return p_[i]; }
  // room for `more` further elements; returns where they start (the caller writes all of them)
  T *grow(size_t more) {
    if (n_ + more > cap_) {
      size_t cap = cap_ ? cap_ : (size_t)1 << 16;
      while (cap < n_ + more) cap *= 2;
      void *q = nullptr;
      if (posix_memalign(&q, 64, cap * sizeof(T)) != 0) throw std::bad_alloc();
#ifdef MADV_HUGEPAGE
      if (cap * sizeof(T) >= ((size_t)4 << 20)) (void)madvise(q, cap * sizeof(T), MADV_HUGEPAGE);
#endif
      if (n_) std::memcpy(q, p_, n_ * sizeof(T));
      std::free(p_);
      p_ = (T *)q;
      cap_ = cap;
    }
    T *at = p_ + n_;
    n_ += more;
    return at;
  }

 private:
  T *p_ = nullptr;
  size_t n_ = 0, cap_ = 0;
};

struct Cold {            // per node, indexed by node index
  double g = kInf;       // copy of the slot's g (heap tie-breaks of re-opened nodes, recoverTraj)
  double h = kInf;       // State::h
  uint64_t key = 0;
  uint32_t parent = 0;   // the node whose expansion created this one (the start node: itself) ...
  int32_t action = -1;   // ... through this control: coordinates = forward_state(parent, U[action])
  int32_t coord = -1;    // index into Planner::coords once the state has been materialised
  int32_t handle = -1;   // heap handle of the latest push (State::heapkey)
  int32_t cache = -1;    // successor lists waiting for this node's expansion (Planner::caches)
  uint32_t pick_stamp = 0;  // launch for which the node was last picked (a re-opened node sits in the heap twice)
  bool opened = false, closed = false;
};

// Mutable binary max-heap on compare_pair (state_space.h:16-34): the top is the smallest f, ties go to the smaller
// min(g, rhs).  Sift rules follow a 2-ary boost::heap::d_ary_heap (mutable): sift-up stops at equality, sift-down
// swaps at equality and prefers the first maximal child.  Entries are addressed through handles like boost's: a node
// that is pushed while an older entry of it is still in the heap (closed, improved twice before its second pop --
// graph_search.h:117-136 pushes instead of updating because iterationclosed is still set) owns two entries, and
// `increase` moves the one of the LATEST push.  The tie-break reads the node's g at comparison time in the reference
// (compare_pair dereferences the state); an entry carries a copy that is exact as long as every node has one entry,
// and from the first double entry on the live value is read instead (`live`).
class OpenList {
 public:
  struct Item { double f, g; uint32_t idx; int32_t handle; };
  bool empty() const { return q_.empty(); }
  size_t size() const { return q_.size(); }
  const Item &top() const { return q_.front(); }
  const std::vector<Item, HugeAlloc<Item>> &items() const { return q_; }
  void reset(const std::vector<Cold, HugeAlloc<Cold>> *cold) { q_.clear(); pos_.clear(); cold_ = cold; live = false; }
  bool live = false;
  int32_t push(double f, double g, uint32_t idx) {
    const int32_t h = (int32_t)pos_.size();
    pos_.push_back((int32_t)q_.size());
    q_.push_back({f, g, idx, h});
    up((int)q_.size() - 1);
    return h;
  }
  void pop() {
    pos_[(size_t)q_.front().handle] = -1;
    if (q_.size() > 1) {
      q_.front() = q_.back();
      q_.pop_back();
      pos_[(size_t)q_.front().handle] = 0;
      down(0);
    } else {
      q_.pop_back();
    }
  }
  // key became better (smaller f): state_space `increase` (graph_search.h:133)
  void increase(int32_t handle, double f, double g) {
    const int i = pos_[(size_t)handle];
    q_[(size_t)i].f = f;
    q_[(size_t)i].g = g;
    up(i);
  }
  int position(int32_t handle) const { return pos_[(size_t)handle]; }
  double tie(const Item &a) const { return live ? (*cold_)[a.idx].g : a.g; }
  bool less(const Item &a, const Item &b) const {
    if (a.f == b.f) return tie(a) > tie(b);
    return a.f > b.f;
  }

 private:
  void swap_at(int i, int j) {
    std::swap(q_[(size_t)i], q_[(size_t)j]);
    pos_[(size_t)q_[(size_t)i].handle] = i;
    pos_[(size_t)q_[(size_t)j].handle] = j;
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
  std::vector<Item, HugeAlloc<Item>> q_;
  std::vector<int32_t, HugeAlloc<int32_t>> pos_;  // handle -> position in q_ (-1 once popped)
  const std::vector<Cold, HugeAlloc<Cold>> *cold_ = nullptr;
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

// forward_state()'s position rows only (the default heuristic of a new node needs nothing else, env_base.h:58-64)
inline void forward_pos(int dim, int control, const double *nd, const double *u, double T, double *out) {
  const int K = (control & 8) ? 4 : (control & 4) ? 3 : (control & 2) ? 2 : 1;
  const double t3 = (T * T) * T;
  for (int i = 0; i < dim; i++) {
    const double p = nd[i], v = nd[dim + i], a = nd[2 * dim + i], j = nd[3 * dim + i], ui = u[i];
    if (K == 1) out[i] = (0.0 + ui * T) + p;
    else if (K == 2) out[i] = ((0.0 + ((ui / 2) * T) * T) + v * T) + p;
    else if (K == 3) out[i] = (((0.0 + (ui / 6) * t3) + ((a / 2) * T) * T) + v * T) + p;
    else out[i] = ((((0.0 + (ui / 24) * (t3 * T)) + (j / 6) * t3) + ((a / 2) * T) * T) + v * T) + p;
  }
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
  // where the wall time went (ms) and what the relaxation loop did
  double t_total = 0, t_provider = 0, t_fill = 0, t_pick = 0, t_relax = 0, t_recover = 0;
  int64_t relaxed = 0, improved = 0, pushes = 0, materialised = 0, heur_from_provider = 0;
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

  struct PredRec { double cost; uint32_t parent; int32_t next; };  // pred_coord / pred_action_cost of state_space.h:49-53
  NodeTable hm;
  std::vector<Cold, HugeAlloc<Cold>> cold;  // nodes in creation order
  RawVec<PredRec> preds;         // every relaxed edge (graph_search.h:97-99), newest first per child
  RawVec<int32_t> pred_act;      // pred_action_id, parallel to preds
  std::vector<double> coords;    // [materialised][4D+2]
  OpenList pq;
  PlanResult last;

  int F() const { return 4 * dim + 2; }

  // order-independent digest of the closed set (sum of mixed lattice hashes): two searches that closed the same nodes agree
  uint64_t closed_checksum() const {
    uint64_t s = 0;
    for (const Cold &nd : cold)
      if (nd.closed) { uint64_t k = nd.key; k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; s += k; }
    return s;
  }
  // PlannerBase::getCloseSet (positions, creation order): fills up to cap points, returns the closed-set size
  int32_t closed_positions(double *pos, int32_t cap) {
    int32_t m = 0;
    for (uint32_t i = 0; i < (uint32_t)cold.size(); i++) {
      if (!cold[i].closed) continue;
      if (pos && m < cap) {
        const double *c = coord_of(i);
        for (int k = 0; k < dim; k++) pos[(size_t)m * dim + k] = c[k];
      }
      m++;
    }
    return m;
  }
  // PlannerBase::getOpenSet walks the heap (planner_base.h:77-81): full states, heap order
  int32_t open_states(double *states, int32_t cap) {
    const int f = F();
    int32_t m = 0;
    for (const OpenList::Item &it : pq.items()) {
      if (states && m < cap) {
        const double *c = coord_of(it.idx);
        for (int k = 0; k < f; k++) states[(size_t)m * f + k] = c[k];
      }
      m++;
    }
    return m;
  }

  double heur(const double *s, const double *goal) const {  // env_base.h:46-64
    return heur_keyed(s, lattice_hash(dim, control, s), goal, lattice_hash(dim, effective_goal_control(), goal));
  }
  double heur_keyed(const double *s, uint64_t s_key, const double *goal, uint64_t goal_key) const {  // s: a full state
    if (s_key == goal_key) return 0;
    return heur_at(s, s[4 * dim + 1], goal);
  }

  // ---- prior trajectory (env_map::set_prior_trajectory, env_map.h:189-226; env_base::get_heur, env_base.h:46-52): a
  // state at time t is guided towards where the prior trajectory is at that time -- h = cal_heur(state, prior(t)) + the
  // prior's remaining cost -- instead of towards the goal.  With a potential map (round 6) the values of the cells the
  // prior trajectory's samples fall into come through `pot` (the device's potential map, or the caller's host copy).
  std::vector<double> prior_pos;   // [n][D] position of traj.evaluate(k dt)
  std::vector<double> prior_togo;  // [n] total_cost - costs[k]
  double prior_goal[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // traj.evaluate(total_time): THE goal while a prior is set
  int prior_control = 0;           // ... and its control flag (the prior trajectory's)
  bool has_prior() const { return !prior_togo.empty(); }
  void clear_prior() { prior_pos.clear(); prior_togo.clear(); }
  // env_base::set_goal (env_base.h:295-298) keeps the goal set_prior_trajectory installed (env_map.h:225): with a prior
  // trajectory the search ends where THAT ends, whatever goal plan() is given.
  const double *effective_goal(const double *goal) const { return has_prior() ? prior_goal : goal; }
  int effective_goal_control() const { return has_prior() ? prior_control : (goal_control ? goal_control : control); }
  // nodes: [segs][4D+2] segment start states, actions: [segs] rows of `pU` (udim entries each), the prior's control flag
  // and primitive duration.  v_max, w, dt and the grid of THIS planner must be set (the reference's call order,
  // test_planner_2d_with_prior_traj.cpp:60-70).
  // pot(user, cell indices, n, values out): the potential map's values at cells inside the map, nullptr = no potential map
  // (env_map::potential_map_ empty); pot_w / grad_w: env_map::potential_weight_ / gradient_weight_.
  typedef int (*pot_lookup_fn)(void *user, const int64_t *idx, int64_t n, int8_t *out);
  int set_prior_trajectory(const double *nodes, const int32_t *actions, int segs, int pcontrol, const double *pU, int pudim, double pdt,
                           pot_lookup_fn pot = nullptr, void *pot_user = nullptr, double pot_w = 0.0, double grad_w = 0.0) {
    clear_prior();
    if (segs <= 0) return 0;
    if (!(dt > 0) || !(pdt > 0) || !(grid.res > 0)) return -1;  // (the loops below step by dt)
    const int f = F(), K = (pcontrol & 8) ? 4 : (pcontrol & 4) ? 3 : (pcontrol & 2) ? 2 : 1;
    // Primitive1D coefficient vectors of every segment and axis (primitive.h:34-50), highest order first
    std::vector<double> c((size_t)segs * dim * 6, 0.0), taus{0.0};
    for (int s = 0; s < segs; s++) {
      const double *nd = nodes + (size_t)s * f, *u = pU + (size_t)actions[s] * pudim;
      for (int i = 0; i < dim; i++) {
        double *ci = &c[((size_t)s * dim + i) * 6];
        ci[5] = nd[i];
        if (K == 1) ci[4] = u[i];
        if (K == 2) { ci[4] = nd[dim + i]; ci[3] = u[i]; }
        if (K == 3) { ci[4] = nd[dim + i]; ci[3] = nd[2 * dim + i]; ci[2] = u[i]; }
        if (K == 4) { ci[4] = nd[dim + i]; ci[3] = nd[2 * dim + i]; ci[2] = nd[3 * dim + i]; ci[1] = u[i]; }
      }
      taus.push_back(pdt + taus.back());  // trajectory.h:52-56
    }
    const double total_time = taus.back();
    auto p_of = [&](int s, int i, double t) {  // Primitive1D::p, primitive.h:128-131 (power() of math.h:197-205)
      const double *ci = &c[((size_t)s * dim + i) * 6];
      auto power = [](double x, int n) { double r = 1; while (n-- > 0) r *= x; return r; };
      return ci[0] / 120 * power(t, 5) + ci[1] / 24 * power(t, 4) + ci[2] / 6 * power(t, 3) + ci[3] / 2 * t * t + ci[4] * t + ci[5];
    };
    auto v_of = [&](int s, int i, double t) {  // Primitive1D::v, primitive.h:133-137
      const double *ci = &c[((size_t)s * dim + i) * 6];
      auto power = [](double x, int n) { double r = 1; while (n-- > 0) r *= x; return r; };
      return ci[0] / 24 * power(t, 4) + ci[1] / 6 * power(t, 3) + ci[2] / 2 * t * t + ci[3] * t + ci[4];
    };
    auto clamp = [&](double tau) { if (tau < 0) tau = 0; if (tau > total_time) tau = total_time; return tau; };
    // traj.sample(n) (trajectory.h:230-236, 99-131): Command k at time k * total / n -- its time stamp (the unclamped
    // time), cell, |vel| (Command::vel = pr.v(tau) / lambda with lambda = 1; Eigen's norm over the D components)
    struct Sample { double t, vnorm; int64_t idx; bool outside, occupied; int8_t pot; };
    std::vector<Sample> pts;
    {
      const int n = (int)std::ceil(v_max * total_time / grid.res);
      const double sdt = total_time / n;
      for (int k = 0; k <= n && n > 0; k++) {
        double tau = clamp(k * sdt), pt[3] = {0, 0, 0}, vv = 0;
        for (int s = 0; s < segs; s++)
          if (tau >= taus[(size_t)s] && tau <= taus[(size_t)s + 1]) {
            tau -= taus[(size_t)s];
            for (int i = 0; i < dim; i++) {
              pt[i] = p_of(s, i, tau);
              const double v = v_of(s, i, tau) / 1.0;
              vv += v * v;
            }
            break;
          }
        int pn[3] = {0, 0, 0};
        grid.to_cell(pt, pn);
        Sample sm;
        sm.t = k * sdt;
        sm.vnorm = std::sqrt(vv);
        sm.idx = (int)grid.index(pn);  // (MapUtil::getIndex is int arithmetic, also for cells outside)
        sm.outside = grid.outside(pn);
        sm.occupied = !sm.outside && grid.is_occupied(pn);
        sm.pot = 0;
        pts.push_back(sm);
      }
    }
    if (pot) {  // the potential map's values at the samples' cells, one call
      std::vector<int64_t> idx;
      std::vector<size_t> who;
      for (size_t k = 0; k < pts.size(); k++)
        if (!pts[k].outside) { idx.push_back(pts[k].idx); who.push_back(k); }
      std::vector<int8_t> val(idx.size());
      if (!idx.empty() && pot(pot_user, idx.data(), (int64_t)idx.size(), val.data()) != 0) return -2;
      for (size_t q = 0; q < who.size(); q++) pts[who[q]].pot = val[q];
    }
    // env_map::traverse_trajectory (env_map.h:229-255): +inf when a sample is outside, occupied (no potential map) or
    // inside an obstacle's core (potential >= 100); the potential + gradient terms of the cells passed through
    double traverse = 0;
    {
      int64_t prev_idx = -1;
      for (const Sample &sm : pts) {
        if (sm.idx == prev_idx) continue;
        prev_idx = sm.idx;
        if (sm.outside) { traverse = kInf; break; }
        if (pot) {
          if (sm.pot < 100 && sm.pot > 0) traverse += pot_w * sm.pot + grad_w * sm.vnorm;
          else if (sm.pot >= 100) { traverse = kInf; break; }
        } else if (sm.occupied) { traverse = kInf; break; }
      }
    }
    const double total_cost = traverse + w * total_time;
    std::vector<double> costs;
    for (double t = 0; t < total_time; t += dt) {  // env_map.h:197-216
      double potential_cost = 0;
      if (pot) {
        int64_t prev_idx = -1;
        for (const Sample &sm : pts) {
          if (sm.t >= t) break;
          if (sm.idx == prev_idx) continue;
          prev_idx = sm.idx;
          // (the reference reads potential_map_[idx] unchecked here: a sample outside the map is undefined there, 0 here)
          potential_cost += pot_w * sm.pot + grad_w * sm.vnorm;
        }
      }
      costs.push_back(w * t + potential_cost);
    }
    for (double t = 0; t < total_time; t += dt) {
      const int id = (int)(t / dt);
      double tau = clamp(t);
      for (int s = 0; s < segs; s++)  // Trajectory::evaluate(t) -> Waypoint, trajectory.h:67-86
        if ((tau >= taus[(size_t)s] && tau < taus[(size_t)s + 1]) || s == segs - 1) {
          tau -= taus[(size_t)s];
          for (int i = 0; i < dim; i++) prior_pos.push_back(p_of(s, i, tau));
          break;
        }
      prior_togo.push_back(total_cost - costs[(size_t)id]);
    }
    // goal_node_ = traj.evaluate(total_time) (env_map.h:225): the last segment at its full duration, the prior's flags
    {
      const int s = segs - 1;
      forward_state(dim, pcontrol, nodes + (size_t)s * f, pU + (size_t)actions[s] * pudim, clamp(total_time) - taus[(size_t)s], prior_goal);
      prior_goal[4 * dim + 1] = 0.0;  // (Trajectory::evaluate returns a fresh Waypoint: t = 0)
      prior_control = pcontrol;
    }
    return 0;
  }
  // env_base::get_heur for a state at position `s` and time `t` whose lattice hash differs from the goal's
  double heur_at(const double *s, double t, const double *goal) const {
    auto linf = [&](const double *b) {
      double m = 0;
      for (int i = 0; i < dim; i++) m = std::max(m, std::abs(s[i] - b[i]));
      return v_max > 0 ? w * m / v_max : w * m;
    };
    const size_t id = t > 0 ? (size_t)(t / dt) : 0;  // env_base.h:48 (a size_t there: times are never negative)
    if (has_prior() && id < prior_togo.size()) return linf(&prior_pos[id * (size_t)dim]) + prior_togo[id];
    return linf(goal);
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

  // The node's state: the one its creating expansion produced (graph_search.h:83-85), rebuilt on first use.
  const double *coord_of(uint32_t idx) {
    Cold &nd = cold[idx];
    if (nd.coord < 0) {
      const int f = F();
      const double *par = coord_of(nd.parent);  // (a creating parent was expanded, so it has its state: depth 1)
      const size_t at = coords.size();
      coords.resize(at + (size_t)f);
      // (coord_of(parent) may have pointed into `coords` before the resize: take it again)
      par = &coords[(size_t)cold[nd.parent].coord * (size_t)f];
      forward_state(dim, control, par, &U[(size_t)cold[idx].action * udim], dt, &coords[at]);
      cold[idx].coord = (int32_t)(at / (size_t)f);
      last.materialised++;
    }
    return &coords[(size_t)cold[idx].coord * (size_t)F()];
  }

  // PlannerBase::plan (A*), planner_base.h:275-325 + GraphSearch::Astar
  int plan(const double *start, const double *goal_given) {
    const double *goal = effective_goal(goal_given);
    using clk = std::chrono::steady_clock;
    auto ms_since = [](clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); };
    const auto t_plan0 = clk::now();
    last = PlanResult();
    t_succ = t_provider = t_fill = t_pick = 0;
    checked_states = 0;
    hm.clear();
    cold.clear();
    coords.clear();
    preds.clear();
    pred_act.clear();
    caches.clear();
    free_caches.clear();
    if (spec.cap) spec.clear();  // (the map may have changed since the last plan)
    spec_now = spec_cfg();
    spec_keys.clear();  // (states a plan with batch > 1 assembled must not ride along in this plan's launches)
    spec_states.clear();
    blob_bytes_check();
    free_blobs.clear();
    for (auto &b : all_blobs) free_blobs.push_back(b.get());  // (buffers of the previous plan are recycled, not freed)
    cur_blob = nullptr;
    cur_group.clear();
    cur_view = PackedView();
    cur_view_n = 0;
    pq.reset(&cold);
    if (!single && !batched && !lists && !packed) return -1;
    int pn[3];
    grid.to_cell(start, pn);
    if (!grid.is_free(pn)) return 0;  // "start is not free": plan() == false
    const int f = F();
    if (is_goal(start, goal)) { last.ok = true; last.cost = 0; return 0; }

    const uint64_t goal_key = lattice_hash(dim, effective_goal_control(), goal);
    {
      bool fresh;
      const uint64_t key = lattice_hash(dim, control, start);
      hm.reserve(1);
      Slot *sl = hm.insert(key, &fresh);
      sl->g = 0;
      sl->idx = 0;
      cold.emplace_back();
      Cold &nd = cold.back();
      nd.key = key;
      nd.g = 0;
      nd.h = eps == 0 ? 0 : heur(start, goal);
      nd.parent = 0;
      nd.coord = 0;
      coords.assign(start, start + f);
      nd.opened = true;
      nd.handle = pq.push(nd.g + eps * nd.h, nd.g, 0);
    }

    v_succ.resize((size_t)nU * f);
    v_cost.resize((size_t)nU);
    v_act.resize((size_t)nU);
    v_keys.resize((size_t)nU);
    r_fin.resize((size_t)nU + 8);  // (+8: the compress-store of pass 0 may touch a full vector past the end)
    r_new.resize((size_t)nU + 1);
    r_imp.resize((size_t)nU + 1);
    r_tent.resize((size_t)nU);
    r_slot.resize((size_t)nU);
    const bool pass_timing = getenv("MPLX_PLAN_PASS_TIMING") != nullptr;  // diagnostic: four time-stamp reads per expansion
    uint64_t pass_tsc[5] = {0, 0, 0, 0, 0};  // ([4]: the record / g part of pass 1)
    int expand_iteration = 0;
    bool reached = false;
    uint32_t curr = 0;
    double sc[14], hs[14];
    for (;;) {
      expand_iteration++;
      curr = pq.top().idx;
      pq.pop();
      cold[curr].closed = true;
      const auto t_s0 = clk::now();
      SuccView sv;
      if (int rc = successors(curr, &sv)) return rc;
      t_succ += ms_since(t_s0);
      // ---- relaxation of the node's successors (graph_search.h:78-141), in four passes over the list.  Written as one
      // loop, every edge costs three branches nobody can predict (blocked? new node? better path?) and each miss throws
      // away the table probes in flight behind it; split up, the per-edge work is branch-free and the rare cases
      // (one edge in five creates a node, one in four improves one) run in loops of their own.  Order is kept where the
      // reference's results depend on it: nodes are numbered, predecessors recorded and heap operations issued in
      // ascending successor index, exactly as the single loop did.
      const int n_succ = sv.m;
      const double g_curr = cold[curr].g;
      const double *c_curr = coord_of(curr);
      const uint64_t tp0 = pass_timing ? __builtin_ia32_rdtsc() : 0;
      // pass 0: the finite edges (graph_search.h:81 skips the blocked ones)
      const int nf = finite_pass(sv.cost, n_succ, r_fin.data());
      const uint64_t *keys = sv.keys;
      if (!keys) {  // a provider without lattice hashes: hash the states it delivered (or their host evaluation)
        for (int j = 0; j < nf; j++) {
          const int s = r_fin[(size_t)j];
          if (sv.state) for (int r = 0; r < f; r++) sc[r] = sv.state[(int64_t)r * sv.fs + (int64_t)s * sv.es];
          else forward_state(dim, control, c_curr, &U[(size_t)sv.act[s] * udim], dt, sc);  // (lists that rode along without hashes)
          v_keys[(size_t)s] = lattice_hash(dim, control, sc);
        }
        keys = v_keys.data();
      }
      const uint64_t tp1 = pass_timing ? __builtin_ia32_rdtsc() : 0;
      // pass 1: find or claim the child's slot, record the edge, lower g -- no branch depends on the data
      hm.reserve((size_t)nf);
      if (cold.capacity() < cold.size() + (size_t)nf) cold.reserve(std::max(cold.capacity() * 2, cold.size() + (size_t)nf));
      const size_t rec0 = preds.size();
      if (rec0 + (size_t)nf > (size_t)0x7fffffff) return -2;  // (32-bit record indices)
      PredRec *const P = preds.grow((size_t)nf);
      int32_t *const A = pred_act.grow((size_t)nf);
      const Cold *const cold0 = cold.data();
      // 1a: the probes.  The only data-dependent branches of the relaxation are the ends of the probe sequences; a
      // mispredicted one discards nothing but other probes.
      constexpr int kAhead = 24;
      for (int j = 0; j < nf && j < kAhead; j++) hm.prefetch(keys[r_fin[(size_t)j]]);
      int n_new = 0, n_imp = 0;
      for (int j = 0; j < nf; j++) {
        if (j + kAhead < nf) hm.prefetch(keys[r_fin[(size_t)(j + kAhead)]]);
        bool fresh;
        r_slot[(size_t)j] = hm.insert(keys[r_fin[(size_t)j]], &fresh);
        r_new[(size_t)n_new] = j;
        n_new += fresh;
      }
      const uint64_t tp1b = pass_timing ? __builtin_ia32_rdtsc() : 0;
      // 1b: the edge into the child's predecessor list, the child's g lowered -- straight-line code
      for (int j = 0; j < nf; j++) {
        const int s = r_fin[(size_t)j];
        Slot *sl = r_slot[(size_t)j];
        const double c_s = sv.cost[s];
        // (the records are written once and read again only by recoverTraj, for a few dozen nodes: streamed past the
        // caches, 20 bytes per edge that would otherwise cost a line fetch for ownership each)
        stream_record(&P[j], c_s, curr, sl->pred_head);
        stream_i32(&A[j], sv.act[s]);
        sl->pred_head = (int32_t)(rec0 + (size_t)j);
        const double tentative = g_curr + c_s;
        const bool better = tentative < sl->g;
        r_tent[(size_t)j] = tentative;
        r_imp[(size_t)n_imp] = j;
        n_imp += better;
        sl->g = better ? tentative : sl->g;
        // the improved child's cold record is needed in pass 3 (a new child's is about to be written anyway)
        __builtin_prefetch(better && sl->idx != NodeTable::kNoIndex ? (const void *)(cold0 + sl->idx) : (const void *)sl);
      }
      last.relaxed += nf;
      last.improved += n_imp;
      const uint64_t tp2 = pass_timing ? __builtin_ia32_rdtsc() : 0;
      // pass 2: the new nodes, numbered in successor order
      for (int q = 0; q < n_new; q++) {
        const int j = r_new[(size_t)q], s = r_fin[(size_t)j];
        Slot *sl = r_slot[(size_t)j];
        const uint64_t key = keys[s];
        sl->idx = (uint32_t)cold.size();
        cold.emplace_back();
        Cold &nd = cold.back();
        nd.key = key;
        nd.parent = curr;
        nd.action = sv.act[s];
        bool have_sc = false;
        if (sv.state) {
          // the provider's own rows are the node's state (and, as a test hook, are compared with the host evaluation)
          for (int r = 0; r < f; r++) sc[r] = sv.state[(int64_t)r * sv.fs + (int64_t)s * sv.es];
          have_sc = true;
          if (check_states) {
            forward_state(dim, control, c_curr, &U[(size_t)sv.act[s] * udim], dt, hs);
            if (check_perturb >= 0 && checked_states++ == check_perturb) {
              uint64_t b;
              std::memcpy(&b, &hs[0], 8);
              b ^= 1ull;
              std::memcpy(&hs[0], &b, 8);
            }
            if (std::memcmp(hs, sc, sizeof(double) * (size_t)f) != 0) last.state_mismatches++;
          }
          nd.coord = (int32_t)(coords.size() / (size_t)f);
          coords.insert(coords.end(), sc, sc + f);
          c_curr = &coords[(size_t)cold[curr].coord * (size_t)f];  // (coords may have moved)
        }
        if (eps == 0 || key == goal_key) nd.h = 0;
        else if (sv.heur && !has_prior()) { nd.h = sv.heur[s]; last.heur_from_provider++; }  // computed where the successor was made
        else {
          if (!have_sc) forward_pos(dim, control, c_curr, &U[(size_t)sv.act[s] * udim], dt, sc);
          nd.h = heur_at(sc, c_curr[4 * dim + 1] + dt, goal);  // (the successor's time: env_map.h:161)
        }
      }
      const uint64_t tp3 = pass_timing ? __builtin_ia32_rdtsc() : 0;
      // pass 3: the improved children into the open list (graph_search.h:108-141), in successor order
      for (int q = 0; q < n_imp; q++) {
        const int j = r_imp[(size_t)q];
        const Slot *sl = r_slot[(size_t)j];
        Cold &ch = cold[sl->idx];
        const double tentative = r_tent[(size_t)j];
        ch.g = tentative;
        const double fval = tentative + eps * ch.h;
        if (ch.opened && !ch.closed) {
          pq.increase(ch.handle, fval, tentative);
        } else {
          // (a closed node whose older entry is still in the heap: two entries of one node from here on)
          if (ch.handle >= 0 && pq.position(ch.handle) >= 0) pq.live = true;
          ch.handle = pq.push(fval, tentative, sl->idx);
          ch.opened = true;
          last.pushes++;
        }
      }
      if (pass_timing) {
        const uint64_t tp4 = __builtin_ia32_rdtsc();
        pass_tsc[0] += tp1 - tp0; pass_tsc[1] += tp2 - tp1; pass_tsc[2] += tp3 - tp2; pass_tsc[3] += tp4 - tp3; pass_tsc[4] += tp2 - tp1b;
      }
      if (is_goal(coord_of(curr), goal)) { reached = true; break; }
      if (max_expand > 0 && expand_iteration >= max_expand) break;
      if (pq.empty()) break;
    }
    last.expansions = expand_iteration;
    last.nodes = (int)hm.size();
    for (const Cold &nd : cold)
      if (nd.closed) last.closed++;
    // PlannerBase::getOpenSet walks the heap (planner_base.h:77-81): a closed node that was pushed again counts
    last.opened = (int)pq.size();
#if defined(__x86_64__)
    _mm_sfence();  // (the streamed predecessor records are globally visible before anything reads them back)
#endif
    const auto t_r0 = clk::now();
    if (reached && recover(curr, start)) { last.ok = true; last.cost = cold[curr].g; }
    last.t_recover = ms_since(t_r0);
    last.t_total = ms_since(t_plan0);
    last.t_provider = t_provider;
    last.t_fill = t_fill;
    last.t_pick = t_pick;
    last.t_relax = last.t_total - t_succ - last.t_recover;
    if (pass_timing) {
      const double tot = (double)(pass_tsc[0] + pass_tsc[1] + pass_tsc[2] + pass_tsc[3]);
      fprintf(stderr, "[host_planner] relaxation passes (share of their sum): finite edges %.3f, table probes %.3f, records + g %.3f, new nodes %.3f, heap %.3f; sum = %.0f Mcycles (tsc)\n",
              pass_tsc[0] / tot, (pass_tsc[1] - pass_tsc[4]) / tot, pass_tsc[4] / tot, pass_tsc[2] / tot, pass_tsc[3] / tot, tot * 1e-6);
    }
    if (getenv("MPLX_PLAN_TIMING"))
      fprintf(stderr, "[host_planner] %.1f ms: successors() %.1f (provider %.1f, cache fill %.1f, candidate pick %.1f), relaxation + heap %.1f, "
              "recover %.2f; %lld edges relaxed, %lld improved, %lld pushes, %zu nodes, %lld states materialised\n",
              last.t_total, t_succ, t_provider, t_fill, t_pick, last.t_relax, last.t_recover, (long long)last.relaxed,
              (long long)last.improved, (long long)last.pushes, cold.size(), (long long)last.materialised);
    return 0;
  }

 private:
  double t_succ = 0, t_provider = 0, t_fill = 0, t_pick = 0;  // MPLX_PLAN_TIMING diagnostics
  int64_t checked_states = 0;
  std::vector<int32_t> r_fin, r_new, r_imp;  // scratch of the relaxation passes (one successor list)
  std::vector<double> r_tent;
  std::vector<Slot *> r_slot;
  static void stream_record(PredRec *p, double cost, uint32_t parent, int32_t next) {
#if defined(__x86_64__)
    static_assert(sizeof(PredRec) == 16, "one 16-byte store");
    uint64_t lo, hi = (uint64_t)parent | ((uint64_t)(uint32_t)next << 32);
    std::memcpy(&lo, &cost, 8);
    _mm_stream_si128((__m128i *)p, _mm_set_epi64x((long long)hi, (long long)lo));
#else
    *p = PredRec{cost, parent, next};
#endif
  }
  static void stream_i32(int32_t *p, int32_t v) {
#if defined(__x86_64__)
    _mm_stream_si32(p, v);
#else
    *p = v;
#endif
  }
  // pass 0 of the relaxation: the indices of the finite entries of a cost list, in order.  One entry in three is
  // blocked, at random: a compare-and-compress per 8 entries where the host has AVX-512 (every EPYC an MI355X sits
  // in), a branch-free scalar loop elsewhere.
  static int finite_scalar(const double *cost, int m, int32_t *fin) {
    int nf = 0;
    for (int s = 0; s < m; s++) {
      fin[nf] = s;
      nf += std::fabs(cost[s]) != kInf;  // (!isinf)
    }
    return nf;
  }
#if defined(__x86_64__)
  __attribute__((target("avx512f,avx512vl"))) static int finite_avx512(const double *cost, int m, int32_t *fin) {
    const __m512d inf = _mm512_set1_pd(kInf);
    __m256i idx = _mm256_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7);
    const __m256i eight = _mm256_set1_epi32(8);
    int nf = 0, s = 0;
    for (; s + 8 <= m; s += 8) {
      const __m512d c = _mm512_abs_pd(_mm512_loadu_pd(cost + s));
      const __mmask8 k = _mm512_cmp_pd_mask(c, inf, _CMP_NEQ_UQ);
      _mm256_mask_compressstoreu_epi32(fin + nf, k, idx);
      nf += __builtin_popcount((unsigned)k);
      idx = _mm256_add_epi32(idx, eight);
    }
    for (; s < m; s++) {
      fin[nf] = s;
      nf += std::fabs(cost[s]) != kInf;
    }
    return nf;
  }
#endif
  int finite_pass(const double *cost, int m, int32_t *fin) const {
#if defined(__x86_64__)
    static const bool wide = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl") && !getenv("MPLX_PLAN_NO_AVX512");
    if (wide) return finite_avx512(cost, m, fin);
#endif
    return finite_scalar(cost, m, fin);
  }

  // ---- successor lists between the launch that produced them and the pop that consumes them.  The provider's landing
  // buffer of the latest launch stays valid until the next one: nodes expanded before that (9 in 10) are read in
  // place; the others are moved to buffers of their own just before the next launch (Planner::keep).
  struct CacheRec {
    int32_t slot = -1;       // index in the landing buffer of launch `batch` ...
    uint32_t batch = 0;
    char *blob = nullptr;    // ... or a recycled buffer [cost m][hash m]?[state (4D+2) x m]?[action m]
    int32_t m = 0;
    bool has_state = false, has_keys = false;
  };
  std::vector<CacheRec> caches;
  std::vector<int32_t> free_caches;
  int32_t take_cache() {
    if (!free_caches.empty()) { const int32_t c = free_caches.back(); free_caches.pop_back(); caches[(size_t)c] = CacheRec(); return c; }
    caches.emplace_back();
    return (int32_t)caches.size() - 1;
  }
  std::vector<int32_t> own_cnt, own_act;  // landing buffer of the providers that fill caller arrays
  std::vector<int64_t> own_offs;
  std::vector<double> own_cost, own_state;
  std::vector<uint64_t> own_hash;
  std::vector<uint8_t> own_status;
  std::vector<double> v_succ, v_cost;  // the current expansion's successors (single provider, speculation store)
  std::vector<int32_t> v_act;
  std::vector<uint64_t> v_keys;
  char *cur_blob = nullptr;            // ... or the packed lists of the node being expanded
  PackedView cur_view;
  int64_t cur_view_n = 0;  // nodes (slots) of the landing buffer
  uint32_t cur_batch = 0, pick_counter = 0;
  std::vector<uint32_t> cur_group;
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
  // Buffers of the kept lists: fixed capacity (a full list), recycled when their node has been expanded and from plan
  // to plan, so that steady state touches no fresh pages (a fresh 40-KB allocation per node cost more than the copy).
  std::vector<std::unique_ptr<char[]>> all_blobs;
  std::vector<char *> free_blobs;
  size_t blob_bytes = 0;
  void blob_bytes_check() {  // a control table or state size that changed since the buffers were made: start over
    const size_t need = (size_t)nU * (size_t)(8 + 8 + 8 * F() + 4);
    if (need != blob_bytes) { all_blobs.clear(); free_blobs.clear(); blob_bytes = need; }
  }
  char *take_blob() {
    if (!free_blobs.empty()) { char *b = free_blobs.back(); free_blobs.pop_back(); return b; }
    all_blobs.emplace_back(new char[blob_bytes]);
    return all_blobs.back().get();
  }
  void keep(CacheRec &c) {  // lists out of the landing buffer into a recycled buffer
    const int f = F();
    const size_t m = (size_t)cur_view.count[c.slot], o = (size_t)cur_view.offs[c.slot];
    c.m = (int32_t)m;
    if (!c.blob) c.blob = take_blob();
    char *b = c.blob;
    std::memcpy(b, cur_view.cost + o, m * 8);
    b += m * 8;
    c.has_keys = cur_view.hash != nullptr;
    if (c.has_keys) { std::memcpy(b, cur_view.hash + o, m * 8); b += m * 8; }
    c.has_state = cur_view.state != nullptr;
    if (c.has_state)
      for (int r = 0; r < f; r++, b += m * 8) std::memcpy(b, cur_view.state + (size_t)r * cur_view.total + o, m * 8);
    std::memcpy(b, cur_view.action + o, m * 4);
    c.slot = -1;
  }

  // One get_succ, possibly served from / filling the batch cache.
  int successors(uint32_t curr, SuccView *v) {
    const int f = F();
    if (single && (batch <= 1 || (!batched && !lists && !packed))) {  // the reference's loop: one node, one call
      last.device_launches++;
      last.pairs += nU;
      int32_t m = 0;
      if (int rc = single(user, coord_of(curr), v_succ.data(), v_cost.data(), v_act.data(), &m)) return rc;
      *v = SuccView{m, v_cost.data(), nullptr, v_act.data(), v_succ.data(), 1, f};
      return 0;
    }
    if (batch <= 1 || (!batched && !lists && !packed)) {
      if (int rc = run_batch({curr})) return rc;
      return fetch(curr, v);
    }
    const int n_spec_parents = spec_now;
    if (cold[curr].cache < 0 && n_spec_parents > 0 && spec.cap) {
      const int e = spec.find(cold[curr].key);
      if (e >= 0 && std::memcmp(&spec.coord[(size_t)e * f], coord_of(curr), sizeof(double) * (size_t)f) == 0) {
        const int32_t m = spec.m[(size_t)e];
        std::copy(&spec.cost[(size_t)e * nU], &spec.cost[(size_t)e * nU] + m, v_cost.begin());
        std::copy(&spec.act[(size_t)e * nU], &spec.act[(size_t)e * nU] + m, v_act.begin());
        if (spec.has_keys) std::copy(&spec.keys[(size_t)e * nU], &spec.keys[(size_t)e * nU] + m, v_keys.begin());
        *v = SuccView{m, v_cost.data(), spec.has_keys ? v_keys.data() : nullptr, v_act.data(), nullptr, 1, f};
        last.spec_hits++;
        return 0;
      }
    }
    if (cold[curr].cache < 0) {
      const auto t_p0 = std::chrono::steady_clock::now();
      // the popped node plus the best open nodes that have no list yet: best-first walk of the heap
      // array (k smallest of a binary heap with an auxiliary heap of positions), O(k log k) whatever
      // the size of the open list
      std::vector<uint32_t> group{curr};
      cold[curr].pick_stamp = ++pick_counter;
      const size_t want = (size_t)batch - 1;
      const auto &h = pq.items();
      auto worse = [&](int a, int b) {  // max-heap on "better", so top() is the best position
        const OpenList::Item &x = h[(size_t)a], &y = h[(size_t)b];
        if (x.f != y.f) return x.f > y.f;
        return pq.tie(x) > pq.tie(y);
      };
      // (the walk stops after ~2 x batch heap positions: what lies deeper is not popped soon enough to be worth a
      // slot, and on a small problem most of the top of the heap already holds its lists -- walking 16 x batch
      // positions per launch, the first version, was 3.0 of C1's 5.3 ms; the position heap lives in a member vector)
      std::vector<int> &aux = aux_buf;
      aux.clear();
      if (!h.empty()) aux.push_back(0);
      // (a visited position reads its node's cold record -- has it lists already? -- a cache miss each at a million
      // nodes: requested when the position enters the walk, two to three visits before it is looked at)
      auto push = [&](int pos) {
        __builtin_prefetch(&cold[h[(size_t)pos].idx]);
        aux.push_back(pos);
        std::push_heap(aux.begin(), aux.end(), worse);
      };
      size_t visited = 0;
      while (!aux.empty() && group.size() - 1 < want && visited < 2 * want + 16) {
        std::pop_heap(aux.begin(), aux.end(), worse);
        const int i = aux.back();
        aux.pop_back();
        visited++;
        Cold &cand = cold[h[(size_t)i].idx];
        if (cand.cache < 0 && cand.pick_stamp != pick_counter) {
          cand.pick_stamp = pick_counter;
          group.push_back(h[(size_t)i].idx);
        }
        if (2 * i + 1 < (int)h.size()) push(2 * i + 1);
        if (2 * i + 2 < (int)h.size()) push(2 * i + 2);
      }
      // children of the first nodes of the launch (the popped one, then the best of the open list)
      spec_states.clear();
      spec_keys.clear();
      if (n_spec_parents > 0) {
        if (spec.cap == 0 || spec.nU != nU || spec.F != f) spec.reset(nU, f, 2048);
        double cs[14], par[14];
        // children of `par`: into the launch -- as nodes when they exist and wait for lists, as bare states otherwise
        auto children = [&](uint32_t pidx) {
          std::memcpy(par, coord_of(pidx), sizeof(double) * (size_t)f);  // (coord_of of a child may move `coords`)
          for (int i = 0; i < nU; i++) {
            forward_state(dim, control, par, &U[(size_t)i * udim], dt, cs);
            const uint64_t key = lattice_hash(dim, control, cs);
            if (const Slot *sl = hm.find(key)) {
              // a node already: closed or served -> nothing to do; open without lists -> an ordinary member of the launch
              Cold &ex = cold[sl->idx];
              if (!ex.closed && ex.cache < 0 && ex.pick_stamp != pick_counter && group.size() < (size_t)batch + 64) {
                ex.pick_stamp = pick_counter;
                group.push_back(sl->idx);
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
        for (size_t gi = 0; gi < n_par; gi++) children(group[gi]);
      }
      t_pick += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_p0).count();
      if (int rc = run_batch(group)) return rc;
    }
    return fetch(curr, v);
  }

  int run_batch(const std::vector<uint32_t> &group) {
    using clk = std::chrono::steady_clock;
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const int f = F();
    // (speculated states only ever ride along into a store of their own shape: a control table or state size that
    // changed since they were assembled would index the store's arrays with the wrong strides)
    if (!spec_keys.empty() && (spec.cap == 0 || spec.nU != nU || spec.F != f)) { spec_keys.clear(); spec_states.clear(); }
    const int64_t ng = (int64_t)group.size(), ns = (int64_t)spec_keys.size();
    const int64_t n = ng + ns;  // the launch: the nodes of the group, then the speculated states
    std::vector<double> &nodes = nodes_buf;
    nodes.resize((size_t)f * n);
    for (int64_t k = 0; k < ng; k++) {
      const double *c = coord_of(group[(size_t)k]);
      for (int r = 0; r < f; r++) nodes[(size_t)r * n + k] = c[r];
    }
    for (int64_t k = 0; k < ns; k++)
      for (int r = 0; r < f; r++) nodes[(size_t)r * n + ng + k] = spec_states[(size_t)k * f + r];
    const int64_t slots = n * nU;
    last.device_launches++;
    last.pairs += slots;
    const auto t_f0 = clk::now();
    for (uint32_t p : cur_group) {  // what the previous launch delivered and the search has not consumed yet
      const int32_t ci = cold[p].cache;  // (a re-opened node is closed AND waiting)
      if (ci >= 0 && caches[(size_t)ci].slot >= 0 && caches[(size_t)ci].batch == cur_batch) keep(caches[(size_t)ci]);
    }
    const auto t_l0 = clk::now();
    t_fill += ms(t_f0, t_l0);
    cur_group.clear();
    cur_batch++;
    if (packed) {
      if (int rc = packed(user, nodes.data(), n, &cur_view)) return rc;
      t_provider += ms(t_l0, clk::now());
    } else if (lists) {
      // compact per-node lists with the device's lattice hashes, in caller arrays: node k owns [k nU, k nU + count[k])
      own_cnt.resize((size_t)n);
      own_act.resize((size_t)slots);
      own_cost.resize((size_t)slots);
      own_hash.resize((size_t)slots);
      own_state.resize((size_t)f * slots);
      own_offs.resize((size_t)n + 1);
      if (int rc = lists(user, nodes.data(), n, own_cnt.data(), own_act.data(), own_cost.data(), own_hash.data(), own_state.data()))
        return rc;
      t_provider += ms(t_l0, clk::now());
      for (int64_t k = 0; k <= n; k++) own_offs[(size_t)k] = k * nU;
      cur_view = PackedView{slots, own_cnt.data(), own_offs.data(), own_cost.data(), own_hash.data(), own_act.data(), own_state.data()};
    } else {
      // dense slots: the emitted ones (finite or blocked), in control order, moved to the front of each node's range
      own_status.resize((size_t)slots);
      own_cnt.resize((size_t)n);
      own_act.resize((size_t)slots);
      own_cost.resize((size_t)slots);
      own_state.resize((size_t)f * slots);
      own_offs.resize((size_t)n + 1);
      if (int rc = batched(user, nodes.data(), n, own_status.data(), own_cost.data(), own_state.data())) return rc;
      const auto t_b1 = clk::now();
      t_provider += ms(t_l0, t_b1);
      for (int64_t k = 0; k < n; k++) {
        int32_t m = 0;
        const int64_t o = k * nU;
        for (int i = 0; i < nU; i++) {
          const uint8_t st = own_status[(size_t)(o + i)];
          if (st != 1 && st != 2) continue;
          if (m != i) {
            own_cost[(size_t)(o + m)] = own_cost[(size_t)(o + i)];
            for (int r = 0; r < f; r++) own_state[(size_t)r * slots + o + m] = own_state[(size_t)r * slots + o + i];
          }
          own_act[(size_t)(o + m)] = i;
          m++;
        }
        own_cnt[(size_t)k] = m;
        own_offs[(size_t)k] = o;
      }
      own_offs[(size_t)n] = slots;
      cur_view = PackedView{slots, own_cnt.data(), own_offs.data(), own_cost.data(), nullptr, own_act.data(), own_state.data()};
      t_fill += ms(t_b1, clk::now());
    }
    cur_view_n = n;
    for (int64_t k = 0; k < ng; k++) {
      Cold &nd = cold[group[(size_t)k]];
      if (nd.cache < 0) nd.cache = take_cache();
      CacheRec &c = caches[(size_t)nd.cache];
      c.slot = (int32_t)k;
      c.batch = cur_batch;
    }
    // lists of the speculated states into the store (entry e <- column ng + k of the launch)
    if (ns) spec.has_keys = cur_view.hash != nullptr;
    for (int64_t k = 0; k < ns; k++) {
      const size_t o = (size_t)cur_view.offs[ng + k];
      const int32_t m = cur_view.count[ng + k];
      const int e = spec.insert(spec_keys[(size_t)k]);
      std::copy(&spec_states[(size_t)k * f], &spec_states[(size_t)k * f] + f, &spec.coord[(size_t)e * f]);
      spec.m[(size_t)e] = m;
      std::copy(cur_view.cost + o, cur_view.cost + o + m, &spec.cost[(size_t)e * nU]);
      std::copy(cur_view.action + o, cur_view.action + o + m, &spec.act[(size_t)e * nU]);
      if (cur_view.hash) std::copy(cur_view.hash + o, cur_view.hash + o + m, &spec.keys[(size_t)e * nU]);
    }
    cur_group.assign(group.begin(), group.end());
    return 0;
  }

  // Hands the cached lists of `n` to the relaxation loop and forgets them: the reference expands a node again
  // when a cheaper path re-opens it after it was closed (graph_search.h:108-141 pushes it a second time; that
  // happens with an inconsistent heuristic, i.e. eps > 1 or the default v_max <= 0), so the cache entry must not
  // outlive its one use -- a re-popped node goes through run_batch again (get_succ is pure: same lists).
  int fetch(uint32_t n, SuccView *v) {
    Cold &nd = cold[n];
    if (nd.cache < 0) return -1;
    const int32_t ci = nd.cache;
    CacheRec &c = caches[(size_t)ci];
    nd.cache = -1;
    if (cur_blob) { free_blobs.push_back(cur_blob); cur_blob = nullptr; }  // the previous expansion's lists are done with
    view_of(c, v);
    if (c.slot >= 0 && c.batch == cur_batch) {
      if (c.blob) free_blobs.push_back(c.blob);
      // The lists of a launch lie in memory the device wrote: the first read of every line is a DRAM miss, and three
      // short streams per node (cost, hash, action: a page or two each) are over before the hardware prefetcher has
      // locked on.  The launch's nodes were picked best first, so they are popped roughly in slot order: ask for the
      // lists two slots further on now.
      for (int d = 1; d <= 2; d++) prefetch_lists(c.slot + d);
    } else {
      cur_blob = c.blob;  // the planner takes the node's lists over for the duration of this expansion
    }
    c = CacheRec();
    free_caches.push_back(ci);
    return 0;
  }
  void prefetch_lists(int64_t slot) const {
    if (slot >= cur_view_n) return;
    const size_t o = (size_t)cur_view.offs[slot];
    const size_t m = (size_t)cur_view.count[slot];
    const char *c = (const char *)(cur_view.cost + o), *h = (const char *)(cur_view.hash + o), *a = (const char *)(cur_view.action + o);
    for (size_t b = 0; b < m * 8; b += 64) __builtin_prefetch(c + b);
    if (cur_view.hash) for (size_t b = 0; b < m * 8; b += 64) __builtin_prefetch(h + b);
    for (size_t b = 0; b < m * 4; b += 64) __builtin_prefetch(a + b);
    if (cur_view.heur) {
      const char *hr = (const char *)(cur_view.heur + o);
      for (size_t b = 0; b < m * 8; b += 64) __builtin_prefetch(hr + b);
    }
  }
  void view_of(const CacheRec &c, SuccView *v) const {
    const int f = F();
    if (c.slot >= 0 && c.batch == cur_batch) {
      // still in the landing buffer of the latest launch: read in place
      const size_t o = (size_t)cur_view.offs[c.slot];
      *v = SuccView{cur_view.count[c.slot], cur_view.cost + o, cur_view.hash ? cur_view.hash + o : nullptr, cur_view.action + o,
                    cur_view.state ? cur_view.state + o : nullptr, cur_view.total, 1, cur_view.heur ? cur_view.heur + o : nullptr};
      return;
    }
    const size_t m = (size_t)c.m;
    const char *b = c.blob;
    const double *cost = (const double *)b;
    b += m * 8;
    const uint64_t *keys = nullptr;
    if (c.has_keys) { keys = (const uint64_t *)b; b += m * 8; }
    const double *state = nullptr;
    if (c.has_state) { state = (const double *)b; b += m * 8 * (size_t)f; }
    *v = SuccView{(int32_t)m, cost, keys, (const int32_t *)b, state, (int64_t)m, 1};
  }

  // GraphSearch::recoverTraj, graph_search.h:369-455
  bool recover(uint32_t curr, const double *start) {
    const int f = F();
    const uint64_t start_key = lattice_hash(dim, control, start);
    {
      const double *c = coord_of(curr);
      last.traj_end.assign(c, c + f);
    }
    std::vector<uint32_t> from;
    std::vector<int32_t> acts, recs;
    bool found = false;
    for (;;) {
      const Slot *sl = hm.find(cold[curr].key);
      if (!sl || sl->pred_head < 0) break;
      recs.clear();
      for (int32_t i = sl->pred_head; i >= 0; i = preds[(size_t)i].next) recs.push_back(i);
      std::reverse(recs.begin(), recs.end());  // insertion order, as the reference's pred_coord vector holds them
      int min_id = -1;
      double min_rhs = kInf, min_g = kInf;
      for (int32_t i : recs) {
        const PredRec &pr = preds[(size_t)i];
        const double pg = cold[pr.parent].g;
        const double v = pg + pr.cost;
        if (min_rhs > v) { min_rhs = v; min_g = pg; min_id = (int)i; }
        else if (!std::isinf(pr.cost) && min_rhs == v) {
          if (min_g < pg) { min_g = pg; min_id = (int)i; }
        }
      }
      if (min_id < 0) break;
      const int a = pred_act[(size_t)min_id];
      curr = preds[(size_t)min_id].parent;
      from.push_back(curr);
      acts.push_back(a);
      if (cold[curr].key == start_key) { found = true; break; }
    }
    if (!found) return false;
    std::reverse(from.begin(), from.end());
    std::reverse(acts.begin(), acts.end());
    // Trajectory(prs): total time and efforts (trajectory.h:52-57, 250-254)
    const int K = (control & 8) ? 4 : (control & 4) ? 3 : (control & 2) ? 2 : 1;
    for (size_t s = 0; s < from.size(); s++) {
      const double *nd = coord_of(from[s]);
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
