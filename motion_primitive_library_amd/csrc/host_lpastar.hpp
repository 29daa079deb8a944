// host_lpastar.hpp -- Lifelong Planning A* of the engine's planner: incremental re-planning around the device get_succ.
//
// A restatement, on the plain types of host_planner.hpp, of
//   GraphSearch::LPAstar                                  reference include/mpl_planner/common/graph_search.h:194-365
//   StateSpace::updateNode / increaseCost / decreaseCost  reference include/mpl_planner/common/state_space.h:197-270
//   StateSpace::getSubStateSpace                          reference include/mpl_planner/common/state_space.h:116-195
//   PlannerBase::plan with setLPAstar(true)               reference include/mpl_planner/common/planner_base.h:170-176, 293-304
//   MapPlanner::getLinkedNodes / updateBlockedNodes / updateClearedNodes
//                                                         reference src/mpl_planner/map_planner.cpp:125-185
// The state space outlives plan(): every node keeps its predecessor AND successor lists (edge costs included), a map
// edit is translated into edge-cost changes through the voxel -> edge table and the next plan() repairs only what
// those changes made inconsistent.
//
// What runs on the device: get_succ of the nodes a plan expands for the first time (batched: the popped node with the
// best open nodes that have no list yet, as in the A* planner), and the edge work of the table and of
// updateClearedNodes -- one mplx_check_edges call for ALL stored edges (cells each edge passes through, map_planner.cpp:
// 139-152) or for all affected ones (is_free + intrinsic cost, state_space.h:236-240), where the reference walks them
// one Primitive at a time.
//
// Order is part of the contract: the reference's results depend on the iteration order of its hash map in two places
// (getSubStateSpace pushes the open nodes in map order; getLinkedNodes lists the edges of a cell in map order, and
// increaseCost / decreaseCost repair them in that order).  The node map here is a std::unordered_map keyed by the
// 64-bit lattice hash -- the same container with the same hash codes and the same insertion sequence as the
// reference compiled against the stand-in Boost of oracle/stub_include (boost::unordered_map is an alias of
// std::unordered_map there), so the two walk their maps in the same order.  The heap follows that stand-in's
// d_ary_heap as well (erase = move the last entry into the hole, then sift up or down).
#ifndef MPLX_HOST_LPASTAR_HPP
#define MPLX_HOST_LPASTAR_HPP

#include "host_planner.hpp"

#include <memory>
#include <unordered_map>
#include <utility>

namespace mplx {
namespace host {

// batched edge re-validation with the shape of mplx_check_edges (include/mplx.h): parents field-major [4D+2][n]
typedef int (*edges_fn)(void *user, const double *parents, const int32_t *actions, int64_t n, uint8_t *free_flag, double *cost,
                        int32_t *cells, int32_t *cell_count, int32_t cell_cap);

class LpaPlanner {
 public:
  struct LNode {  // State<Coord>, state_space.h:37-70
    double coord[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t key = 0;
    double g = kInf, rhs = kInf, h = kInf;
    bool opened = false, closed = false;
    int handle = -1;  // heapkey
    std::vector<LNode *> pred;  // pred_coord (the node of that coordinate), pred_action_cost, pred_action_id
    std::vector<double> pred_cost;
    std::vector<int32_t> pred_act;
    std::vector<LNode *> succ;  // succ_coord, succ_action_cost, succ_action_id
    std::vector<double> succ_cost;
    std::vector<int32_t> succ_act;
  };
  typedef std::unordered_map<uint64_t, LNode *> NodeMapT;  // hm_ (state_space.h:78)

  // d_ary_heap<pair<fval, node>, mutable, arity 2, compare_pair> of the reference with the erase() LPA* needs
  class Heap {
   public:
    struct Item { double f; LNode *n; int handle; };
    bool empty() const { return q_.empty(); }
    size_t size() const { return q_.size(); }
    const Item &top() const { return q_.front(); }
    const std::vector<Item> &items() const { return q_; }
    void clear() { q_.clear(); pos_.clear(); }
    int push(double f, LNode *n) {
      const int h = (int)pos_.size();
      pos_.push_back((int)q_.size());
      q_.push_back({f, n, h});
      up((int)q_.size() - 1);
      return h;
    }
    void pop() { remove_at(0); }
    void erase(int handle) { remove_at(pos_[(size_t)handle]); }

   private:
    static bool less(const Item &a, const Item &b) {  // compare_pair, state_space.h:16-34
      if (a.f == b.f) return std::min(a.n->g, a.n->rhs) > std::min(b.n->g, b.n->rhs);
      return a.f > b.f;
    }
    void place(int i, const Item &it) { q_[(size_t)i] = it; pos_[(size_t)it.handle] = i; }
    void swap_at(int i, int j) {
      const Item a = q_[(size_t)i], b = q_[(size_t)j];
      place(i, b);
      place(j, a);
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
    void remove_at(int i) {
      const int last = (int)q_.size() - 1;
      pos_[(size_t)q_[(size_t)i].handle] = -1;
      if (i != last) {
        place(i, q_[(size_t)last]);
        q_.pop_back();
        if (i > 0 && less(q_[(size_t)(i - 1) / 2], q_[(size_t)i])) up(i); else down(i);
      } else {
        q_.pop_back();
      }
    }
    std::vector<Item> q_;
    std::vector<int> pos_;
  };

  // ---- what the owner (planner_capi.cpp) wires up
  Planner *cfg = nullptr;   // dimension, control, dt, w, v_max, eps, tolerances, U, grid, the successor providers
  edges_fn edges = nullptr;
  void *edges_user = nullptr;

  PlanResult last;
  bool initialized() const { return ready_; }
  void reset() {  // PlannerBase::reset
    hm_.clear();
    pool_.clear();
    pq_.clear();
    best_child_.clear();
    lhm_.clear();
    ready_ = false;
    start_g_ = start_rhs_ = start_t_ = 0;
    expand_iteration_ = 0;
    n_closed_ = 0;
    spec_.clear();
  }

  // PlannerBase::plan with use_lpastar_ (planner_base.h:275-325)
  int plan(const double *start, const double *goal_given) {
    Planner &P = *cfg;
    const double *goal = P.effective_goal(goal_given);  // (a prior trajectory's end, env_base.h:295-298)
    last = PlanResult();
    const auto t0 = std::chrono::steady_clock::now();
    int pn[3];
    P.grid.to_cell(start, pn);
    if (!P.grid.is_free(pn)) return 0;  // "start is not free"
    if (!ready_) { reset(); ready_ = true; }
    spec_.clear();  // (lists that rode along in an earlier plan: the map may have been edited since)
    const int rc = lpastar(start, goal);
    last.t_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    last.nodes = (int)hm_.size();
    last.closed = n_closed_;  // PlannerBase::getCloseSet's size, kept up to date by set_closed()
    last.opened = (int)pq_.size();
    return rc;
  }

  // MapPlanner::getLinkedNodes (map_planner.cpp:125-157): the voxel -> edge table; returns the number of linked points
  // and (optionally) their positions, D doubles each
  int linked_nodes(std::vector<double> *points, int64_t *n_points) {
    Planner &P = *cfg;
    const int D = P.dim, f = P.F();
    lhm_.clear();
    // every stored edge, in the map's iteration order and each node's predecessor order
    std::vector<LNode *> owner;
    std::vector<int32_t> slot, acts;
    std::vector<const LNode *> par;
    for (const auto &kv : hm_) {
      LNode *nd = kv.second;
      if (!nd) continue;
      for (size_t i = 0; i < nd->pred.size(); i++) {
        owner.push_back(nd);
        slot.push_back((int32_t)i);
        par.push_back(nd->pred[i]);
        acts.push_back(nd->pred_act[i]);
      }
    }
    const int64_t n = (int64_t)acts.size();
    int64_t total = 0;
    if (points) points->clear();
    if (n > 0) {
      if (!edges) return -1;
      std::vector<double> parents((size_t)f * n);
      for (int64_t e = 0; e < n; e++)
        for (int r = 0; r < f; r++) parents[(size_t)r * n + e] = par[(size_t)e]->coord[r];
      const double vb = P.v_max > 0 ? P.v_max : 4.0;
      int cap = (int)std::ceil(vb * P.dt / P.grid.res) + 2;
      std::vector<int32_t> cells, cnt((size_t)n);
      for (;;) {
        cells.assign((size_t)n * cap, 0);
        if (int rc = edges(edges_user, parents.data(), acts.data(), n, nullptr, nullptr, cells.data(), cnt.data(), cap)) return rc;
        int need = 0;
        for (int64_t e = 0; e < n; e++) need = std::max(need, cnt[(size_t)e]);
        if (need <= cap) break;
        cap = need;  // a row was truncated: once more with room for the longest
      }
      // sweep 1: the distinct cells, sorted; sweep 2: every cell's entries in the order the edges were met
      for (int64_t e = 0; e < n; e++) total += cnt[(size_t)e];
      std::vector<int32_t> ids;
      ids.reserve((size_t)total);
      for (int64_t e = 0; e < n; e++)
        for (int k = 0; k < cnt[(size_t)e]; k++) ids.push_back(cells[(size_t)e * cap + k]);
      // The distinct cells in ascending order and every cell's entry range: a counting pass over the span of cell
      // indices the edges touch when that span is moderate (the usual case: a sort of a million indices was 40 of this
      // call's 54 ms), a sort otherwise.
      int32_t id_lo = 0x7fffffff, id_hi = (int32_t)0x80000000;
      for (int32_t id : ids) { id_lo = id < id_lo ? id : id_lo; id_hi = id > id_hi ? id : id_hi; }
      // (two span-sized arrays: only while the span stays a small multiple of the entries -- stored edges on a 512^3 map
      // can span tens of millions of cell indices while touching a few thousand cells)
      const bool dense = !ids.empty() && (int64_t)id_hi - id_lo < ((int64_t)1 << 26) && (int64_t)id_hi - id_lo <= (int64_t)16 * (int64_t)ids.size();
      std::vector<int64_t> fill;
      std::vector<int32_t> rk(ids.size());
      if (dense) {
        const size_t span = (size_t)((int64_t)id_hi - id_lo + 1);
        std::vector<uint32_t> per_cell(span, 0u);
        for (int32_t id : ids) per_cell[(size_t)(id - id_lo)]++;
        std::vector<int32_t> rank(span, -1);
        lhm_.key.clear();
        fill.assign(1, 0);
        for (size_t c = 0; c < span; c++)
          if (per_cell[c]) {
            rank[c] = (int32_t)lhm_.key.size();
            lhm_.key.push_back((int32_t)((int64_t)c + id_lo));
            fill.push_back(fill.back() + per_cell[c]);
          }
        for (size_t q = 0; q < ids.size(); q++) rk[q] = rank[(size_t)(ids[q] - id_lo)];
      } else {
        lhm_.key = ids;
        std::sort(lhm_.key.begin(), lhm_.key.end());
        lhm_.key.erase(std::unique(lhm_.key.begin(), lhm_.key.end()), lhm_.key.end());
        fill.assign(lhm_.key.size() + 1, 0);
        for (size_t q = 0; q < ids.size(); q++) {
          rk[q] = (int32_t)(std::lower_bound(lhm_.key.begin(), lhm_.key.end(), ids[q]) - lhm_.key.begin());
          fill[(size_t)rk[q] + 1]++;
        }
        for (size_t k = 0; k + 1 < fill.size(); k++) fill[k + 1] += fill[k];
      }
      lhm_.off = fill;
      lhm_.ent.resize(ids.size());
      {
        size_t q = 0;
        for (int64_t e = 0; e < n; e++)
          for (int k = 0; k < cnt[(size_t)e]; k++, q++) lhm_.ent[(size_t)fill[(size_t)rk[q]]++] = std::make_pair(owner[(size_t)e], slot[(size_t)e]);
      }
      if (points) {  // intToFloat(floatToInt(w.pos)), map_util.h:110-113: the cell's centre, in the order of the edges
        points->reserve(ids.size() * (size_t)D);
        for (int32_t id : ids) {
          int64_t r = id;
          for (int i = 0; i < D; i++) {
            const int64_t c = r % P.grid.n[i];
            r /= P.grid.n[i];
            points->push_back(((double)c + 0.5) * P.grid.res + P.grid.origin[i]);
          }
        }
      }
    }
    if (n_points) *n_points = total;
    return 0;
  }
  size_t linked_cells() const { return lhm_.cells(); }
  int64_t linked_entries() const { return (int64_t)lhm_.ent.size(); }

  // MapPlanner::updateBlockedNodes (map_planner.cpp:160-171) + StateSpace::increaseCost (state_space.h:198-220).
  // cells: [n][D] integer cell coordinates.
  int update_blocked(const int32_t *cells, int64_t n) {
    std::vector<std::pair<LNode *, int32_t>> hit;
    collect(cells, n, &hit);
    for (const auto &h : hit) {
      LNode *nd = h.first;
      const int i = h.second;
      if (!std::isinf(nd->pred_cost[(size_t)i])) {
        nd->pred_cost[(size_t)i] = kInf;
        update_node(nd);
        LNode *par = nd->pred[(size_t)i];
        const int32_t a = nd->pred_act[(size_t)i];
        for (size_t j = 0; j < par->succ_act.size(); j++)
          if (a == par->succ_act[j]) { par->succ_cost[j] = kInf; break; }
      }
    }
    spec_.clear();
    return 0;
  }
  // MapPlanner::updateClearedNodes (map_planner.cpp:174-185) + StateSpace::decreaseCost (state_space.h:222-253): the
  // is_free / calculate_intrinsic_cost of every blocked edge among the affected ones in ONE device call (both are pure
  // functions of the edge and the map), then the reference's loop over them in its order.
  int update_cleared(const int32_t *cells, int64_t n) {
    Planner &P = *cfg;
    const int f = P.F();
    std::vector<std::pair<LNode *, int32_t>> hit;
    collect(cells, n, &hit);
    std::vector<int64_t> which;
    for (size_t k = 0; k < hit.size(); k++)
      if (std::isinf(hit[k].first->pred_cost[(size_t)hit[k].second])) which.push_back((int64_t)k);
    const int64_t m = (int64_t)which.size();
    std::vector<uint8_t> fr((size_t)m);
    std::vector<double> cost((size_t)m);
    if (m > 0) {
      if (!edges) return -1;
      std::vector<double> parents((size_t)f * m);
      std::vector<int32_t> acts((size_t)m);
      for (int64_t e = 0; e < m; e++) {
        const LNode *nd = hit[(size_t)which[(size_t)e]].first;
        const int i = hit[(size_t)which[(size_t)e]].second;
        for (int r = 0; r < f; r++) parents[(size_t)r * m + e] = nd->pred[(size_t)i]->coord[r];
        acts[(size_t)e] = nd->pred_act[(size_t)i];
      }
      if (int rc = edges(edges_user, parents.data(), acts.data(), m, fr.data(), cost.data(), nullptr, nullptr, 0)) return rc;
    }
    int64_t e = 0;
    for (size_t k = 0; k < hit.size(); k++) {
      LNode *nd = hit[k].first;
      const int i = hit[k].second;
      const bool was_queried = e < m && which[(size_t)e] == (int64_t)k;
      // (an edge that an earlier entry of this very loop has already repaired is finite by now: skipped, as in the reference)
      if (std::isinf(nd->pred_cost[(size_t)i]) && was_queried && fr[(size_t)e]) {
        nd->pred_cost[(size_t)i] = cost[(size_t)e];
        update_node(nd);
        LNode *par = nd->pred[(size_t)i];
        const int32_t a = nd->pred_act[(size_t)i];
        for (size_t j = 0; j < par->succ_act.size(); j++)
          if (a == par->succ_act[j]) { par->succ_cost[j] = nd->pred_cost[(size_t)i]; break; }
      }
      if (was_queried) e++;
    }
    spec_.clear();
    return 0;
  }

  // StateSpace::getSubStateSpace (state_space.h:116-195): re-root the tree at best_child_[time_step]
  int sub_state_space(int time_step) {
    if (best_child_.empty()) return 0;
    if (time_step < 0 || (size_t)time_step >= best_child_.size()) return -1;
    Planner &P = *cfg;
    LNode *curr = best_child_[(size_t)time_step];
    start_g_ = curr->g;
    start_rhs_ = curr->rhs;
    start_t_ = curr->coord[4 * P.dim + 1];
    for (auto &kv : hm_) {
      LNode *nd = kv.second;
      nd->g = nd->rhs = kInf;
      nd->pred.clear();
      nd->pred_cost.clear();
      nd->pred_act.clear();
    }
    curr->g = start_g_;
    curr->rhs = start_rhs_;
    NodeMapT new_hm;
    Heap epq;  // ordered by rhs through the same comparison
    curr->handle = epq.push(curr->rhs, curr);
    new_hm[curr->key] = curr;
    while (!epq.empty()) {
      curr = epq.top().n;
      epq.pop();
      for (size_t i = 0; i < curr->succ.size(); i++) {
        // new_hm[succ_coord], else hm_[succ_coord] (state_space.h:152-153).  A stored pointer to a node that an EARLIER
        // re-rooting dropped from hm_ is not that node any more: the reference prints "critical bug" and dereferences
        // a null pointer there; here the coordinate gets a fresh node (g = rhs = inf, never opened).
        LNode *&slot = new_hm[curr->succ[i]->key];
        if (!slot) {
          const auto it = hm_.find(curr->succ[i]->key);
          if (it != hm_.end() && it->second) slot = it->second;
          else {
            slot = make(curr->succ[i]->coord, curr->succ[i]->key);
            slot->h = curr->succ[i]->h;  // (same coordinate, same goal: the heuristic of the node it replaces)
          }
        }
        LNode *sn = slot;
        curr->succ[i] = sn;
        if (std::find(sn->pred.begin(), sn->pred.end(), curr) == sn->pred.end()) {
          sn->pred.push_back(curr);
          sn->pred_cost.push_back(curr->succ_cost[i]);
          sn->pred_act.push_back(curr->succ_act[i]);
        }
        const double tentative = curr->rhs + curr->succ_cost[i];
        if (tentative < sn->rhs) {
          sn->rhs = tentative;
          if (sn->closed) {
            sn->g = sn->rhs;
            sn->handle = epq.push(sn->rhs, sn);
          }
        }
      }
    }
    hm_.swap(new_hm);
    pq_.clear();
    n_closed_ = 0;
    for (auto &kv : hm_) {
      LNode *nd = kv.second;
      if (nd->closed) n_closed_++;
      if (nd->opened && !nd->closed) nd->handle = pq_.push(key_of(nd), nd);
    }
    lhm_.clear();
    spec_.clear();
    return 0;
  }

  // PlannerBase::getCloseSet: positions of the closed nodes, map order
  int32_t closed_positions(double *pos, int32_t cap) const {
    int32_t m = 0;
    for (const auto &kv : hm_) {
      if (!kv.second || !kv.second->closed) continue;
      if (pos && m < cap)
        for (int k = 0; k < cfg->dim; k++) pos[(size_t)m * cfg->dim + k] = kv.second->coord[k];
      m++;
    }
    return m;
  }
  int32_t open_states(double *states, int32_t cap) const {
    const int f = cfg->F();
    int32_t m = 0;
    for (const Heap::Item &it : pq_.items()) {
      if (states && m < cap)
        for (int k = 0; k < f; k++) states[(size_t)m * f + k] = it.n->coord[k];
      m++;
    }
    return m;
  }

 private:
  NodeMapT hm_;
  std::vector<std::unique_ptr<LNode>> pool_;
  Heap pq_;
  std::vector<LNode *> best_child_;
  // linkedHashMap (map_planner.h:15-17: cell index -> the (node, predecessor slot) pairs whose edge passes through the
  // cell, in the order getLinkedNodes met them).  Only ever looked up by cell, never iterated: a compressed row
  // storage built in two sweeps over the device's cell lists (1 M entries in a few ms where a hash map of vectors took
  // 40) -- lhm_key_ sorted cell indices of the non-empty cells, lhm_off_ their entry ranges.
  struct LinkedTable {
    std::vector<int32_t> key;
    std::vector<int64_t> off;  // [key.size() + 1]
    std::vector<std::pair<LNode *, int32_t>> ent;
    void clear() { key.clear(); off.clear(); ent.clear(); }
    size_t cells() const { return key.size(); }
    // entries of cell `id`: [first, last)
    std::pair<const std::pair<LNode *, int32_t> *, const std::pair<LNode *, int32_t> *> find(int32_t id) const {
      const auto it = std::lower_bound(key.begin(), key.end(), id);
      if (it == key.end() || *it != id) return {nullptr, nullptr};
      const size_t k = (size_t)(it - key.begin());
      return {ent.data() + off[k], ent.data() + off[k + 1]};
    }
  } lhm_;
  bool ready_ = false;
  double start_g_ = 0, start_rhs_ = 0, start_t_ = 0;
  int expand_iteration_ = 0;  // StateSpace::expand_iteration_
  int n_closed_ = 0;          // nodes with iterationclosed set
  void set_closed(LNode *nd, bool v) { n_closed_ += (int)v - (int)nd->closed; nd->closed = v; }

  struct List { std::vector<double> coord, cost; std::vector<int32_t> act; std::vector<uint64_t> key; };
  std::unordered_map<const LNode *, List> spec_;  // get_succ results that rode along in a launch (valid for this plan)

  LNode *make(const double *coord, uint64_t key) {
    pool_.emplace_back(new LNode());
    LNode *nd = pool_.back().get();
    std::copy(coord, coord + cfg->F(), nd->coord);
    nd->key = key;
    return nd;
  }
  double key_of(const LNode *nd) const { return std::min(nd->g, nd->rhs) + cfg->eps * nd->h; }  // calculateKey

  // StateSpace::updateNode, state_space.h:255-281
  void update_node(LNode *nd) {
    if (nd->rhs != start_rhs_) {
      nd->rhs = kInf;
      for (size_t i = 0; i < nd->pred.size(); i++) {
        const double v = nd->pred[i]->g + nd->pred_cost[i];
        if (nd->rhs > v) nd->rhs = v;
      }
    }
    if (nd->opened && !nd->closed) {
      pq_.erase(nd->handle);
      set_closed(nd, true);
    }
    if (nd->g != nd->rhs) {
      nd->handle = pq_.push(key_of(nd), nd);
      nd->opened = true;
      set_closed(nd, false);
    }
  }

  void collect(const int32_t *cells, int64_t n, std::vector<std::pair<LNode *, int32_t>> *hit) const {
    const Planner &P = *cfg;
    for (int64_t k = 0; k < n; k++) {
      int c[3] = {0, 0, 0};
      for (int i = 0; i < P.dim; i++) c[i] = cells[(size_t)k * P.dim + i];
      const int32_t id = (int32_t)P.grid.index(c);  // MapUtil::getIndex
      const auto range = lhm_.find(id);
      for (const auto *e = range.first; e != range.second; e++) hit->push_back(*e);
    }
  }

  // env_base::get_succ for `nd` (env_map.h:147-172), blocked successors included: from the lists that rode along in
  // an earlier launch of this plan, or from a launch of `nd` together with the best open nodes that were never expanded
  int get_succ(LNode *nd, List *out) {
    Planner &P = *cfg;
    auto it = spec_.find(nd);
    if (it != spec_.end()) {
      *out = std::move(it->second);
      spec_.erase(it);
      last.spec_hits++;
      return 0;
    }
    std::vector<LNode *> group{nd};
    if (P.batch > 1 && (P.packed || P.lists || P.batched)) {
      // the heap array is ordered well enough near its front: the first entries that have no successor list yet
      const auto &h = pq_.items();
      for (size_t i = 0; i < h.size() && group.size() < (size_t)P.batch && i < (size_t)4 * P.batch; i++) {
        LNode *c = h[i].n;
        if (c != nd && c->succ.empty() && !spec_.count(c) && std::find(group.begin(), group.end(), c) == group.end()) group.push_back(c);
      }
    }
    std::vector<List> res;
    if (int rc = expand(group, &res)) return rc;
    *out = std::move(res[0]);
    for (size_t k = 1; k < group.size(); k++) spec_[group[k]] = std::move(res[k]);
    return 0;
  }

  // one provider call for `group`: per node the emitted successors (finite or blocked) in control order
  int expand(const std::vector<LNode *> &group, std::vector<List> *res) {
    Planner &P = *cfg;
    const int f = P.F(), nU = P.nU;
    const int64_t n = (int64_t)group.size();
    res->assign((size_t)n, List());
    last.device_launches++;
    last.pairs += n * nU;
    auto finish = [&](List &L, const double *parent) {  // keys (and states) the provider did not deliver
      const size_t m = L.act.size();
      if (L.coord.empty()) {
        L.coord.resize(m * (size_t)f);
        for (size_t s = 0; s < m; s++) forward_state(P.dim, P.control, parent, &P.U[(size_t)L.act[s] * P.udim], P.dt, &L.coord[s * (size_t)f]);
      }
      if (L.key.empty()) {
        L.key.resize(m);
        for (size_t s = 0; s < m; s++) L.key[s] = lattice_hash(P.dim, P.control, &L.coord[s * (size_t)f]);
      }
    };
    if (n == 1 && P.single && !P.packed) {
      std::vector<double> succ((size_t)nU * f), cost((size_t)nU);
      std::vector<int32_t> act((size_t)nU);
      int32_t m = 0;
      if (int rc = P.single(P.user, group[0]->coord, succ.data(), cost.data(), act.data(), &m)) return rc;
      List &L = (*res)[0];
      L.coord.assign(succ.begin(), succ.begin() + (size_t)m * f);
      L.cost.assign(cost.begin(), cost.begin() + m);
      L.act.assign(act.begin(), act.begin() + m);
      finish(L, group[0]->coord);
      return 0;
    }
    std::vector<double> nodes((size_t)f * n);
    for (int64_t k = 0; k < n; k++)
      for (int r = 0; r < f; r++) nodes[(size_t)r * n + k] = group[(size_t)k]->coord[r];
    if (P.packed) {
      PackedView v;
      if (int rc = P.packed(P.user, nodes.data(), n, &v)) return rc;
      for (int64_t k = 0; k < n; k++) {
        List &L = (*res)[(size_t)k];
        const size_t o = (size_t)v.offs[k], m = (size_t)v.count[k];
        L.cost.assign(v.cost + o, v.cost + o + m);
        L.act.assign(v.action + o, v.action + o + m);
        if (v.hash) L.key.assign(v.hash + o, v.hash + o + m);
        if (v.state) {
          L.coord.resize(m * (size_t)f);
          for (size_t s = 0; s < m; s++)
            for (int r = 0; r < f; r++) L.coord[s * (size_t)f + r] = v.state[(size_t)r * v.total + o + s];
        }
        finish(L, group[(size_t)k]->coord);
      }
      return 0;
    }
    if (P.batched) {
      const int64_t slots = n * nU;
      std::vector<uint8_t> st((size_t)slots);
      std::vector<double> cs((size_t)slots), state((size_t)f * slots);
      if (int rc = P.batched(P.user, nodes.data(), n, st.data(), cs.data(), state.data())) return rc;
      for (int64_t k = 0; k < n; k++) {
        List &L = (*res)[(size_t)k];
        for (int i = 0; i < nU; i++) {
          const int64_t sl = k * nU + i;
          if (st[(size_t)sl] != 1 && st[(size_t)sl] != 2) continue;
          for (int r = 0; r < f; r++) L.coord.push_back(state[(size_t)r * slots + sl]);
          L.cost.push_back(cs[(size_t)sl]);
          L.act.push_back(i);
        }
        finish(L, group[(size_t)k]->coord);
      }
      return 0;
    }
    for (int64_t k = 0; k < n; k++) {  // the single-node provider only: one call per node
      std::vector<LNode *> one{group[(size_t)k]};
      std::vector<List> r1;
      if (int rc = expand(one, &r1)) return rc;
      (*res)[(size_t)k] = std::move(r1[0]);
      last.device_launches--;  // (counted once above)
      last.pairs -= nU;
    }
    last.device_launches += (int)n - 1;
    last.pairs += (n - 1) * nU;
    return 0;
  }

  // GraphSearch::LPAstar, graph_search.h:194-365
  int lpastar(const double *start, const double *goal) {
    Planner &P = *cfg;
    if (P.is_goal(start, goal)) { last.ok = true; last.cost = 0; return 0; }
    const uint64_t start_key = lattice_hash(P.dim, P.control, start);
    LNode *&sslot = hm_[start_key];
    LNode *curr = sslot;
    if (!curr) {
      curr = make(start, start_key);
      curr->g = kInf;
      curr->rhs = 0;
      curr->h = P.eps == 0 ? 0 : P.heur(start, goal);
      curr->handle = pq_.push(key_of(curr), curr);
      curr->opened = true;
      set_closed(curr, false);
      sslot = curr;
    }
    LNode dummy;  // the goal node until one is reached: g = rhs = inf, h = 0
    LNode *goal_node = &dummy;
    if (!best_child_.empty() && P.is_goal(best_child_.back()->coord, goal)) goal_node = best_child_.back();
    else { dummy.g = dummy.rhs = kInf; dummy.h = 0; }
    int expand_iteration = 0;
    double cost = kInf;
    bool finished = false;
    for (;;) {
      // (the reference reads pq_.top() of an EMPTY queue here when nothing is left to repair: undefined there.  Here:
      // a goal node that is consistent with a finite g needs no repair -- its trajectory is recovered again; anything
      // else is "no trajectory")
      if (pq_.empty()) {
        finished = goal_node != &dummy && goal_node->g == goal_node->rhs && !std::isinf(goal_node->g);
        break;
      }
      if (!(pq_.top().f < key_of(goal_node) || goal_node->rhs != goal_node->g)) { finished = true; break; }
      expand_iteration++;
      curr = pq_.top().n;
      pq_.pop();
      set_closed(curr, true);
      if (curr->g > curr->rhs) curr->g = curr->rhs;
      else { curr->g = kInf; update_node(curr); }
      List L;
      const bool explored = !curr->succ.empty();
      if (!explored) {
        if (int rc = get_succ(curr, &L)) return rc;
      }
      const size_t m = explored ? curr->succ.size() : L.act.size();
      if (!explored) {
        curr->succ.resize(m);
        curr->succ_cost.resize(m);
        curr->succ_act.resize(m);
      }
      const int f = P.F();
      for (size_t s = 0; s < m; s++) {
        LNode *sn;
        double c_s;
        int32_t a_s;
        if (explored) {
          // hm_[succ_coord[s]] of the reference (graph_search.h:285-290), not the stored pointer: getSubStateSpace drops
          // the nodes it did not reach from hm_ and clears the queue, and a re-opened parent may still list one of them
          // -- the reference then starts a FRESH State for that coordinate (g = rhs = inf, never opened).  The dropped
          // node keeps opened / handle of the cleared queue: relaxing it would erase a stale heap position.
          LNode *&slot = hm_[curr->succ[s]->key];
          if (!slot) {
            slot = make(curr->succ[s]->coord, curr->succ[s]->key);
            slot->h = P.eps == 0 ? 0 : P.heur(slot->coord, goal);
          }
          sn = slot;
          curr->succ[s] = sn;
          c_s = curr->succ_cost[s];
          a_s = curr->succ_act[s];
        } else {
          LNode *&slot = hm_[L.key[s]];
          if (!slot) {
            slot = make(&L.coord[s * (size_t)f], L.key[s]);
            slot->h = P.eps == 0 ? 0 : P.heur(slot->coord, goal);
          }
          sn = slot;
          c_s = L.cost[s];
          a_s = L.act[s];
          curr->succ[s] = sn;
          curr->succ_cost[s] = c_s;
          curr->succ_act[s] = a_s;
        }
        if (std::find(sn->pred.begin(), sn->pred.end(), curr) == sn->pred.end()) {
          sn->pred.push_back(curr);
          sn->pred_cost.push_back(c_s);
          sn->pred_act.push_back(a_s);
        }
        update_node(sn);
        last.relaxed++;
      }
      if (P.is_goal(curr->coord, goal)) goal_node = curr;
      if (P.max_expand > 0 && expand_iteration >= P.max_expand) break;
      if (pq_.empty()) break;
    }
    // (expand_iteration_ is assigned after the loop only, graph_search.h:346: a plan that gives up inside the loop leaves
    // getExpandedNum at the previous plan's figure)
    if (finished) expand_iteration_ = expand_iteration;
    last.expansions = expand_iteration_;
    if (!finished) return 0;  // max expansions or an empty queue: infinite cost, no trajectory in `last` (ok = false)
    if (recover(goal_node, start_key)) {
      cost = goal_node->g - start_g_;
      last.ok = !std::isinf(cost);
      last.cost = cost;
    }
    return 0;
  }

  // GraphSearch::recoverTraj, graph_search.h:369-455 (also maintains best_child_)
  bool recover(LNode *curr, uint64_t start_key) {
    Planner &P = *cfg;
    const int f = P.F();
    best_child_.clear();
    last.traj_end.assign(curr->coord, curr->coord + f);
    std::vector<LNode *> from;
    std::vector<int32_t> acts;
    bool found = false;
    while (!curr->pred.empty()) {
      best_child_.push_back(curr);
      int min_id = -1;
      double min_rhs = kInf, min_g = kInf;
      for (size_t i = 0; i < curr->pred.size(); i++) {
        const double pg = curr->pred[i]->g, v = pg + curr->pred_cost[i];
        if (min_rhs > v) { min_rhs = v; min_g = pg; min_id = (int)i; }
        else if (!std::isinf(curr->pred_cost[i]) && min_rhs == v) {
          if (min_g < pg) { min_g = pg; min_id = (int)i; }
        }
      }
      if (min_id < 0) break;
      const int32_t a = curr->pred_act[(size_t)min_id];
      curr = curr->pred[(size_t)min_id];
      from.push_back(curr);
      acts.push_back(a);
      if (curr->key == start_key) {
        best_child_.push_back(curr);
        found = true;
        break;
      }
    }
    std::reverse(best_child_.begin(), best_child_.end());
    if (!found) return false;
    std::reverse(from.begin(), from.end());
    std::reverse(acts.begin(), acts.end());
    const int K = (P.control & 8) ? 4 : (P.control & 4) ? 3 : (P.control & 2) ? 2 : 1;
    for (size_t s = 0; s < from.size(); s++) {
      const double *nd = from[s]->coord;
      const double *u = &P.U[(size_t)acts[s] * P.udim];
      last.total_time += P.dt;
      for (int order = 1; order <= 4; order++) {
        double j = 0;
        for (int i = 0; i < P.dim; i++) {
          double c[6] = {0, 0, 0, 0, 0, 0};
          c[5] = nd[i];
          if (K == 1) c[4] = u[i];
          if (K == 2) { c[4] = nd[P.dim + i]; c[3] = u[i]; }
          if (K == 3) { c[4] = nd[P.dim + i]; c[3] = nd[2 * P.dim + i]; c[2] = u[i]; }
          if (K == 4) { c[4] = nd[P.dim + i]; c[3] = nd[2 * P.dim + i]; c[2] = nd[3 * P.dim + i]; c[1] = u[i]; }
          j += effort_1d(c, P.dt, order);
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
