// mplx_env_map.hpp -- the reference-side binding of libmplx.so.
//
// This is the file a maintainer of sikang/motion_primitive_library adds to the
// reference tree to make MapPlanner::plan() run its successor expansion on an
// MI355X.  It is compiled against the REFERENCE's headers (Eigen, Boost and
// all) and links only against the C ABI of include/mplx.h:
//
//   MPL::env_map_hip<Dim>   : MPL::env_map<Dim>
//        overrides the virtual get_succ  (env_base.h:358-362, env_map.h:147-172)
//        and forwards it to mplx_get_succ / mplx_expand;
//   MPL::GpuMapPlanner<Dim> : MPL::MapPlanner<Dim>
//        overrides the virtual setMapUtil (map_planner.h:29,
//        map_planner.cpp:14-18) to install env_map_hip as ENV_.
//
// Everything else of PlannerBase / MapPlanner / GraphSearch / StateSpace is the
// reference's own, unchanged code: they only reach the env through ENV_.
//
// Parameter tracking.  env_base's setters are not virtual (env_base.h:234-306),
// so the adapter re-reads the env members and pushes what changed:
//   * scalars and U_ are compared with a cached copy on every get_succ (cheap);
//   * the map, the potential map and the search region are uploaded when they
//     are set through a virtual (set_potential_map) or at the start of a plan:
//     PlannerBase::plan calls the virtual ENV_->is_free(start.pos) first
//     (planner_base.h:283), which marks them stale.  Call notify_map_changed()
//     after editing the MapUtil in place outside of these paths.
// Speculative batching (set_batch(n), n > 1).  Astar asks for one node at a time
// (graph_search.h:75), which costs one launch + round trip per expansion.  get_succ
// is a pure function of the node, so the adapter may compute lists ahead of time: on
// a miss it expands the requested node TOGETHER with the most promising successors
// it has handed out and that have not been asked for yet (ranked by g + h like the
// search's own open list, with g accumulated along the edges it returned and h =
// get_heur), in one mplx_expand_lists call, and serves later requests from that
// cache.  The lists returned are identical with and without batching; only the
// number of launches changes.  The cache is dropped whenever parameters, controls or
// maps change, and at the start of every plan().
// Incremental re-planning (LPA*).  MapPlanner::getLinkedNodes and updateClearedNodes
// (map_planner.cpp:125-157, 174-185) walk every stored edge with Primitive::sample /
// env_map::is_free(Primitive); GpuMapPlanner shadows both with ONE batched device call
// each (mplx_check_edges) and leaves lhm_ and the state space exactly as the
// reference's loops would; updateBlockedNodes runs the reference's bookkeeping unchanged.  All three also hand the
// edited cells to the device copy of the map (mplx_edit_map), so a re-plan after an edit of k cells uploads k cells.
// Errors never throw: a failed device call prints the engine's message, latches it
// (device_ok() / device_error(); GpuMapPlanner::plan then returns false with the
// message instead of looking like "no trajectory exists") and returns an empty
// successor list (the reference's own error convention is printf + sentinel,
// graph_search.h:149-161).  There is no CPU fallback.
#ifndef MPLX_ENV_MAP_HPP
#define MPLX_ENV_MAP_HPP

#include <mpl_planner/env/env_map.h>
#include <mpl_planner/planner/map_planner.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <queue>
#include <string>
#include <unordered_map>
#include <vector>

#include "mplx.h"

namespace MPL {

template <int Dim>
class env_map_hip : public env_map<Dim> {
 public:
  explicit env_map_hip(std::shared_ptr<MapUtil<Dim>> map_util, int device = 0)
      : env_map<Dim>(map_util) {
    if (mplx_create(Dim, device, &ctx_) != MPLX_OK) {
      printf(ANSI_COLOR_RED "[env_map_hip] %s\n" ANSI_COLOR_RESET, mplx_last_error(nullptr));
      ctx_ = nullptr;
    }
  }
  ~env_map_hip() {
    if (getenv("MPLX_ADAPTER_TIMING"))
      printf("[env_map_hip] get_succ total %.1f ms (device launches %.1f ms) over %d launches\n", t_total_ * 1e3, t_launch_ * 1e3, launches_);
    mplx_destroy(ctx_);
  }
  env_map_hip(const env_map_hip &) = delete;
  env_map_hip &operator=(const env_map_hip &) = delete;

  bool ok() const { return ctx_ != nullptr; }
  /// False once any device call of this env has failed (sticky until clear_device_error()); the text of the
  /// first failure.  A search that ran into a device failure sees empty successor lists, i.e. "no path": callers
  /// tell the two apart here (GpuMapPlanner::plan does).
  bool device_ok() const { return ctx_ != nullptr && first_error_.empty(); }
  const std::string &device_error() const { return first_error_; }
  void clear_device_error() { first_error_.clear(); }
  /// Nodes per device launch (1 = one launch per get_succ, the default).
  void set_batch(int n) { batch_ = n < 1 ? 1 : n; drop_cache(); }
  /// env_map::get_succ also records every finite edge as a Primitive in expanded_edges_ (env_map.h:166, read by
  /// PlannerBase::getExpandedEdges for drawing): a 200-byte object per relaxed edge, a fifth of the reference's
  /// plan() time on 3D problems.  Kept by default without batching (identical side effects); with batching it is
  /// off unless asked for.
  void set_record_edges(bool on) { record_edges_ = on ? 1 : 0; }
  bool record_edges() const { return record_edges_ < 0 ? batch_ <= 1 : record_edges_ != 0; }
  int launches() const { return launches_; }
  /// Re-upload the map / potential / region before the next expansion.
  void notify_map_changed() { maps_stale_ = true; cells_synced_ = false; }
  /// The application has edited exactly these cells of the MapUtil since the maps were last in step (the incremental
  /// loop of map_planner.cpp:160-185: edit, updateBlockedNodes / updateClearedNodes, plan): the device copy is patched
  /// with their current values (mplx_edit_map: 9 bytes per cell, blocked bits included) instead of being uploaded
  /// whole, and the NEXT plan() does not mark the maps stale -- it has just been told what changed.  A plan() without
  /// such a call before it uploads the map as before (an application may have edited the MapUtil in place).
  bool edit_cells(const vec_Veci<Dim> &pns) const {
    if (!ctx_) return latch("no device context (mplx_create failed)");
    if (maps_stale_) return true;  // a whole upload is pending (never uploaded, or notify_map_changed): it carries the edit
    const Tmap &cells = peek_map();
    std::vector<int64_t> idx;
    std::vector<int8_t> val;
    idx.reserve(pns.size());
    val.reserve(pns.size());
    for (const auto &pn : pns) {
      if (this->map_util_->isOutside(pn)) continue;
      const int id = this->map_util_->getIndex(pn);
      idx.push_back(id);
      val.push_back(cells[(size_t)id]);
    }
    drop_cache();
    if (!idx.empty() && mplx_edit_map(ctx_, idx.data(), val.data(), (int64_t)idx.size()) != MPLX_OK) return complain();
    cells_synced_ = true;
    return true;
  }
  /// Host -> device bytes this env's map calls have moved so far (mplx_map_upload_bytes).
  uint64_t map_upload_bytes() const {
    uint64_t b = 0;
    if (ctx_) mplx_map_upload_bytes(ctx_, &b);
    return b;
  }

  /// First virtual call of every PlannerBase::plan (planner_base.h:283).
  bool is_free(const Vecf<Dim> &pt) const override {
    if (cells_synced_) cells_synced_ = false;  // (one plan: the edit the application named is on the device already)
    else maps_stale_ = true;
    return env_map<Dim>::is_free(pt);
  }
  bool is_free(const Primitive<Dim> &pr) const override { return env_map<Dim>::is_free(pr); }

  void set_potential_map(const std::vector<int8_t> &map) override {
    env_map<Dim>::set_potential_map(map);
    maps_stale_ = true;
  }

  /// The hot path: same contract as env_map<Dim>::get_succ.
  void get_succ(const Waypoint<Dim> &curr, vec_E<Waypoint<Dim>> &succ, std::vector<decimal_t> &succ_cost,
                std::vector<int> &action_idx) const override {
    struct Tick {
      double &acc;
      std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
      ~Tick() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
    } tick{t_total_};
    succ.clear();
    succ_cost.clear();
    action_idx.clear();
    this->expanded_nodes_.push_back(curr.pos);
    if (!ctx_) { latch("no device context (mplx_create failed)"); return; }
    if (!sync(curr.control)) return;
    constexpr int F = 4 * Dim + 2;
    const int nU = (int)this->U_.size();
    double node[F];
    pack(curr, node);
    int32_t n = 0;
    if (batch_ > 1) {
      const uint64_t key = lattice_hash(curr.control, node);
      auto it = cache_.find(key);
      if (it == cache_.end()) {
        if (!expand_group(curr.control, key, node)) return;
        it = cache_.find(key);
      }
      Cached &c = it->second;
      n = (int32_t)c.act.size();
      buf_succ_.swap(c.succ);
      buf_cost_.swap(c.cost);
      buf_act_.swap(c.act);
      // its successors become candidates for the next speculative launch
      const double g = c.g;
      for (int m = 0; m < n; m++) {
        if (std::isinf(buf_cost_[(size_t)m])) continue;
        const uint64_t ck = c.keys[(size_t)m];
        const double cg = g + buf_cost_[(size_t)m];
        if (m + 12 < n) g_est_.prefetch(c.keys[(size_t)m + 12]);
        const double *gi = g_est_.find(ck);
        if (gi && *gi <= cg) continue;
        g_est_.put(ck, cg);
        Cand cand;
        cand.f = cg + rank_heur(&buf_succ_[(size_t)m * F], tn_scratch_);
        cand.g = cg;
        cand.key = ck;
        cand.row = (uint32_t)(rows_.size() / F);
        rows_.insert(rows_.end(), &buf_succ_[(size_t)m * F], &buf_succ_[(size_t)m * F] + F);
        shadow_.push(cand);
      }
      cache_.erase(it);
      asked_[key] = true;
    } else {
      buf_succ_.resize((size_t)nU * F);
      buf_cost_.resize((size_t)nU);
      buf_act_.resize((size_t)nU);
      launches_++;
      if (mplx_get_succ(ctx_, node, buf_succ_.data(), buf_cost_.data(), buf_act_.data(), &n) != MPLX_OK) {
        complain();
        return;
      }
    }
    const bool rec = record_edges();
    succ.reserve((size_t)n);
    succ_cost.reserve((size_t)n);
    action_idx.reserve((size_t)n);
    for (int m = 0; m < n; m++) {
      Waypoint<Dim> tn(curr.control);
      unpack(&buf_succ_[(size_t)m * F], tn);
      succ.push_back(tn);
      succ_cost.push_back(buf_cost_[(size_t)m]);
      action_idx.push_back(buf_act_[(size_t)m]);
      if (rec && !std::isinf(buf_cost_[(size_t)m]))  // debug side effect of env_map.h:166
        this->expanded_edges_.push_back(Primitive<Dim>(curr, this->U_[buf_act_[(size_t)m]], this->dt_));
    }
  }

  /// MapPlanner::updatePotentialMap on the device (map_planner.cpp:286-391): the map held by
  /// `map_util` is dilated in HBM, read back, installed in the MapUtil and as potential map.
  bool device_update_potential_map(const Vecf<Dim> &pos, const Vecf<Dim> &radius, const Vecf<Dim> &range,
                                   decimal_t pow, const std::shared_ptr<MapUtil<Dim>> &map_util) {
    maps_stale_ = true;
    if (!ctx_) return latch("no device context (mplx_create failed)");
    if (!sync_maps()) return false;
    double p[3] = {0, 0, 0}, r[3] = {0, 0, 0}, g[3] = {0, 0, 0};
    for (int i = 0; i < Dim; i++) { p[i] = pos(i); r[i] = radius(i); g[i] = range(i); }
    Tmap dmap(map_util->getMap().size());
    if (mplx_update_potential_map(ctx_, p, r, g, pow, dmap.data()) != MPLX_OK) return complain();
    map_util->setMap(map_util->getOrigin(), map_util->getDim(), dmap, map_util->getRes());
    env_map<Dim>::set_potential_map(map_util->getMap());
    maps_stale_ = false;  // the device already holds exactly this map and potential
    return true;
  }

  /// MapPlanner::setSearchRegion on the device (map_planner.cpp:46-95).
  bool device_set_search_region(const vec_Vecf<Dim> &path, bool dense, const Vecf<Dim> &search_radius) {
    if (!ctx_) return latch("no device context (mplx_create failed)");
    if (!sync_maps()) return false;
    std::vector<double> pts(path.size() * Dim);
    for (size_t k = 0; k < path.size(); k++)
      for (int i = 0; i < Dim; i++) pts[k * Dim + i] = path[k](i);
    double sr[3] = {0, 0, 0};
    for (int i = 0; i < Dim; i++) sr[i] = search_radius(i);
    std::vector<uint8_t> bytes(this->map_util_->getMap().size());
    if (mplx_set_search_region_path(ctx_, pts.data(), (int32_t)path.size(), dense ? 1 : 0, sr, bytes.data()) != MPLX_OK)
      return complain();
    std::vector<bool> in_region(bytes.size());
    for (size_t i = 0; i < bytes.size(); i++) in_region[i] = bytes[i] != 0;
    env_map<Dim>::set_search_region(in_region);
    region_fp_ = region_fingerprint();  // the device already holds exactly this region
    return true;
  }

  /// pack() for callers outside the class (GpuMapPlanner's edge batches)
  static void pack_row(const Waypoint<Dim> &w, double *row) { pack(w, row); }

  /// Batched edge re-validation (mplx_check_edges; env_map::is_free(Primitive) env_map.h:60-76, intrinsic cost
  /// env_base.h:343-345, the cells of MapPlanner::getLinkedNodes) for edges (parent waypoint, action id).  parents
  /// are field-major [4D+2][n].  Any output may be null; `cells` rows hold *cell_cap entries (grown until no row is
  /// truncated).
  bool device_check_edges(int control, const std::vector<double> &parents_fm, const std::vector<int32_t> &actions,
                          std::vector<uint8_t> *free_flag, std::vector<double> *cost, std::vector<int32_t> *cells,
                          std::vector<int32_t> *cell_count, int *cell_cap, std::vector<uint8_t> *outside) const {
    if (!ctx_) return latch("no device context (mplx_create failed)");
    if (!sync(control)) return false;
    const int64_t n = (int64_t)actions.size();
    if (free_flag) free_flag->assign((size_t)n, 0);
    if (cost) cost->assign((size_t)n, 0.0);
    if (outside) outside->assign((size_t)n, 0);
    std::vector<int32_t> cnt((size_t)n, 0);
    int cap = 0;
    if (cells) {
      const double vb = this->v_max_ > 0 ? this->v_max_ : 4.0;
      cap = (int)std::ceil(vb * this->dt_ / this->map_util_->getRes()) + 2;
    }
    for (;;) {
      mplx_edges_out o{};
      o.free_flag = free_flag ? free_flag->data() : nullptr;
      o.cost = cost ? cost->data() : nullptr;
      o.outside = outside ? outside->data() : nullptr;
      if (cells) {
        cells->assign((size_t)n * cap, 0);
        o.cells = cells->data();
        o.cell_count = cnt.data();
        o.cell_cap = cap;
      }
      if (mplx_check_edges(ctx_, parents_fm.data(), actions.data(), n, n, &o) != MPLX_OK) return complain();
      int need = 0;
      for (int64_t e = 0; e < n; e++) need = cnt[(size_t)e] > need ? cnt[(size_t)e] : need;
      if (!cells || need <= cap) break;
      cap = need;  // a row was truncated: once more with room for the longest
    }
    if (cell_count) *cell_count = cnt;
    if (cell_cap) *cell_cap = cap;
    return true;
  }

  /// Batched form: dense slots for `nodes` (see mplx_expand); host buffers.
  bool expand(const vec_E<Waypoint<Dim>> &nodes, std::vector<uint8_t> &status, std::vector<decimal_t> &cost,
              std::vector<uint64_t> &hash, std::vector<decimal_t> &state) const {
    if (!ctx_ || nodes.empty() || !sync(nodes.front().control)) return false;
    constexpr int F = 4 * Dim + 2;
    const int64_t n = (int64_t)nodes.size(), slots = n * (int64_t)this->U_.size();
    std::vector<double> packed((size_t)F * n);
    for (int64_t k = 0; k < n; k++) {
      double row[F];
      pack(nodes[(size_t)k], row);
      for (int f = 0; f < F; f++) packed[(size_t)f * n + k] = row[f];
    }
    status.resize((size_t)slots);
    cost.resize((size_t)slots);
    hash.resize((size_t)slots);
    state.resize((size_t)F * slots);
    mplx_succ o{};
    o.status = status.data();
    o.cost = cost.data();
    o.hash = hash.data();
    o.state = state.data();
    o.state_stride = slots;
    return mplx_expand(ctx_, packed.data(), n, n, &o) == MPLX_OK;
  }

 private:
  // lattice hash -> best path cost seen so far (open addressing, linear probing; NaN marks an empty slot)
  class GMap {
   public:
    double *find(uint64_t key) {
      if (!cap_) return nullptr;
      for (size_t i = slot(key);; i = (i + 1) & (cap_ - 1)) {
        if (used_[i] == 0) return nullptr;
        if (keys_[i] == key) return &vals_[i];
      }
    }
    void put(uint64_t key, double v) {
      if ((n_ + 1) * 10 > cap_ * 6) grow();
      size_t i = slot(key);
      while (used_[i] && keys_[i] != key) i = (i + 1) & (cap_ - 1);
      if (!used_[i]) { used_[i] = 1; keys_[i] = key; n_++; }
      vals_[i] = v;
    }
    void prefetch(uint64_t key) const {
      if (cap_) { const size_t i = slot(key); __builtin_prefetch(&keys_[i]); __builtin_prefetch(&used_[i]); __builtin_prefetch(&vals_[i]); }
    }
    void clear() { keys_.clear(); vals_.clear(); used_.clear(); cap_ = n_ = 0; }

   private:
    size_t slot(uint64_t k) const {
      k ^= k >> 33;
      k *= 0xff51afd7ed558ccdULL;
      k ^= k >> 33;
      return (size_t)k & (cap_ - 1);
    }
    void grow() {
      std::vector<uint64_t> ok;
      std::vector<double> ov;
      std::vector<uint8_t> ou;
      ok.swap(keys_); ov.swap(vals_); ou.swap(used_);
      const size_t old_cap = cap_;
      cap_ = cap_ ? cap_ * 2 : 4096;
      keys_.assign(cap_, 0); vals_.assign(cap_, 0.0); used_.assign(cap_, 0);
      for (size_t j = 0; j < old_cap; j++)
        if (ou[j]) {
          size_t i = slot(ok[j]);
          while (used_[i]) i = (i + 1) & (cap_ - 1);
          used_[i] = 1; keys_[i] = ok[j]; vals_[i] = ov[j];
        }
    }
    std::vector<uint64_t> keys_;
    std::vector<double> vals_;
    std::vector<uint8_t> used_;
    size_t cap_ = 0, n_ = 0;
  };

  struct Cached {
    std::vector<double> succ, cost;  // succ: [n][4D+2]
    std::vector<int32_t> act;
    std::vector<uint64_t> keys;
    double g = 0;
  };
  struct Cand {
    double f, g;
    uint64_t key;
    uint32_t row;  // index into rows_ (the state, 4D+2 doubles)
    bool operator<(const Cand &o) const { return f > o.f; }  // smallest f on top
  };

  // f-value used to rank speculation candidates: the search's own heuristic.  The default one
  // (env_base.h:58-64) is evaluated in place -- get_heur would hash the state and the goal first
  // (env_base.h:47) -- any other goes through the virtual.  Ranking only decides which nodes share a
  // launch, never a result.
  double rank_heur(const double *row, Waypoint<Dim> &scratch) const {
    if (this->heur_ignore_dynamics_ && this->prior_traj_.empty()) {
      double m = 0;
      for (int i = 0; i < Dim; i++) m = std::max(m, std::abs(row[i] - this->goal_node_.pos(i)));
      return this->v_max_ > 0 ? this->w_ * m / this->v_max_ : this->w_ * m;
    }
    unpack(row, scratch);
    return this->get_heur(scratch);
  }

  // the engine's lattice hash (waypoint.h:93-125 with the classic hash_combine), for cache keys only
  static uint64_t lattice_hash(int control, const double *w) {
    uint64_t h = 0;
    auto fold = [&h](int id) { h ^= (uint64_t)(int64_t)id + 0x9e3779b9ULL + (h << 6) + (h >> 2); };
    for (int i = 0; i < Dim; i++) {
      if (control & 1) fold((int)std::round(w[0 * Dim + i] / 0.01));
      if (control & 2) fold((int)std::round(w[1 * Dim + i] / 0.1));
      if (control & 4) fold((int)std::round(w[2 * Dim + i] / 0.1));
      if (control & 8) fold((int)std::round(w[3 * Dim + i] / 0.1));
    }
    if (control & 16) fold((int)std::round(w[4 * Dim] / 0.1));
    return h;
  }

  void drop_cache() const {
    cache_.clear();
    g_est_.clear();
    asked_.clear();
    rows_.clear();
    shadow_ = std::priority_queue<Cand>();
  }

  // one launch: the requested node plus the best candidates that have no list yet
  bool expand_group(int control, uint64_t key, const double *node) const {
    constexpr int F = 4 * Dim + 2;
    const int nU = (int)this->U_.size();
    std::vector<uint64_t> gk{key};
    std::vector<double> rows(node, node + F), gg;
    {
      const double *gi = g_est_.find(key);
      gg.push_back(gi ? *gi : 0.0);
    }
    while ((int)gk.size() < batch_ && !shadow_.empty()) {
      const Cand c = shadow_.top();
      shadow_.pop();
      const double *gi = g_est_.find(c.key);
      if (gi && *gi < c.g) continue;                                    // superseded by a better path
      if (c.key == key || asked_.count(c.key) || cache_.count(c.key)) continue;  // already served / cached
      gk.push_back(c.key);
      gg.push_back(c.g);
      rows.insert(rows.end(), &rows_[(size_t)c.row * F], &rows_[(size_t)c.row * F] + F);
    }
    const int64_t n = (int64_t)gk.size(), slots = n * nU;
    nodes_fm_.resize((size_t)F * n);
    for (int64_t k = 0; k < n; k++)
      for (int f = 0; f < F; f++) nodes_fm_[(size_t)f * n + k] = rows[(size_t)k * F + f];
    l_count_.resize((size_t)n);
    l_act_.resize((size_t)slots);
    l_cost_.resize((size_t)slots);
    l_hash_.resize((size_t)slots);
    l_state_.resize((size_t)F * slots);
    mplx_succ_lists o{};
    o.count = l_count_.data();
    o.action = l_act_.data();
    o.cost = l_cost_.data();
    o.hash = l_hash_.data();
    o.state = l_state_.data();
    o.state_stride = slots;
    launches_++;
    {
      const auto t0 = std::chrono::steady_clock::now();
      const int rc = mplx_expand_lists(ctx_, nodes_fm_.data(), n, n, &o);
      t_launch_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (rc != MPLX_OK) return complain();
    }
    for (int64_t k = 0; k < n; k++) {
      Cached &c = cache_[gk[(size_t)k]];
      const int32_t m = l_count_[(size_t)k];
      const int64_t base = k * nU;
      c.g = gg[(size_t)k];
      c.succ.resize((size_t)m * F);
      for (int f = 0; f < F; f++) {
        const double *row = l_state_.data() + (size_t)f * slots + base;
        for (int32_t j = 0; j < m; j++) c.succ[(size_t)j * F + f] = row[j];
      }
      c.cost.assign(l_cost_.begin() + base, l_cost_.begin() + base + m);
      c.act.assign(l_act_.begin() + base, l_act_.begin() + base + m);
      c.keys.assign(l_hash_.begin() + base, l_hash_.begin() + base + m);
    }
    (void)control;
    return true;
  }

  static void pack(const Waypoint<Dim> &w, double *row) {
    for (int i = 0; i < Dim; i++) {
      row[0 * Dim + i] = w.pos(i);
      row[1 * Dim + i] = w.vel(i);
      row[2 * Dim + i] = w.acc(i);
      row[3 * Dim + i] = w.jrk(i);
    }
    row[4 * Dim] = w.yaw;
    row[4 * Dim + 1] = w.t;
  }
  static void unpack(const double *row, Waypoint<Dim> &w) {
    for (int i = 0; i < Dim; i++) {
      w.pos(i) = row[0 * Dim + i];
      w.vel(i) = row[1 * Dim + i];
      w.acc(i) = row[2 * Dim + i];
      w.jrk(i) = row[3 * Dim + i];
    }
    w.yaw = row[4 * Dim];
    w.t = row[4 * Dim + 1];
  }

  // env_base::set_search_region is not virtual (env_base.h:301-303), so a region installed behind the adapter's back
  // is noticed by a fingerprint (size + 1024 strided probes, ~1 us) taken at every expansion.  A plan() marks the maps
  // stale anyway (is_free(start)); an edit that changes none of the probed bits needs notify_map_changed().
  uint64_t region_fingerprint() const {
    const std::vector<bool> &r = this->search_region_;
    uint64_t h = r.size() * 0x9e3779b97f4a7c15ULL;
    if (r.empty()) return h;
    const size_t step = r.size() / 1024 + 1;
    for (size_t i = 0; i < r.size(); i += step) h = (h << 1 | h >> 63) ^ (r[i] ? 0xd6e8feb86659fd93ULL : 0x1ULL);
    return h;
  }

  // MapUtil::map_ without the copy getMap() makes: a pointer to the protected member, formed through a derived class
  // (the standard's own rule for protected access), applied to the MapUtil itself
  struct MapPeek : MapUtil<Dim> {
    static const Tmap &of(const MapUtil<Dim> &m) { return m.*(&MapPeek::map_); }
  };
  const Tmap &peek_map() const { return MapPeek::of(*this->map_util_); }

  bool sync_maps() const {
    {
      const uint64_t fp = region_fingerprint();
      if (fp != region_fp_) maps_stale_ = true;
      region_fp_ = fp;
    }
    if (maps_stale_) {
      drop_cache();
      const Veci<Dim> dim = this->map_util_->getDim();
      const Vecf<Dim> ori = this->map_util_->getOrigin();
      const Tmap &cells = peek_map();  // (MapUtil::getMap returns a copy: 128 MiB at 512^3)
      int32_t d[3] = {1, 1, 1};
      double o[3] = {0, 0, 0};
      for (int i = 0; i < Dim; i++) { d[i] = dim(i); o[i] = ori(i); }
      if (mplx_set_map(ctx_, cells.data(), d, o, this->map_util_->getRes()) != MPLX_OK) return complain();
      if (mplx_set_potential(ctx_, this->potential_map_.empty() ? nullptr : this->potential_map_.data()) != MPLX_OK)
        return complain();
      if (this->search_region_.empty()) {
        if (mplx_set_region(ctx_, nullptr) != MPLX_OK) return complain();
      } else {
        std::vector<uint8_t> bytes(this->search_region_.size());
        for (size_t i = 0; i < bytes.size(); i++) bytes[i] = this->search_region_[i] ? 1 : 0;
        if (mplx_set_region(ctx_, bytes.data()) != MPLX_OK) return complain();
      }
      maps_stale_ = false;
    }
    return true;
  }

  bool sync(int control) const {
    if (!sync_maps()) return false;
    mplx_params p{};
    p.control = control;
    p.dt = this->dt_;
    p.w = this->w_;
    p.wyaw = this->wyaw_;
    p.v_max = this->v_max_;
    p.a_max = this->a_max_;
    p.j_max = this->j_max_;
    p.yaw_max = this->yaw_max_;
    p.potential_weight = this->potential_weight_;
    p.gradient_weight = this->gradient_weight_;
    if (!have_params_ || std::memcmp(&p, &params_, sizeof p) != 0) {
      drop_cache();
      if (mplx_set_params(ctx_, &p) != MPLX_OK) return complain();
      params_ = p;
      have_params_ = true;
    }
    const int nU = (int)this->U_.size();
    const int udim = nU ? (int)this->U_[0].size() : 0;
    flatU_.resize((size_t)nU * udim);
    for (int i = 0; i < nU; i++)
      for (int k = 0; k < udim; k++) flatU_[(size_t)i * udim + k] = this->U_[i](k);
    if (flatU_ != sentU_) {
      drop_cache();
      if (nU == 0 || mplx_set_controls(ctx_, flatU_.data(), nU, udim) != MPLX_OK) return complain();
      sentU_ = flatU_;
    }
    return true;
  }
  bool complain() const { return latch(mplx_last_error(ctx_)); }
  bool latch(const char *msg) const {
    printf(ANSI_COLOR_RED "[env_map_hip] %s\n" ANSI_COLOR_RESET, msg);
    if (first_error_.empty()) first_error_ = msg;
    return false;
  }

  mplx_ctx *ctx_ = nullptr;
  mutable std::string first_error_;
  mutable uint64_t region_fp_ = 0;
  mutable bool maps_stale_ = true, have_params_ = false;
  mutable bool cells_synced_ = false;  // edit_cells has patched the device copy since the last plan() started
  mutable mplx_params params_{};
  mutable std::vector<double> flatU_, sentU_, buf_succ_, buf_cost_;
  mutable std::vector<int32_t> buf_act_;
  // speculative batching
  int batch_ = 1;
  int record_edges_ = -1;  // -1: automatic (on without batching)
  mutable int launches_ = 0;
  mutable double t_total_ = 0, t_launch_ = 0;  // MPLX_ADAPTER_TIMING diagnostics
  mutable std::unordered_map<uint64_t, Cached> cache_;
  mutable GMap g_est_;
  mutable std::unordered_map<uint64_t, bool> asked_;
  mutable std::priority_queue<Cand> shadow_;
  mutable std::vector<double> rows_;  // states of the candidates in shadow_
  mutable Waypoint<Dim> tn_scratch_;
  mutable std::vector<double> nodes_fm_, l_cost_, l_state_;
  mutable std::vector<int32_t> l_count_, l_act_;
  mutable std::vector<uint64_t> l_hash_;
};

/// MapPlanner whose ENV_ expands successors on the GPU; everything else is the
/// reference's MapPlanner (setSearchRegion, updatePotentialMap, iterativePlan,
/// plan, getTraj, ...).
template <int Dim>
class GpuMapPlanner : public MapPlanner<Dim> {
 public:
  explicit GpuMapPlanner(bool verbose = false, int device = 0, int batch = 1)
      : MapPlanner<Dim>(verbose), device_(device), batch_(batch) {}

  void setMapUtil(const std::shared_ptr<MapUtil<Dim>> &map_util) override {
    env_map_hip<Dim> *env = new env_map_hip<Dim>(map_util, device_);
    env->set_batch(batch_);
    this->ENV_.reset(env);
    this->map_util_ = map_util;
  }
  /// Nodes per device launch of the speculative batching (see the file header); 1 = off.
  void setBatch(int n) {
    batch_ = n;
    if (this->ENV_) static_cast<env_map_hip<Dim> *>(this->ENV_.get())->set_batch(n);
  }
  int deviceLaunches() const { return this->ENV_ ? static_cast<env_map_hip<Dim> *>(this->ENV_.get())->launches() : 0; }

  /// Shadows MapPlanner::updatePotentialMap (not virtual in the reference): same result, the
  /// dilation runs on the GPU (mplx_update_potential_map).
  void updatePotentialMap(const Vecf<Dim> &pos) {
    env_map_hip<Dim> *env = static_cast<env_map_hip<Dim> *>(this->ENV_.get());
    env->device_update_potential_map(pos, this->potential_radius_, this->potential_map_range_, this->pow_,
                                     this->map_util_);
  }

  /// Shadows MapPlanner::setSearchRegion: the tunnel mask is built on the GPU.  (iterativePlan
  /// calls the base version through the base class; that still reaches the device through
  /// env_base::set_search_region.)
  void setSearchRegion(const vec_Vecf<Dim> &path, bool dense = false) {
    env_map_hip<Dim> *env = static_cast<env_map_hip<Dim> *>(this->ENV_.get());
    env->device_set_search_region(path, dense, this->search_radius_);
  }
  void set_pow(decimal_t p) { this->pow_ = p; }  // the reference has no setter for pow_ (map_planner.h:113)

  /// Device failures of the env since the last clear (see env_map_hip::device_ok).
  bool deviceOk() const { return this->ENV_ && env()->device_ok(); }
  const std::string &deviceError() const { return env()->device_error(); }

  /// Shadows PlannerBase::plan (not virtual): the reference's plan, then the check a CPU env never needs -- a
  /// device failure during the search (empty successor lists) must not pass for "no trajectory exists".
  bool plan(const Waypoint<Dim> &start, const Waypoint<Dim> &goal) {
    const bool ok = MapPlanner<Dim>::plan(start, goal);
    if (!deviceOk()) {
      printf(ANSI_COLOR_RED "[GpuMapPlanner] device error during plan(): %s\n" ANSI_COLOR_RESET, deviceError().c_str());
      return false;
    }
    return ok;
  }

  /// Shadows MapPlanner::getLinkedNodes (map_planner.cpp:125-157): the voxel -> edge table lhm_ and the list of
  /// linked cell centres, from ONE batched device call over every stored edge instead of a Primitive::sample per
  /// edge.  Edges are visited in the reference's own order (hm_ iteration, then pred index), so lhm_'s per-cell
  /// vectors come out in the same order and updateBlockedNodes / updateClearedNodes behave identically.
  vec_Vecf<Dim> getLinkedNodes() const {
    constexpr int F = 4 * Dim + 2;
    this->lhm_.clear();
    vec_Vecf<Dim> linked_pts;
    struct EdgeRef { const Waypoint<Dim> *node; const Waypoint<Dim> *parent; int i, action; };
    std::vector<EdgeRef> edges;
    for (const auto &it : this->ss_ptr_->hm_) {
      if (!it.second) continue;
      for (unsigned int i = 0; i < it.second->pred_coord.size(); i++)
        edges.push_back({&it.second->coord, &this->ss_ptr_->hm_[it.second->pred_coord[i]]->coord, (int)i,
                         it.second->pred_action_id[i]});
    }
    const size_t n = edges.size();
    if (n == 0) return linked_pts;
    std::vector<double> parents((size_t)F * n);
    std::vector<int32_t> actions(n), cells, count;
    std::vector<uint8_t> outside;
    for (size_t e = 0; e < n; e++) {
      double row[F];
      env_map_hip<Dim>::pack_row(*edges[e].parent, row);
      for (int f = 0; f < F; f++) parents[(size_t)f * n + e] = row[f];
      actions[e] = edges[e].action;
    }
    int cap = 0;
    if (!env()->device_check_edges(edges[0].parent->control, parents, actions, nullptr, nullptr, &cells, &count, &cap, &outside))
      return linked_pts;
    const Veci<Dim> dim = this->map_util_->getDim();
    for (size_t e = 0; e < n; e++) {
      if (outside[e]) {
        // a sample left the map: the cell centres of outside cells cannot be told from the (reference-identical)
        // int index alone, so this edge's points follow the reference's own statements (map_planner.cpp:140-153)
        Primitive<Dim> pr;
        this->ENV_->forward_action(*edges[e].parent, edges[e].action, pr);
        decimal_t max_v = 0;
        for (int k = 0; k < Dim; k++) max_v = std::max(max_v, pr.max_vel(k));
        const int ns = 1.0 * std::ceil(max_v * pr.t() / this->map_util_->getRes());
        int prev_id = -1;
        for (const auto &w : pr.sample(ns)) {
          const int id = this->map_util_->getIndex(this->map_util_->floatToInt(w.pos));
          if (id != prev_id) {
            linked_pts.push_back(this->map_util_->intToFloat(this->map_util_->floatToInt(w.pos)));
            this->lhm_[id].push_back(std::make_pair(*edges[e].node, edges[e].i));
            prev_id = id;
          }
        }
        continue;
      }
      for (int k = 0; k < count[e]; k++) {
        const int id = cells[e * (size_t)cap + k];
        Veci<Dim> pn;
        int rem = id;
        for (int a = 0; a < Dim; a++) {
          pn(a) = a + 1 < Dim ? rem % dim(a) : rem;
          rem /= dim(a);
        }
        linked_pts.push_back(this->map_util_->intToFloat(pn));
        this->lhm_[id].push_back(std::make_pair(*edges[e].node, edges[e].i));
      }
    }
    return linked_pts;
  }

  /// Shadows MapPlanner::updateBlockedNodes (map_planner.cpp:160-171): the reference's host bookkeeping unchanged
  /// (increaseCost over lhm_), plus the cells themselves -- the application has just blocked them in the MapUtil --
  /// patched into the device copy (env_map_hip::edit_cells) instead of a whole-map upload at the next plan().
  void updateBlockedNodes(const vec_Veci<Dim> &blocked_pns) {
    MapPlanner<Dim>::updateBlockedNodes(blocked_pns);
    env()->edit_cells(blocked_pns);
  }
  /// Host -> device bytes of the env's map calls so far (a re-plan after an edit of k cells moves 9 k).
  uint64_t mapUploadBytes() const { return this->ENV_ ? env()->map_upload_bytes() : 0; }

  /// Shadows MapPlanner::updateClearedNodes (map_planner.cpp:174-185) + StateSpace::decreaseCost
  /// (state_space.h:230-253): the affected edges are re-validated in ONE device call (is_free(Primitive) and
  /// calculate_intrinsic_cost), then the state space is updated by the reference's own statements in the
  /// reference's own order.  The map edit must already be in the MapUtil (as for the reference).
  void updateClearedNodes(const vec_Veci<Dim> &cleared_pns) {
    constexpr int F = 4 * Dim + 2;
    std::vector<std::pair<Waypoint<Dim>, int>> cleared_nodes;
    for (const auto &it : cleared_pns) {
      const int id = this->map_util_->getIndex(it);
      auto search = this->lhm_.find(id);
      if (search != this->lhm_.end())
        for (const auto &node : search->second) cleared_nodes.push_back(node);
    }
    // the cells were cleared in the MapUtil: patched into the device copy (not a whole-map upload) before the edges
    // through them are re-validated there
    env()->edit_cells(cleared_pns);
    const size_t n = cleared_nodes.size();
    if (n == 0) return;
    auto &hm = this->ss_ptr_->hm_;
    std::vector<double> parents((size_t)F * n);
    std::vector<int32_t> actions(n);
    for (size_t e = 0; e < n; e++) {
      const auto &succ = hm[cleared_nodes[e].first];
      const int i = cleared_nodes[e].second;
      double row[F];
      env_map_hip<Dim>::pack_row(succ->pred_coord[i], row);  // decreaseCost forwards from the stored parent coord
      for (int f = 0; f < F; f++) parents[(size_t)f * n + e] = row[f];
      actions[e] = succ->pred_action_id[i];
    }
    std::vector<uint8_t> free_flag;
    std::vector<double> cost;
    if (!env()->device_check_edges(cleared_nodes[0].first.control, parents, actions, &free_flag, &cost, nullptr, nullptr,
                                   nullptr, nullptr))
      return;
    for (size_t e = 0; e < n; e++) {  // state_space.h:232-252
      StatePtr<Waypoint<Dim>> &succNode_ptr = hm[cleared_nodes[e].first];
      const int i = cleared_nodes[e].second;
      if (std::isinf(succNode_ptr->pred_action_cost[i])) {
        const Waypoint<Dim> parent_key = succNode_ptr->pred_coord[i];
        if (free_flag[e]) {
          succNode_ptr->pred_action_cost[i] = cost[e];
          this->ss_ptr_->updateNode(succNode_ptr);
          const int succ_act_id = succNode_ptr->pred_action_id[i];
          for (size_t j = 0; j < hm[parent_key]->succ_action_id.size(); j++) {
            if (succ_act_id == hm[parent_key]->succ_action_id[j]) {
              hm[parent_key]->succ_action_cost[j] = succNode_ptr->pred_action_cost[i];
              break;
            }
          }
        }
      }
    }
  }

 private:
  env_map_hip<Dim> *env() const { return static_cast<env_map_hip<Dim> *>(this->ENV_.get()); }
  int device_;
  int batch_ = 1;
};

typedef GpuMapPlanner<2> GpuOccMapPlanner;
typedef GpuMapPlanner<3> GpuVoxelMapPlanner;

}  // namespace MPL
#endif
