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
// Errors never throw: a failed device call prints the engine's message and
// returns an empty successor list (the reference's own error convention is
// printf + sentinel, graph_search.h:149-161).  There is no CPU fallback.
#ifndef MPLX_ENV_MAP_HPP
#define MPLX_ENV_MAP_HPP

#include <mpl_planner/env/env_map.h>
#include <mpl_planner/planner/map_planner.h>

#include <cmath>
#include <cstdio>
#include <cstring>
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
  ~env_map_hip() { mplx_destroy(ctx_); }
  env_map_hip(const env_map_hip &) = delete;
  env_map_hip &operator=(const env_map_hip &) = delete;

  bool ok() const { return ctx_ != nullptr; }
  /// Re-upload the map / potential / region before the next expansion.
  void notify_map_changed() { maps_stale_ = true; }

  /// First virtual call of every PlannerBase::plan (planner_base.h:283).
  bool is_free(const Vecf<Dim> &pt) const override {
    maps_stale_ = true;
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
    succ.clear();
    succ_cost.clear();
    action_idx.clear();
    this->expanded_nodes_.push_back(curr.pos);
    if (!ctx_ || !sync(curr.control)) return;
    constexpr int F = 4 * Dim + 2;
    const int nU = (int)this->U_.size();
    double node[F];
    pack(curr, node);
    buf_succ_.resize((size_t)nU * F);
    buf_cost_.resize((size_t)nU);
    buf_act_.resize((size_t)nU);
    int32_t n = 0;
    if (mplx_get_succ(ctx_, node, buf_succ_.data(), buf_cost_.data(), buf_act_.data(), &n) != MPLX_OK) {
      printf(ANSI_COLOR_RED "[env_map_hip] get_succ: %s\n" ANSI_COLOR_RESET, mplx_last_error(ctx_));
      return;
    }
    for (int m = 0; m < n; m++) {
      Waypoint<Dim> tn(curr.control);
      unpack(&buf_succ_[(size_t)m * F], tn);
      succ.push_back(tn);
      succ_cost.push_back(buf_cost_[(size_t)m]);
      action_idx.push_back(buf_act_[(size_t)m]);
      if (!std::isinf(buf_cost_[(size_t)m]))  // debug side effect of env_map.h:166
        this->expanded_edges_.push_back(Primitive<Dim>(curr, this->U_[buf_act_[(size_t)m]], this->dt_));
    }
  }

  /// MapPlanner::updatePotentialMap on the device (map_planner.cpp:286-391): the map held by
  /// `map_util` is dilated in HBM, read back, installed in the MapUtil and as potential map.
  bool device_update_potential_map(const Vecf<Dim> &pos, const Vecf<Dim> &radius, const Vecf<Dim> &range,
                                   decimal_t pow, const std::shared_ptr<MapUtil<Dim>> &map_util) {
    maps_stale_ = true;
    if (!ctx_ || !sync_maps()) return false;
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
    if (!ctx_ || !sync_maps()) return false;
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

  bool sync_maps() const {
    if (maps_stale_) {
      const Veci<Dim> dim = this->map_util_->getDim();
      const Vecf<Dim> ori = this->map_util_->getOrigin();
      const Tmap cells = this->map_util_->getMap();
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
      if (nU == 0 || mplx_set_controls(ctx_, flatU_.data(), nU, udim) != MPLX_OK) return complain();
      sentU_ = flatU_;
    }
    return true;
  }
  bool complain() const {
    printf(ANSI_COLOR_RED "[env_map_hip] %s\n" ANSI_COLOR_RESET, mplx_last_error(ctx_));
    return false;
  }

  mplx_ctx *ctx_ = nullptr;
  mutable bool maps_stale_ = true, have_params_ = false;
  mutable mplx_params params_{};
  mutable std::vector<double> flatU_, sentU_, buf_succ_, buf_cost_;
  mutable std::vector<int32_t> buf_act_;
};

/// MapPlanner whose ENV_ expands successors on the GPU; everything else is the
/// reference's MapPlanner (setSearchRegion, updatePotentialMap, iterativePlan,
/// plan, getTraj, ...).
template <int Dim>
class GpuMapPlanner : public MapPlanner<Dim> {
 public:
  explicit GpuMapPlanner(bool verbose = false, int device = 0) : MapPlanner<Dim>(verbose), device_(device) {}

  void setMapUtil(const std::shared_ptr<MapUtil<Dim>> &map_util) override {
    this->ENV_.reset(new env_map_hip<Dim>(map_util, device_));
    this->map_util_ = map_util;
  }

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

 private:
  int device_;
};

typedef GpuMapPlanner<2> GpuOccMapPlanner;
typedef GpuMapPlanner<3> GpuVoxelMapPlanner;

}  // namespace MPL
#endif
