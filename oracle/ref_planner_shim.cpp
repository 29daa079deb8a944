/*
 * ref_planner_shim.cpp -- runs the REFERENCE'S OWN MapPlanner<Dim>::plan
 * (PlannerBase::plan + GraphSearch::Astar + StateSpace, unmodified headers and
 * src/mpl_planner/map_planner.cpp compiled where they lie) either
 *   (a) with the reference's env_map<Dim>   -> pins our host search, or
 *   (b) with MPL::GpuMapPlanner<Dim> from include/mplx_env_map.hpp, i.e. the
 *       drop-in adapter over libmplx.so   -> the true drop-in test on a GPU.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Eigen / Boost come from oracle/stub_include
 * (stand-ins, incl. a binary mutable heap for boost::heap::d_ary_heap).  Output:
 * oracle/_ref/libmpl_ref_planner.so (git-ignored, prebuilt file travels to the
 * GPU box; links libmplx.so for variant (b)).
 */
#include <mpl_planner/planner/map_planner.h>

#include <algorithm>
#include <chrono>
#include <memory>

#include "../include/mplx_env_map.hpp"
#include "mpl_oracle.h"

/* the reference's translation unit, from where it lies */
#include "../../reference/src/mpl_planner/map_planner.cpp"

extern "C" {
typedef struct {
  int32_t ok, closed, opened, expansions, segments, hm_size;
  double cost, total_time, J[4];
  double wall_ms;
} mpl_ref_plan_out;
}

namespace {

template <int D>
int run_plan(const mpl_oracle_env *e, const double *start_row, const double *goal_row, int use_gpu,
             double epsilon, int reps, mpl_ref_plan_out *out) {
  std::shared_ptr<MPL::MapUtil<D>> mu = std::make_shared<MPL::MapUtil<D>>();
  Vecf<D> ori;
  Veci<D> dim;
  size_t n = 1;
  for (int i = 0; i < D; i++) { ori(i) = e->origin[i]; dim(i) = e->map_dim[i]; n *= (size_t)e->map_dim[i]; }
  mu->setMap(ori, dim, MPL::Tmap(e->map, e->map + n), e->res);

  std::unique_ptr<MPL::MapPlanner<D>> planner;
  if (use_gpu) planner.reset(new MPL::GpuMapPlanner<D>(false, 0, use_gpu > 1 ? use_gpu : 1));  // use_gpu > 1: speculative batch size
  else planner.reset(new MPL::MapPlanner<D>(false));
  planner->setMapUtil(mu);
  planner->setVmax(e->v_max);
  planner->setAmax(e->a_max);
  planner->setJmax(e->j_max);
  planner->setYawmax(e->yaw_max);
  planner->setDt(e->dt);
  planner->setW(e->w);
  planner->setWyaw(e->wyaw);
  planner->setEpsilon(epsilon);
  vec_E<VecDf> U;
  for (int i = 0; i < e->nU; i++) {
    VecDf u(e->udim);
    for (int k = 0; k < e->udim; k++) u(k) = e->U[(size_t)i * e->udim + k];
    U.push_back(u);
  }
  planner->setU(U);

  auto load = [&](const double *r) {
    Waypoint<D> w((Control::Control)e->control);
    for (int i = 0; i < D; i++) {
      w.pos(i) = r[i]; w.vel(i) = r[D + i]; w.acc(i) = r[2 * D + i]; w.jrk(i) = r[3 * D + i];
    }
    w.yaw = r[4 * D];
    w.t = r[4 * D + 1];
    return w;
  };
  const Waypoint<D> start = load(start_row), goal = load(goal_row);
  bool ok = false;
  double best = 1e300;
  for (int r = 0; r < (reps < 1 ? 1 : reps); r++) {
    auto t0 = std::chrono::steady_clock::now();
    ok = planner->plan(start, goal);
    best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
  out->ok = ok ? 1 : 0;
  out->wall_ms = best;
  out->closed = (int32_t)planner->getCloseSet().size();
  out->opened = (int32_t)planner->getOpenSet().size();
  out->expansions = planner->getExpandedNum();
  const Trajectory<D> traj = planner->getTraj();
  out->segments = (int32_t)traj.getPrimitives().size();
  out->total_time = traj.getTotalTime();
  out->J[0] = traj.J(Control::VEL);
  out->J[1] = traj.J(Control::ACC);
  out->J[2] = traj.J(Control::JRK);
  out->J[3] = traj.J(Control::SNP);
  out->cost = planner->getTrajCost();
  out->hm_size = use_gpu ? static_cast<MPL::GpuMapPlanner<D> *>(planner.get())->deviceLaunches() : 0;  // device launches
  return 0;
}

}  // namespace

extern "C" int mpl_ref_plan(const mpl_oracle_env *env, const double *start, const double *goal, int use_gpu,
                            double epsilon, int reps, mpl_ref_plan_out *out) {
  if (!env || !start || !goal || !out) return -1;
  if (env->dim == 2) return run_plan<2>(env, start, goal, use_gpu, epsilon, reps, out);
  if (env->dim == 3) return run_plan<3>(env, start, goal, use_gpu, epsilon, reps, out);
  return -1;
}

/* ---- the scenarios of the reference's own tests end to end (test/test_distance_map_planner_2d.cpp:48-98,
 * ..._with_yaw.cpp:48-104, ..._iterative.cpp:48-89, test_planner_2d_with_yaw.cpp:29-67,
 * test_planner_2d_with_prior_traj.cpp:29-102): e.g. plan on the raw map, then a second planner with a search
 * region around that trajectory and a potential map; with yaw the second stage is iterativePlan over a control
 * table extended by three yaw rates.  PlannerT = MPL::MapPlanner<D> (CPU) or MPL::GpuMapPlanner<D> (the drop-in:
 * get_succ, updatePotentialMap and setSearchRegion on the device). ---- */
namespace {

template <int D> int planner_launches(MPL::MapPlanner<D> &);
template <int D> int planner_launches(MPL::GpuMapPlanner<D> &);
template <int D> void set_batch(MPL::MapPlanner<D> &, int);
template <int D> void set_batch(MPL::GpuMapPlanner<D> &, int);
template <int D> struct ViaBase;
template <int D> int planner_launches(ViaBase<D> &);
template <int D> void set_batch(ViaBase<D> &, int);

template <int D>
void fill_out(MPL::MapPlanner<D> &pl, bool ok, double ms, int launches, mpl_ref_plan_out *o, double *checksum) {
  o->ok = ok ? 1 : 0;
  o->wall_ms = ms;
  o->closed = (int32_t)pl.getCloseSet().size();
  o->opened = (int32_t)pl.getOpenSet().size();
  o->expansions = pl.getExpandedNum();
  const Trajectory<D> traj = pl.getTraj();
  o->segments = (int32_t)traj.getPrimitives().size();
  o->total_time = traj.getTotalTime();
  o->J[0] = traj.J(Control::VEL);
  o->J[1] = traj.J(Control::ACC);
  o->J[2] = traj.J(Control::JRK);
  o->J[3] = traj.J(Control::SNP);
  o->cost = pl.getTrajCost();
  o->hm_size = launches;
  double c = 0;
  int k = 1;
  for (const auto &w : traj.getWaypoints()) {
    for (int i = 0; i < D; i++) c += k * (w.pos(i) + 3.0 * w.vel(i));
    c += 7.0 * k * w.yaw;
    k++;
  }
  *checksum = c;
}

double g_last_prep_ms = 0;  // wall time of the last stage-2 updatePotentialMap (mpl_ref_last_prep_ms)

/* mode: 0 test_distance_map_planner_2d, 1 ..._with_yaw, 2 ..._iterative, 3 test_planner_2d_with_yaw (one stage),
 *       4 test_planner_2d_with_prior_traj (VEL plan, then JRK-state plan guided by it)
 *       5 the same with a potential map installed in the second planner BEFORE setPriorTrajectory (updatePotentialMap,
 *         potential weight 0.5, gradient weight = env->gradient_weight): env_map::set_prior_trajectory /
 *         traverse_trajectory with potential_map_ (env_map.h:197-216, 241-249) */
template <int D, class PlannerT>
int run_scenario(const mpl_oracle_env *e, const double *start_row, const double *goal_row, int mode, int batch,
                 mpl_ref_plan_out *out, double *checksum, int64_t *region_cells, int64_t *potential_sum) {
  std::shared_ptr<MPL::MapUtil<D>> mu = std::make_shared<MPL::MapUtil<D>>();
  Vecf<D> ori;
  Veci<D> dim;
  size_t n = 1;
  for (int i = 0; i < D; i++) { ori(i) = e->origin[i]; dim(i) = e->map_dim[i]; n *= (size_t)e->map_dim[i]; }
  mu->setMap(ori, dim, MPL::Tmap(e->map, e->map + n), e->res);
  vec_E<VecDf> U, U_yaw, U_unit;
  for (int i = 0; i < e->nU; i++) {
    VecDf u(D), uy(D + 1), u1(D);
    for (int k = 0; k < D; k++) { u(k) = uy(k) = e->U[(size_t)i * e->udim + k]; u1(k) = 2.0 * u(k); }
    U.push_back(u);
    U_unit.push_back(u1);  // {-1, 0, 1}^D when U is {-0.5, 0, 0.5}^D (test_planner_2d_with_prior_traj.cpp:47-52)
    for (int y = -1; y <= 1; y++) { uy(D) = 0.5 * y; U_yaw.push_back(uy); }
  }
  auto load = [&](const double *r, int control) {
    Waypoint<D> w((Control::Control)control);
    for (int i = 0; i < D; i++) {
      w.pos(i) = r[i]; w.vel(i) = r[D + i]; w.acc(i) = r[2 * D + i]; w.jrk(i) = r[3 * D + i];
    }
    w.yaw = r[4 * D];
    w.t = r[4 * D + 1];
    return w;
  };
  const int base = e->control & 0x0f;
  *region_cells = 0;
  *potential_sum = 0;
  out[0] = out[1] = mpl_ref_plan_out{};
  checksum[0] = checksum[1] = 0;
  auto make = [&](PlannerT &pl, const vec_E<VecDf> &u) {
    pl.setMapUtil(mu);
    pl.setVmax(e->v_max);
    pl.setAmax(e->a_max);
    pl.setDt(e->dt);
    pl.setU(u);
  };
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms_since = [&](std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(now() - t0).count();
  };

  if (mode == 3) {
    // test_planner_2d_with_yaw.cpp:29-67: ACCxYAW from yaw = pi/2, yaw_max = 0.7
    Waypoint<D> start = load(start_row, base | 0x10);
    start.yaw = M_PI / 2;
    Waypoint<D> goal = load(goal_row, base | 0x10);
    PlannerT pl(false);
    set_batch(pl, batch);
    make(pl, U_yaw);
    pl.setYawmax(0.7);
    auto t0 = now();
    const bool ok = pl.plan(start, goal);
    fill_out<D>(pl, ok, ms_since(t0), planner_launches(pl), &out[0], &checksum[0]);
    return 0;
  }
  if (mode == 4 || mode == 5) {
    // test_planner_2d_with_prior_traj.cpp:29-102
    Waypoint<D> start = load(start_row, 0x01), goal = load(goal_row, 0x01);
    PlannerT first(false);
    set_batch(first, batch);
    make(first, U_unit);
    auto t0 = now();
    bool ok = first.plan(start, goal);
    fill_out<D>(first, ok, ms_since(t0), planner_launches(first), &out[0], &checksum[0]);
    if (!ok) return 0;
    const Trajectory<D> prior = first.getTraj();
    start.use_vel = true;
    start.use_acc = true;  // goal keeps the VEL flag (…prior_traj.cpp:45)
    PlannerT second(false);
    set_batch(second, batch);
    second.setMapUtil(mu);
    second.setEpsilon(1.0);
    second.setVmax(e->v_max);
    second.setAmax(e->a_max);
    second.setDt(e->dt);
    second.setW(10);
    second.setU(U);
    second.setTol(0.5);
    if (mode == 5) {
      Vecf<D> rad;
      for (int i = 0; i < D; i++) rad(i) = 1.0;
      second.setPotentialRadius(rad);
      second.setPotentialWeight(0.5);
      second.setGradientWeight(e->gradient_weight);
      second.updatePotentialMap(start.pos);
      int64_t ps = 0;
      for (const auto v : mu->getMap()) ps += v;
      *potential_sum = ps;
    }
    second.setPriorTrajectory(prior);
    t0 = now();
    ok = second.plan(start, goal);
    fill_out<D>(second, ok, ms_since(t0), planner_launches(second), &out[1], &checksum[1]);
    return 0;
  }

  const bool with_yaw = mode == 1;
  Waypoint<D> start = load(start_row, base);
  const Waypoint<D> goal = load(goal_row, base);  // the goal keeps the control without yaw (…_with_yaw.cpp:47, 97)
  // stage 1
  PlannerT first(false);
  set_batch(first, batch);
  make(first, U);
  auto t0 = now();
  bool ok = first.plan(start, goal);
  fill_out<D>(first, ok, ms_since(t0), planner_launches(first), &out[0], &checksum[0]);
  if (!ok) return 0;
  const Trajectory<D> traj = first.getTraj();
  vec_Vecf<D> path;
  for (const auto &w : traj.getWaypoints()) path.push_back(w.pos);
  // stage 2
  PlannerT second(false);
  set_batch(second, batch);
  make(second, with_yaw ? U_yaw : U);
  second.setEpsilon(1.0);
  Vecf<D> rad;
  for (int i = 0; i < D; i++) rad(i) = 0.5;
  second.setSearchRadius(rad);
  if (mode == 0) second.setSearchRegion(path);
  for (int i = 0; i < D; i++) rad(i) = 1.0;
  second.setPotentialRadius(rad);
  second.setPotentialWeight(0.5);
  second.setGradientWeight(0);
  t0 = now();
  second.updatePotentialMap(start.pos);
  g_last_prep_ms = ms_since(t0);
  t0 = now();
  if (with_yaw) {
    start.use_yaw = true;
    second.setYawmax(0.5);
  }
  if (mode == 0) ok = second.plan(start, goal);
  else ok = second.iterativePlan(start, goal, traj, 10);
  fill_out<D>(second, ok, ms_since(t0), planner_launches(second), &out[1], &checksum[1]);
  *region_cells = (int64_t)second.getSearchRegion().size();
  int64_t ps = 0;
  for (const auto v : mu->getMap()) ps += v;  // updatePotentialMap rewrites the MapUtil's map (map_planner.cpp:387)
  *potential_sum = ps;
  return 0;
}

/* The drop-in held the way an application that only swaps the constructor holds it: through a MapPlanner<D>
 * pointer.  plan / setSearchRegion / updatePotentialMap / iterativePlan are NOT virtual in the reference
 * (map_planner.h, planner_base.h), so through this pointer the BASE versions run -- host preprocessing, then the
 * virtual / fingerprinted hand-over to the device env (env_map_hip::set_potential_map, is_free, region fingerprint)
 * -- and only get_succ is the device's.  INTEGRATION.md states which calls need the derived type; this type makes
 * the run_scenario template exercise exactly the base-pointer route. */
template <int D>
struct ViaBase {
  std::unique_ptr<MPL::MapPlanner<D>> p;
  explicit ViaBase(bool verbose) : p(new MPL::GpuMapPlanner<D>(verbose, 0, 1)) {}
  operator MPL::MapPlanner<D> &() { return *p; }
  void setMapUtil(const std::shared_ptr<MPL::MapUtil<D>> &mu) { p->setMapUtil(mu); }  // virtual: installs env_map_hip
  void setVmax(decimal_t v) { p->setVmax(v); }
  void setAmax(decimal_t v) { p->setAmax(v); }
  void setDt(decimal_t v) { p->setDt(v); }
  void setW(decimal_t v) { p->setW(v); }
  void setEpsilon(decimal_t v) { p->setEpsilon(v); }
  void setYawmax(decimal_t v) { p->setYawmax(v); }
  void setTol(decimal_t v) { p->setTol(v); }
  void setU(const vec_E<VecDf> &u) { p->setU(u); }
  void setPriorTrajectory(const Trajectory<D> &t) { p->setPriorTrajectory(t); }
  void setSearchRadius(const Vecf<D> &r) { p->setSearchRadius(r); }
  void setSearchRegion(const vec_Vecf<D> &path) { p->setSearchRegion(path); }   // BASE version (host ray trace)
  void setPotentialRadius(const Vecf<D> &r) { p->setPotentialRadius(r); }
  void setPotentialWeight(decimal_t w) { p->setPotentialWeight(w); }
  void setGradientWeight(decimal_t w) { p->setGradientWeight(w); }
  void updatePotentialMap(const Vecf<D> &pos) { p->updatePotentialMap(pos); }   // BASE version (host scatter)
  bool plan(const Waypoint<D> &s, const Waypoint<D> &g) { return p->plan(s, g); }  // BASE version (no device check)
  bool iterativePlan(const Waypoint<D> &s, const Waypoint<D> &g, const Trajectory<D> &t, int n) {
    return p->iterativePlan(s, g, t, n);
  }
  Trajectory<D> getTraj() const { return p->getTraj(); }
  vec_Vecf<D> getSearchRegion() const { return p->getSearchRegion(); }
  MPL::GpuMapPlanner<D> &derived() { return *static_cast<MPL::GpuMapPlanner<D> *>(p.get()); }
};
template <int D> int planner_launches(ViaBase<D> &p) { return p.derived().deviceLaunches(); }
template <int D> void set_batch(ViaBase<D> &p, int b) { p.derived().setBatch(b > 1 ? b : 1); }

template <int D>
int planner_launches(MPL::MapPlanner<D> &) { return 0; }
template <int D>
int planner_launches(MPL::GpuMapPlanner<D> &p) { return p.deviceLaunches(); }
template <int D>
void set_batch(MPL::MapPlanner<D> &, int) {}
template <int D>
void set_batch(MPL::GpuMapPlanner<D> &p, int b) { p.setBatch(b > 1 ? b : 1); }

}  // namespace

extern "C" int mpl_ref_scenario(const mpl_oracle_env *env, const double *start, const double *goal, int use_gpu,
                                int mode, mpl_ref_plan_out *out2, double *checksum2, int64_t *region_cells,
                                int64_t *potential_sum) {
  if (!env || !start || !goal || !out2 || (env->dim != 2 && env->dim != 3) || mode < 0 || mode > 5) return -1;
  if (env->dim == 3) {
    // the same flows on a voxel map (BASELINE config 5's planner: potential map + ACCxYAW, 3^3 x 3 yaw rates = 81 controls)
    if (use_gpu)
      return run_scenario<3, MPL::GpuMapPlanner<3>>(env, start, goal, mode, use_gpu, out2, checksum2, region_cells,
                                                    potential_sum);
    return run_scenario<3, MPL::MapPlanner<3>>(env, start, goal, mode, 1, out2, checksum2, region_cells, potential_sum);
  }
  if (use_gpu)
    return run_scenario<2, MPL::GpuMapPlanner<2>>(env, start, goal, mode, use_gpu, out2, checksum2, region_cells,
                                                  potential_sum);
  return run_scenario<2, MPL::MapPlanner<2>>(env, start, goal, mode, 1, out2, checksum2, region_cells, potential_sum);
}

/* ---- map preprocessing through the reference's own MapPlanner (map_planner.cpp:46-95, 246-391) ---- */
namespace {

template <int D>
struct PrepPlanner : MPL::MapPlanner<D> {
  PrepPlanner() : MPL::MapPlanner<D>(false) {}
  void set_pow(double p) { this->pow_ = p; }
};

template <int D>
std::shared_ptr<MPL::MapUtil<D>> make_map(const int8_t *cells, const int32_t *map_dim, const double *origin, double res) {
  auto mu = std::make_shared<MPL::MapUtil<D>>();
  Vecf<D> ori;
  Veci<D> dim;
  size_t n = 1;
  for (int i = 0; i < D; i++) { ori(i) = origin[i]; dim(i) = map_dim[i]; n *= (size_t)map_dim[i]; }
  if (cells) mu->setMap(ori, dim, MPL::Tmap(cells, cells + n), res);
  else mu->setMap(ori, dim, MPL::Tmap(n, 0), res);
  return mu;
}

template <int D, class Planner>
int ref_potential(const int8_t *map_in, const int32_t *map_dim, const double *origin, double res, const double *pos,
                  const double *radius, const double *range, double pow_, int8_t *map_out) {
  auto mu = make_map<D>(map_in, map_dim, origin, res);
  Planner pl;
  pl.setMapUtil(mu);
  Vecf<D> r, g, p;
  for (int i = 0; i < D; i++) { r(i) = radius[i]; g(i) = range[i]; p(i) = pos[i]; }
  pl.setPotentialRadius(r);
  pl.setPotentialMapRange(g);
  pl.set_pow(pow_);
  pl.updatePotentialMap(p);
  const auto m = mu->getMap();
  std::copy(m.begin(), m.end(), map_out);
  return 0;
}

template <int D, class Planner>
int ref_region(const int32_t *map_dim, const double *origin, double res, const double *path, int n_points, int dense,
               const double *search_radius, uint8_t *region_out) {
  auto mu = make_map<D>(nullptr, map_dim, origin, res);
  Planner pl;
  pl.setMapUtil(mu);
  Vecf<D> sr;
  for (int i = 0; i < D; i++) sr(i) = search_radius[i];
  pl.setSearchRadius(sr);
  vec_Vecf<D> pts;
  for (int k = 0; k < n_points; k++) {
    Vecf<D> p;
    for (int i = 0; i < D; i++) p(i) = path[(size_t)k * D + i];
    pts.push_back(p);
  }
  pl.setSearchRegion(pts, dense != 0);
  /* getSearchRegion returns the in-region cell centres (map_planner.cpp:98-122) */
  size_t n = 1;
  for (int i = 0; i < D; i++) n *= (size_t)map_dim[i];
  std::fill(region_out, region_out + n, (uint8_t)0);
  for (const auto &c : pl.getSearchRegion()) region_out[mu->getIndex(mu->floatToInt(c))] = 1;
  return 0;
}

}  // namespace

extern "C" int mpl_ref_update_potential_map(int32_t dim, const int8_t *map_in, const int32_t *map_dim,
                                            const double *origin, double res, const double *pos, const double *radius,
                                            const double *range, double pow_, int8_t *map_out) {
  if (dim == 2) return ref_potential<2, PrepPlanner<2>>(map_in, map_dim, origin, res, pos, radius, range, pow_, map_out);
  if (dim == 3) return ref_potential<3, PrepPlanner<3>>(map_in, map_dim, origin, res, pos, radius, range, pow_, map_out);
  return -1;
}

/* the same calls on MPL::GpuMapPlanner (include/mplx_env_map.hpp): the drop-in adapter, needs a GPU */
extern "C" int mpl_gpu_update_potential_map(int32_t dim, const int8_t *map_in, const int32_t *map_dim,
                                            const double *origin, double res, const double *pos, const double *radius,
                                            const double *range, double pow_, int8_t *map_out) {
  if (dim == 2) return ref_potential<2, MPL::GpuMapPlanner<2>>(map_in, map_dim, origin, res, pos, radius, range, pow_, map_out);
  if (dim == 3) return ref_potential<3, MPL::GpuMapPlanner<3>>(map_in, map_dim, origin, res, pos, radius, range, pow_, map_out);
  return -1;
}

extern "C" int mpl_gpu_search_region(int32_t dim, const int32_t *map_dim, const double *origin, double res,
                                     const double *path, int32_t n_points, int32_t dense, const double *search_radius,
                                     uint8_t *region_out) {
  if (dim == 2) return ref_region<2, MPL::GpuMapPlanner<2>>(map_dim, origin, res, path, n_points, dense, search_radius, region_out);
  if (dim == 3) return ref_region<3, MPL::GpuMapPlanner<3>>(map_dim, origin, res, path, n_points, dense, search_radius, region_out);
  return -1;
}

extern "C" int mpl_ref_search_region(int32_t dim, const int32_t *map_dim, const double *origin, double res,
                                     const double *path, int32_t n_points, int32_t dense, const double *search_radius,
                                     uint8_t *region_out) {
  if (dim == 2) return ref_region<2, PrepPlanner<2>>(map_dim, origin, res, path, n_points, dense, search_radius, region_out);
  if (dim == 3) return ref_region<3, PrepPlanner<3>>(map_dim, origin, res, path, n_points, dense, search_radius, region_out);
  return -1;
}

/* ---- incremental re-planning: the reference's LPA* (PlannerBase::setLPAstar, graph_search.h:194-365) with a map
 * edit between plans, through MapPlanner::getLinkedNodes / updateBlockedNodes / updateClearedNodes
 * (map_planner.cpp:125-185) -- on MPL::MapPlanner (CPU) or MPL::GpuMapPlanner (edge work batched on the device). ---- */
namespace {

template <class Base>
struct PeekPlanner : Base {  // lhm_ is protected
  PeekPlanner() : Base(false) {}
  using Base::lhm_;
};

// host -> device bytes the planner's env has moved for its maps (MPL::GpuMapPlanner::mapUploadBytes; 0 for the CPU planner)
template <class P>
auto upload_bytes(const P &p, int) -> decltype(p.mapUploadBytes()) { return p.mapUploadBytes(); }
template <class P>
uint64_t upload_bytes(const P &, long) { return 0; }

template <int D, class PlannerT>
int run_lpastar(const mpl_oracle_env *e, const double *start_row, const double *goal_row, int box_half,
                mpl_ref_plan_out *out3, double *checksum3, int64_t *stats /* [10] */) {
  std::shared_ptr<MPL::MapUtil<D>> mu = std::make_shared<MPL::MapUtil<D>>();
  Vecf<D> ori;
  Veci<D> dim;
  size_t n = 1;
  for (int i = 0; i < D; i++) { ori(i) = e->origin[i]; dim(i) = e->map_dim[i]; n *= (size_t)e->map_dim[i]; }
  MPL::Tmap cells(e->map, e->map + n);
  mu->setMap(ori, dim, cells, e->res);
  vec_E<VecDf> U;
  for (int i = 0; i < e->nU; i++) {
    VecDf u(e->udim);
    for (int k = 0; k < e->udim; k++) u(k) = e->U[(size_t)i * e->udim + k];
    U.push_back(u);
  }
  auto load = [&](const double *r) {
    Waypoint<D> w((Control::Control)e->control);
    for (int i = 0; i < D; i++) { w.pos(i) = r[i]; w.vel(i) = r[D + i]; w.acc(i) = r[2 * D + i]; w.jrk(i) = r[3 * D + i]; }
    w.yaw = r[4 * D];
    w.t = r[4 * D + 1];
    return w;
  };
  const Waypoint<D> start = load(start_row), goal = load(goal_row);
  PlannerT pl;
  pl.setMapUtil(mu);
  pl.setVmax(e->v_max);
  pl.setAmax(e->a_max);
  pl.setJmax(e->j_max);
  pl.setDt(e->dt);
  pl.setW(e->w);
  pl.setEpsilon(1.0);
  pl.setU(U);
  pl.setLPAstar(true);
  for (int i = 0; i < 3; i++) { out3[i] = mpl_ref_plan_out{}; checksum3[i] = 0; }
  for (int i = 0; i < 10; i++) stats[i] = 0;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms_since = [&](std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(now() - t0).count();
  };
  // ---- plan 1
  auto t0 = now();
  bool ok = pl.plan(start, goal);
  fill_out<D>(pl, ok, ms_since(t0), 0, &out3[0], &checksum3[0]);
  if (!ok) return 0;
  // ---- the voxel -> edge table
  t0 = now();
  const vec_Vecf<D> linked = pl.getLinkedNodes();
  stats[6] = (int64_t)(ms_since(t0) * 1000.0);  // microseconds
  stats[0] = (int64_t)pl.lhm_.size();
  uint64_t h = 1469598103934665603ull;
  auto mix = [&h](uint64_t v) { h ^= v; h *= 1099511628211ull; };
  int64_t entries = 0;
  {
    // order-independent over cells (the map's iteration order is an implementation detail), order-DEPENDENT
    // within a cell (updateBlockedNodes / updateClearedNodes walk that vector front to back)
    uint64_t acc = 0;
    for (const auto &kv : pl.lhm_) {
      h = 1469598103934665603ull;
      mix((uint64_t)(int64_t)kv.first);
      for (const auto &pr : kv.second) {
        mix((uint64_t)hash_value(pr.first));
        mix((uint64_t)pr.second);
        entries++;
      }
      acc += h;
    }
    stats[2] = (int64_t)acc;
  }
  stats[1] = entries;
  stats[3] = (int64_t)linked.size();
  {
    double c = 0;
    size_t k = 1;
    for (const auto &p : linked) {
      for (int i = 0; i < D; i++) c += (double)((k * 2654435761u) % 1009) * p(i);
      k++;
    }
    std::memcpy(&stats[4], &c, 8);
  }
  // ---- block a box of free cells around the middle of the trajectory
  const Trajectory<D> traj = pl.getTraj();
  const auto wps = traj.getWaypoints();
  const Veci<D> mid = mu->floatToInt(wps[wps.size() / 2].pos);
  const Veci<D> sc = mu->floatToInt(start.pos), gc = mu->floatToInt(goal.pos);
  vec_Veci<D> edit;
  {
    Veci<D> pn;
    const int w = 2 * box_half + 1;
    int total = 1;
    for (int i = 0; i < D; i++) total *= w;
    for (int q = 0; q < total; q++) {
      int r = q;
      for (int i = 0; i < D; i++) { pn(i) = mid(i) + (r % w) - box_half; r /= w; }
      if (mu->isOutside(pn) || !mu->isFree(pn)) continue;
      bool keep = true;
      for (int i = 0; i < D && keep; i++) keep = std::abs(pn(i) - sc(i)) > 2 || std::abs(pn(i) - gc(i)) > 2;
      bool near_s = true, near_g = true;
      for (int i = 0; i < D; i++) { near_s = near_s && std::abs(pn(i) - sc(i)) <= 2; near_g = near_g && std::abs(pn(i) - gc(i)) <= 2; }
      if (near_s || near_g) continue;
      edit.push_back(pn);
    }
  }
  stats[5] = (int64_t)edit.size();
  for (const auto &pn : edit) cells[(size_t)mu->getIndex(pn)] = 100;
  mu->setMap(ori, dim, cells, e->res);
  const uint64_t up0 = upload_bytes(pl, 0);
  pl.updateBlockedNodes(edit);
  t0 = now();
  ok = pl.plan(start, goal);
  fill_out<D>(pl, ok, ms_since(t0), 0, &out3[1], &checksum3[1]);
  stats[8] = (int64_t)(upload_bytes(pl, 0) - up0);  // map bytes moved to the device by the edit + the re-plan
  // ---- clear them again
  pl.getLinkedNodes();  // the table of the grown graph
  for (const auto &pn : edit) cells[(size_t)mu->getIndex(pn)] = 0;
  mu->setMap(ori, dim, cells, e->res);
  const uint64_t up1 = upload_bytes(pl, 0);
  t0 = now();
  pl.updateClearedNodes(edit);
  stats[7] = (int64_t)(ms_since(t0) * 1000.0);
  t0 = now();
  ok = pl.plan(start, goal);
  fill_out<D>(pl, ok, ms_since(t0), 0, &out3[2], &checksum3[2]);
  stats[9] = (int64_t)(upload_bytes(pl, 0) - up1);
  return 0;
}

}  // namespace

/* ---- StateSpace::getSubStateSpace (state_space.h:116-195) through PlannerBase::getSubStateSpace (planner_base.h:155): plan
 * with LPA*, re-root the tree at way point `time_step` of the trajectory (a robot that has executed that many
 * primitives), plan again from there.  out2[0]: the first plan, out2[1]: the plan from the new root. ---- */
namespace {
template <int D>
int run_substate(const mpl_oracle_env *e, const double *start_row, const double *goal_row, int time_step, mpl_ref_plan_out *out2,
                 double *checksum2) {
  std::shared_ptr<MPL::MapUtil<D>> mu = std::make_shared<MPL::MapUtil<D>>();
  Vecf<D> ori;
  Veci<D> dim;
  size_t n = 1;
  for (int i = 0; i < D; i++) { ori(i) = e->origin[i]; dim(i) = e->map_dim[i]; n *= (size_t)e->map_dim[i]; }
  mu->setMap(ori, dim, MPL::Tmap(e->map, e->map + n), e->res);
  vec_E<VecDf> U;
  for (int i = 0; i < e->nU; i++) {
    VecDf u(e->udim);
    for (int k = 0; k < e->udim; k++) u(k) = e->U[(size_t)i * e->udim + k];
    U.push_back(u);
  }
  auto load = [&](const double *r) {
    Waypoint<D> w((Control::Control)e->control);
    for (int i = 0; i < D; i++) { w.pos(i) = r[i]; w.vel(i) = r[D + i]; w.acc(i) = r[2 * D + i]; w.jrk(i) = r[3 * D + i]; }
    w.yaw = r[4 * D];
    w.t = r[4 * D + 1];
    return w;
  };
  const Waypoint<D> start = load(start_row), goal = load(goal_row);
  MPL::MapPlanner<D> pl(false);
  pl.setMapUtil(mu);
  pl.setVmax(e->v_max);
  pl.setAmax(e->a_max);
  pl.setJmax(e->j_max);
  pl.setDt(e->dt);
  pl.setW(e->w);
  pl.setEpsilon(1.0);
  pl.setU(U);
  pl.setLPAstar(true);
  for (int i = 0; i < 2; i++) { out2[i] = mpl_ref_plan_out{}; checksum2[i] = 0; }
  auto t0 = std::chrono::steady_clock::now();
  auto ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
  bool ok = pl.plan(start, goal);
  fill_out<D>(pl, ok, ms(), 0, &out2[0], &checksum2[0]);
  if (!ok) return 0;
  const auto wps = pl.getTraj().getWaypoints();
  if (time_step < 0 || (size_t)time_step >= wps.size()) return -2;
  pl.getSubStateSpace(time_step);
  Waypoint<D> from = wps[(size_t)time_step];
  from.control = start.control;  // (Trajectory::getWaypoints carries the primitives' control flag: the same here)
  t0 = std::chrono::steady_clock::now();
  ok = pl.plan(from, goal);
  fill_out<D>(pl, ok, ms(), 0, &out2[1], &checksum2[1]);
  return 0;
}
}  // namespace

/* ---- a map edit AND a re-rooting between two plans: plan, getLinkedNodes, block a box of free cells around the middle of
 * the trajectory + updateBlockedNodes, getSubStateSpace(time_step), plan from way point `time_step`.  getSubStateSpace drops
 * the nodes it does not reach from the hash map while re-opened parents still list them as successors
 * (state_space.h:116-195, graph_search.h:285-290 look them up again).  out2[0]: first plan, out2[1]: the plan after both. ---- */
namespace {
template <int D>
int run_edit_substate(const mpl_oracle_env *e, const double *start_row, const double *goal_row, int box_half, int time_step,
                      mpl_ref_plan_out *out2, double *checksum2, int64_t *edited) {
  std::shared_ptr<MPL::MapUtil<D>> mu = std::make_shared<MPL::MapUtil<D>>();
  Vecf<D> ori;
  Veci<D> dim;
  size_t n = 1;
  for (int i = 0; i < D; i++) { ori(i) = e->origin[i]; dim(i) = e->map_dim[i]; n *= (size_t)e->map_dim[i]; }
  MPL::Tmap cells(e->map, e->map + n);
  mu->setMap(ori, dim, cells, e->res);
  vec_E<VecDf> U;
  for (int i = 0; i < e->nU; i++) {
    VecDf u(e->udim);
    for (int k = 0; k < e->udim; k++) u(k) = e->U[(size_t)i * e->udim + k];
    U.push_back(u);
  }
  auto load = [&](const double *r) {
    Waypoint<D> w((Control::Control)e->control);
    for (int i = 0; i < D; i++) { w.pos(i) = r[i]; w.vel(i) = r[D + i]; w.acc(i) = r[2 * D + i]; w.jrk(i) = r[3 * D + i]; }
    w.yaw = r[4 * D];
    w.t = r[4 * D + 1];
    return w;
  };
  const Waypoint<D> start = load(start_row), goal = load(goal_row);
  MPL::MapPlanner<D> pl(false);
  pl.setMapUtil(mu);
  pl.setVmax(e->v_max);
  pl.setAmax(e->a_max);
  pl.setJmax(e->j_max);
  pl.setDt(e->dt);
  pl.setW(e->w);
  pl.setEpsilon(1.0);
  pl.setU(U);
  pl.setLPAstar(true);
  for (int i = 0; i < 2; i++) { out2[i] = mpl_ref_plan_out{}; checksum2[i] = 0; }
  *edited = 0;
  auto t0 = std::chrono::steady_clock::now();
  auto ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
  bool ok = pl.plan(start, goal);
  fill_out<D>(pl, ok, ms(), 0, &out2[0], &checksum2[0]);
  if (!ok) return 0;
  const auto wps = pl.getTraj().getWaypoints();
  if (time_step < 0 || (size_t)time_step >= wps.size()) return -2;
  pl.getLinkedNodes();
  const Veci<D> mid = mu->floatToInt(wps[wps.size() / 2].pos);
  const Veci<D> sc = mu->floatToInt(wps[(size_t)time_step].pos), gc = mu->floatToInt(goal.pos);
  vec_Veci<D> edit;
  {
    Veci<D> pn;
    const int w = 2 * box_half + 1;
    int total = 1;
    for (int i = 0; i < D; i++) total *= w;
    for (int q = 0; q < total; q++) {
      int r = q;
      for (int i = 0; i < D; i++) { pn(i) = mid(i) + (r % w) - box_half; r /= w; }
      if (mu->isOutside(pn) || !mu->isFree(pn)) continue;
      bool near_s = true, near_g = true;
      for (int i = 0; i < D; i++) { near_s = near_s && std::abs(pn(i) - sc(i)) <= 2; near_g = near_g && std::abs(pn(i) - gc(i)) <= 2; }
      if (near_s || near_g) continue;
      edit.push_back(pn);
    }
  }
  *edited = (int64_t)edit.size();
  for (const auto &pn : edit) cells[(size_t)mu->getIndex(pn)] = 100;
  mu->setMap(ori, dim, cells, e->res);
  pl.updateBlockedNodes(edit);
  pl.getSubStateSpace(time_step);
  Waypoint<D> from = wps[(size_t)time_step];
  from.control = start.control;
  t0 = std::chrono::steady_clock::now();
  ok = pl.plan(from, goal);
  fill_out<D>(pl, ok, ms(), 0, &out2[1], &checksum2[1]);
  return 0;
}
}  // namespace

extern "C" int mpl_ref_lpastar_edit_substate(const mpl_oracle_env *env, const double *start, const double *goal, int box_half,
                                             int time_step, mpl_ref_plan_out *out2, double *checksum2, int64_t *edited) {
  if (!env || !start || !goal || !out2 || !checksum2 || !edited) return -1;
  if (env->dim == 2) return run_edit_substate<2>(env, start, goal, box_half, time_step, out2, checksum2, edited);
  if (env->dim == 3) return run_edit_substate<3>(env, start, goal, box_half, time_step, out2, checksum2, edited);
  return -1;
}

extern "C" int mpl_ref_lpastar_substate(const mpl_oracle_env *env, const double *start, const double *goal, int time_step,
                                        mpl_ref_plan_out *out2, double *checksum2) {
  if (!env || !start || !goal || !out2 || !checksum2) return -1;
  if (env->dim == 2) return run_substate<2>(env, start, goal, time_step, out2, checksum2);
  if (env->dim == 3) return run_substate<3>(env, start, goal, time_step, out2, checksum2);
  return -1;
}

extern "C" int mpl_ref_lpastar(const mpl_oracle_env *env, const double *start, const double *goal, int use_gpu,
                               int box_half, mpl_ref_plan_out *out3, double *checksum3, int64_t *stats8 /* [10] */) {
  if (!env || !start || !goal || !out3 || !checksum3 || !stats8) return -1;
  if (env->dim == 2) {
    if (use_gpu) return run_lpastar<2, PeekPlanner<MPL::GpuMapPlanner<2>>>(env, start, goal, box_half, out3, checksum3, stats8);
    return run_lpastar<2, PeekPlanner<MPL::MapPlanner<2>>>(env, start, goal, box_half, out3, checksum3, stats8);
  }
  if (env->dim == 3) {
    if (use_gpu) return run_lpastar<3, PeekPlanner<MPL::GpuMapPlanner<3>>>(env, start, goal, box_half, out3, checksum3, stats8);
    return run_lpastar<3, PeekPlanner<MPL::MapPlanner<3>>>(env, start, goal, box_half, out3, checksum3, stats8);
  }
  return -1;
}

/* the same scenarios with the drop-in held through a MapPlanner<D> base pointer (ViaBase above) */
/* wall time (ms) of updatePotentialMap in the last mpl_ref_scenario* call of this process (modes 0 - 2) */
extern "C" double mpl_ref_last_prep_ms(void) { return g_last_prep_ms; }

extern "C" int mpl_ref_scenario_via_base(const mpl_oracle_env *env, const double *start, const double *goal, int batch,
                                         int mode, mpl_ref_plan_out *out2, double *checksum2, int64_t *region_cells,
                                         int64_t *potential_sum) {
  if (!env || !start || !goal || !out2 || env->dim != 2 || mode < 0 || mode > 5) return -1;
  return run_scenario<2, ViaBase<2>>(env, start, goal, mode, batch, out2, checksum2, region_cells, potential_sum);
}

/* ---- adapter robustness: a device failure must be distinguishable from "no trajectory" ---- */
extern "C" int mpl_gpu_plan_on_device(const mpl_oracle_env *e, const double *start_row, const double *goal_row,
                                      int device, int32_t *plan_ok, int32_t *device_ok, char *err, int err_cap) {
  if (!e || e->dim != 2 || !plan_ok || !device_ok) return -1;
  constexpr int D = 2;
  std::shared_ptr<MPL::MapUtil<D>> mu = std::make_shared<MPL::MapUtil<D>>();
  Vecf<D> ori;
  Veci<D> dim;
  size_t n = 1;
  for (int i = 0; i < D; i++) { ori(i) = e->origin[i]; dim(i) = e->map_dim[i]; n *= (size_t)e->map_dim[i]; }
  mu->setMap(ori, dim, MPL::Tmap(e->map, e->map + n), e->res);
  MPL::GpuMapPlanner<D> pl(false, device, 1);
  pl.setMapUtil(mu);
  pl.setVmax(e->v_max);
  pl.setAmax(e->a_max);
  pl.setDt(e->dt);
  vec_E<VecDf> U;
  for (int i = 0; i < e->nU; i++) {
    VecDf u(e->udim);
    for (int k = 0; k < e->udim; k++) u(k) = e->U[(size_t)i * e->udim + k];
    U.push_back(u);
  }
  pl.setU(U);
  Waypoint<D> s((Control::Control)e->control), g((Control::Control)e->control);
  for (int i = 0; i < D; i++) { s.pos(i) = start_row[i]; g.pos(i) = goal_row[i]; }
  *plan_ok = pl.plan(s, g) ? 1 : 0;
  *device_ok = pl.deviceOk() ? 1 : 0;
  if (err && err_cap > 0) snprintf(err, (size_t)err_cap, "%s", pl.deviceError().c_str());
  return 0;
}

/* the same through a MapPlanner<2> pointer: the base plan() cannot check the device (it returns "no trajectory"),
 * the latch is still there for an application that asks the derived type afterwards */
extern "C" int mpl_gpu_plan_on_device_via_base(const mpl_oracle_env *e, const double *start_row, const double *goal_row,
                                               int device, int32_t *plan_ok, int32_t *device_ok, char *err, int err_cap) {
  if (!e || e->dim != 2 || !plan_ok || !device_ok) return -1;
  constexpr int D = 2;
  std::shared_ptr<MPL::MapUtil<D>> mu = std::make_shared<MPL::MapUtil<D>>();
  Vecf<D> ori;
  Veci<D> dim;
  size_t n = 1;
  for (int i = 0; i < D; i++) { ori(i) = e->origin[i]; dim(i) = e->map_dim[i]; n *= (size_t)e->map_dim[i]; }
  mu->setMap(ori, dim, MPL::Tmap(e->map, e->map + n), e->res);
  std::unique_ptr<MPL::MapPlanner<D>> pl(new MPL::GpuMapPlanner<D>(false, device, 1));
  pl->setMapUtil(mu);
  pl->setVmax(e->v_max);
  pl->setAmax(e->a_max);
  pl->setDt(e->dt);
  vec_E<VecDf> U;
  for (int i = 0; i < e->nU; i++) {
    VecDf u(e->udim);
    for (int k = 0; k < e->udim; k++) u(k) = e->U[(size_t)i * e->udim + k];
    U.push_back(u);
  }
  pl->setU(U);
  Waypoint<D> s((Control::Control)e->control), g((Control::Control)e->control);
  for (int i = 0; i < D; i++) { s.pos(i) = start_row[i]; g.pos(i) = goal_row[i]; }
  *plan_ok = pl->plan(s, g) ? 1 : 0;  // PlannerBase::plan
  MPL::GpuMapPlanner<D> *gp = static_cast<MPL::GpuMapPlanner<D> *>(pl.get());
  *device_ok = gp->deviceOk() ? 1 : 0;
  if (err && err_cap > 0) snprintf(err, (size_t)err_cap, "%s", gp->deviceError().c_str());
  return 0;
}
