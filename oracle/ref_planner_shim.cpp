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
  if (use_gpu) planner.reset(new MPL::GpuMapPlanner<D>(false, 0));
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
  out->hm_size = 0;
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
