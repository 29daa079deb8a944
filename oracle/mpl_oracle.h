/*
 * mpl_oracle.h -- C interface of the CPU parity oracle.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library.  The shipped path is the
 * HIP engine behind include/mplx.h; it never links or calls anything here.
 *
 * The oracle is a CPU restatement of the successor-expansion hot path of
 * sikang/motion_primitive_library (MPL v1.2):
 *   include/mpl_planner/env/env_map.h:90-132   traverse_primitive
 *   include/mpl_planner/env/env_map.h:147-172  get_succ
 * together with the primitive algebra, waypoint hash and map look-ups those
 * two functions reach.  Every function in mpl_oracle.cpp cites the reference
 * lines it follows.
 *
 * Parity pinning: see the header comment of mpl_oracle.cpp.
 */
#ifndef MPL_ORACLE_H
#define MPL_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Slot status codes (shared with include/mplx.h). */
enum {
  MPL_SLOT_SKIP_SAME = 0,   /* tn == curr by lattice hash: not emitted      */
  MPL_SLOT_FINITE = 1,      /* emitted, finite edge cost                    */
  MPL_SLOT_BLOCKED = 2,     /* emitted, cost = +inf (collision / outside)   */
  MPL_SLOT_SKIP_DYN = 3     /* validate_primitive failed: not emitted       */
};

typedef struct {
  int32_t dim;     /* 2 or 3 */
  int32_t control; /* Control::Control bit flags, control.h:10-20 */
  double dt, w, wyaw;
  double v_max, a_max, j_max, yaw_max;
  double potential_weight, gradient_weight;
  int32_t map_dim[3];
  double origin[3];
  double res;
  const int8_t *map;       /* occupancy cells, x fastest (map_util.h:34-41)   */
  const int8_t *potential; /* potential cells or NULL (env_map.h:113-118)     */
  const uint8_t *region;   /* search region, one byte per cell, or NULL       */
  const double *U;         /* [nU][udim] row-major                            */
  int32_t nU, udim;
} mpl_oracle_env;

/* Dense output, one slot per (node, control): slot = node * nU + control.
 * Any pointer may be NULL.  state is field-major: state[f * n_slots + slot]
 * with rows pos[0..D) vel[0..D) acc[0..D) jrk[0..D) yaw t  (4D+2 rows).      */
typedef struct {
  uint8_t *status;
  double *cost;    /* +inf unless status == MPL_SLOT_FINITE                   */
  uint64_t *hash;  /* lattice hash of the successor                           */
  double *state;
  int32_t *iters;  /* executed sample-loop iterations (0 if not traversed)    */
} mpl_oracle_out;

typedef struct {
  int64_t pairs, emitted, finite, skip_same, skip_dyn;
  int64_t samples; /* executed sample-loop iterations, incl. the blocking one */
  double sum_finite_cost;
} mpl_oracle_stats;

/* nodes is field-major [4D+2][n_nodes] with the same row order as `state`.  */
int mpl_oracle_expand(const mpl_oracle_env *env, const double *nodes,
                      int64_t n_nodes, mpl_oracle_out *out, int threads,
                      mpl_oracle_stats *stats);

/* Timing entry for bench.py's cpu_baseline: runs the reference-structured
 * per-node get_succ over the frontier (no dense scatter), `reps` times on
 * `threads` std::threads; returns the best wall time in seconds.             */
double mpl_oracle_time_expand(const mpl_oracle_env *env, const double *nodes,
                              int64_t n_nodes, int threads, int reps,
                              mpl_oracle_stats *stats);

/* env_base<Dim>::get_succ for one node in the reference's own list form
 * (env_map.h:147-172): successors in ascending control index, blocked ones
 * included with cost = +inf.  `user` is a const mpl_oracle_env*.  node: 4D+2
 * doubles; succ: [nU][4D+2].  Signature = mplx_succ_fn of include/mplx.h so a
 * test can plug the oracle into the host search.                             */
int mpl_oracle_get_succ(void *user, const double *node, double *succ, double *cost,
                        int32_t *action, int32_t *n_succ);
/* Batched dense form with the signature of mplx_batch_fn.                    */
int mpl_oracle_batch(void *user, const double *nodes, int64_t n, uint8_t *status, double *cost,
                     double *state);

/* Lattice hash of one waypoint given as 4D+2 doubles (waypoint.h:93-125).   */
uint64_t mpl_oracle_hash(int32_t dim, int32_t control, const double *wp);

/* Default heuristic, env_base.h:46-64 (heur_ignore_dynamics branch).        */
double mpl_oracle_heur(int32_t dim, int32_t control, double w, double v_max,
                       const double *wp, const double *goal);

/* env_map<Dim>::is_goal without its ray trace (env_map.h:25-37).             */
int32_t mpl_oracle_goal_tol(int32_t dim, const double *wp, const double *goal, double tol_pos,
                            double tol_vel, double tol_acc, double tol_yaw);

/* Sample-loop iteration count of `for (t = 0; t < T; t += T/n)`
 * (env_map.h:97-99): n or n+1.                                              */
int32_t mpl_oracle_loop_count(double T, int32_t n);

/* Batched edge re-validation: env_base::forward_action
 * (env_base.h:228-231), env_map::is_free(Primitive) (env_map.h:60-76),
 * calculate_intrinsic_cost (env_base.h:343-345; +inf when not free) and the
 * linked cells of MapPlanner::getLinkedNodes (map_planner.cpp:125-157).
 * parents: field-major [4D+2][n]; cells: [n][cell_cap] or NULL.               */
int mpl_oracle_check_edges(const mpl_oracle_env *env, const double *parents, const int32_t *actions,
                           int64_t n, uint8_t *free_out, double *cost_out, int32_t *cells,
                           int32_t *cell_count, int32_t cell_cap);

/* Map preprocessing.  MapPlanner<Dim>::updatePotentialMap with
 * createMask (src/mpl_planner/map_planner.cpp:246-283, 286-391): map_out
 * receives the map with the potential field stamped in.  radius / range / pos
 * have `dim` entries; an all-zero range means the whole map.                 */
int mpl_oracle_update_potential_map(int32_t dim, const int8_t *map_in, const int32_t *map_dim,
                                    const double *origin, double res, const double *pos,
                                    const double *radius, const double *range, double pow_,
                                    int8_t *map_out);
/* MapPlanner<Dim>::setSearchRegion (map_planner.cpp:46-95) with
 * MapUtil::rayTrace (map_util.h:117-135): one byte per cell, 1 = in region.  */
int mpl_oracle_search_region(int32_t dim, const int32_t *map_dim, const double *origin, double res,
                             const double *path, int32_t n_points, int32_t dense,
                             const double *search_radius, uint8_t *region_out);

#ifdef __cplusplus
}
#endif
#endif
