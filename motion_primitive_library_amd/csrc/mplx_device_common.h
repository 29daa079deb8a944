// mplx_device_common.h -- device helpers shared by the list-producing kernels
// (expand_tile_kernel.hip, expand_grid_kernel.hip): the hoisted exact division,
// the lattice hash (reference include/mpl_basis/waypoint.h:93-125) and the
// per-axis polynomial of a forward primitive (reference
// include/mpl_basis/primitive.h:34-50,128-145,353-394).  Derivations of every
// expression are in expand_kernel.hip.  Compile with -ffp-contract=off; the only
// fused operations are the explicit fma's of the division tail.
#ifndef MPLX_DEVICE_COMMON_H
#define MPLX_DEVICE_COMMON_H

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace mplx {
namespace dev {

// ------------------------------------------------------------------ division
// q = y / d given R = refined reciprocal of d (see make_tables_kernel).
__device__ __forceinline__ double div_by(double y, double d, double R) {
  const double q0 = y * R;
  const double rem = __builtin_fma(-d, q0, y);
  return __builtin_fma(rem, R, q0);
}
// the reciprocal exactly as hipcc's f64 division refines it
__device__ __forceinline__ double refined_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  return r;
}

__device__ __forceinline__ void fold(uint64_t &seed, int id) {
  seed ^= (uint64_t)(int64_t)id + 0x9e3779b9ULL + (seed << 6) + (seed >> 2);
}
// `int id = std::round(x / q)` (waypoint.h:95-112), division by a constant
__device__ __forceinline__ int quantise(double x, double q, double Rq) {
  return (int)round(div_by(x, q, Rq));
}

// ---- wave-wide min / max of an int over all 64 lanes (all lanes must be active), result uniform.  Seven DPP moves
// (row_shr 1, 2, 3, 4, 8, row_bcast 15, 31: the gfx9 cross-lane reduction) instead of six rounds of ds_bpermute -- no
// trip through the LDS crossbar, no dependent ~100-cycle waits.
template <bool MAX>
__device__ __forceinline__ int wave_reduce_minmax(int v) {
  const int id = MAX ? (int)0x80000000 : 0x7fffffff;
  auto op = [](int a, int b) { return MAX ? (a > b ? a : b) : (a < b ? a : b); };
  int x = v;
  x = op(x, __builtin_amdgcn_update_dpp(id, v, 0x111, 0xf, 0xf, false));  // row_shr:1
  x = op(x, __builtin_amdgcn_update_dpp(id, v, 0x112, 0xf, 0xf, false));  // row_shr:2
  x = op(x, __builtin_amdgcn_update_dpp(id, v, 0x113, 0xf, 0xf, false));  // row_shr:3
  x = op(x, __builtin_amdgcn_update_dpp(id, x, 0x114, 0xf, 0xe, false));  // row_shr:4, banks 1-3
  x = op(x, __builtin_amdgcn_update_dpp(id, x, 0x118, 0xf, 0xc, false));  // row_shr:8, banks 2-3: lane 15 of a row = the row
  x = op(x, __builtin_amdgcn_update_dpp(id, x, 0x142, 0xa, 0xf, false));  // row_bcast:15 into rows 1, 3
  x = op(x, __builtin_amdgcn_update_dpp(id, x, 0x143, 0xc, 0xf, false));  // row_bcast:31 into rows 2, 3: lane 63 = the wave
  return __builtin_amdgcn_readlane(x, 63);
}

// wave-wide OR of an unsigned over all 64 lanes (all lanes active), result uniform: the same seven DPP moves
__device__ __forceinline__ unsigned int wave_reduce_or(unsigned int v) {
  int x = (int)v;
  x |= __builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
  x |= __builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
  x |= __builtin_amdgcn_update_dpp(0, (int)v, 0x113, 0xf, 0xf, false);  // row_shr:3
  x |= __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xe, false);       // row_shr:4
  x |= __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xc, false);       // row_shr:8: lane 15 of a row = the row
  x |= __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);       // row_bcast:15 into rows 1, 3
  x |= __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);       // row_bcast:31 into rows 2, 3: lane 63 = the wave
  return (unsigned int)__builtin_amdgcn_readlane(x, 63);
}

// ---- yaw pinning (YawPin, mplx_internal.h): is the heading-limit decision `d < cos_lim` of validate_yaw
// (primitive.h:504-525) within rounding noise of its threshold?  One case is exempt because it is an exact tie under
// ANY libm with an even cosine: velocity along x (vy == 0: the y term is an exact zero and vx / |v| is exactly +-1)
// with |yaw| == yaw_max gives d = +-cos(yaw) against cos(yaw_max) = cos(|yaw|) -- equal, or 2 cos(yaw_max) apart.
// Lattice searches hit that tie all the time (yaw_max and the yaw steps are the same multiples of 0.5), so it must
// not count as ambiguous; it is decided identically by the device and by the host -- PROVIDED the host's cosine of
// the pair sincos(+-yaw_max) (what GCC makes of primitive.h:519-520) is bit-equal to its stand-alone cos(yaw_max)
// (primitive.h:521).  The host checks exactly that once per launch set-up (yaw_slot, mplx_api.cpp) and passes
// tie_yaw = yaw_max when it holds, NaN when it does not: then nothing compares equal and the tie goes through the
// host-pinned pass like every other decision inside the band.
__device__ __forceinline__ bool near_limit(double d, double cos_lim, double margin, double vy, double yaw,
                                           double tie_yaw) {
  if (vy == 0 && fabs(yaw) == tie_yaw) return false;
  return fabs(d - cos_lim) <= margin;
}
__device__ __forceinline__ void flag_node(int32_t *amb, int cap, int64_t node, int32_t *any_host) {
  const int k = atomicAdd(&amb[0], 1);
  if (k < cap) amb[1 + k] = (int32_t)node;
  if (any_host) __hip_atomic_store(any_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- heading cost of one sample (env_map.h:121-129): v.normalized() of the sample's planar velocity, and whether
// v.norm() > 1e-5.  The heading COST is the one quantity of this path held to north_star's 1e-6 and not to the bit (its
// cos / sin are the device's, not glibc's: DESIGN.md 4.3), so the unit vector is v * rsqrt(v.v) -- v_rsq_f64 and two
// Newton steps, within a few ulp of v / sqrt(v.v) -- instead of a square root and two divisions (~75 instructions per
// pair and sample: a tenth of C5's launch).  The DECISION norm > 1e-5 stays the reference's: on v.v except within 1e-8
// (relative) of the threshold, where the square root itself decides.  Every kernel that adds a heading cost calls this.
__device__ __forceinline__ bool heading_unit(double vx, double vy, double &ux, double &uy) {
  const double s2 = vx * vx + vy * vy;
  bool go = s2 >= 1.00000001e-10;
  const bool band = s2 > 0.99999999e-10 && !go;
  // (a wave-uniform branch: the square root's ~25 instructions must be SKIPPED, not predicated, when no lane is in the band)
  if (__builtin_expect(__ballot(band) != 0ull, 0)) {
    if (band) go = sqrt(s2) > 1e-5;
  }
  if (!go) return false;
  double y = __builtin_amdgcn_rsq(s2);
  y = y * (1.5 - 0.5 * s2 * y * y);
  y = y * (1.5 - 0.5 * s2 * y * y);
  ux = vx * y;
  uy = vy * y;
  return true;
}

template <int D, int K>
__device__ __forceinline__ uint64_t lattice_hash(const double *pos, const double *vel, const double *acc,
                                                 const double *jrk, double R001, double R01) {
  uint64_t h = 0;
#pragma unroll
  for (int i = 0; i < D; i++) {
    fold(h, quantise(pos[i], 0.01, R001));
    if (K >= 2) fold(h, quantise(vel[i], 0.1, R01));
    if (K >= 3) fold(h, quantise(acc[i], 0.1, R01));
    if (K >= 4) fold(h, quantise(jrk[i], 0.1, R01));
  }
  return h;
}

// Per-axis polynomial of a forward primitive, see expand_kernel.hip for the
// derivation of every expression (primitive.h:128-145, 353-394).
template <int K>
struct Ax {
  double c1, c2, c3, c4, c5;
  __device__ __forceinline__ void init(double p, double v, double a, double j, double u) {
    c1 = c2 = c3 = c4 = 0.0;
    c5 = p;
    if (K == 1) { c4 = u; }
    if (K == 2) { c4 = v; c3 = u; }
    if (K == 3) { c4 = v; c3 = a; c2 = u; }
    if (K == 4) { c4 = v; c3 = a; c2 = j; c1 = u; }
  }
  template <bool EXACT>
  __device__ __forceinline__ double pos(double t) const {
    double s;
    if (K == 1) { s = c4 * t; if (EXACT) s = 0.0 + s; return s + c5; }
    if (K == 2) { s = ((c3 / 2) * t) * t; if (EXACT) s = 0.0 + s; return (s + c4 * t) + c5; }
    if (K == 3) {
      s = (c2 / 6) * ((t * t) * t);
      if (EXACT) s = 0.0 + s;
      return ((s + ((c3 / 2) * t) * t) + c4 * t) + c5;
    }
    const double t3 = (t * t) * t;
    s = (c1 / 24) * (t3 * t);
    if (EXACT) s = 0.0 + s;
    return (((s + (c2 / 6) * t3) + ((c3 / 2) * t) * t) + c4 * t) + c5;
  }
  // pos<false>(t) with the quotient of the highest-order coefficient given (c2 / 6 for K = 3, c1 / 24 for K = 4): it
  // depends only on the control value, so the row builder takes it from a per-entry table instead of dividing per
  // sample (an f64 division by 6 is ~30 instructions; the value is the same one, so is the result)
  __device__ __forceinline__ double pos_q(double t, double qtop) const {
    if (K <= 2) return pos<false>(t);
    if (K == 3) return ((qtop * ((t * t) * t) + ((c3 / 2) * t) * t) + c4 * t) + c5;
    const double t3 = (t * t) * t;
    return (((qtop * (t3 * t) + (c2 / 6) * t3) + ((c3 / 2) * t) * t) + c4 * t) + c5;
  }
  __device__ __forceinline__ double top_quotient() const { return K == 3 ? c2 / 6 : (K == 4 ? c1 / 24 : 0.0); }
  template <bool EXACT>
  __device__ __forceinline__ double vel(double t) const {
    double s;
    if (K == 1) { return EXACT ? 0.0 + c4 : c4; }
    if (K == 2) { s = c3 * t; if (EXACT) s = 0.0 + s; return s + c4; }
    if (K == 3) { s = ((c2 / 2) * t) * t; if (EXACT) s = 0.0 + s; return (s + c3 * t) + c4; }
    s = (c1 / 6) * ((t * t) * t);
    if (EXACT) s = 0.0 + s;
    return ((s + ((c2 / 2) * t) * t) + c3 * t) + c4;
  }
  template <bool EXACT>
  __device__ __forceinline__ double acc(double t) const {
    double s;
    if (K == 1) return 0.0;
    if (K == 2) { return EXACT ? 0.0 + c3 : c3; }
    if (K == 3) { s = c2 * t; if (EXACT) s = 0.0 + s; return s + c3; }
    s = ((c1 / 2) * t) * t;
    if (EXACT) s = 0.0 + s;
    return (s + c2 * t) + c3;
  }
  template <bool EXACT>
  __device__ __forceinline__ double jrk(double t) const {
    double s;
    if (K <= 2) return 0.0;
    if (K == 3) { return EXACT ? 0.0 + c2 : c2; }
    s = c1 * t;
    if (EXACT) s = 0.0 + s;
    return s + c2;
  }
  __device__ __forceinline__ double max_vel(double T) const {
    const double v0 = fabs(c4), vT = fabs(vel<false>(T));
    double m = (v0 < vT) ? vT : v0;
    if (K == 3) {
      if (c2 != 0) {
        const double r = -c3 / c2;
        if (r > 0 && r < T) { const double v = fabs(vel<false>(r)); m = v > m ? v : m; }
      }
    }
    if (K == 4) {
      const double b = c1 / 2;
      if (b != 0) {
        const double disc = c2 * c2 - 4 * b * c3;
        if (!(disc < 0)) {
          const double sq = sqrt(disc);
          const double r1 = (-c2 - sq) / (2 * b);
          const double r2 = (-c2 + sq) / (2 * b);
          bool go_on = true;
          if (r1 > 0 && r1 < T) { const double v = fabs(vel<false>(r1)); m = v > m ? v : m; }
          else if (r1 >= T) go_on = false;
          if (go_on && r2 > 0 && r2 < T) { const double v = fabs(vel<false>(r2)); m = v > m ? v : m; }
        }
      } else if (c2 != 0) {
        const double r = -c3 / c2;
        if (r > 0 && r < T) { const double v = fabs(vel<false>(r)); m = v > m ? v : m; }
      }
    }
    return m;
  }
  __device__ __forceinline__ double max_acc(double T) const {
    const double a0 = fabs(c3), aT = fabs(acc<false>(T));
    double m = (a0 < aT) ? aT : a0;
    if (K == 4) {
      if (c1 != 0) {
        const double r = -c2 / c1;
        if (r > 0 && r < T) { const double a = fabs(acc<false>(r)); m = a > m ? a : m; }
      }
    }
    return m;
  }
  __device__ __forceinline__ double max_jrk(double T) const {
    const double j0 = fabs(c2), jT = fabs(jrk<false>(T));
    return (j0 < jT) ? jT : j0;
  }
  __device__ __forceinline__ double effort(double T) const {
    const double u = (K == 1) ? c4 : (K == 2) ? c3 : (K == 3) ? c2 : c1;
    return u * u * T;
  }
};

// Heuristic and goal flags of ONE successor (PostFuse of mplx_internal.h), from the values its state rows hold --
// the arithmetic of post_kernel.hip::post_lists_kernel, which reads those rows back from memory:
//   heur   env_base.h:46-64, default branch: 0 for the goal's own lattice state (a hash comparison, :47), else
//          w * |pos - goal.pos|_inf / v_max (w * |.|_inf when v_max <= 0)
//   flags  bit 0: env_map.h:25-37 without the ray trace; bit 1: the goal's lattice state
// pos / vel / acc: the D values of the successor's rows 0..D-1, D..2D-1, 2D..3D-1; yaw: row 4D.
struct PostGoal {  // PostFuse's scalars in registers (the kernel arguments live in the constant address space)
  double g[14];
  uint64_t goal_hash;
  double w, v_max, tol_pos, tol_vel, tol_acc, tol_yaw;
};
#define MPLX_POST_GOAL(PG, PF, D_)                                                                         \
  dev::PostGoal PG;                                                                                        \
  _Pragma("unroll") for (int i_ = 0; i_ < 3 * (D_); i_++) PG.g[i_] = (PF).goal[i_];                        \
  PG.g[4 * (D_)] = (PF).goal[4 * (D_)];                                                                    \
  PG.goal_hash = (PF).goal_hash; PG.w = (PF).w; PG.v_max = (PF).v_max; PG.tol_pos = (PF).tol_pos;          \
  PG.tol_vel = (PF).tol_vel; PG.tol_acc = (PF).tol_acc; PG.tol_yaw = (PF).tol_yaw;

template <int D>
__device__ __forceinline__ void post_eval(const PostGoal &P, uint64_t h, const double *pos, const double *vel, const double *acc,
                                          double yaw, double *heur, unsigned int *flags) {
  const bool is_goal_state = (h == P.goal_hash);
  double m = 0;  // lpNorm<Infinity> of pos - goal.pos
#pragma unroll
  for (int i = 0; i < D; i++) {
    const double d = fabs(pos[i] - P.g[i]);
    m = d > m ? d : m;
  }
  *heur = is_goal_state ? 0.0 : (P.v_max > 0 ? P.w * m / P.v_max : P.w * m);
  bool goaled = m <= P.tol_pos;  // env_map.h:26-28
  if (goaled && P.tol_vel >= 0) {
    double mv = 0;
#pragma unroll
    for (int i = 0; i < D; i++) {
      const double d = fabs(vel[i] - P.g[D + i]);
      mv = d > mv ? d : mv;
    }
    goaled = mv <= P.tol_vel;
  }
  if (goaled && P.tol_acc >= 0) {
    double ma = 0;
#pragma unroll
    for (int i = 0; i < D; i++) {
      const double d = fabs(acc[i] - P.g[2 * D + i]);
      ma = d > ma ? d : ma;
    }
    goaled = ma <= P.tol_acc;
  }
  if (goaled && P.tol_yaw >= 0) goaled = fabs(yaw - P.g[4 * D]) <= P.tol_yaw;
  *flags = (goaled ? 1u : 0u) | (is_goal_state ? 2u : 0u);
}

}  // namespace dev
}  // namespace mplx
#endif
