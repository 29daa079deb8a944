// expand_kernel.hip -- batched successor expansion for gfx950 (MI355X).
//
// One lane evaluates one (frontier node, control input) pair; a wavefront
// therefore covers 64 consecutive controls of (almost always) one node, so the
// node state is a broadcast load, the control table and every output row are
// read / written as contiguous 512-byte segments, and the 64 map look-ups of a
// sample step fall into one small neighbourhood of the voxel grid.
//
// This is the device side of MPL::env_map<Dim>::get_succ
//   reference include/mpl_planner/env/env_map.h:147-172 (get_succ)
//   reference include/mpl_planner/env/env_map.h:90-132  (traverse_primitive)
// written against the arithmetic specification in SURVEY.md Appendix A.  The
// result must be bit-identical to the CPU path, so:
//   * the file is compiled with -ffp-contract=off (no FMA contraction) and
//     without any fast-math flag; every product / sum below is one IEEE
//     binary64 operation in the reference's evaluation order;
//   * divisions are true divisions (never multiplication by a reciprocal);
//   * the sample time is accumulated (t += dt), never computed as k*dt
//     (env_map.h:97-99 runs n or n+1 iterations depending on rounding);
//   * structurally-zero coefficient terms are dropped only where that cannot
//     change a bit of the result; the emitted
//     successor state keeps the leading `0.0 +` so that signed zeros match.
//
// Polynomial layout (reference include/mpl_basis/primitive.h:34-50): per axis
//   p(t) = c0/120 t^5 + c1/24 t^4 + c2/6 t^3 + c3/2 t^2 + c4 t + c5
// and for a forward primitive of control order K (1 VEL, 2 ACC, 3 JRK, 4 SNP)
// only c[5-K] .. c5 are non-zero: c5 = pos, c4 = vel (or u for K = 1), ...
#include "mplx_internal.h"
#include "mplx_device_common.h"  // near_limit / flag_node of the yaw pinning

#include <math.h>

namespace mplx {
namespace {

constexpr int kBlock = 256;

template <typename T>
__device__ __forceinline__ void st_out(T *p, T v, bool stream) {
  if (stream) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// ---------------------------------------------------------------- lattice hash
// boost::hash_combine in its classic (< 1.81) form with hash_value(int) =
// sign-extending conversion; reference include/mpl_basis/waypoint.h:93-125.
__device__ __forceinline__ void fold(uint64_t &seed, int id) {
  seed ^= (uint64_t)(int64_t)id + 0x9e3779b9ULL + (seed << 6) + (seed >> 2);
}
// `int id = std::round(x / q)`: round half away from zero, then value-convert.
__device__ __forceinline__ int quantise(double x, double q) { return (int)round(x / q); }

template <int D, int K, bool YAW>
__device__ __forceinline__ uint64_t lattice_hash(const double *pos, const double *vel,
                                                 const double *acc, const double *jrk, double yaw) {
  uint64_t h = 0;
#pragma unroll
  for (int i = 0; i < D; i++) {
    fold(h, quantise(pos[i], 0.01));
    if (K >= 2) fold(h, quantise(vel[i], 0.1));
    if (K >= 3) fold(h, quantise(acc[i], 0.1));
    if (K >= 4) fold(h, quantise(jrk[i], 0.1));
  }
  if (YAW) fold(h, quantise(yaw, 0.1));
  return h;
}

// reference include/mpl_basis/math.h:15-19
__device__ __forceinline__ double wrap_angle(double a) {
  while (a > M_PI) a -= 2.0 * M_PI;
  while (a < -M_PI) a += 2.0 * M_PI;
  return a;
}

// ------------------------------------------------------------ one axis, order K
// Hoisted per-pair constants of one axis.  c1..c5 are the reference
// coefficients; the quotients are the `c_k / const` sub-expressions of
// primitive.h:128-145, each computed once (bit-exact hoist).
template <int K>
struct Axis {
  double c1, c2, c3, c4, c5;
  double c1_24, c1_6, c1_2, c2_6, c2_2, c3_2;

  __device__ __forceinline__ void init(double p, double v, double a, double j, double u) {
    c1 = c2 = c3 = c4 = 0.0;
    c5 = p;
    if (K == 1) { c4 = u; }
    if (K == 2) { c4 = v; c3 = u; }
    if (K == 3) { c4 = v; c3 = a; c2 = u; }
    if (K == 4) { c4 = v; c3 = a; c2 = j; c1 = u; }
    c3_2 = c3 / 2;
    c2_6 = c2 / 6;
    c2_2 = c2 / 2;
    c1_24 = c1 / 24;
    c1_6 = c1 / 6;
    c1_2 = c1 / 2;
  }

  // Z = 0.0 gives the exact reference value including the sign of a zero
  // result (the dropped leading terms sum to +0.0); pass Z = -0.0 (the additive
  // identity, folded away) on paths where the sign of zero cannot matter.
  // primitive.h:128-131
  template <bool EXACT>
  __device__ __forceinline__ double pos(double t) const {
    double s;
    if (K == 1) { s = c4 * t; if (EXACT) s = 0.0 + s; return s + c5; }
    if (K == 2) { s = (c3_2 * t) * t; if (EXACT) s = 0.0 + s; return (s + c4 * t) + c5; }
    if (K == 3) {
      s = c2_6 * ((t * t) * t);
      if (EXACT) s = 0.0 + s;
      return ((s + (c3_2 * t) * t) + c4 * t) + c5;
    }
    const double t3 = (t * t) * t;
    s = c1_24 * (t3 * t);
    if (EXACT) s = 0.0 + s;
    return (((s + c2_6 * t3) + (c3_2 * t) * t) + c4 * t) + c5;
  }
  // primitive.h:134-137
  template <bool EXACT>
  __device__ __forceinline__ double vel(double t) const {
    double s;
    if (K == 1) { return EXACT ? 0.0 + c4 : c4; }
    if (K == 2) { s = c3 * t; if (EXACT) s = 0.0 + s; return s + c4; }
    if (K == 3) { s = (c2_2 * t) * t; if (EXACT) s = 0.0 + s; return (s + c3 * t) + c4; }
    s = c1_6 * ((t * t) * t);
    if (EXACT) s = 0.0 + s;
    return ((s + (c2_2 * t) * t) + c3 * t) + c4;
  }
  // primitive.h:140-142
  template <bool EXACT>
  __device__ __forceinline__ double acc(double t) const {
    double s;
    if (K == 1) return 0.0;
    if (K == 2) { return EXACT ? 0.0 + c3 : c3; }
    if (K == 3) { s = c2 * t; if (EXACT) s = 0.0 + s; return s + c3; }
    s = (c1_2 * t) * t;
    if (EXACT) s = 0.0 + s;
    return (s + c2 * t) + c3;
  }
  // primitive.h:145
  template <bool EXACT>
  __device__ __forceinline__ double jrk(double t) const {
    double s;
    if (K <= 2) return 0.0;
    if (K == 3) { return EXACT ? 0.0 + c2 : c2; }
    s = c1 * t;
    if (EXACT) s = 0.0 + s;
    return s + c2;
  }

  // primitive.h:353-363 with extrema_v :152-162 and solve/quad math.h:117-131,
  // :22-32.  |v(0)| is |c4| for every K (the other terms are exact zeros).
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
      if (c1_2 != 0) {
        const double disc = c2 * c2 - 4 * c1_2 * c3;
        if (!(disc < 0)) {
          const double sq = sqrt(disc);
          const double r1 = (-c2 - sq) / (2 * c1_2);
          const double r2 = (-c2 + sq) / (2 * c1_2);
          // roots are visited in solver order; the scan stops at the first
          // root >= T (primitive.h:156-160)
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
  // primitive.h:369-379 with extrema_a :169-179
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
  // primitive.h:384-394; extrema_j :186-193 has no root because c0 == 0
  __device__ __forceinline__ double max_jrk(double T) const {
    const double j0 = fabs(c2), jT = fabs(jrk<false>(T));
    return (j0 < jT) ? jT : j0;
  }
  // Control effort of this axis, primitive.h:92-122.  For a forward primitive
  // every term but the last is an exact +/-0 and the partial sums stay +0, so
  // J = (u*u)*T bit-for-bit.
  __device__ __forceinline__ double effort(double T) const {
    const double u = (K == 1) ? c4 : (K == 2) ? c3 : (K == 3) ? c2 : c1;
    return u * u * T;
  }
};

// primitive.h:504-525, one end of the primitive.  cs: the host libm's {cos(yaw), sin(yaw)} in the override pass of
// the yaw pinning (YawPin, mplx_internal.h), else null; *amb: the decision is within `margin` of the threshold.
__device__ __forceinline__ bool heading_ok(double vx, double vy, double yaw, double cos_lim, const double *cs,
                                           double margin, double tie_yaw, bool *amb) {
  if (vx != 0 || vy != 0) {
    const double s = sqrt(vx * vx + vy * vy);
    const double c = cs ? cs[0] : cos(yaw), sn = cs ? cs[1] : sin(yaw);
    const double d = vx / s * c + vy / s * sn;
    *amb = *amb || mplx::dev::near_limit(d, cos_lim, margin, vy, yaw, tie_yaw);
    if (d < cos_lim) return false;
  }
  return true;
}

// ------------------------------------------------------------------ the kernel
template <int D, int K, bool YAW>
__global__ __launch_bounds__(kBlock) void expand_kernel(const ExpandArgs A) {
  const int64_t n_slots = A.n_nodes * (int64_t)A.nU;
  const int64_t slot = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (slot >= n_slots) return;
  const int64_t it = slot / A.nU;  // position in the launch; the override pass of the yaw pinning walks a node list
  const int ci = (int)(slot - it * A.nU);
  const int64_t node = (YAW && A.yaw.node_list) ? (int64_t)A.yaw.node_list[it] : it;

  // ---- load the node (broadcast within the wave) and the control (coalesced)
  const double *nd = A.nodes + node;
  const int64_t ns = A.node_stride;
  double cpos[D], cvel[D], cacc[D], cjrk[D];
#pragma unroll
  for (int i = 0; i < D; i++) {
    cpos[i] = nd[(0 * D + i) * ns];
    cvel[i] = (K >= 2) ? nd[(1 * D + i) * ns] : 0.0;
    cacc[i] = (K >= 3) ? nd[(2 * D + i) * ns] : 0.0;
    cjrk[i] = (K >= 4) ? nd[(3 * D + i) * ns] : 0.0;
  }
  const double cyaw = YAW ? nd[(4 * D) * ns] : 0.0;
  const double ct = nd[(4 * D + 1) * ns];
  const double *u = A.U + (int64_t)ci * A.udim;

  const double T = A.dt;
  Axis<K> ax[D];
#pragma unroll
  for (int i = 0; i < D; i++) ax[i].init(cpos[i], cvel[i], cacc[i], cjrk[i], u[i]);
  const double uyaw = YAW ? u[D] : 0.0;

  // ---- successor state tn = pr.evaluate(dt)  (env_map.h:157, primitive.h:321-331)
  double npos[D], nvel[D], nacc[D], njrk[D];
#pragma unroll
  for (int i = 0; i < D; i++) {
    npos[i] = ax[i].template pos<true>(T);
    nvel[i] = ax[i].template vel<true>(T);
    nacc[i] = ax[i].template acc<true>(T);
    njrk[i] = ax[i].template jrk<true>(T);
  }
  double nyaw = 0.0;
  if (YAW) nyaw = wrap_angle((0.0 + uyaw * T) + cyaw);

  const uint64_t h_next = lattice_hash<D, K, YAW>(npos, nvel, nacc, njrk, nyaw);
  const uint64_t h_curr = lattice_hash<D, K, YAW>(cpos, cvel, cacc, cjrk, cyaw);

  // ---- dynamic limits (primitive.h:450-475); max_vel is needed again by the
  //      traversal, so it is computed once here
  double mv[D];
  double max_v = 0;
#pragma unroll
  for (int i = 0; i < D; i++) {
    mv[i] = ax[i].max_vel(T);
    if (mv[i] > max_v) max_v = mv[i];
  }
  bool valid = true;
  if (YAW && A.yaw_max > 0) {
    // override pass (YawPin): per listed node {cos, sin} of yaw(0), then of yaw(T) for every control, and cos(yaw_max),
    // all from the host libm
    const double *tab = A.yaw.tab ? A.yaw.tab + it * A.yaw.tab_stride : nullptr;
    const double cos_lim = tab ? A.yaw.cos_lim : cos(A.yaw_max);
    // evaluate(0): vel = 0.0 + c4 terms, yaw = wrap(0.0 + uyaw*0 + yaw)
    const double y0 = wrap_angle((0.0 + uyaw * 0.0) + cyaw);
    bool amb = false;
    const bool ok0 = heading_ok(ax[0].template vel<true>(0.0), ax[1].template vel<true>(0.0), y0, cos_lim,
                                tab ? tab : nullptr, A.yaw.margin, A.yaw.tie_yaw, &amb);
    const bool okT = heading_ok(nvel[0], nvel[1], nyaw, cos_lim, tab ? tab + 2 + 2 * ci : nullptr, A.yaw.margin,
                                A.yaw.tie_yaw, &amb);
    valid = ok0 && okT;
    if (amb && A.yaw.amb) mplx::dev::flag_node(A.yaw.amb, A.yaw.amb_cap, node, A.yaw.any_host);
  }
  if (K >= 2 && A.v_max > 0) {
#pragma unroll
    for (int i = 0; i < D; i++) valid = valid && !(mv[i] > A.v_max);
  }
  if (K >= 3 && A.a_max > 0) {
#pragma unroll
    for (int i = 0; i < D; i++) valid = valid && !(ax[i].max_acc(T) > A.a_max);
  }
  if (K >= 4 && A.j_max > 0) {
#pragma unroll
    for (int i = 0; i < D; i++) valid = valid && !(ax[i].max_jrk(T) > A.j_max);
  }

  uint8_t st;
  double cost = INFINITY;
  int iters = 0;
  if (h_next == h_curr) {
    st = 0;  // MPLX_SLOT_SKIP_SAME
  } else if (!valid) {
    st = 3;  // MPLX_SLOT_SKIP_DYN
  } else {
    // ---- traverse_primitive (env_map.h:90-132), skipped when the position
    //      does not change at all (env_map.h:163)
    bool same_pos = true;
#pragma unroll
    for (int i = 0; i < D; i++) same_pos = same_pos && (cpos[i] == npos[i]);
    double c = 0;
    bool blocked = false;
    if (!same_pos) {
      int n = (int)ceil(max_v * T / A.res);
      n = n < 5 ? 5 : n;
      const double sdt = T / n;
      const double org[3] = {A.org0, A.org1, A.org2};
      const int dims[3] = {A.dim0, A.dim1, A.dim2};
      const bool want_vel = (A.pot != nullptr && A.grad_w != 0) || (YAW && A.wyaw > 0);
      for (double t = 0; t < T; t += sdt) {
        iters++;
        int cell[D];
        bool outside = false;
#pragma unroll
        for (int i = 0; i < D; i++) {
          // map_util.h:103-108
          cell[i] = (int)round((ax[i].template pos<false>(t) - org[i]) / A.res - 0.5);
          outside = outside || cell[i] < 0 || cell[i] >= dims[i];
        }
        if (outside) { blocked = true; break; }
        int64_t idx = cell[0] + (int64_t)dims[0] * cell[1];
        if (D == 3) idx += (int64_t)dims[0] * dims[1] * cell[2];
        if (A.region != nullptr && !((A.region[idx >> 5] >> (idx & 31)) & 1u)) { blocked = true; break; }
        double vs[D];
        if (want_vel) {
#pragma unroll
          for (int i = 0; i < D; i++) vs[i] = ax[i].template vel<false>(t);
        }
        if (A.pot != nullptr) {
          const int pv = A.pot[idx];
          if (pv < 100 && pv > 0) {
            double gterm = 0;
            if (A.grad_w != 0) {
              double q = 0;
#pragma unroll
              for (int i = 0; i < D; i++) q += vs[i] * vs[i];
              gterm = A.grad_w * sqrt(q);
              c += sdt * (A.pot_w * pv + gterm);
            } else {
              // gradient_weight * norm is an exact +0 when the weight is 0
              c += sdt * (A.pot_w * pv + 0.0);
            }
          } else if (pv >= 100) { blocked = true; break; }
        } else if (A.map[idx] == 100) { blocked = true; break; }
        if (YAW && A.wyaw > 0) {
          double ux, uy;
          if (mplx::dev::heading_unit(vs[0], vs[1], ux, uy)) {  // (mplx_device_common.h: the same unit vector in every kernel)
            const double yw = wrap_angle(uyaw * t + cyaw);
            const double v_value = 1 - (ux * cos(yw) + uy * sin(yw));
            c += A.wyaw * v_value * sdt;
          }
        }
      }
    }
    if (blocked) {
      st = 2;  // MPLX_SLOT_BLOCKED
    } else {
      // env_map.h:164-165, env_base.h:343-345: cost += J + w*dt
      double J = 0;
#pragma unroll
      for (int i = 0; i < D; i++) J += ax[i].effort(T);
      cost = c + (J + A.w * A.dt);
      st = 1;  // MPLX_SLOT_FINITE
    }
  }

  // ---- dense, coalesced slot writes
  // final outputs stream past L2 (never re-read by this kernel); scratch for the compaction stays cached
  const bool stream = A.stream_out != 0;
  const int64_t oslot = node * A.nU + ci;  // == slot except in the override pass of the yaw pinning
  if (A.status) st_out(&A.status[oslot], st, stream);
  if (A.cost) st_out(&A.cost[oslot], cost, stream);
  if (A.hash) st_out(&A.hash[oslot], h_next, stream);
  if (A.iters) st_out(&A.iters[oslot], iters, stream);
  if (A.state) {
    double *o = A.state + oslot;
    const int64_t ss = A.state_stride;
#pragma unroll
    for (int i = 0; i < D; i++) {
      st_out(&o[(0 * D + i) * ss], npos[i], stream);
      st_out(&o[(1 * D + i) * ss], nvel[i], stream);
      st_out(&o[(2 * D + i) * ss], nacc[i], stream);
      st_out(&o[(3 * D + i) * ss], njrk[i], stream);
    }
    st_out(&o[(4 * D) * ss], nyaw, stream);
    st_out(&o[(4 * D + 1) * ss], ct + A.dt, stream);  // env_map.h:161
  }
}

template <int D, int K, bool YAW>
hipError_t launch_one(const ExpandArgs &a, hipStream_t stream) {
  const int64_t n_slots = a.n_nodes * (int64_t)a.nU;
  if (n_slots == 0) return hipSuccess;
  const int64_t blocks = (n_slots + kBlock - 1) / kBlock;
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL((expand_kernel<D, K, YAW>), dim3((unsigned)blocks), dim3(kBlock), 0, stream, a);
  return hipGetLastError();
}

template <int D>
hipError_t launch_dim(int control, const ExpandArgs &a, hipStream_t s) {
  switch (control) {
    case 0x01: return launch_one<D, 1, false>(a, s);
    case 0x03: return launch_one<D, 2, false>(a, s);
    case 0x07: return launch_one<D, 3, false>(a, s);
    case 0x0f: return launch_one<D, 4, false>(a, s);
    case 0x11: return launch_one<D, 1, true>(a, s);
    case 0x13: return launch_one<D, 2, true>(a, s);
    case 0x17: return launch_one<D, 3, true>(a, s);
    case 0x1f: return launch_one<D, 4, true>(a, s);
    default: return hipErrorInvalidValue;
  }
}

__global__ void math_probe_kernel(int op, const double *a, const double *b, double *out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = a[i];
  double r;
  switch (op) {
    case 0: r = x / b[i]; break;
    case 1: r = sqrt(x); break;
    case 2: r = cos(x); break;
    case 3: r = sin(x); break;
    case 4: r = round(x); break;
    default: r = ceil(x); break;
  }
  out[i] = r;
}

__global__ void pack_region_kernel(const uint8_t *bytes, uint32_t *bits, int64_t n_cells) {
  const int64_t wordi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n_words = (n_cells + 31) >> 5;
  if (wordi >= n_words) return;
  uint32_t w = 0;
  const int64_t base = wordi << 5;
  for (int k = 0; k < 32; k++) {
    const int64_t c = base + k;
    if (c < n_cells && bytes[c]) w |= (1u << k);
  }
  bits[wordi] = w;
}

// One workgroup per node: ordered compaction of the emitted slots (status 1/2).
__global__ __launch_bounds__(256) void compact_lists_kernel(const CompactArgs A) {
  __shared__ int s_wsum[4];
  const int64_t nl = blockIdx.x;
  const int64_t node = A.node_offset + nl;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int emitted = 0;
  for (int base = 0; base < A.nU; base += 256) {
    const int ci = base + tid;
    const int64_t slot = nl * A.nU + ci;
    uint8_t st = 0;
    if (ci < A.nU) st = A.status[slot];
    const bool emit = (st == 1 || st == 2);
    const unsigned long long m = __ballot(emit);
    const int within = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_wsum[wv] = __popcll(m);
    __syncthreads();
    int pre = 0, tot = 0;
    for (int i = 0; i < 4; i++) { const int c = s_wsum[i]; if (i < wv) pre += c; tot += c; }
    __syncthreads();
    if (emit) {
      const int64_t idx = node * A.l_nstride + emitted + pre + within;
      if (A.l_action) __builtin_nontemporal_store(ci, &A.l_action[idx]);
      if (A.l_cost) __builtin_nontemporal_store(A.cost[slot], &A.l_cost[idx]);
      if (A.l_hash) __builtin_nontemporal_store(A.hash[slot], &A.l_hash[idx]);
      if (A.l_iters && A.iters) __builtin_nontemporal_store(A.iters[slot], &A.l_iters[idx]);
      if (A.l_state)
        for (int f = 0; f < A.n_fields; f++)
          __builtin_nontemporal_store(A.state[(int64_t)f * A.chunk_slots + slot], &A.l_state[(int64_t)f * A.l_stride + idx]);
    }
    emitted += tot;
  }
  if (tid == 0 && A.l_count) A.l_count[node] = emitted;
}

}  // namespace

hipError_t launch_compact_lists(const CompactArgs &a, hipStream_t stream) {
  if (a.n_nodes_chunk <= 0) return hipSuccess;
  hipLaunchKernelGGL(compact_lists_kernel, dim3((unsigned)a.n_nodes_chunk), dim3(256), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_expand(int dim, int control, const ExpandArgs &args, hipStream_t stream) {
  if (dim == 2) return launch_dim<2>(control, args, stream);
  if (dim == 3) return launch_dim<3>(control, args, stream);
  return hipErrorInvalidValue;
}

hipError_t launch_math_probe(int op, const double *a, const double *b, double *out, int64_t n,
                             hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  const int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(math_probe_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, op, a, b, out, n);
  return hipGetLastError();
}

hipError_t launch_pack_region(const uint8_t *bytes, uint32_t *bits, int64_t n_cells, hipStream_t stream) {
  const int64_t n_words = (n_cells + 31) >> 5;
  if (n_words <= 0) return hipSuccess;
  const int64_t blocks = (n_words + 255) / 256;
  hipLaunchKernelGGL(pack_region_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, bytes, bits, n_cells);
  return hipGetLastError();
}

}  // namespace mplx
