// expand_pair_kernel.hip -- the factorised successor expansion for yaw controls on a potential map (BASELINE config 5),
// TWO NODES PER WAVEFRONT.
//
// Same function, same results as expand_grid_kernel.hip<D, K, YAW = true, POT = true>
//   MPL::env_map<Dim>::get_succ, reference include/mpl_planner/env/env_map.h:147-172 with traverse_primitive :90-132
//   (potential map :113-118, heading cost :121-129), validate_yaw include/mpl_basis/primitive.h:504-525,
// for the launches that kernel serves worst: a large frontier of which the pre-screen (grid_prescreen_kernel) leaves a
// few thousand live nodes.  What round 6 measured on C5 (profiles/r06_c5_sharing_variants.txt): 5.2 k live nodes on the
// 4 096 waves a 125-VGPR kernel keeps resident; every wave runs ONE node's dependent chain of ~25 us, a quarter of them
// a second one -- two rounds; 46 % of a node's instructions are set-up phases with 3 - 14 of 64 lanes busy; the
// per-sample tables of the heading cost take 6.4 of a node's 8.7 KB of LDS.  Here
//   * a node is a GROUP of 32 lanes (its tables have 9 - 20 entries, its list <= 81 - 128 pairs): a wave carries two
//     nodes through every phase, so 5.2 k nodes are 2.6 k wave tasks -- ONE round on 3 072 resident waves (3 per SIMD:
//     the kernel may use 168 VGPRs) -- and the low-lane phases run at twice the utilisation;
//   * the velocity of a sample is computed where it is used (K = 2: one multiply-add per axis from the sample time)
//     instead of being staged per (entry, sample) in LDS: a node needs 6 KB, 24 nodes fit a CU;
//   * a lane per COMBINATION of axis entries walks the samples once for all of its yaw rates (same cells, same potential
//     terms, one v.normalized() per sample; the heading term per yaw rate), the costs wait in LDS at their list positions
//     and a dense pass writes every row of the list in whole lines.
// Everything that decides a RESULT is the arithmetic of expand_grid_kernel.hip, expression for expression
// (mplx_grid_common.h, mplx_device_common.h; -ffp-contract=off): the two kernels write bit-identical lists
// (tests/test_gpu_fullsize.py runs C5, C5 with a tunnel and the 2D variant through both; tests/test_gpu_lists.py).
//
// Scope: Dim 2 / 3, ACCxYAW / JRKxYAW, potential map (with or without a search region, gradient weight, heading cost),
// lexicographic control table with D * ndp <= 16 entries and <= 4 yaw rates, a pre-screened frontier (GridArgs::live).
// Everything else -- and the host-libm fix pass of the yaw pinning -- stays with expand_grid_kernel.hip.
#include "mplx_grid_common.h"

namespace mplx {
namespace {

static_assert(MPLX_GRID_TT_RESIDENT == 1, "the sample times are read from the workgroup's resident table");
constexpr int kPairGS = 32;                 // lanes per node
constexpr int kPairNG = 64 / kPairGS;       // nodes per wave
constexpr int kPairWPB = 4;                 // waves per workgroup
constexpr int kPairBT = 64 * kPairWPB;
constexpr int kPairNYMax = 4;               // yaw rates a lane carries through its sample loop (instantiations NY = 2, 3, 4)
#ifndef MPLX_PAIR_UB
#define MPLX_PAIR_UB 4
#endif
constexpr int kPairUB = MPLX_PAIR_UB;       // samples per step of the sample loop (C5: 4 = 8 = 16 in time, 4 is the fewest registers)

typedef const GridArgs __attribute__((address_space(4))) *GridKernargPtr;

// ---- A node belongs to a GROUP of GS lanes (GS = 32: two nodes per wave).  vl = lane within the group, gsh = the group's
// first lane.  The phases below are expand_grid_kernel.hip's, statement for statement, with the wave's lanes replaced by the
// group's: ballots are masked to the group, loops step by GS, what was wave-uniform is group-uniform.
template <int GS>
__device__ __forceinline__ unsigned long long gballot(int gsh, bool p) {  // ballot over the group, as bits 0 .. GS - 1
  const unsigned long long b = __ballot(p);
  if (GS == 64) return b;
  return (b >> gsh) & ((1ull << GS) - 1ull);
}

struct NodeTabs {  // where one node's tables live (LDS)
  double *node, *est;
  uint64_t *hp;
  int *eq, *eflag, *misc;
  double *uq, *yawT, *ycs;
  int *yq;
  unsigned short *hmask;
};

#define A (*Ak)
// Phases T1 (axis entries, the node's lattice integers, the yaw values, the valid lists) and the prefix / heading-mask
// tables (expand_grid_kernel.hip, "phase T1" .. "validate_yaw ... for every (x entry, y entry)").  Returns this lane's
// entry flag; rb_lo / rb_hi: the cells its p(t) spans (unused here).  report: flag the node for the host-libm pass when a
// heading decision is within rounding noise of its threshold (false for a group that only repeats another group's node).
template <int D, int K, bool YAW, int GS>
__device__ __forceinline__ int grid_node_setup(GridKernargPtr Ak, const NodeTabs &t, const double *s_uval, const double *s_uyaw,
                                               int vl, int gsh, int64_t node, const double *ytab, bool pinned, double cos_lim,
                                               bool report, bool yaw_amb, int &rb_lo, int &rb_hi) {
  constexpr int KQ = K == 3 ? 4 : K;
  constexpr int NQ0 = GS == 64 ? 48 : 20;  // first lane of the node's own lattice integers (GS = 32: D * ndp <= 20)
  const int ndp = A.ndp;
  const int EN = D * ndp, PN = (D == 3) ? ndp * ndp : ndp;
  const int ndy = YAW ? A.ndy : 0;
  const double T = A.dt;
  const double org[3] = {A.org0, A.org1, A.org2};
  const int nd[3] = {A.nd0, A.nd1, A.nd2};
    // ---- phase T1: axis entries; the node's own lattice integers (lanes NQ0 ..)
    int flag = 0;
    rb_lo = 0x7fffffff;  // this lane's entry: cells its p(t) spans (free-box query)
    rb_hi = (int)0x80000000;
    if (vl < EN) {
      const int ax = vl / ndp, jv = vl - ax * ndp;
      if (jv < nd[ax]) {
        const double p = t.node[0 * D + ax];
        const double v = (K >= 2) ? t.node[1 * D + ax] : 0.0;
        const double a = (K >= 3) ? t.node[2 * D + ax] : 0.0;
        const double j = (K >= 4) ? t.node[3 * D + ax] : 0.0;
        const double u = s_uval[vl];
        Ax<K> q;
        q.init(p, v, a, j, u);
        const double mv = q.max_vel(T);
        bool valid = true;
        if (K >= 2 && A.v_max > 0) valid = valid && !(mv > A.v_max);
        if (K >= 3 && A.a_max > 0) valid = valid && !(q.max_acc(T) > A.a_max);
        if (K >= 4 && A.j_max > 0) valid = valid && !(q.max_jrk(T) > A.j_max);
        // env_map.h:95, one axis' share of n = max(5, (int)ceil(max_v * T / res))
        int n = (int)ceil(div_by(mv * T, A.res, A.Rres));
        n = n < 5 ? 5 : (n > A.n_max ? A.n_max : n);
        const double np_ = q.template pos<true>(T);
        const double nv_ = q.template vel<true>(T);
        const double na_ = q.template acc<true>(T);
        const double nj_ = q.template jrk<true>(T);
        // fields of order < K - 1; order K - 1 is (0.0 + u*T) + x0, order K is 0.0 + u, higher ones are 0
        // (primitive.h:128-145; the same expressions Ax<K>::pos/vel/acc/jrk<true> evaluate)
        if (K >= 3) t.uq[vl] = q.top_quotient();
        if (K >= 2) t.est[vl * (K - 1) + 0] = np_;
        if (K >= 3) t.est[vl * (K - 1) + 1] = nv_;
        if (K >= 4) t.est[vl * (K - 1) + 2] = na_;
        t.eq[vl * KQ + 0] = quantise(np_, 0.01, A.R001);
        if (K >= 2) t.eq[vl * KQ + 1] = quantise(nv_, 0.1, A.R01);
        if (K >= 3) t.eq[vl * KQ + 2] = quantise(na_, 0.1, A.R01);
        if (K >= 4) t.eq[vl * KQ + 3] = quantise(nj_, 0.1, A.R01);
        flag = (valid ? 1 : 0) | ((p == np_) ? 2 : 0) | (n << 8);
        if (A.sat != nullptr && valid) {
          // range of p(t) over [0, T] of this entry, as cells with one cell of slack on both sides
          // (free-box shortcut below); K = 1, 2: exact extrema; K = 3: |p - p0| <= max_vel * T
          double pmin = p < np_ ? p : np_, pmax = p < np_ ? np_ : p;
          if (K == 2 && u != 0) {
            const double ts = -v / u;
            if (ts > 0 && ts < T) {
              const double pe = q.template pos<false>(ts);
              pmin = pe < pmin ? pe : pmin;
              pmax = pe > pmax ? pe : pmax;
            }
          }
          if (K >= 3) { pmin = p - mv * T; pmax = p + mv * T; }
          rb_lo = (int)floor(div_by(pmin - org[ax], A.res, A.Rres)) - 1;  // (same quotient as `/`)
          rb_hi = (int)floor(div_by(pmax - org[ax], A.res, A.Rres)) + 1;
        }
        if (jv == 0) {
          // the node's own cell on this axis (map_util.h:103-108); the codes are offsets from it.
          // Every negative cell is outside the map alike, so -1 stands for all of them.
          const double qd = div_by(p - org[ax], A.res, A.Rres);
          t.misc[M_BASE + ax] = (qd - 0.5 > -0.5) ? (int)qd : -1;
        }
      }
      t.eflag[vl] = flag;
    } else if (vl >= NQ0 && vl < NQ0 + 4 * D) {
      // lattice integers of the node itself (waypoint.h:93-125), one field per lane
      const int i = (vl - NQ0) >> 2, f = (vl - NQ0) & 3;
      if (f < K) {
        const double x = t.node[f * D + i];
        t.misc[M_NODEQ + i * 4 + f] = f == 0 ? quantise(x, 0.01, A.R001) : quantise(x, 0.1, A.R01);
      }
    }
    if (YAW) {
      // per yaw value: yaw(T) = wrap(p(T)) of the yaw polynomial (primitive.h:329), its lattice integer
      // (waypoint.h:113-116) and, for the heading limit, its cos / sin
      const double cyaw = t.node[4 * D];
      if (vl < ndy) {
        const double yT = wrap_angle((0.0 + s_uyaw[vl] * T) + cyaw);
        t.yawT[vl] = yT;
        t.yq[vl] = quantise(yT, 0.1, A.R01);
        if (A.yaw_max > 0) {
          double sn_, cs_;
          if (pinned) { cs_ = ytab[2 + vl]; sn_ = ytab[2 + 16 + vl]; }
          else sincos(yT, &sn_, &cs_);  // same values as cos() / sin() (one argument reduction instead of two)
          t.ycs[vl * 2 + 0] = cs_;
          t.ycs[vl * 2 + 1] = sn_;
        }
      }
      if (vl == GS - 1) t.misc[M_YQ] = quantise(cyaw, 0.1, A.R01);
    }
    {
      // per axis, the values that pass the limits, in order: the only entries whose samples are ever needed
      const unsigned long long vm = gballot<GS>(gsh, (flag & 1) != 0);
      if (vl < EN) {
        const int ax = vl / ndp, jv = vl - ax * ndp;
        const unsigned long long am = (((1ull << ndp) - 1ull) << (ax * ndp)) & vm;
        if (flag & 1) ((unsigned char *)(t.misc + M_VL))[ax * 16 + __popcll(am & ((1ull << vl) - 1ull))] = (unsigned char)jv;
        if (jv == 0) t.misc[M_NV + ax] = __popcll(am);
      }
    }
    wave_sync();

    // ---- prefix tables over the first D-1 axes
    for (int x = vl; x < PN; x += GS) {
      int e0, e1 = 0;
      bool ok;
      if (D == 3) {
        const int j0 = x / ndp, j1 = x - j0 * ndp;
        ok = j0 < nd[0] && j1 < nd[1];
        e0 = j0;
        e1 = ndp + j1;
      } else {
        ok = x < nd[0];
        e0 = x;
      }
      if (ok) {
        uint64_t h = 0;
        fold_entry<K>(h, t.eq, e0);
        if (D == 3) fold_entry<K>(h, t.eq, e1);
        t.hp[x] = h;
      }
    }
    if (YAW) {
      // validate_yaw (primitive.h:504-525) at t = 0 and t = T for every (x entry, y entry): the set of yaw
      // values whose heading stays within yaw_max of the velocity direction
      const bool lim = A.yaw_max > 0;
      const double y0 = wrap_angle((0.0 + 0.0) + t.node[4 * D]);  // yaw polynomial at t = 0: (0.0 + u_yaw * 0.0) + yaw
      double c0 = 0.0, s0 = 0.0;
      if (lim) { if (pinned) { c0 = ytab[0]; s0 = ytab[1]; } else sincos(y0, &s0, &c0); }
      const float inv_n1 = 1.0f / (float)nd[1];
      for (int x = vl; x < nd[0] * nd[1]; x += GS) {
        const int j0 = (int)(((float)x + 0.5f) * inv_n1), j1 = x - j0 * nd[1];
        unsigned int mask = 0xffffu;
        if (lim) {
          Ax<K> qx, qy;
          qx.init(t.node[0], (K >= 2) ? t.node[1 * D] : 0.0, (K >= 3) ? t.node[2 * D] : 0.0, (K >= 4) ? t.node[3 * D] : 0.0, s_uval[j0]);
          qy.init(t.node[1], (K >= 2) ? t.node[1 * D + 1] : 0.0, (K >= 3) ? t.node[2 * D + 1] : 0.0, (K >= 4) ? t.node[3 * D + 1] : 0.0, s_uval[ndp + j1]);
          const double vx0 = qx.template vel<true>(0.0), vy0 = qy.template vel<true>(0.0);
          if (vx0 != 0 || vy0 != 0) {
            const double sn = sqrt(vx0 * vx0 + vy0 * vy0);
            const double d = vx0 / sn * c0 + vy0 / sn * s0;
            if (d < cos_lim) mask = 0;
            yaw_amb = yaw_amb || near_limit(d, cos_lim, A.yaw.margin, vy0, y0, A.yaw.tie_yaw);
          }
          const double vxT = qx.template vel<true>(T), vyT = qy.template vel<true>(T);
          if (vxT != 0 || vyT != 0) {
            const double sn = sqrt(vxT * vxT + vyT * vyT);
            const double nx = vxT / sn, ny = vyT / sn;
            for (int jy = 0; jy < ndy; jy++) {
              const double d = nx * t.ycs[jy * 2] + ny * t.ycs[jy * 2 + 1];
              if (d < cos_lim) mask &= ~(1u << jy);
              yaw_amb = yaw_amb || near_limit(d, cos_lim, A.yaw.margin, vyT, t.yawT[jy], A.yaw.tie_yaw);
            }
          }
        }
        t.hmask[j0 * ndp + j1] = (unsigned short)mask;
      }
      if (report && A.yaw.amb && gballot<GS>(gsh, yaw_amb) != 0ull && vl == 0) flag_node(A.yaw.amb, A.yaw.amb_cap, node, A.yaw.any_host);
    }
    wave_sync();
  return flag;
}

#undef A

#define A (*Ak)
template <int D, int K, int kPairNY>
__global__ __launch_bounds__(kPairBT) __attribute__((amdgpu_waves_per_eu(3)))
void expand_pair_kernel(const GridArgs A_kernarg) {
  extern __shared__ __align__(16) unsigned char smem[];
  GridKernargPtr Ak = (GridKernargPtr)__builtin_amdgcn_kernarg_segment_ptr();  // (read where it lies: expand_grid_kernel.hip)
  (void)A_kernarg;
  constexpr int F = 4 * D + 2, GS = kPairGS, NG = kPairNG;
  const int nU = A.nU, ndp = A.ndp, RM = A.rmax, ndy = A.ndy;
  const bool ycost = A.wyaw > 0;      // env_map.h:121: per-sample heading cost
  const bool gcost = A.grad_w != 0;   // env_map.h:116: gradient_weight * |vel| per sample inside the potential field
  const GridLds L(D, K, kPairWPB * NG, nU, ndp, A.n_max, RM, 64, pair_lds_mode(ycost), ndy, 1);
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int vl = lane & (GS - 1), gsh = lane & ~(GS - 1), grp = lane / GS;
  const double *s_uval = (const double *)(smem + L.o_uval);
  const unsigned char *s_tc = smem + L.o_tc;
  const double *s_tt = (const double *)(smem + L.o_tt);
  const double *s_uyaw = (const double *)(smem + L.o_uyaw);
  unsigned char *wb = smem + L.o_wave0 + (wv * NG + grp) * L.wave_bytes;  // the GROUP's block
  double *s_node = (double *)(wb + L.w_node);
  double *s_est = (double *)(wb + L.w_est);
  uint64_t *s_hp = (uint64_t *)(wb + L.w_hp);
  int *s_eq = (int *)(wb + L.w_eq);
  int *s_eflag = (int *)(wb + L.w_eflag);
  int *s_misc = (int *)(wb + L.w_misc);
  unsigned short *s_rowmap = (unsigned short *)(wb + L.w_rowmap);
  // per emitted pair (list position): the cost and iteration count its combination's lane found; per combination: entries,
  // sample count, yaw mask, first list position; the list itself (packed entry indices per position)
  double *s_pc = (double *)(wb + L.w_list);
  unsigned int *s_cmb = (unsigned int *)(wb + L.w_list + nU * 8);
  unsigned short *s_ceb = (unsigned short *)(wb + L.w_list + nU * 8 + L.PNC * 4);
  unsigned short *s_list = (unsigned short *)(wb + L.w_list + nU * 8 + L.PNC * 6 + (L.PNC & 1) * 2);
  unsigned short *s_pi = s_list + ((nU + 1) & ~1);
  unsigned char *s_cell = wb + L.w_cell;
  double *s_yawT = (double *)(wb + L.w_yaw);
  double *s_ycs = (double *)(wb + L.w_ycs);
  int *s_yq = (int *)(wb + L.w_yq);
  unsigned short *s_hmask = (unsigned short *)(wb + L.w_hmask);
  double *s_uq = (double *)(wb + L.w_uq);
  double *s_ycsr = (double *)(wb + L.w_ycsr);
  const NodeTabs tabs{s_node, s_est, s_hp, s_eq, s_eflag, s_misc, s_uq, s_yawT, s_ycs, s_yq, s_hmask};

  const int tts = L.tts, EN = L.EN;
  const int rowcap = RM * tts;
  const int half = A.n_max + 2;  // cell-offset code = offset from the node's cell + half
  const double T = A.dt;
  const double org[3] = {A.org0, A.org1, A.org2};
  const int dims[3] = {A.dim0, A.dim1, A.dim2};
  const int nd[3] = {A.nd0, A.nd1, A.nd2};

  // ---- once per (persistent) workgroup: shared read-only tables
  {
    double *uv = (double *)(smem + L.o_uval);
    for (int i = threadIdx.x; i < EN; i += kPairBT) {
      const int ax = i / ndp, j = i - ax * ndp;
      uv[i] = A.uvals[ax * 16 + j];
    }
    if ((int)threadIdx.x < ndy) ((double *)(smem + L.o_uyaw))[threadIdx.x] = A.uvals[3 * 16 + threadIdx.x];
    if (threadIdx.x < 64) smem[L.o_tc + threadIdx.x] = A.tcnt[threadIdx.x];
    double *tt = (double *)(smem + L.o_tt);
    const int ntt = (A.n_max + 1) * tts;
    for (int i = threadIdx.x; i < ntt; i += kPairBT) {
      const int nn = i / tts, k = i - nn * tts;
      tt[i] = A.ttab[nn * kTabStride + k];
    }
  }
  __syncthreads();  // the only workgroup barrier

  const double cos_lim = A.yaw_max > 0 ? cos(A.yaw_max) : 0.0;  // primitive.h:521
  const int NN = (int)*A.live_n;            // the pre-screen's survivors
  const int NI = (NN + NG - 1) / NG;        // wave tasks: NG survivors each
  const int wave_id = (int)blockIdx.x * kPairWPB + wv;
  const int wave_stride = (int)gridDim.x * kPairWPB;
  // Fewer tasks than waves (the usual case: the launch is sized for them): the tasks are SPREAD over the waves -- wave w
  // takes task floor(w NI / W) if that is a new one -- so that the idle waves are everywhere and every CU carries the same
  // share, instead of the first NI waves working on three resident workgroups per CU and the last ones on two.
  int it_first = wave_id, it_step = wave_stride;
  if (NI <= wave_stride) {
    const int t0 = (int)(((int64_t)wave_id * NI) / wave_stride), t1 = (int)(((int64_t)(wave_id + 1) * NI) / wave_stride);
    it_first = t1 > t0 ? t0 : NI;
    it_step = NI;
  }
  for (int it = it_first; it < NI; it += it_step) {
    asm volatile("" : "+s"(Ak));  // (the argument loads stay inside the iteration: expand_grid_kernel.hip)
    const int li = it * NG + grp;
    const bool real = li < NN;  // (an odd survivor count: the last task's second group repeats the first one's node and stores nothing)
    const int64_t node = (int64_t)A.live[real ? li : NN - 1];
    // ---- phase 0: node state into LDS
    wave_prio(0);
    wave_sync();
    if (vl < F) s_node[vl] = A.nodes[(int64_t)vl * A.node_stride + node];
    wave_sync();
    // ---- phases T1 .. heading masks
    int rb_lo, rb_hi;
    const int flag = grid_node_setup<D, K, true, GS>(Ak, tabs, s_uval, s_uyaw, vl, gsh, node, nullptr, false, cos_lim, real, false, rb_lo, rb_hi);
    uint64_t hcur = 0;  // hash of the node, folded by every lane alike
#pragma unroll
    for (int i = 0; i < D; i++) {
      const int4 q = *(const int4 *)(s_misc + M_NODEQ + i * 4);
      fold(hcur, q.x);
      if (K >= 2) fold(hcur, q.y);
      if (K >= 3) fold(hcur, q.z);
    }
    fold(hcur, s_misc[M_YQ]);
    const double node_t = s_node[4 * D + 1];
    int base_c[3];
#pragma unroll
    for (int i = 0; i < 3; i++) base_c[i] = (i < D) ? s_misc[M_BASE + i] : 0;
    // ---- phase A: every COMBINATION of the entries inside the limits (x, y, z; lexicographic, z fastest): which of its
    // yaw values are emitted -- the heading mask of its (x, y) entries, and the successor must differ from the node
    // (env_map.h:158: a hash comparison) -- its sample count, and its first position in the node's list: the emitted
    // pairs in ascending control order are the combinations in order, each with its emitted yaw values in order.
    // s_cmb[x] = j0 | j1 << 4 | j2 << 8 | n << 12 | yaw mask << 18 (ndy <= 14), s_ceb[x] = first list position.
    const int nv0 = s_misc[M_NV + 0], nv1 = s_misc[M_NV + 1], nv2 = (D == 3) ? s_misc[M_NV + 2] : 1;
    const int n12 = nv1 * nv2, ncomb = nv0 * n12;
    const unsigned char *vl_ = (const unsigned char *)(s_misc + M_VL);
    int E = 0;  // emitted successors of the node (uniform over the group)
    unsigned long long nm = 0;
    {
      const float r_n12 = 1.0f / (float)(n12 > 0 ? n12 : 1), r_n2 = 1.0f / (float)(nv2 > 0 ? nv2 : 1);
      unsigned int nm_lo = 0, nm_hi = 0;
      for (int x0 = 0; x0 < ncomb; x0 += GS) {
        const int x = x0 + vl;
        unsigned int mask = 0, word = 0;
        if (x < ncomb) {
          const int a_ = (int)(((float)x + 0.5f) * r_n12);  // exact: x < 2^12
          const int ra = x - a_ * n12;
          const int b_ = (D == 3) ? (int)(((float)ra + 0.5f) * r_n2) : ra;
          const int j0 = vl_[a_], j1 = vl_[16 + b_], j2 = (D == 3) ? (int)vl_[32 + ra - b_ * nv2] : 0;
          const int px = (D == 3) ? __umul24(j0, ndp) + j1 : j0;
          const int eL = (D - 1) * ndp + ((D == 3) ? j2 : j1);
          uint64_t hc = s_hp[px];
          fold_entry<K>(hc, s_eq, eL);
          const unsigned int hm = s_hmask[__umul24(j0, ndp) + j1];
          const int fl = pair_flags<D>(s_eflag, ndp, j0, j1, j2);  // (valid by construction: fl & 1)
          const int n = (fl & 2) ? 0 : (fl >> 8);  // unchanged position: not traversed (env_map.h:163)
          for (int jy = 0; jy < ndy; jy++) {
            uint64_t h = hc;
            fold(h, s_yq[jy]);
            if (((hm >> jy) & 1u) && h != hcur) mask |= 1u << jy;
          }
          word = (unsigned)j0 | ((unsigned)j1 << 4) | ((unsigned)j2 << 8) | ((unsigned)n << 12) | (mask << 18);
          if (mask && n) { if (n < 32) nm_lo |= 1u << n; else nm_hi |= 1u << (n - 32); }
        }
        // first list position: the emitted pairs of the combinations before this one (exclusive scan over the group)
        const int cnt = __popc(mask);
        int incl = cnt;
#pragma unroll
        for (int d = 1; d < GS; d <<= 1) {
          const int o = __shfl_up(incl, d, GS);
          if (vl >= d) incl += o;
        }
        if (x < ncomb) {
          const int eb = E + incl - cnt;
          s_cmb[x] = word;
          s_ceb[x] = (unsigned short)eb;
          int rank = 0;
          for (int jy = 0; jy < ndy; jy++)  // the list: packed entry indices of every emitted pair, in order
            if ((mask >> jy) & 1u) s_list[eb + rank++] = (unsigned short)((word & 0xfffu) | ((unsigned)jy << 12));
        }
        E += __shfl(incl, GS - 1, GS);
      }
#pragma unroll
      for (int d = GS >> 1; d > 0; d >>= 1) {
        nm_lo |= (unsigned int)__shfl_xor((int)nm_lo, d, 64);
        nm_hi |= (unsigned int)__shfl_xor((int)nm_hi, d, 64);
      }
      nm = (unsigned long long)nm_lo | ((unsigned long long)nm_hi << 32);
    }
    wave_sync();
    if (vl == 0 && real && A.l_count) A.l_count[node] = E;
    if (A.dbg & 1) nm = 0;  // timing ablation: no sampling
    // the node's own state and this lane's controls, for the velocity of a sample (Waypoint::vel, primitive.h:321-331)
    wave_prio(1);
    // ---- rounds of as many of the smallest pending sample counts as fit rowcap slots per entry
    for (int pass = 0; pass == 0 || nm != 0ull; pass++) {
      unsigned long long sub = 0;
      {
        int used = 0;
        for (unsigned long long t = nm; t; t &= t - 1ull) {
          const int nn = __ffsll((long long)t) - 1;
          const int cn = (int)s_tc[nn];
          if (used + cn > rowcap && sub) break;  // (the first count always fits: cn <= tts <= rowcap)
          sub |= 1ull << nn;
          used += cn;
        }
      }
      nm &= ~sub;
      wave_sync();
      {
        // offset of every selected count's row: slots of the selected counts below it (count nn is lane nn % GS's)
        int off = 0;
        unsigned short mine_off[NG];
#pragma unroll
        for (int h = 0; h < NG; h++) mine_off[h] = 0xffff;
        for (unsigned long long t = sub; t; t &= t - 1ull) {
          const int nn = __ffsll((long long)t) - 1;
#pragma unroll
          for (int h = 0; h < NG; h++)
            if (nn == vl + GS * h) mine_off[h] = (unsigned short)off;
          off += (int)s_tc[nn];
        }
#pragma unroll
        for (int h = 0; h < NG; h++)
          if (vl + GS * h <= A.n_max) s_rowmap[vl + GS * h] = mine_off[h];  // (n_max <= 61)
      }
      wave_sync();
      // rows: cell-offset codes of every needed axis entry at t_0 .. t_{cnt-1} of each sample count (expand_grid_kernel.hip,
      // "rows"); the heading of every yaw value at those times
      {
        int row = 0;
        for (unsigned long long t = sub; t; t &= t - 1ull) {
          const int nn = __ffsll((long long)t) - 1;
          const int cn = (int)s_tc[nn];
          const float inv_cn = 1.0f / (float)cn;
          const double *trow = s_tt + nn * tts;
          // the row of (entry, count nn) is only read by pairs whose count IS nn, i.e. by entries with n_entry <= nn
          const unsigned long long fm = gballot<GS>(gsh, vl < EN && (flag & 1) && (flag >> 8) <= nn);
          wave_sync();  // (the previous count's list has been read)
          if (vl < EN && ((fm >> vl) & 1ull)) {
            const int ax_ = vl / ndp;
            const unsigned long long am = (((1ull << ndp) - 1ull) << (ax_ * ndp)) & fm;
            ((unsigned char *)(s_misc + M_VLC))[ax_ * 16 + __popcll(am & ((1ull << vl) - 1ull))] = (unsigned char)(vl - ax_ * ndp);
          }
          wave_sync();
#pragma unroll
          for (int ax = 0; ax < D; ax++) {
            const double p0 = s_node[0 * D + ax];
            const double v0 = s_node[1 * D + ax];
            const double a0 = (K >= 3) ? s_node[2 * D + ax] : 0.0;
            const int shift = half - base_c[ax];
            const int nv = __popcll(fm & (((1ull << ndp) - 1ull) << (ax * ndp)));
            const unsigned char *vlc = (const unsigned char *)(s_misc + M_VLC) + ax * 16;
            for (int x = vl; x < nv * cn; x += GS) {
              const int vi = (int)(((float)x + 0.5f) * inv_cn);  // exact: x < 2^12
              const int k = x - __umul24(vi, cn);
              const int aj = ax * ndp + (int)vlc[vi];
              Ax<K> q;
              q.init(p0, v0, a0, 0.0, s_uval[aj]);
              // map_util.h:103-108: cell = round((pos - origin) / res - 0.5), see expand_grid_kernel.hip
              const double qd = div_by(q.pos_q(trow[k], K >= 3 ? s_uq[aj] : 0.0) - org[ax], A.res, A.Rres);
              const int c = (qd - 0.5 > -0.5) ? (int)qd : -1;
              s_cell[__umul24(aj, rowcap) + row + k] = (unsigned char)(c + shift);  // 0 < code < 2 * half (exact velocity maxima for K <= 3)
            }
          }
          if (ycost) {
            // heading of every yaw value at the sample times: wrap(p_yaw(t)), then cos / sin once per node
            const double cyaw = s_node[4 * D];
            for (int x = vl; x < ndy * cn; x += GS) {
              const int jy = (int)(((float)x + 0.5f) * inv_cn);
              const int k = x - __umul24(jy, cn);
              const double yw = wrap_angle(s_uyaw[jy] * trow[k] + cyaw);
              double *o = s_ycsr + (__umul24(jy, rowcap) + row + k) * 2;
              double sn_, cs_;
              sincos(yw, &sn_, &cs_);
              o[0] = cs_;
              o[1] = sn_;
            }
          }
          row += cn;
        }
      }
      wave_sync();

      // ---- phase D1: the combinations, GS at a time; a lane walks the samples of its combination ONCE for all of its
      // emitted yaw values (same cells, same potential terms, same unit velocity; the heading term per yaw value) and
      // leaves cost and iteration count of each in LDS, at the pair's list position
      for (int x0 = 0; x0 < ncomb; x0 += GS) {
        const int x = x0 + vl;
        const unsigned int cw = x < ncomb ? s_cmb[x] : 0u;
        const int j0 = cw & 15, j1 = (cw >> 4) & 15, j2 = (cw >> 8) & 15;
        const int n = (int)((cw >> 12) & 63u);
        const unsigned int ymask = cw >> 18;
        const int eb = x < ncomb ? (int)s_ceb[x] : 0;
        const int en[3] = {j0, ndp + j1, 2 * ndp + j2};
        const bool smp = ymask != 0u && n != 0 && ((sub >> n) & 1ull) != 0ull;
        const int cntl = smp ? (int)s_tc[n] : 0;  // iterations of `for (t = 0; t < T; t += T/n)`
        int fb = -1;                              // first blocked sample
        double csum[kPairNY];                     // traverse_primitive's accumulated cost, per yaw value
#pragma unroll
        for (int y = 0; y < kPairNY; y++) csum[y] = 0.0;
        {
          const int r = smp ? (int)s_rowmap[n] : 0;  // slot offset of the combination's rows
          int ptr[3] = {0, 0, 0};
#pragma unroll
          for (int i = 0; i < D; i++) ptr[i] = __umul24(en[i], rowcap) + r;
          bool done = !smp || (A.dbg & 32);
          const double sdt = smp ? T / n : 0.0;  // env_map.h:96
          const double *trow = s_tt + (smp ? n : 0) * tts;
          // Waypoint::vel of a sample (primitive.h:321-331) from its time: the expression the velocity rows of
          // expand_grid_kernel.hip hold (Ax::vel<false> at the table's accumulated time)
          Ax<K> qv[D];
#pragma unroll
          for (int i = 0; i < D; i++) qv[i].init(s_node[0 * D + i], s_node[1 * D + i], (K >= 3) ? s_node[2 * D + i] : 0.0, 0.0, s_uval[en[i]]);
          for (int k0 = 0; __ballot(!done) != 0ull; k0 += kPairUB) {
            int val[kPairUB];
            bool bad[kPairUB];
#pragma unroll
            for (int q = 0; q < kPairUB; q++) {
              int k = k0 + q;
              k = k < cntl ? k : (cntl > 0 ? cntl - 1 : 0);
              bool inside = !done;
              int64_t cell = 0, mul = 1;
#pragma unroll
              for (int i = 0; i < D; i++) {
                const int c = base_c[i] + (done ? 0 : (int)s_cell[ptr[i] + k]) - half;
                inside = inside && c >= 0 && c < dims[i];
                cell += mul * c;
                mul *= dims[i];
              }
              const int64_t ci_ = inside ? cell : 0;
              const bool in_reg = A.region == nullptr || ((A.region[ci_ >> 5] >> (ci_ & 31)) & 1u);
              val[q] = A.pot[ci_];
              bad[q] = !inside || !in_reg;
            }
#pragma unroll
            for (int q = 0; q < kPairUB; q++) {
              if (!done && k0 + q < cntl) {
                if (bad[q] || val[q] >= 100) { fb = k0 + q; done = true; }
                else {
                  const int k = k0 + q;
                  if (val[q] > 0) {
                    double term;
                    if (gcost) {  // env_map.h:115-116: dt * (potential_weight * value + gradient_weight * vel.norm())
                      const double tk = trow[k];
                      double vn = 0;
#pragma unroll
                      for (int i = 0; i < D; i++) {
                        const double vi_ = qv[i].template vel<false>(tk);
                        vn += vi_ * vi_;
                      }
                      term = sdt * (A.pot_w * val[q] + A.grad_w * sqrt(vn));
                    } else {
                      term = sdt * (A.pot_w * val[q]);
                    }
#pragma unroll
                    for (int y = 0; y < kPairNY; y++) csum[y] += term;
                  }
                  if (ycost) {
                    // env_map.h:121-129: the heading cost of the sample, after its potential term; v.normalized() once,
                    // the heading of every yaw value from the rows
                    const double tk = trow[k];
                    const double vx = qv[0].template vel<false>(tk), vy = qv[1].template vel<false>(tk);
                    double ux, uy;
                    if (heading_unit(vx, vy, ux, uy)) {
#pragma unroll
                      for (int y = 0; y < kPairNY; y++) {
                        if (y < ndy) {
                          const double *cs = s_ycsr + (__umul24(y, rowcap) + r + k) * 2;
                          const double v_value = 1 - (ux * cs[0] + uy * cs[1]);
                          csum[y] += A.wyaw * v_value * sdt;
                        }
                      }
                    }
                  }
                }
              }
            }
            if (k0 + kPairUB >= cntl) done = true;
          }
        }
        if (smp) {
          int rank = 0;
#pragma unroll
          for (int y = 0; y < kPairNY; y++) {
            if (y < ndy && ((ymask >> y) & 1u)) {
              s_pc[eb + rank] = csum[y];
              s_pi[eb + rank] = (unsigned short)(fb >= 0 ? 0x8000 | (fb + 1) : cntl);  // bit 15: blocked at that iteration
              rank++;
            }
          }
        }
      }
      wave_sync();

      // ---- phase D2: the list, GS dense lanes at a time: every row of the node's successor list written in whole lines
      for (int e0 = 0; e0 < E; e0 += GS) {
        const int e = e0 + vl;
        const bool act = e < E;
        const unsigned int pk = act ? (unsigned int)s_list[e] : 0u;  // (lexicographic table: the packed entry indices)
        const int j0 = pk & 15, j1 = (pk >> 4) & 15, j2 = (pk >> 8) & 15, jy = (int)((pk >> 12) & 15);
        int ci = (D == 3) ? (int)__umul24(__umul24(j0, nd[1]) + j1, nd[2]) + j2 : (int)__umul24(j0, nd[1]) + j1;
        ci = (int)__umul24(ci, ndy) + jy;
        const int en[3] = {j0, ndp + j1, 2 * ndp + j2};
        const int px = (D == 3) ? __umul24(j0, ndp) + j1 : j0;
        const int fl = pair_flags<D>(s_eflag, ndp, j0, j1, j2);
        const int n = (fl & 2) ? 0 : (fl >> 8);
        const bool mine = real && act && (n ? ((sub >> n) & 1ull) != 0ull : pass == 0);
        const int64_t idx = node * A.l_nstride + e;
        // line padding (expand_grid_kernel.hip): the lanes just past the end of the list complete its last 128-byte lines
        const bool pad16 = real && A.l_pad && !act && pass == 0 && e < ((E + 15) & ~15);  // 8-byte entries
        const bool pad32 = real && A.l_pad && !act && pass == 0 && e < ((E + 31) & ~31);  // 4-byte entries
        wave_prio(3);
        if ((mine || pad32) && !(A.dbg & 2)) {
          uint64_t h = s_hp[px];
          fold_entry<K>(h, s_eq, en[D - 1]);
          fold(h, s_yq[jy]);
          if (A.l_action) st_stream(mine ? ci : -1, &A.l_action[idx]);
          if (A.l_hash && (mine || pad16)) st_stream(h, &A.l_hash[idx]);
          if (A.l_state && (mine || pad16)) {
            double *o = A.l_state + idx;
            const int64_t ss = A.l_stride;
#pragma unroll
            for (int i = 0; i < D; i++) {
              const double *st = s_est + en[i] * (K - 1);
              const double u = s_uval[en[i]];
              const double uK = 0.0 + u;                                     // field of order K
              const double top = (0.0 + u * T) + s_node[(K - 1) * D + i];    // field of order K - 1
              st_stream((double)st[0], &o[(0 * D + i) * ss]);
              st_stream((double)((K >= 3) ? st[1] : top), &o[(1 * D + i) * ss]);
              st_stream((double)((K == 3) ? top : uK), &o[(2 * D + i) * ss]);
              st_stream((double)((K == 3) ? uK : 0.0), &o[(3 * D + i) * ss]);
            }
            st_stream(s_yawT[jy], &o[(4 * D) * ss]);
            st_stream(node_t + A.dt, &o[(4 * D + 1) * ss]);  // env_map.h:161
          }
          // what the search computes for the successor next (graph_search.h:84-88), while it is in registers
          if ((A.post.heur || A.post.flags) && (mine || pad16)) {
            double pp[D], vv[D], aa[D];
#pragma unroll
            for (int i = 0; i < D; i++) {
              const double *st = s_est + en[i] * (K - 1);
              const double u = s_uval[en[i]];
              const double uK = 0.0 + u;
              const double top = (0.0 + u * T) + s_node[(K - 1) * D + i];
              pp[i] = st[0];
              vv[i] = (K >= 3) ? st[1] : top;
              aa[i] = (K == 3) ? top : uK;
            }
            double hv;
            unsigned int fv;
            MPLX_POST_GOAL(pg, A.post, D)
            post_eval<D>(pg, h, pp, vv, aa, s_yawT[jy], &hv, &fv);
            if (A.post.heur) st_stream(hv, &A.post.heur[idx]);
            if (A.post.flags && mine) A.post.flags[idx] = (uint8_t)fv;
          }
        }
        // ---- cost (env_map.h:162-169) and iteration count
        if ((mine || pad32) && !(A.dbg & 4)) {
          const bool smp = mine && n != 0;
          const unsigned int pi = smp ? (unsigned int)s_pi[e] : 0u;
          const bool blocked = (pi & 0x8000u) != 0u;
          double J = 0;
#pragma unroll
          for (int i = 0; i < D; i++) {  // Primitive::J of a forward primitive: u*u*T per axis (see expand_kernel.hip)
            const double u = s_uval[en[i]];
            J += u * u * T;
          }
          const double csum = smp ? s_pc[e] : 0.0;
          const double cost = blocked ? INFINITY : csum + (J + A.w * A.dt);
          if (A.l_cost && (mine || pad16)) st_stream(cost, &A.l_cost[idx]);
          if (A.l_iters) st_stream((int)(pi & 0x7fffu), &A.l_iters[idx]);
        }
        wave_prio(1);
      }
    }
  }
  if (A.done.flag != nullptr) {
    // small synchronous batch: the host spins on a pinned word (DoneSignal, mplx_internal.h)
    __threadfence_system();
    if (lane == 0) {
      const unsigned int waves = gridDim.x * (unsigned int)kPairWPB;
      if (__hip_atomic_fetch_add(A.done.count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == waves - 1u) {
        __hip_atomic_store(A.done.count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(A.done.flag, A.done.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}
#undef A

template <int D_, int K_, int NY_>
struct PairInst {
  static constexpr int D = D_, K = K_, NY = NY_;
};

template <int D, int K, int NY>
hipError_t pair_inst_attr() {
  static bool attr_set[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 63;
  if (!attr_set[dev] || dev == 63) {
    hipError_t e = hipFuncSetAttribute((const void *)expand_pair_kernel<D, K, NY>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  return hipSuccess;
}

template <int D, int K, int NY>
hipError_t launch_pair_inst(const GridArgs &a, hipStream_t stream) {
  if (a.n_nodes == 0) return hipSuccess;
  const size_t lds = pair_lds_bytes(D, K, a.nU, a.ndp, a.n_max, a.rmax, a.wyaw > 0, a.ndy);
  if (hipError_t e = pair_inst_attr<D, K, NY>()) return e;
  hipLaunchKernelGGL((expand_pair_kernel<D, K, NY>), dim3((unsigned)a.grid_limit), dim3(kPairBT), lds, stream, a);
  return hipGetLastError();
}

template <int D, int K, int NY>
int pair_resident_inst(size_t lds) {
  if (pair_inst_attr<D, K, NY>() != hipSuccess) return 0;
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)expand_pair_kernel<D, K, NY>, kPairBT, lds) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return nb;
}

// (dim, control, yaw rates) -> instantiation: the sample loop carries NY = 2, 3 or 4 accumulators per lane
template <class R, class F>
R dispatch_pair(int dim, int control, int ndy, R none, F &&f) {
  const int ny = ndy <= 2 ? 2 : (ndy == 3 ? 3 : 4);
#define MPLX_PI(D, K) (ny == 2 ? f(PairInst<D, K, 2>{}) : (ny == 3 ? f(PairInst<D, K, 3>{}) : f(PairInst<D, K, 4>{})))
  if (dim == 2 && control == 0x13) return MPLX_PI(2, 2);
  if (dim == 2 && control == 0x17) return MPLX_PI(2, 3);
  if (dim == 3 && control == 0x13) return MPLX_PI(3, 2);
  if (dim == 3 && control == 0x17) return MPLX_PI(3, 3);
#undef MPLX_PI
  return none;
}

}  // namespace

size_t pair_lds_bytes(int dim, int order, int nU, int ndp, int n_max, int rmax, bool ycost, int ndy) {
  return (size_t)GridLds(dim, order, kPairWPB * kPairNG, nU, ndp, n_max, rmax, 64, pair_lds_mode(ycost), ndy, 1).total;
}
int pair_waves_per_block() { return kPairWPB; }
int pair_nodes_per_wave() { return kPairNG; }
int pair_max_yaw_rates() { return kPairNYMax; }

// the configurations this kernel has an instantiation for (the host checks the rest of the scope)
bool pair_covers(int dim, int control) { return (dim == 2 || dim == 3) && (control == 0x13 || control == 0x17); }

hipError_t launch_expand_pair(int dim, int control, const GridArgs &a, hipStream_t s) {
  if (a.ndy < 1 || a.ndy > kPairNYMax) return hipErrorInvalidValue;
  return dispatch_pair<hipError_t>(dim, control, a.ndy, hipErrorInvalidValue, [&](auto t) {
    using T = decltype(t);
    return launch_pair_inst<T::D, T::K, T::NY>(a, s);
  });
}

int pair_resident_blocks(int dim, int control, int ndy, size_t lds) {
  return dispatch_pair<int>(dim, control, ndy, 0, [&](auto t) {
    using T = decltype(t);
    return pair_resident_inst<T::D, T::K, T::NY>(lds);
  });
}

}  // namespace mplx
