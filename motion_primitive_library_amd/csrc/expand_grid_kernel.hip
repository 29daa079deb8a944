// expand_grid_kernel.hip -- factorised, list-producing successor expansion for
// gfx950 (MI355X): one WAVEFRONT owns one frontier node.
//
// Same function as expand_kernel.hip / expand_tile_kernel.hip
//   MPL::env_map<Dim>::get_succ, reference include/mpl_planner/env/env_map.h:147-172
//   with traverse_primitive :90-132,
// organised around one property of the reference's Primitive<Dim>: it is Dim
// INDEPENDENT Primitive1D polynomials (include/mpl_basis/primitive.h:220-256), so
// everything get_succ evaluates per (node, control) pair is a combination of
// per-axis quantities that depend only on (node, axis, u_axis):
//   max_vel/acc/jrk and the limit test   primitive.h:353-407, 483-496
//   the end state p(T), v(T), a(T), j(T)  primitive.h:128-145, 321-331
//   the lattice integers round(x / q)     waypoint.h:93-125
//   the effort term u*u*T                 primitive.h:92-122
//   the cell coordinate of a sample       map_util.h:103-108
// A control table of |U| = 9^3 entries has only 9 distinct values per axis, so
// the wave evaluates D*9 "axis entries" per node instead of 729*D, keeps them in
// LDS, and the per-pair / per-sample work becomes table look-ups and integer
// hashing.  The host detects the distinct values of any control table
// (mplx_set_controls); tables with more than 16 distinct values on an axis run
// expand_tile_kernel.hip instead.
//
// What the earlier versions taught (profiles/README.md): (1) with the f64
// arithmetic factorised away the workgroup-wide version was barrier / latency
// bound, so here a wave never waits for another wave: all phases are
// wave-synchronous (LDS is in-order per wave) and a workgroup is only a
// container for kWPB independent waves; (2) the wave-per-node version was VALU
// ISSUE bound (a wave64 VALU instruction of the kinds this kernel is made of occupies
// the SIMD for 4 cycles: profiles/r03_valu_issue_rates.txt) with ~60 instructions per map sample, most of them address
// arithmetic of the global look-up, so here the occupancy bits of the node's
// reachable box are staged into LDS once per node and a sample is 3 byte
// look-ups + 1 word look-up; (3) the list stores (2.7 GB per launch on C4) are
// full 128-byte lines, carry the `sc1 nt` policy (st_stream), and the waves'
// priority follows their progress through a node (wave_prio) so that store bursts
// are issued ahead of set-up work; nodes whose whole reach box is free
// (summed-area table) skip R and the sample loops altogether.  (4) What bounds
// the result (round 3, profiles/r03_c4_ablation.txt: same box, same allocation,
// variants alternating): VALU ISSUE, still -- 206 M wave instructions per C4
// launch = 0.336 ms of the 0.504; without any list store the kernel takes 0.378
// ms, without sampling 0.285, with neither 0.201, and the parts add up almost
// serially.  Instructions per node (3 189 on C4: 1 578 set-up + pair phase, 667
// rows + staging, 615 sample loops, 329 stores) are the figure of merit.
//
// Per node (one wave):
//   T1  axis entries (axis, value): limits, n_axis, end state, lattice integers,
//       u*u*T; prefix tables over the first D-1 axes (partial hash, flags)
//   A   64 pairs per step: valid = AND of flags, n = max n_axis, hash = prefix
//       hash folded with the last axis, emit = valid && hash != hash(node)
//       (env_map.h:158); the control indices of the emitted pairs are appended,
//       in order, to an LDS list; the set of sample counts n in use
//   R   per round of up to `rmax` sample counts: the rows of cell-offset codes of
//       every axis entry at the reference's accumulated sample times t_k
//       (`for (t = 0; t < T; t += T/n)`, env_map.h:97-99); the bounding box of
//       the codes of the entries that pass the limits; the blocked bits of that
//       box copied from the blocked-bit map into LDS (one row of x per (y, z))
//   D   the list, 64 dense lanes at a time: action / hash / Waypoint written to
//       the node's successor list; the sample loop of traverse_primitive with
//       lane = pair, eight samples per step; cost = J + w*dt or +inf
//       (env_map.h:162-169)
// The blocked-bit map (built on the device from the map and the search region
// whenever they change) has 1 bit per cell, x fastest like the map itself:
// blocked = occupied or outside the region (env_map.h:104-119).  A box that does
// not fit the LDS budget (possible only for far larger per-step displacements
// than the BASELINE configurations have) is sampled straight from that map.
//
// Bit-exactness: n = max(5, ceil(max_v*T/res)) with max_v = max over axes equals
// the max over axes of the per-axis counts because *, / by a positive constant
// and ceil are monotone; everything else is the same per-axis arithmetic as the
// other kernels, evaluated once instead of once per pair.
//
// Yaw controls (VELxYAW .. JRKxYAW): the yaw rate is one more factor of the
// control table.  Per node and yaw value: yaw(T), its lattice integer, cos / sin;
// per (x entry, y entry): the 16-bit set of yaw values whose heading constraint
// holds at both ends (primitive.h:504-525); per (yaw value, sample) cos / sin and
// per (x / y entry, sample) the velocity, so that the per-sample heading cost of
// traverse_primitive (env_map.h:121-129) is table look-ups, one sqrt and two
// divisions.  Potential maps with gradient_weight == 0 read the int8 values per
// sample (env_map.h:113-118); with gradient_weight != 0 the velocity rows of every
// axis are built next to the cell rows and a sample inside the field adds
// gradient_weight * |vel|.
//
// Scope: v_max > 0 (or VEL), Dim 2/3, K = 1..4, with or without yaw and potential
// maps, n_max <= 61, <= 16 distinct control values per axis, <= 1024 controls.
#include "mplx_grid_common.h"

namespace mplx {


#ifdef MPLX_PHASE_TIMING
// Diagnostic build only (python -m motion_primitive_library_amd.build --define MPLX_PHASE_TIMING --out ...): shader
// clock ticks spent between the phase markers of expand_grid_kernel, summed over all waves (profiles/micro/phase_times.py).
__device__ unsigned long long g_phase_ticks[16];
#define PT_DECL unsigned long long pt_last = __builtin_readcyclecounter(), pt_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define PT(i) do { const unsigned long long pt_now = __builtin_readcyclecounter(); pt_acc[i] += pt_now - pt_last; pt_last = pt_now; } while (0)
#define PT_FLUSH do { if (lane == 0) for (int pi = 0; pi < 10; pi++) atomicAdd(&g_phase_ticks[pi], pt_acc[pi]); } while (0)
#elif defined(MPLX_PHASE_MARK)
// static phase split: `hipcc -S -DMPLX_PHASE_MARK` leaves "; PTMARK i" comments where the markers are
// (profiles/micro/isa_phase_count.py counts the instructions between them)
#define PT_DECL
#define PT(i) asm volatile("; PTMARK " #i)
#define PT_FLUSH
#else
#define PT_DECL
#define PT(i)
#define PT_FLUSH
#endif

namespace {

using namespace dev;

// Resident waves per SIMD the register allocation has to allow (amdgpu_waves_per_eu).  The heading instantiations
// need 120 - 143 VGPRs when unconstrained: past 128 only 3 waves per SIMD fit, and the kernel is latency bound there.
#ifndef MPLX_OCC_YAW
#define MPLX_OCC_YAW 4
#endif
#ifndef MPLX_OCC_K3
#define MPLX_OCC_K3 1
#endif
#ifndef MPLX_OCC_PLAIN
#define MPLX_OCC_PLAIN 1
#endif
template <int K, bool YAW>
constexpr int grid_min_waves() { return YAW ? MPLX_OCC_YAW : (K >= 3 ? MPLX_OCC_K3 : MPLX_OCC_PLAIN); }

template <int D, int K, bool YAW, bool POT>
__global__ __launch_bounds__(kBT) __attribute__((amdgpu_waves_per_eu(grid_min_waves<K, YAW>())))
void expand_grid_kernel(const GridArgs A_kernarg) {
  extern __shared__ __align__(16) unsigned char smem[];
  // The argument block is ~100 SGPRs' worth and the kernel's SGPR budget is 102: preloaded, it alone forces hundreds
  // of v_writelane / v_readlane spills.  So it is read where it lies, through the kernarg segment pointer (scalar
  // loads from the constant cache), and the pointer is laundered at the top of every node so that the loads stay
  // inside the iteration, next to their uses, instead of being hoisted and kept live across the whole kernel.
  // (the pointer keeps the CONSTANT address space: only then are the loads scalar s_load's)
  typedef const GridArgs __attribute__((address_space(4))) *KernargPtr;
  KernargPtr Ak = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
  (void)A_kernarg;
#define A (*Ak)
  constexpr int F = 4 * D + 2;
  const int nU = A.nU, ndp = A.ndp, RM = A.rmax;
  const bool ycost = YAW && A.wyaw > 0;  // env_map.h:121: per-sample heading cost
  // gather mode: no box staging -- the sample loops read the blocked-bit map (L2) directly.  Staging pays when a
  // node's pairs make many more look-ups than its box has rows (|U| = 729: ~5 200 samples against ~600 rows); with a
  // small control table it is the other way round (|U| = 125: ~700 samples against up to 1 156 rows).
  const bool gather = A.gather != 0;
  const int ndy = YAW ? A.ndy : 0;
  const bool gcost = POT && A.grad_w != 0;  // env_map.h:116: gradient_weight * |vel| per sample inside the potential field
  const GridLds L(D, K, kWPB, nU, ndp, A.n_max, RM, A.boxcap, (YAW ? (ycost ? 2 : 1) : 0) | (gcost ? 4 : 0), ndy, A.ulex);
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const double *s_uval = (const double *)(smem + L.o_uval);
  const unsigned short *s_uidx = (const unsigned short *)(smem + L.o_uidx);
  const unsigned char *s_tc = smem + L.o_tc;
  unsigned char *wb = smem + L.o_wave0 + wv * L.wave_bytes;
  double *s_node = (double *)(wb + L.w_node);
  double *s_est = (double *)(wb + L.w_est);
  uint64_t *s_hp = (uint64_t *)(wb + L.w_hp);
  int *s_eq = (int *)(wb + L.w_eq);
  int *s_eflag = (int *)(wb + L.w_eflag);
  unsigned int *s_box = (unsigned int *)(wb + L.w_box);
  double *s_trow = (double *)(wb + L.w_box);  // [RM][tts] sample times, live only while the rows are built
  int *s_misc = (int *)(wb + L.w_misc);
  unsigned short *s_rowmap = (unsigned short *)(wb + L.w_rowmap);
  unsigned short *s_list = (unsigned short *)(wb + L.w_list);
  unsigned char *s_cell = wb + L.w_cell;
  const double *s_uyaw = (const double *)(smem + L.o_uyaw);
  double *s_yawT = (double *)(wb + L.w_yaw);
  double *s_ycs = (double *)(wb + L.w_ycs);
  int *s_yq = (int *)(wb + L.w_yq);
  unsigned short *s_hmask = (unsigned short *)(wb + L.w_hmask);
  double *s_uq = (double *)(wb + L.w_uq);
  double *s_vs = (double *)(wb + L.w_vs);
  double *s_ycsr = (double *)(wb + L.w_ycsr);

  const int tts = L.tts, EN = L.EN, PN = L.PN;
  // Rows are PACKED: the row of sample count n takes tc[n] (= n or n + 1) slots, and a pass takes as many of the
  // smallest pending counts as fit `rowcap` slots per entry -- with the fixed [RM][tts] layout of round 1 a pass held
  // RM counts whatever their length, and a JRK node (counts 5 .. 31) needed ~3 passes, each of which re-runs the row
  // builder, the box staging and a sparsely occupied sweep over the node's list.
  const int rowcap = RM * tts;
  constexpr int KQ = K == 3 ? 4 : K;
  const int half = A.n_max + 2;  // cell-offset code = offset from the node's cell + half
  const double T = A.dt;
  const double org[3] = {A.org0, A.org1, A.org2};
  const int dims[3] = {A.dim0, A.dim1, A.dim2};
  const int nd[3] = {A.nd0, A.nd1, A.nd2};

  // ---- which nodes are this wave's, and the first one's state on its way BEFORE the shared tables are fetched: the
  // two round trips to memory overlap (a launch's first node otherwise waits for the tables, then for its own state).
  // Static: wave w takes nodes w, w + W, ...  Dynamic (GridArgs::work): chunks of `ck` nodes; chunk w is wave w's, the
  // later ones are claimed from this workgroup's counter.  The chunk after the current one is always claimed already
  // (its first node is being prefetched), so the atomic's round trip is never waited for.
  const bool pinned = YAW && A.yaw.tab != nullptr;
  // (node indices and chunk bookkeeping in 32 bits: a frontier has fewer than 2^31 nodes -- checked by the host -- and
  // every 64-bit uniform costs two of the kernel's 102 SGPRs for its whole life)
  const int wave_id = (int)blockIdx.x * kWPB + wv;
  const int wave_stride = (int)gridDim.x * kWPB;
  // the override pass of the yaw pinning walks a list of nodes; everything else the whole frontier in order
  // (or, after a pre-screen launch, the list of nodes whose own heading passed validate_yaw at t = 0)
  const bool screened = YAW && A.live != nullptr;
  const int NN = screened ? (int)*A.live_n : (int)A.n_nodes;
  auto node_of = [&](int it) -> int64_t {
    return screened ? (int64_t)A.live[it] : ((YAW && A.yaw.node_list) ? (int64_t)A.yaw.node_list[it] : (int64_t)it);
  };
  const bool dyn = A.work != nullptr;
  const int ck = dyn ? A.work_chunk : 1;
  int dyn_beg = 0, dyn_len = 0, dyn_step = 1;  // this counter's share of the claimable chunks
  unsigned int *ctr = nullptr;
  if (dyn) {
    if (blockIdx.x == 0 && threadIdx.x < kWorkCounters) A.work_zero[threadIdx.x * 32] = 0u;  // for the next launch
    const int n_chunks = (NN + ck - 1) / ck, n_dyn = n_chunks > wave_stride ? n_chunks - wave_stride : 0;
    const int nc = gridDim.x < (unsigned)kWorkCounters ? (int)gridDim.x : kWorkCounters;  // counters in use
    const int cx = (int)(blockIdx.x % nc);
    const int base = n_dyn / nc, rem = n_dyn % nc;
    dyn_len = base + (cx < rem ? 1 : 0);
    // Which chunks are this counter's: dealt round-robin (chunk W + j * nc + cx is its j-th).  All counters advance
    // at about the same pace, so at any moment the whole launch works inside ONE window of the frontier and of every
    // output row, instead of 64 windows x 17 rows (work_blocked: a contiguous block per counter, the first version).
    if (A.work_blocked) { dyn_beg = wave_stride + cx * base + (cx < rem ? cx : rem); dyn_step = 1; }
    else { dyn_beg = wave_stride + cx; dyn_step = nc; }
    ctr = A.work + cx * 32;
  }
  auto claim = [&]() -> int {  // first node of the next chunk of this counter, or past the end
    unsigned int v = 0;
    if (lane == 0) v = atomicAdd(ctr, 1u);
    const int j = __builtin_amdgcn_readfirstlane((int)v);
    return (j >= 0 && j < dyn_len) ? (dyn_beg + j * dyn_step) * ck : NN;
  };
  const int it0 = wave_id * ck;
  int chunk_end = it0 + ck < NN ? it0 + ck : NN;
  double nxt = 0.0;  // lanes < F: one field of the next node (prefetched)
  if (it0 < NN && lane < F) nxt = A.nodes[(int64_t)lane * A.node_stride + node_of(it0)];
  int next_chunk = NN;  // (dynamic) first node of the chunk claimed ahead
  if (dyn && it0 < NN) next_chunk = claim();

  // ---- once per (persistent) workgroup: shared read-only tables
  {
    double *uv = (double *)(smem + L.o_uval);
    for (int i = threadIdx.x; i < EN; i += kBT) {
      const int ax = i / ndp, j = i - ax * ndp;
      uv[i] = A.uvals[ax * 16 + j];
    }
    unsigned short *ui = (unsigned short *)(smem + L.o_uidx);
    for (int i = threadIdx.x; i < (A.ulex ? 0 : nU); i += kBT) {
      const unsigned int pk = A.uidx[i];  // j0 | j1 << 8 | j2 << 16, each < 16
      ui[i] = (unsigned short)((pk & 15u) | (((pk >> 8) & 15u) << 4) | (((pk >> 16) & 15u) << 8) |
                               (((pk >> 24) & 15u) << 12));  // bits 12..15: the yaw value
    }
    if (YAW && threadIdx.x < ndy) ((double *)(smem + L.o_uyaw))[threadIdx.x] = A.uvals[3 * 16 + threadIdx.x];
    if (threadIdx.x < 64) smem[L.o_tc + threadIdx.x] = A.tcnt[threadIdx.x];
    if (MPLX_GRID_TT_RESIDENT) {
      double *tt = (double *)(smem + L.o_tt);
      const int ntt = (A.n_max + 1) * L.tts;
      for (int i = threadIdx.x; i < ntt; i += kBT) {
        const int nn = i / L.tts, k = i - nn * L.tts;
        tt[i] = A.ttab[nn * kTabStride + k];
      }
    }
  }
  __syncthreads();  // the only workgroup barrier
  asm volatile("" ::"v"(nxt));  // arrived before the loop: no wait for it at the loop head (see the pin after phase A)

  // primitive.h:521; in the override pass the host libm's value (see YawPin in mplx_internal.h)
  const double cos_lim = (YAW && A.yaw_max > 0) ? (pinned ? A.yaw.cos_lim : cos(A.yaw_max)) : 0.0;
  int it_next = 0;

  PT_DECL;
  for (int it = it0; it < NN; it = it_next) {
    PT(9);  // (loop overhead / tail of the previous node)
    asm volatile("" : "+s"(Ak));  // see the top of the kernel
    if (!dyn) {
      it_next = it + wave_stride;
    } else if (it + 1 < chunk_end) {
      it_next = it + 1;
    } else {  // last node of the chunk: move on to the chunk claimed ahead and claim the one after it
      it_next = next_chunk;
      chunk_end = it_next + ck < NN ? it_next + ck : NN;
      next_chunk = it_next < NN ? claim() : NN;
    }
    const int64_t node = node_of(it);
    const double *ytab = pinned ? A.yaw.tab + (int64_t)it * A.yaw.tab_stride : nullptr;  // [c0, s0, cT[16], sT[16]]
    bool yaw_amb = false;  // a heading-limit decision of this node is within rounding noise of the threshold
    // ---- phase 0: node state into LDS, prefetch of the next node
    wave_prio(0);
    wave_sync();
    if (lane < F) s_node[lane] = nxt;
    if (it_next < NN && lane < F)
      nxt = A.nodes[(int64_t)lane * A.node_stride + node_of(it_next)];
    wave_sync();

    if (YAW && K >= 2 && A.yaw_max > 0 && !screened) {
      // validate_yaw at t = 0 (primitive.h:509-523) does not depend on the control when the state carries a
      // velocity: evaluate(0).vel = 0.0 + v and yaw(0) = wrap(yaw).  A node that fails it has no successor at all.
      // (Large frontiers: grid_prescreen_kernel has done this test lane-per-node and only the survivors are here.)
      const double vx0 = 0.0 + s_node[1 * D], vy0 = 0.0 + s_node[1 * D + 1];
      bool dead = false;
      if (vx0 != 0 || vy0 != 0) {
        double c0, s0;
        const double y0 = wrap_angle((0.0 + 0.0) + s_node[4 * D]);
        if (pinned) { c0 = ytab[0]; s0 = ytab[1]; } else sincos(y0, &s0, &c0);
        const double sn = sqrt(vx0 * vx0 + vy0 * vy0);
        const double d = vx0 / sn * c0 + vy0 / sn * s0;
        dead = d < cos_lim;
        yaw_amb = near_limit(d, cos_lim, A.yaw.margin, vy0, y0, A.yaw.tie_yaw);
      }
      if (dead) {  // uniform
        if (lane == 0 && A.l_count) A.l_count[node] = 0;
        if (lane == 0 && yaw_amb && A.yaw.amb) flag_node(A.yaw.amb, A.yaw.amb_cap, node, A.yaw.any_host);
        continue;
      }
    }

    PT(0);
    // ---- phase T1: axis entries; the node's own lattice integers (lanes 48..)
    int flag = 0;
    int rb_lo = 0x7fffffff, rb_hi = (int)0x80000000;  // this lane's entry: cells its p(t) spans (free-box query)
    if (lane < EN) {
      const int ax = lane / ndp, jv = lane - ax * ndp;
      if (jv < nd[ax]) {
        const double p = s_node[0 * D + ax];
        const double v = (K >= 2) ? s_node[1 * D + ax] : 0.0;
        const double a = (K >= 3) ? s_node[2 * D + ax] : 0.0;
        const double j = (K >= 4) ? s_node[3 * D + ax] : 0.0;
        const double u = s_uval[lane];
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
        if (K >= 3) s_uq[lane] = q.top_quotient();
        if (K >= 2) s_est[lane * (K - 1) + 0] = np_;
        if (K >= 3) s_est[lane * (K - 1) + 1] = nv_;
        if (K >= 4) s_est[lane * (K - 1) + 2] = na_;
        s_eq[lane * KQ + 0] = quantise(np_, 0.01, A.R001);
        if (K >= 2) s_eq[lane * KQ + 1] = quantise(nv_, 0.1, A.R01);
        if (K >= 3) s_eq[lane * KQ + 2] = quantise(na_, 0.1, A.R01);
        if (K >= 4) s_eq[lane * KQ + 3] = quantise(nj_, 0.1, A.R01);
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
          s_misc[M_BASE + ax] = (qd - 0.5 > -0.5) ? (int)qd : -1;
        }
      }
      s_eflag[lane] = flag;
    } else if (lane >= 48 && lane < 48 + 4 * D) {
      // lattice integers of the node itself (waypoint.h:93-125), one field per lane
      const int i = (lane - 48) >> 2, f = (lane - 48) & 3;
      if (f < K) {
        const double x = s_node[f * D + i];
        s_misc[M_NODEQ + i * 4 + f] = f == 0 ? quantise(x, 0.01, A.R001) : quantise(x, 0.1, A.R01);
      }
    }
    if (YAW) {
      // per yaw value: yaw(T) = wrap(p(T)) of the yaw polynomial (primitive.h:329), its lattice integer
      // (waypoint.h:113-116) and, for the heading limit, its cos / sin
      const double cyaw = s_node[4 * D];
      if (lane < ndy) {
        const double yT = wrap_angle((0.0 + s_uyaw[lane] * T) + cyaw);
        s_yawT[lane] = yT;
        s_yq[lane] = quantise(yT, 0.1, A.R01);
        if (A.yaw_max > 0) {
          double sn_, cs_;
          if (pinned) { cs_ = ytab[2 + lane]; sn_ = ytab[2 + 16 + lane]; }
          else sincos(yT, &sn_, &cs_);  // same values as cos() / sin() (one argument reduction instead of two)
          s_ycs[lane * 2 + 0] = cs_;
          s_ycs[lane * 2 + 1] = sn_;
        }
      }
      if (lane == 63) s_misc[M_YQ] = quantise(cyaw, 0.1, A.R01);
    }
    {
      // per axis, the values that pass the limits, in order: the only entries whose samples are ever needed
      const unsigned long long vm = __ballot((flag & 1) != 0);
      if (lane < EN) {
        const int ax = lane / ndp, jv = lane - ax * ndp;
        const unsigned long long am = (((1ull << ndp) - 1ull) << (ax * ndp)) & vm;
        if (flag & 1) ((unsigned char *)(s_misc + M_VL))[ax * 16 + __popcll(am & ((1ull << lane) - 1ull))] = (unsigned char)jv;
        if (jv == 0) s_misc[M_NV + ax] = __popcll(am);
      }
    }
    wave_sync();
    PT(1);

    // ---- prefix tables over the first D-1 axes
    for (int x = lane; x < PN; x += 64) {
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
        fold_entry<K>(h, s_eq, e0);
        if (D == 3) fold_entry<K>(h, s_eq, e1);
        s_hp[x] = h;
      }
    }
    if (YAW) {
      // validate_yaw (primitive.h:504-525) at t = 0 and t = T for every (x entry, y entry): the set of yaw
      // values whose heading stays within yaw_max of the velocity direction
      const bool lim = A.yaw_max > 0;
      const double y0 = wrap_angle((0.0 + 0.0) + s_node[4 * D]);  // yaw polynomial at t = 0: (0.0 + u_yaw * 0.0) + yaw
      double c0 = 0.0, s0 = 0.0;
      if (lim) { if (pinned) { c0 = ytab[0]; s0 = ytab[1]; } else sincos(y0, &s0, &c0); }
      const float inv_n1 = 1.0f / (float)nd[1];
      for (int x = lane; x < nd[0] * nd[1]; x += 64) {
        const int j0 = (int)(((float)x + 0.5f) * inv_n1), j1 = x - j0 * nd[1];
        unsigned int mask = 0xffffu;
        if (lim) {
          Ax<K> qx, qy;
          qx.init(s_node[0], (K >= 2) ? s_node[1 * D] : 0.0, (K >= 3) ? s_node[2 * D] : 0.0, (K >= 4) ? s_node[3 * D] : 0.0, s_uval[j0]);
          qy.init(s_node[1], (K >= 2) ? s_node[1 * D + 1] : 0.0, (K >= 3) ? s_node[2 * D + 1] : 0.0, (K >= 4) ? s_node[3 * D + 1] : 0.0, s_uval[ndp + j1]);
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
              const double d = nx * s_ycs[jy * 2] + ny * s_ycs[jy * 2 + 1];
              if (d < cos_lim) mask &= ~(1u << jy);
              yaw_amb = yaw_amb || near_limit(d, cos_lim, A.yaw.margin, vyT, s_yawT[jy], A.yaw.tie_yaw);
            }
          }
        }
        s_hmask[j0 * ndp + j1] = (unsigned short)mask;
      }
      if (A.yaw.amb && __ballot(yaw_amb) != 0ull && lane == 0) flag_node(A.yaw.amb, A.yaw.amb_cap, node, A.yaw.any_host);
    }
    wave_sync();
    PT(2);

    uint64_t hcur = 0;  // hash of the node, folded by every lane alike (no divergence)
#pragma unroll
    for (int i = 0; i < D; i++) {
      const int4 q = *(const int4 *)(s_misc + M_NODEQ + i * 4);
      fold(hcur, q.x);
      if (K >= 2) fold(hcur, q.y);
      if (K >= 3) fold(hcur, q.z);
      if (K >= 4) fold(hcur, q.w);
    }
    if (YAW) fold(hcur, s_misc[M_YQ]);
    // ---- free-box shortcut: the summed-area table of the blocked-bit map answers "is the whole box the
    // node can reach in T free?" with 2^D look-ups.  If it is, every sample of every valid pair is free and
    // rows, box staging and the sample loops are skipped for this node.  The loads are issued here and
    // consumed after phase A.
    unsigned int sat_v = 0;
    bool sat_inside = false;
    if (A.sat != nullptr) {
      int rlo[3] = {0, 0, 0}, rhi[3] = {0, 0, 0};
      bool inside = true;
#pragma unroll
      for (int i = 0; i < D; i++) {
        // per axis the span of the entries inside the limits: a DPP min / max over the lanes of that axis (the lanes
        // hold their entries' spans since T1; the others carry the identities) -- no loop, no trip through LDS
        const bool mine_ax = lane >= i * ndp && lane < (i + 1) * ndp;
        const int lo_ = wave_reduce_minmax<false>(mine_ax ? rb_lo : 0x7fffffff);
        const int hi_ = wave_reduce_minmax<true>(mine_ax ? rb_hi : (int)0x80000000);
        rlo[i] = lo_;
        rhi[i] = hi_;
        inside = inside && hi_ >= lo_ && lo_ >= 0 && hi_ < dims[i];  // (hi < lo: no entry of the axis is inside the limits)
      }
      sat_inside = inside;
      if (inside && lane < (1 << D)) {
        // corner `lane` of the inclusion-exclusion sum over [rlo, rhi] (table has a zero border at index 0)
        const int cx = (lane & 1) ? rhi[0] + 1 : rlo[0];
        const int cy = (lane & 2) ? rhi[1] + 1 : rlo[1];
        const int cz = (D == 3) ? ((lane & 4) ? rhi[2] + 1 : rlo[2]) : 1;  // 2D: the one real plane sits above the border plane
        const int64_t idx = cx + (int64_t)(dims[0] + 1) * (cy + (int64_t)(dims[1] + 1) * cz);
        sat_v = A.sat[idx];
      }
    }
    const double node_t = s_node[4 * D + 1];
    int base_c[3];
#pragma unroll
    for (int i = 0; i < 3; i++) base_c[i] = (i < D) ? __builtin_amdgcn_readfirstlane(s_misc[M_BASE + i]) : 0;

    PT(3);
    // ---- phase A: every pair; ordered list of the emitted ones; sample counts in use.
    // When the control table enumerates its per-axis values in lexicographic order (A.ulex: the nested loops every
    // reference test builds U with, test/test_planner_2d.cpp:52-53), only the combinations of entries that pass
    // the limits are enumerated, in the same ascending control order: 43 % of C4's pairs instead of all of them.
    int E = 0;  // emitted successors of the node (uniform)
    unsigned int nm_lo = 0, nm_hi = 0;  // this lane's share of the set of sample counts in use (OR-reduced after the loop)
    const int ny_ = YAW ? ndy : 1;
    const int nv0 = __builtin_amdgcn_readfirstlane(s_misc[M_NV + 0]), nv1 = __builtin_amdgcn_readfirstlane(s_misc[M_NV + 1]);
    const int nv2 = (D == 3) ? __builtin_amdgcn_readfirstlane(s_misc[M_NV + 2]) : 1;
    const int in1 = nv2 * ny_, in0 = nv1 * in1;  // combinations per step of the second / first axis
    const int nA = A.ulex ? nv0 * in0 : nU;
    const float r_in0 = 1.0f / (float)(in0 > 0 ? in0 : 1), r_in1 = 1.0f / (float)(in1 > 0 ? in1 : 1), r_ny = 1.0f / (float)ny_;
    const unsigned char *vl_ = (const unsigned char *)(s_misc + M_VL);
    for (int base = 0; base < nA; base += 64) {
      const int x = base + lane;
      int ci = x;
      unsigned int lpk = 0;  // what the list holds: the control index, or (ulex) the packed entry indices it follows from
      bool emit = false;
      int n = 0;
      if (x < nA) {
        int j0, j1, j2 = 0, jy = 0;
        if (A.ulex) {
          const int a = (int)(((float)x + 0.5f) * r_in0);  // exact: x < 2^12
          const int ra = x - a * in0;
          const int b = (int)(((float)ra + 0.5f) * r_in1);
          int rb = ra - b * in1;
          if (YAW) {
            const int c = (int)(((float)rb + 0.5f) * r_ny);
            jy = rb - c * ny_;
            rb = c;
          }
          j0 = vl_[a];
          j1 = vl_[16 + b];
          if (D == 3) j2 = vl_[32 + rb];
          ci = (D == 3) ? (j0 * nd[1] + j1) * nd[2] + j2 : j0 * nd[1] + j1;
          if (YAW) ci = ci * ny_ + jy;
          lpk = (unsigned)j0 | ((unsigned)j1 << 4) | ((unsigned)j2 << 8) | ((unsigned)jy << 12);
        } else {
          const unsigned int pk = s_uidx[x];
          j0 = pk & 15, j1 = (pk >> 4) & 15, j2 = (pk >> 8) & 15;
          if (YAW) jy = (pk >> 12) & 15;
          lpk = (unsigned)ci;
        }
        const int px = (D == 3) ? __umul24(j0, ndp) + j1 : j0;
        const int eL = (D - 1) * ndp + ((D == 3) ? j2 : j1);
        uint64_t h = s_hp[px];
        fold_entry<K>(h, s_eq, eL);
        bool head = true;
        if (YAW) {
          fold(h, s_yq[jy]);
          head = (s_hmask[__umul24(j0, ndp) + j1] >> jy) & 1;
        }
        const int fl = pair_flags<D>(s_eflag, ndp, j0, j1, j2);
        n = (fl & 2) ? 0 : (fl >> 8);  // unchanged position: not traversed (env_map.h:163)
        emit = (fl & 1) && head && (h != hcur);  // env_map.h:158: `tn == curr` is a hash comparison
      }
      const unsigned long long m = __ballot(emit);
      if (emit) {
        s_list[E + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)lpk;
        // (an LDS atomicOr per emitting lane here -- the first version -- is processed lane by lane: five 64-lane
        // atomics per C4 node; the set is OR-ed in registers and reduced once per node instead)
        if (n) { if (n < 32) nm_lo |= 1u << n; else nm_hi |= 1u << (n - 32); }
      }
      E += __popcll(m);
    }
    wave_sync();
    if (lane == 0 && A.l_count) A.l_count[node] = E;
    unsigned long long nm = (unsigned long long)wave_reduce_or(nm_lo) | ((unsigned long long)wave_reduce_or(nm_hi) << 32);
    if (A.dbg & 1) nm = 0;  // timing ablation: no sampling
    // The next node's state (issued at the top of this node) has certainly arrived by now; pin that here,
    // where only loads are in flight, so that the wait does not end up at the top of the next node behind
    // this node's ~90 stores (vmcnt retires in order: waiting there means waiting for every store's ack).
    asm volatile("" ::"v"(nxt));
    bool safe = false;  // the node's whole reach box is free (uniform)
    if (A.sat != nullptr) {
      // inclusion-exclusion: + for corners with an even number of low coordinates
      unsigned int term = (lane < (1 << D)) ? sat_v : 0u;  // modulo 2^32 throughout: the box sum itself is small
      if ((D - __popc((unsigned)lane & ((1u << D) - 1u))) & 1) term = 0u - term;
#pragma unroll
      for (int d = 1; d < (1 << D); d <<= 1) term += (unsigned int)__shfl_xor((int)term, d, 64);
      safe = sat_inside && !ycost && __builtin_amdgcn_readfirstlane((int)term) == 0;
    }

    PT(4);
    wave_prio(1);
    // ---- rounds of up to RM sample counts
    for (int pass = 0; pass == 0 || nm != 0ull; pass++) {
      // this round's sample counts and their rows
      unsigned long long sub = 0;
      if (safe) {
        sub = ~0ull;  // nothing to sample: every successor is handled in this one pass
      } else {
        int used = 0;  // slots taken so far (uniform)
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
        // offset of every selected count's row: slots of the selected counts below it
        int off = 0;
        unsigned short mine_off = 0xffff;
        for (unsigned long long t = safe ? 0ull : sub; t; t &= t - 1ull) {
          const int nn = __ffsll((long long)t) - 1;
          if (nn == lane) mine_off = (unsigned short)off;
          off += (int)s_tc[nn];
        }
        if (lane <= A.n_max) s_rowmap[lane] = mine_off;  // (n_max <= 61)
      }
      if (!safe) {
        // the accumulated sample times of this pass' rows (independent loads: one round trip for all of them)
        int off = 0;
        for (unsigned long long t = sub; t; t &= t - 1ull) {
          const int nn = __ffsll((long long)t) - 1;
          const int cn = (int)s_tc[nn];
          if (!MPLX_GRID_TT_RESIDENT && lane < cn) s_trow[off + lane] = A.ttab[nn * kTabStride + lane];
          off += cn;
        }
      }
      wave_sync();
      PT(5);
      // rows: cell-offset codes of every axis entry at t_0 .. t_{cnt-1} of each sample count;
      // lanes = (value, k) of ONE axis at a time, so everything per axis is scalar
      int lo_l[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi_l[3] = {-1, -1, -1};  // per lane, reduced below
      bool ovf = false;
      {
        int row = 0;  // slot offset of the row being built
        for (unsigned long long t = safe ? 0ull : sub; t; t &= t - 1ull) {
          const int nn = __ffsll((long long)t) - 1;
          const int cn = (int)s_tc[nn];
          const float inv_cn = 1.0f / (float)cn;
          const double *trow = MPLX_GRID_TT_RESIDENT ? (const double *)(smem + L.o_tt) + nn * tts : s_trow + row;
          // The row of (entry, count nn) is only ever read by a pair whose count max(n_axis) IS nn, i.e. by entries
          // with n_entry <= nn: the small counts of a node need the rows of few entries (a JRK node uses 4 - 8 counts
          // between 5 and 31; building every entry's row for every count was ~40 % of C3's instructions).  Per count:
          // the entries to build, per axis, in value order (lanes = entries, as in T1).
          const unsigned long long fm = __ballot(lane < EN && (flag & 1) && (flag >> 8) <= nn);
          wave_sync();  // (the previous count's list has been read)
          if (lane < EN && ((fm >> lane) & 1ull)) {
            const int ax_ = lane / ndp;
            const unsigned long long am = (((1ull << ndp) - 1ull) << (ax_ * ndp)) & fm;
            ((unsigned char *)(s_misc + M_VLC))[ax_ * 16 + __popcll(am & ((1ull << lane) - 1ull))] = (unsigned char)(lane - ax_ * ndp);
          }
          wave_sync();
#pragma unroll
          for (int ax = 0; ax < D; ax++) {
            const double p0 = s_node[0 * D + ax];
            const double v0 = (K >= 2) ? s_node[1 * D + ax] : 0.0;
            const double a0 = (K >= 3) ? s_node[2 * D + ax] : 0.0;
            const double j0 = (K >= 4) ? s_node[3 * D + ax] : 0.0;
            const int shift = half - base_c[ax];
            const int nv = __popcll(fm & (((1ull << ndp) - 1ull) << (ax * ndp)));
            const unsigned char *vl = (const unsigned char *)(s_misc + M_VLC) + ax * 16;
            for (int x = lane; x < nv * cn; x += 64) {
              const int vi = (int)(((float)x + 0.5f) * inv_cn);  // exact: x < 2^12
              const int k = x - __umul24(vi, cn);
              const int aj = ax * ndp + (int)vl[vi];
              Ax<K> q;
              q.init(p0, v0, a0, j0, s_uval[aj]);
              // map_util.h:103-108: cell = round((pos - origin) / res - 0.5)
              const double qd = div_by(q.pos_q(trow[k], K >= 3 ? s_uq[aj] : 0.0) - org[ax], A.res, A.Rres);
              // qd - 0.5 > -0.5 <=> the rounded cell is >= 0; then qd > 0 and (qd - 0.5 being exact
              // for qd >= 0.5) round-half-away(qd - 0.5) == trunc(qd).  Negative cells: see M_BASE.
              const int c = (qd - 0.5 > -0.5) ? (int)qd : -1;
              // An entry within the limits moves at most n_max + 1 cells in T, so 0 < code < 2 * half --
              // as long as max_vel is the true maximum.  The reference's root loop can miss an extremum
              // of a SNP primitive (it stops at the first root >= T, primitive.h:158-159), so the range
              // is checked and a pass with an escaped code samples by direct evaluation instead.
              const int code = c + shift;
              ovf = ovf || code < 0 || code > 2 * half;
              s_cell[__umul24(aj, rowcap) + row + k] = (unsigned char)code;
              lo_l[ax] = code < lo_l[ax] ? code : lo_l[ax];
              hi_l[ax] = code > hi_l[ax] ? code : hi_l[ax];
              if ((YAW && ax < 2 && ycost) || gcost)  // Waypoint::vel of the sample (primitive.h:321-331): x and y for the heading cost, every axis for |vel|
                s_vs[__umul24(aj, rowcap) + row + k] = q.template vel<false>(trow[k]);
            }
          }
          if (YAW && ycost) {
            // heading of every yaw value at the sample times: wrap(p_yaw(t)), then cos / sin once per node
            const double cyaw = s_node[4 * D];
            for (int x = lane; x < ndy * cn; x += 64) {
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
      PT(6);
      // the box of codes the valid entries reach: DPP min / max over the wave (no LDS atomics: hipcc serialises
      // a divergent LDS atomic into a 64-trip scalar loop; no ds_bpermute butterfly: 36 dependent LDS round trips)
      int lo[3] = {0, 0, 0}, nb[3] = {1, 1, 1};
      bool have_box = !gather;
#pragma unroll
      for (int i = 0; i < D && !gather; i++) {
        lo[i] = wave_reduce_minmax<false>(lo_l[i]);
        const int hi = wave_reduce_minmax<true>(hi_l[i]);
        if (hi < lo[i]) have_box = false;
        nb[i] = hi - lo[i] + 1;
      }
      wave_sync();  // rows complete; the sample times (aliasing the box) are no longer needed
      const int WX = (nb[0] + 31) >> 5;
      const int n_rows = nb[1] * nb[2];
      // Only a SNP primitive's code can leave its range (the entries of K <= 3 have exact velocity maxima): the
      // direct-evaluation path is compiled into the K = 4 instantiations alone -- in the others it was dead code that
      // cost the yaw / potential instantiations 52 - 96 bytes of scratch per lane (C5 +7 %).
      constexpr bool kDirectPossible = K == 4;
      const bool direct = kDirectPossible && ((__ballot(ovf) != 0ull) || (A.dbg & 64));  // dbg 64: test hook, force direct evaluation
      const bool fits = !safe && !direct && !POT && have_box && n_rows * WX <= A.boxcap;
      if (fits && sub) {  // (never for a safe node)
        const float inv_ny = 1.0f / (float)nb[1];
        const int ax0 = base_c[0] + lo[0] - half;
#ifndef MPLX_GRID_SU
#define MPLX_GRID_SU 4
#endif
        constexpr int SU = MPLX_GRID_SU;  // rows per lane with their loads in flight together
        const int ayb = base_c[1] + lo[1] - half, azb = (D == 3) ? base_c[2] + lo[2] - half : 0;
        for (int w = 0; w < WX; w++) {
          const int xw = ax0 + 32 * w;
          const int vlo = xw < 0 ? -xw : 0;
          const int vhi = (dims[0] - xw) < 32 ? (dims[0] - xw) : 32;
          const unsigned int mask = (vhi >= 32 ? 0xffffffffu : ((1u << (vhi > 0 ? vhi : 0)) - 1u)) & ~((1u << vlo) - 1u);
          for (int r0 = lane; r0 < n_rows; r0 += 64 * SU) {
            unsigned int a0[SU], a1[SU], shf[SU];
            bool in[SU];
#pragma unroll
            for (int u = 0; u < SU; u++) {
              const int r = r0 + 64 * u;
              const int rz = (D == 3) ? (int)(((float)r + 0.5f) * inv_ny) : 0;  // exact: r < 2^14
              const int ry = r - rz * nb[1];
              const int ay = ayb + ry, az = azb + rz;
              in[u] = r < n_rows && vhi > vlo && ay >= 0 && ay < dims[1] && (D == 2 || (az >= 0 && az < dims[2]));
              const int64_t off = in[u] ? ((int64_t)az * dims[1] + ay) * (int64_t)dims[0] + xw : 0;
              const int64_t wi = off >> 5;
              shf[u] = (unsigned)(off & 31);
              const int64_t w0 = wi < 0 ? 0 : wi;
              const int64_t w1 = wi + 1 >= A.blk_words ? A.blk_words - 1 : wi + 1;
              a0[u] = (A.dbg & 8) ? 0u : A.blk[w0];  // dbg 8: timing ablation, no staging loads
              a1[u] = (A.dbg & 8) ? 0u : A.blk[w1 < 0 ? 0 : w1];
            }
#pragma unroll
            for (int u = 0; u < SU; u++) {
              const int r = r0 + 64 * u;
              const unsigned int val = in[u] ? (__builtin_amdgcn_alignbit(a1[u], a0[u], shf[u]) | ~mask) : 0xffffffffu;
              if (r < n_rows) s_box[r * WX + w] = val;
            }
          }
        }
      }
      wave_sync();
      PT(7);

      // ---- phase D: the list, 64 dense lanes at a time
      const int rowc = lo[1] + nb[1] * lo[2];
      for (int e0 = 0; e0 < E; e0 += 64) {
        const int e = e0 + lane;
        const bool act = e < E;
        const unsigned int le = act ? (unsigned int)s_list[e] : 0u;
        const unsigned int pk = A.ulex ? le : (unsigned int)s_uidx[le];
        const int j0 = pk & 15, j1 = (pk >> 4) & 15, j2 = (pk >> 8) & 15;
        int ci = (int)le;
        if (A.ulex) {
          ci = (D == 3) ? (int)__umul24(__umul24(j0, nd[1]) + j1, nd[2]) + j2 : (int)__umul24(j0, nd[1]) + j1;
          if (YAW) ci = (int)__umul24(ci, ndy) + (int)((pk >> 12) & 15);
        }
        const int en[3] = {j0, ndp + j1, 2 * ndp + j2};
        const int jy = YAW ? (int)((pk >> 12) & 15) : 0;
        const int px = (D == 3) ? __umul24(j0, ndp) + j1 : j0;
        const int fl = pair_flags<D>(s_eflag, ndp, j0, j1, j2);
        const int n = (fl & 2) ? 0 : (fl >> 8);
        const bool mine = act && (n ? ((sub >> n) & 1ull) != 0ull : pass == 0);
        const int64_t idx = node * A.l_nstride + e;
        // Line padding (when the node stride is a multiple of 32; MPLX_NO_LINE_PAD turns it off): a list that
        // ends inside a 128-byte line leaves a partial-line write in every output row, and those cost far
        // more than their bytes (a store-only kernel writing C4's lists: 3.5-4 TB/s with them, 6+ TB/s when
        // every list ends on a line, profiles/micro/write_pattern.hip).  The lanes just past the end of the
        // list therefore store too (unspecified values, inside the node's own region), completing the lines:
        // +2.7 % bytes, -9 % kernel time on C4 once the kernel is store bound (it made no difference while
        // the kernel was still compute bound).
        const bool pad16 = A.l_pad && !act && pass == 0 && e < ((E + 15) & ~15);  // 8-byte entries
        const bool pad32 = A.l_pad && !act && pass == 0 && e < ((E + 31) & ~31);  // 4-byte entries
        wave_prio(3);
        if ((mine || pad32) && !(A.dbg & 2)) {
          uint64_t h = s_hp[px];
          fold_entry<K>(h, s_eq, en[D - 1]);
          if (YAW) fold(h, s_yq[jy]);
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
              st_stream((double)((K >= 2) ? st[0] : top), &o[(0 * D + i) * ss]);
              st_stream((double)((K >= 3) ? st[1] : (K == 2 ? top : uK)), &o[(1 * D + i) * ss]);
              st_stream((double)((K >= 4) ? st[2] : (K == 3 ? top : (K == 2 ? uK : 0.0))), &o[(2 * D + i) * ss]);
              st_stream((double)((K == 4) ? top : (K == 3 ? uK : 0.0)), &o[(3 * D + i) * ss]);
            }
            // Waypoint::yaw: 0 for a control without yaw (primitive.h:322)
            st_stream(YAW ? s_yawT[jy] : 0.0, &o[(4 * D) * ss]);
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
              pp[i] = (K >= 2) ? st[0] : top;
              vv[i] = (K >= 3) ? st[1] : (K == 2 ? top : uK);
              aa[i] = (K >= 4) ? st[2] : (K == 3 ? top : (K == 2 ? uK : 0.0));
            }
            double hv;
            unsigned int fv;
            MPLX_POST_GOAL(pg, A.post, D)
            post_eval<D>(pg, h, pp, vv, aa, YAW ? s_yawT[jy] : 0.0, &hv, &fv);
            if (A.post.heur) st_stream(hv, &A.post.heur[idx]);
            if (A.post.flags && mine) A.post.flags[idx] = (uint8_t)fv;
          }
        }
        // ---- the sample loop of traverse_primitive (env_map.h:97-120)
        const bool smp = mine && n != 0;
        const int cntl = smp ? (int)s_tc[n] : 0;  // iterations of `for (t = 0; t < T; t += T/n)`
        int fb = -1;                              // first blocked sample
        double csum = 0.0;                        // traverse_primitive's accumulated cost (potential maps)
        {
          const int r = smp ? (int)s_rowmap[n] : 0;  // slot offset of the pair's row
          int ptr[3] = {0, 0, 0};
#pragma unroll
          for (int i = 0; i < D; i++) ptr[i] = __umul24(en[i], rowcap) + r;
          bool done = !smp || safe || (A.dbg & 32);  // a node whose whole reach box is free has nothing to look up (dbg 32: timing ablation, rows built but no sample loop)
          const double sdt = (smp && (POT || ycost)) ? T / n : 0.0;  // env_map.h:96
          // env_map.h:121-129: heading cost of sample k (after the potential term of the same sample)
          const int pyr = YAW ? __umul24(jy, rowcap) + r : 0;
          auto heading_cost = [&](int k) {
            const double vx = s_vs[ptr[0] + k], vy = s_vs[ptr[1] + k];
            double ux, uy;
            if (heading_unit(vx, vy, ux, uy)) {
              const double v_value = 1 - (ux * s_ycsr[(pyr + k) * 2] + uy * s_ycsr[(pyr + k) * 2 + 1]);
              csum += A.wyaw * v_value * sdt;
            }
          };
          if (POT && !direct) {
            // potential map (env_map.h:113-118): the values are needed, not
            // just a bit, so the samples read the int8 cells from HBM / L2 (8 in flight per lane) and the cost is
            // accumulated in the reference's order
            for (int k0 = 0; __ballot(!done) != 0ull; k0 += kUB) {
              int val[kUB];
              bool bad[kUB];
#pragma unroll
              for (int q = 0; q < kUB; q++) {
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
              for (int q = 0; q < kUB; q++) {
                if (!done && k0 + q < cntl) {
                  if (bad[q] || val[q] >= 100) { fb = k0 + q; done = true; }
                  else {
                    if (val[q] > 0) {
                      if (gcost) {  // env_map.h:115-116: dt * (potential_weight * value + gradient_weight * vel.norm())
                        double vv = 0;
#pragma unroll
                        for (int i = 0; i < D; i++) {
                          const double vi_ = s_vs[ptr[i] + k0 + q];
                          vv += vi_ * vi_;
                        }
                        csum += sdt * (A.pot_w * val[q] + A.grad_w * sqrt(vv));
                      } else {
                        csum += sdt * (A.pot_w * val[q]);
                      }
                    }
                    if (YAW && ycost) heading_cost(k0 + q);
                  }
                }
              }
              if (k0 + kUB >= cntl) done = true;
            }
          } else if (fits) {
            for (int k0 = 0; __ballot(!done) != 0ull; k0 += kUB) {
              unsigned int m = 0;
              if (WX == 1) {  // the box is at most 32 cells wide: one word per (y, z) row
#pragma unroll
                for (int q = 0; q < kUB; q++) {
                  const int k = k0 + q;
                  const int ex = s_cell[ptr[0] + k];
                  const int ey = s_cell[ptr[1] + k];
                  const int ez = (D == 3) ? (int)s_cell[ptr[2] + k] : 0;
                  const unsigned int word = s_box[(D == 3 ? __umul24(nb[1], ez) : 0) + ey - rowc];
                  m |= ((word >> (ex - lo[0])) & 1u) << q;
                }
              } else {
#pragma unroll
                for (int q = 0; q < kUB; q++) {
                  const int k = k0 + q;
                  const int ex = s_cell[ptr[0] + k];
                  const int ey = s_cell[ptr[1] + k];
                  const int ez = (D == 3) ? (int)s_cell[ptr[2] + k] : 0;
                  const int dx = ex - lo[0];
                  const unsigned int word = s_box[__umul24((D == 3 ? __umul24(nb[1], ez) : 0) + ey - rowc, WX) + (dx >> 5)];
                  m |= ((word >> (dx & 31)) & 1u) << q;
                }
              }
              const int left = cntl - k0;
              if (left < kUB) m &= (1u << (left > 0 ? left : 0)) - 1u;
              if (!done && m) { fb = k0 + __ffs((int)m) - 1; done = true; }
              if (left <= kUB) done = true;
            }
          } else if (direct) {
            // a cell code left its range (only a SNP primitive can do that: the reference's root loop stops at the first
            // root >= T, primitive.h:158-159): every sample of the pair by direct evaluation, with everything the row
            // paths do -- occupancy bit or potential value + search region, the potential / |vel| cost of
            // env_map.h:113-118 and the heading cost of :121-129 -- in the reference's order, on the same expressions
            // as the rows (Ax::pos / vel<false> at the table's accumulated times; wrap(u_yaw * t + yaw))
            double p0[D], v0[D], a0[D], j0d[D], uu[D];
#pragma unroll
            for (int i = 0; i < D; i++) {
              p0[i] = s_node[0 * D + i];
              v0[i] = (K >= 2) ? s_node[1 * D + i] : 0.0;
              a0[i] = (K >= 3) ? s_node[2 * D + i] : 0.0;
              j0d[i] = (K >= 4) ? s_node[3 * D + i] : 0.0;
              uu[i] = s_uval[en[i]];
            }
            const double cyaw_d = YAW ? s_node[4 * D] : 0.0, uyaw_d = YAW ? s_uyaw[jy] : 0.0;
            for (int k = 0; __ballot(!done) != 0ull; k++) {
              if (!done) {
                const double t_k = A.ttab[n * kTabStride + k];
                bool inside = true;
                int64_t cell = 0, mul = 1;
                double vk[D];
#pragma unroll
                for (int i = 0; i < D; i++) {
                  Ax<K> q;
                  q.init(p0[i], v0[i], a0[i], j0d[i], uu[i]);
                  const double qd = div_by(q.template pos<false>(t_k) - org[i], A.res, A.Rres);
                  const int c = (qd - 0.5 > -0.5) ? (int)qd : -1;
                  inside = inside && c >= 0 && c < dims[i];
                  cell += mul * c;
                  mul *= dims[i];
                  vk[i] = (POT || (YAW && ycost)) ? q.template vel<false>(t_k) : 0.0;
                }
                bool blocked;
                if (POT) {
                  const int64_t ci_ = inside ? cell : 0;
                  const bool in_reg = A.region == nullptr || ((A.region[ci_ >> 5] >> (ci_ & 31)) & 1u);
                  const int pv = A.pot[ci_];
                  blocked = !inside || !in_reg || pv >= 100;
                  if (!blocked) {
                    if (pv > 0) {
                      if (gcost) {
                        double vv = 0;
#pragma unroll
                        for (int i = 0; i < D; i++) vv += vk[i] * vk[i];
                        csum += sdt * (A.pot_w * pv + A.grad_w * sqrt(vv));
                      } else {
                        csum += sdt * (A.pot_w * pv);
                      }
                    }
                    if (YAW && ycost) {
                      double ux, uy;
                      if (heading_unit(vk[0], vk[1], ux, uy)) {
                        double sn_, cs_;
                        sincos(wrap_angle(uyaw_d * t_k + cyaw_d), &sn_, &cs_);
                        const double v_value = 1 - (ux * cs_ + uy * sn_);
                        csum += A.wyaw * v_value * sdt;
                      }
                    }
                  }
                } else {
                  blocked = !inside || ((A.blk[inside ? (cell >> 5) : 0] >> (cell & 31)) & 1u);
                }
                if (blocked) { fb = k; done = true; }
                if (k + 1 >= cntl) done = true;
              }
            }
          } else {
            // straight from the blocked-bit map (gather mode, or a box too large for LDS): kUB samples per step,
            // their look-ups in flight together -- one round trip to L2 per kUB samples
            for (int k0 = 0; __ballot(!done) != 0ull; k0 += kUB) {
              unsigned int wd[kUB];
              int sh[kUB];
#pragma unroll
              for (int q = 0; q < kUB; q++) {
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
                sh[q] = inside ? (int)(cell & 31) : -1;  // -1: outside the map = blocked (env_map.h:104)
                wd[q] = A.blk[inside ? (cell >> 5) : 0];
              }
#pragma unroll
              for (int q = 0; q < kUB; q++) {
                if (!done && k0 + q < cntl && (sh[q] < 0 || ((wd[q] >> sh[q]) & 1u))) { fb = k0 + q; done = true; }
              }
              if (k0 + kUB >= cntl) done = true;
            }
          }
        }
        if (YAW && ycost && !POT) {
          // occupancy map: the heading cost only matters for a primitive that is not blocked
          const bool go = smp && fb < 0;
          int ptr[3] = {0, 0, 0};
          const int r = smp ? (int)s_rowmap[n] : 0;
#pragma unroll
          for (int i = 0; i < 2; i++) ptr[i] = __umul24(en[i], rowcap) + r;
          const int pyr = __umul24(jy, rowcap) + r;
          const double sdt = go ? T / n : 0.0;
          Ax<K> qx_d, qy_d;  // direct evaluation (a pass whose cell codes left their range): no velocity / trig rows
          if (direct) {
            qx_d.init(s_node[0], (K >= 2) ? s_node[1 * D] : 0.0, (K >= 3) ? s_node[2 * D] : 0.0, (K >= 4) ? s_node[3 * D] : 0.0, s_uval[en[0]]);
            qy_d.init(s_node[1], (K >= 2) ? s_node[1 * D + 1] : 0.0, (K >= 3) ? s_node[2 * D + 1] : 0.0, (K >= 4) ? s_node[3 * D + 1] : 0.0, s_uval[en[1]]);
          }
          for (int k = 0; __ballot(go && k < cntl) != 0ull; k++) {
            if (go && k < cntl && direct) {
              const double t_k = A.ttab[n * kTabStride + k];
              const double vx = qx_d.template vel<false>(t_k), vy = qy_d.template vel<false>(t_k);
              double ux, uy;
              if (heading_unit(vx, vy, ux, uy)) {
                double sn_, cs_;
                sincos(wrap_angle(s_uyaw[jy] * t_k + s_node[4 * D]), &sn_, &cs_);
                const double v_value = 1 - (ux * cs_ + uy * sn_);
                csum += A.wyaw * v_value * sdt;
              }
            } else if (go && k < cntl) {
              const double vx = s_vs[ptr[0] + k], vy = s_vs[ptr[1] + k];
              double ux, uy;
              if (heading_unit(vx, vy, ux, uy)) {
                const double v_value = 1 - (ux * s_ycsr[(pyr + k) * 2] + uy * s_ycsr[(pyr + k) * 2 + 1]);
                csum += A.wyaw * v_value * sdt;
              }
            }
          }
        }
        // ---- cost (env_map.h:162-169) and iteration count
        if ((mine || pad32) && !(A.dbg & 4)) {
          const bool blocked = fb >= 0;
          double J = 0;
#pragma unroll
          for (int i = 0; i < D; i++) {  // Primitive::J of a forward primitive: u*u*T per axis (see expand_kernel.hip)
            const double u = s_uval[en[i]];
            J += u * u * T;
          }
          const double cost = blocked ? INFINITY : csum + (J + A.w * A.dt);
          if (A.l_cost && (mine || pad16)) st_stream(cost, &A.l_cost[idx]);
          if (A.l_iters) st_stream(blocked ? fb + 1 : cntl, &A.l_iters[idx]);
        }
        wave_prio(1);
      }
      PT(8);
    }
  }
  PT_FLUSH;
  if (A.done.flag != nullptr) {
    // small synchronous batch: the host spins on a pinned word instead of synchronising the stream.  Every wave's
    // stores are out (and written back) before it counts itself; the last one resets the counter and signals.
    __threadfence_system();
    if (lane == 0) {
      const unsigned int waves = gridDim.x * (unsigned int)kWPB;
      if (__hip_atomic_fetch_add(A.done.count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == waves - 1u) {
        __hip_atomic_store(A.done.count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(A.done.flag, A.done.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
#undef A
}

// Pre-screen of yaw controls, lane per node.  validate_yaw at t = 0 (primitive.h:509-523) depends on the node alone
// when the state carries a velocity (evaluate(0).vel = 0.0 + v, yaw(0) = wrap(yaw)): a node whose own heading is off
// its velocity direction by more than yaw_max has no successor at all.  On a synthetic frontier with random headings
// that is most nodes (C5: 84 %), and in the main kernel each of them costs a WAVE a sincos, a square root, two
// divisions and a trip round the node loop; here 64 nodes share those instructions.  The survivors are appended to
// `live` in frontier order within a workgroup (ballot + prefix; workgroups append in arrival order -- the output of a
// node does not depend on where it sits in the list), the dead get their empty list here.  Same expressions as the
// main kernel's own test (and -ffp-contract=off), so the decision is the same bit for bit; decisions within rounding
// noise of the threshold are flagged for the host-libm pass exactly as there.
template <int D, int K>
__global__ __launch_bounds__(256) void grid_prescreen_kernel(const double *nodes, int64_t n_nodes, int64_t node_stride,
                                                             double yaw_max, YawPin yaw, int32_t *l_count, int32_t *live,
                                                             uint32_t *live_n, uint32_t *live_zero) {
  __shared__ uint32_t s_wave[4], s_base;
  if (blockIdx.x == 0 && threadIdx.x == 0) *live_zero = 0u;  // the counter of the next pre-screen launch of the stream
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  bool alive = false;
  if (i < n_nodes) {
    alive = true;
    const double vx0 = 0.0 + nodes[(int64_t)(1 * D) * node_stride + i], vy0 = 0.0 + nodes[(int64_t)(1 * D + 1) * node_stride + i];
    if (vx0 != 0 || vy0 != 0) {
      const double cos_lim = cos(yaw_max);
      double c0, s0;
      const double y0 = wrap_angle((0.0 + 0.0) + nodes[(int64_t)(4 * D) * node_stride + i]);
      sincos(y0, &s0, &c0);
      const double sn = sqrt(vx0 * vx0 + vy0 * vy0);
      const double d = vx0 / sn * c0 + vy0 / sn * s0;
      if (d < cos_lim) {
        alive = false;
        if (l_count) l_count[i] = 0;
        if (yaw.amb && near_limit(d, cos_lim, yaw.margin, vy0, y0, yaw.tie_yaw)) flag_node(yaw.amb, yaw.amb_cap, i, yaw.any_host);
      }
    }
  }
  const unsigned long long m = __ballot(alive);
  if (lane == 0) s_wave[wv] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t t = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    s_base = t ? atomicAdd(live_n, t) : 0u;
  }
  __syncthreads();
  if (alive) {
    uint32_t pos = s_base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    for (int w = 0; w < wv; w++) pos += s_wave[w];
    live[pos] = (int32_t)i;
  }
}

// Summed-area table of the blocked-bit map: sat[z][y][x] (sizes d+1, zero border at index 0) = number of
// blocked cells with coordinates < (x, y, z).  Built once per map / region change.
__global__ void sat_seed_kernel(const uint32_t *blk, int d0, int d1, int d2, uint32_t *sat) {
  const int64_t n = (int64_t)d0 * d1 * d2;
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const int x = (int)(g % d0);
  const int64_t r = g / d0;
  const int y = (int)(r % d1), z = (int)(r / d1);
  sat[((int64_t)(z + 1) * (d1 + 1) + (y + 1)) * (d0 + 1) + (x + 1)] = (blk[g >> 5] >> (g & 31)) & 1u;
}
// x: one wave per (y, z) row, 64 cells per step, wave-level inclusive scan
__global__ __launch_bounds__(256) void sat_scan_x_kernel(int d0, int64_t n_rows_with_border, uint32_t *sat) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows_with_border) return;
  uint32_t *r = sat + row * (d0 + 1) + 1;
  uint32_t carry = 0;
  for (int x0 = 0; x0 < d0; x0 += 64) {
    const int x = x0 + lane;
    uint32_t v = x < d0 ? r[x] : 0u;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = (uint32_t)__shfl_up((int)v, d, 64);
      if (lane >= d) v += o;
    }
    v += carry;
    if (x < d0) r[x] = v;
    carry = (uint32_t)__shfl((int)v, 63, 64);
  }
}
// y (and z): one thread per column, consecutive threads on consecutive x
__global__ void sat_scan_y_kernel(int d0, int d1, int d2p, uint32_t *sat) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (int64_t)(d0 + 1) * d2p) return;
  const int x = (int)(g % (d0 + 1));
  const int64_t z = g / (d0 + 1);
  uint32_t *c = sat + z * (int64_t)(d1 + 1) * (d0 + 1) + x;
  uint32_t acc = 0;
  for (int y = 1; y <= d1; y++) {
    acc += c[(int64_t)y * (d0 + 1)];
    c[(int64_t)y * (d0 + 1)] = acc;
  }
}
__global__ void sat_scan_z_kernel(int d0, int d1, int d2, uint32_t *sat) {
  const int64_t plane = (int64_t)(d0 + 1) * (d1 + 1);
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= plane) return;
  uint32_t *c = sat + g;
  uint32_t acc = 0;
  for (int z = 1; z <= d2; z++) {
    acc += c[(int64_t)z * plane];
    c[(int64_t)z * plane] = acc;
  }
}

// Blocked-bit map: 1 bit per cell in the map's own order (x fastest), 1 = occupied
// (map == 100) or outside the search region.
__global__ void build_blocked_bits_kernel(const int8_t *map, const uint32_t *region, int64_t n_cells, int64_t n_words,
                                          int potential, uint32_t *out) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_words) return;
  uint32_t bits = 0;
  const uint32_t reg = region ? region[g] : 0xffffffffu;
  for (int b = 0; b < 32; b++) {
    const int64_t idx = g * 32 + b;
    // occupancy: blocked cells; potential map: every cell that blocks OR costs (value > 0, env_map.h:113-118)
    const bool blocked = idx >= n_cells || (potential ? map[idx] > 0 : map[idx] == 100) || !((reg >> b) & 1u);
    bits |= (blocked ? 1u : 0u) << b;
  }
  out[g] = bits;
}

// the dynamic-LDS ceiling is a per-device attribute of the kernel: set it once per (instantiation, device), not
// once per process -- a second context on another GPU of the same process needs it too
template <int D, int K, bool YAW, bool POT>
hipError_t grid_inst_attr() {
  static bool attr_set[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 63;
  if (!attr_set[dev] || dev == 63) {
    hipError_t e = hipFuncSetAttribute((const void *)expand_grid_kernel<D, K, YAW, POT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  return hipSuccess;
}

template <int D, int K, bool YAW, bool POT>
hipError_t launch_grid_inst(const GridArgs &a, hipStream_t stream) {
  if (a.n_nodes == 0) return hipSuccess;
  const int64_t n_wg = (a.n_nodes + kWPB - 1) / kWPB;
  const int64_t blocks = n_wg < (int64_t)a.grid_limit ? n_wg : (int64_t)a.grid_limit;
  const size_t lds = grid_lds_bytes(D, K, a.nU, a.ndp, a.n_max, a.rmax, a.boxcap,
                                    (YAW ? (a.wyaw > 0 ? 2 : 1) : 0) | ((POT && a.grad_w != 0) ? 4 : 0), a.ndy, a.ulex);
  if (hipError_t e = grid_inst_attr<D, K, YAW, POT>()) return e;
  hipLaunchKernelGGL((expand_grid_kernel<D, K, YAW, POT>), dim3((unsigned)blocks), dim3(kBT), lds, stream, a);
  return hipGetLastError();
}

// Workgroups of this instantiation one CU keeps resident with `lds` bytes of dynamic LDS each (registers, LDS
// granules and wave slots as the runtime accounts them); 0 if the runtime cannot tell.
template <int D, int K, bool YAW, bool POT>
int resident_inst(size_t lds) {
  if (grid_inst_attr<D, K, YAW, POT>() != hipSuccess) return 0;
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)expand_grid_kernel<D, K, YAW, POT>, kBT, lds) !=
      hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return nb;
}

template <int D_, int K_, bool YAW_, bool POT_>
struct GridInst {
  static constexpr int D = D_, K = K_;
  static constexpr bool YAW = YAW_, POT = POT_;
};

// (dim, control, potential map?) -> instantiation; f(GridInst<...>{}) or `none` when the kernel has none
template <class R, class F>
R dispatch_grid(int dim, int control, bool pot, R none, F &&f) {
#define MPLX_GI(D, K, Y) (pot ? f(GridInst<D, K, Y, true>{}) : f(GridInst<D, K, Y, false>{}))
  if (dim == 2) {
    switch (control) {
      case 0x01: return MPLX_GI(2, 1, false);
      case 0x03: return MPLX_GI(2, 2, false);
      case 0x07: return MPLX_GI(2, 3, false);
      case 0x0f: return MPLX_GI(2, 4, false);
      case 0x11: return MPLX_GI(2, 1, true);
      case 0x13: return MPLX_GI(2, 2, true);
      case 0x17: return MPLX_GI(2, 3, true);
      case 0x1f: return MPLX_GI(2, 4, true);
    }
  } else if (dim == 3) {
    switch (control) {
      case 0x01: return MPLX_GI(3, 1, false);
      case 0x03: return MPLX_GI(3, 2, false);
      case 0x07: return MPLX_GI(3, 3, false);
      case 0x0f: return MPLX_GI(3, 4, false);
      case 0x11: return MPLX_GI(3, 1, true);
      case 0x13: return MPLX_GI(3, 2, true);
      case 0x17: return MPLX_GI(3, 3, true);
      case 0x1f: return MPLX_GI(3, 4, true);
    }
  }
#undef MPLX_GI
  return none;
}

}  // namespace

#ifdef MPLX_PHASE_TIMING
extern "C" int mplx_debug_phase_ticks(unsigned long long *out16, int reset) {
  if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_phase_ticks), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase_ticks), z, sizeof z) != hipSuccess) return -1;
  }
  return 0;
}
#endif

size_t grid_lds_bytes(int dim, int order, int nU, int ndp, int n_max, int rmax, int boxcap, int yaw_mode, int ndy,
                      int ulex) {
  return (size_t)GridLds(dim, order, kWPB, nU, ndp, n_max, rmax, boxcap, yaw_mode, ndy, ulex).total;
}
int grid_waves_per_block() { return kWPB; }

hipError_t launch_build_blocked_bits(const int8_t *map, const uint32_t *region, int64_t n_cells, int potential,
                                     uint32_t *out, hipStream_t stream) {
  const int64_t n_words = (n_cells + 31) >> 5;
  const unsigned blocks = (unsigned)((n_words + 255) / 256);
  hipLaunchKernelGGL(build_blocked_bits_kernel, dim3(blocks), dim3(256), 0, stream, map, region, n_cells, n_words, potential, out);
  return hipGetLastError();
}

hipError_t launch_build_sat(int dim, const uint32_t *blk, const int32_t *mdim, uint32_t *sat, hipStream_t stream) {
  const int d0 = mdim[0], d1 = mdim[1], d2 = dim == 3 ? mdim[2] : 1;
  const int d2p = dim == 3 ? d2 + 1 : 2;  // planes incl. the zero border plane (2D: border + the one real plane)
  const int64_t total = (int64_t)(d0 + 1) * (d1 + 1) * d2p;
  hipError_t e = hipMemsetAsync(sat, 0, (size_t)total * 4, stream);
  if (e != hipSuccess) return e;
  const int64_t n = (int64_t)d0 * d1 * d2;
  hipLaunchKernelGGL(sat_seed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, blk, d0, d1, d2, sat);
  const int64_t rows = (int64_t)(d1 + 1) * d2p;
  hipLaunchKernelGGL(sat_scan_x_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, d0, rows, sat);
  const int64_t cols = (int64_t)(d0 + 1) * d2p;
  hipLaunchKernelGGL(sat_scan_y_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, stream, d0, d1, d2p, sat);
  if (dim == 3) {
    const int64_t plane = (int64_t)(d0 + 1) * (d1 + 1);
    hipLaunchKernelGGL(sat_scan_z_kernel, dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, stream, d0, d1, d2, sat);
  }
  return hipGetLastError();
}

hipError_t launch_expand_grid(int dim, int control, const GridArgs &a, hipStream_t s) {
  return dispatch_grid<hipError_t>(dim, control, a.pot != nullptr, hipErrorInvalidValue, [&](auto t) {
    using T = decltype(t);
    return launch_grid_inst<T::D, T::K, T::YAW, T::POT>(a, s);
  });
}

hipError_t launch_grid_prescreen(int dim, int control, const GridArgs &a, int32_t *live, uint32_t *live_n, uint32_t *live_zero,
                                 hipStream_t s) {
  if (a.n_nodes == 0) return hipErrorInvalidValue;  // (the caller only pre-screens large frontiers; the counters must change hands)
  const unsigned blocks = (unsigned)((a.n_nodes + 255) / 256);
#define MPLX_PS(D, K) hipLaunchKernelGGL((grid_prescreen_kernel<D, K>), dim3(blocks), dim3(256), 0, s, a.nodes, a.n_nodes, \
                                         a.node_stride, a.yaw_max, a.yaw, a.l_count, live, live_n, live_zero)
  if (dim == 2 && control == 0x13) MPLX_PS(2, 2);
  else if (dim == 2 && control == 0x17) MPLX_PS(2, 3);
  else if (dim == 3 && control == 0x13) MPLX_PS(3, 2);
  else if (dim == 3 && control == 0x17) MPLX_PS(3, 3);
  else if (dim == 2 && control == 0x1f) MPLX_PS(2, 4);
  else if (dim == 3 && control == 0x1f) MPLX_PS(3, 4);
  else return hipErrorInvalidValue;
#undef MPLX_PS
  return hipGetLastError();
}

int grid_resident_blocks(int dim, int control, bool pot, size_t lds) {
  return dispatch_grid<int>(dim, control, pot, 0, [&](auto t) {
    using T = decltype(t);
    return resident_inst<T::D, T::K, T::YAW, T::POT>(lds);
  });
}

}  // namespace mplx
