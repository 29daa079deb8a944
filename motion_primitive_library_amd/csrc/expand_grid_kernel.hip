// expand_grid_kernel.hip -- factorised, list-producing successor expansion for
// gfx950 (MI355X).
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
// a workgroup that owns whole nodes evaluates D*9 axis entries per node instead
// of 729*D, keeps them in LDS, and the per-pair / per-sample work becomes table
// look-ups, integer hashing and the map gather.  The host detects the distinct
// values of any control table (mplx_set_controls); tables with more than 16
// distinct values on an axis run expand_tile_kernel.hip instead.
//
// Phases of one tile (npb whole nodes, one 512-thread workgroup, persistent):
//   T1  axis entries (node, axis, value): limits, n_axis, end state, lattice
//       integers, u*u*T                                          -> LDS
//   A   every pair: valid = AND of entry flags, n = max n_axis, hash = fold of
//       the entries' integers, emit = valid && hash != hash(node)
//       (env_map.h:158); ordered compaction; action / hash / Waypoint written
//       to the node's list; the set of sample counts n in use per node
//   T2  cell tables: for every n in use, axis entry and sample k the cell
//       coordinate of p(t_k) (or -1 outside), t_k = the reference's accumulated
//       `for (t = 0; t < T; t += T/n)` times (env_map.h:97-99)      -> LDS
//   W/B work list of (pair, k) samples, dense lanes, kUB samples in flight per
//       lane: three LDS look-ups, one map byte (+ region bit), LDS atomicMin of
//       the first blocked k
//   C   cost = J + w*dt or +inf (env_map.h:162-169), iteration count
// Bit-exactness: n = max(5, ceil(max_v*T/res)) with max_v = max over axes equals
// the max over axes of the per-axis counts because *, / by a positive constant
// and ceil are monotone; everything else is the same per-axis arithmetic as the
// other kernels, evaluated once instead of once per pair.
//
// Scope: controls without yaw, no potential map, v_max > 0 (or VEL), Dim 2/3,
// K = 1..4, map dims <= 32767 per axis (int16 cell table).
#include "mplx_internal.h"
#include "mplx_device_common.h"

namespace mplx {
namespace {

using namespace dev;

constexpr int kBT = 512;
constexpr int kWaves = kBT / 64;
constexpr int kTabStride = 64;  // row stride of the global time table (launch_make_tables)
constexpr int kUB = 4;          // samples in flight per lane in phase B

// Ordered exclusive prefix of a per-thread flag over the workgroup.
__device__ __forceinline__ int block_scan(bool f, int *total, int *s_wsum) {
  const unsigned long long m = __ballot(f);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int within = __popcll(m & ((1ull << lane) - 1ull));
  if (lane == 0) s_wsum[wv] = __popcll(m);
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < kWaves; i++) {
    const int c = s_wsum[i];
    if (i < wv) base += c;
    tot += c;
  }
  __syncthreads();
  *total = tot;
  return base + within;
}

}  // namespace

// LDS carve-up, shared by host (size) and device (offsets).
struct GridLds {
  int o_hcur, o_nmask, o_tt, o_node, o_uval, o_est, o_eJ, o_eq, o_eflag, o_uidx, o_einfo, o_fb, o_misc,
      o_ncnt, o_nbase, o_rowinfo, o_wl, o_cell, o_tc, total;
  int F, EN, NR, tts, P_cap;
  __host__ __device__ GridLds(int D, int npb, int nU, int ndp, int n_max, int wl_cap) {
    F = 4 * D + 2;
    EN = npb * D * ndp;
    NR = n_max - 4;
    tts = n_max + 1;
    P_cap = npb * nU;
    int b = 0;
    o_hcur = b; b += npb * 8;
    o_nmask = b; b += npb * 8;
    o_tt = b; b += (n_max + 1) * tts * 8;
    o_node = b; b += npb * F * 8;
    o_uval = b; b += D * ndp * 8;
    o_est = b; b += EN * 4 * 8;
    o_eJ = b; b += EN * 8;
    b = (b + 15) & ~15;
    o_eq = b; b += EN * 4 * 4;
    o_eflag = b; b += EN * 4;
    o_uidx = b; b += nU * 4;
    o_einfo = b; b += P_cap * 4;
    o_fb = b; b += P_cap * 4;
    o_misc = b; b += 32 * 4;
    o_ncnt = b; b += npb * 4;
    o_nbase = b; b += npb * 4;
    o_rowinfo = b; b += ((npb * NR + 1) & ~1) * 2;
    o_wl = b; b += ((wl_cap + 1) & ~1) * 2;
    o_cell = b; b += ((EN * NR * tts + 1) & ~1) * 2;
    o_tc = b; b += 64;
    total = (b + 15) & ~15;
  }
};

namespace {

// pair index inside the tile -> (local node, control); exact for p < 2^20
__device__ __forceinline__ void split_pair(int p, int nU, float inv_nU, int npb, int *nl, int *ci) {
  if (npb == 1) { *nl = 0; *ci = p; return; }
  const int q = (int)(((float)p + 0.5f) * inv_nU);
  *nl = q;
  *ci = p - q * nU;
}

template <int D, int K>
__global__ __launch_bounds__(kBT, 6) void expand_grid_kernel(const GridArgs A) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int F = 4 * D + 2;
  const int npb = A.npb, nU = A.nU, ndp = A.ndp;
  const GridLds L(D, npb, nU, ndp, A.n_max, A.wl_cap);
  uint64_t *s_hcur = (uint64_t *)(smem + L.o_hcur);
  unsigned long long *s_nmask = (unsigned long long *)(smem + L.o_nmask);
  double *s_tt = (double *)(smem + L.o_tt);
  double *s_node = (double *)(smem + L.o_node);
  double *s_uval = (double *)(smem + L.o_uval);
  double *s_est = (double *)(smem + L.o_est);
  double *s_eJ = (double *)(smem + L.o_eJ);
  int *s_eq = (int *)(smem + L.o_eq);
  int *s_eflag = (int *)(smem + L.o_eflag);
  unsigned int *s_uidx = (unsigned int *)(smem + L.o_uidx);
  unsigned int *s_einfo = (unsigned int *)(smem + L.o_einfo);
  unsigned int *s_fb = (unsigned int *)(smem + L.o_fb);
  int *s_misc = (int *)(smem + L.o_misc);
  int *s_ncnt = (int *)(smem + L.o_ncnt);
  int *s_nbase = (int *)(smem + L.o_nbase);
  unsigned short *s_rowinfo = (unsigned short *)(smem + L.o_rowinfo);
  unsigned short *s_wl = (unsigned short *)(smem + L.o_wl);
  short *s_cell = (short *)(smem + L.o_cell);
  unsigned char *s_tc = smem + L.o_tc;
  int *s_wcount = s_misc + 0;
  int *s_rows = s_misc + 1;
  int *s_wsum = s_misc + 8;

  const int tid = threadIdx.x;
  const int tts = L.tts, NR = L.NR;
  const int CS = NR * tts;  // cell-table entries of one axis entry
  const double T = A.dt;
  const float inv_nU = A.inv_nU;
  const double org[3] = {A.org0, A.org1, A.org2};
  const int dims[3] = {A.dim0, A.dim1, A.dim2};
  const int nd[3] = {A.nd0, A.nd1, A.nd2};

  // ---- once per (persistent) workgroup: time tables, control factorisation
  if (tid < 64) s_tc[tid] = A.tcnt[tid];
  for (int i = tid; i < (A.n_max + 1) * tts; i += kBT) {
    const int n = i / tts, k = i - n * tts;
    s_tt[i] = A.ttab[n * kTabStride + k];
  }
  for (int i = tid; i < D * ndp; i += kBT) {
    const int ax = i / ndp, j = i - ax * ndp;
    s_uval[i] = A.uvals[ax * 16 + j];
  }
  for (int i = tid; i < nU; i += kBT) s_uidx[i] = A.uidx[i];

  const int64_t n_tiles = (A.n_nodes + npb - 1) / npb;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    __syncthreads();  // LDS of the previous tile is free; tables visible
    const int64_t node0 = tile * npb;
    const int nn = (int)((A.n_nodes - node0) < (int64_t)npb ? (A.n_nodes - node0) : (int64_t)npb);
    const int P = nn * nU;

    // ---- phase 0: node states into LDS
    for (int i = tid; i < nn * F; i += kBT) {
      const int r = i / nn, nl = i - r * nn;  // consecutive lanes -> consecutive nodes of one row
      s_node[nl * F + r] = A.nodes[(int64_t)r * A.node_stride + node0 + nl];
    }
    __syncthreads();

    // ---- phase T1: node hashes and axis entries
    for (int nl = tid; nl < nn; nl += kBT) {
      const double *nd_ = s_node + nl * F;
      double p[D], v[D], a[D], j[D];
#pragma unroll
      for (int i = 0; i < D; i++) {
        p[i] = nd_[0 * D + i];
        v[i] = (K >= 2) ? nd_[1 * D + i] : 0.0;
        a[i] = (K >= 3) ? nd_[2 * D + i] : 0.0;
        j[i] = (K >= 4) ? nd_[3 * D + i] : 0.0;
      }
      s_hcur[nl] = lattice_hash<D, K>(p, v, a, j, A.R001, A.R01);
      s_nmask[nl] = 0ull;
      s_ncnt[nl] = 0;
    }
    for (int en = tid; en < nn * D * ndp; en += kBT) {
      const int row = en / ndp, jv = en - row * ndp;  // row = nl*D + axis
      const int nl = row / D, ax = row - nl * D;
      int flag = 0;
      if (jv < nd[ax]) {
        const double *nd_ = s_node + nl * F;
        const double p = nd_[0 * D + ax];
        const double v = (K >= 2) ? nd_[1 * D + ax] : 0.0;
        const double a = (K >= 3) ? nd_[2 * D + ax] : 0.0;
        const double j = (K >= 4) ? nd_[3 * D + ax] : 0.0;
        const double u = s_uval[ax * ndp + jv];
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
        s_est[en * 4 + 0] = np_;
        s_est[en * 4 + 1] = nv_;
        s_est[en * 4 + 2] = na_;
        s_est[en * 4 + 3] = nj_;
        s_eq[en * 4 + 0] = quantise(np_, 0.01, A.R001);
        s_eq[en * 4 + 1] = (K >= 2) ? quantise(nv_, 0.1, A.R01) : 0;
        s_eq[en * 4 + 2] = (K >= 3) ? quantise(na_, 0.1, A.R01) : 0;
        s_eq[en * 4 + 3] = (K >= 4) ? quantise(nj_, 0.1, A.R01) : 0;
        s_eJ[en] = u * u * T;  // Primitive::J of a forward primitive (see expand_kernel.hip)
        flag = (valid ? 1 : 0) | ((p == np_) ? 2 : 0) | (n << 8);
      }
      s_eflag[en] = flag;
    }
    __syncthreads();

    // ---- phase A: every pair; ordered compaction of the emitted successors
    int E = 0;  // emitted so far in the tile (uniform)
    for (int base = 0; base < P; base += kBT) {
      const int p = base + tid;
      bool emit = false;
      int nl = 0, ci = 0, n = 0;
      int en[D];
      uint64_t h = 0;
      bool same_pos = false;
      if (p < P) {
        split_pair(p, nU, inv_nU, npb, &nl, &ci);
        const unsigned int pk = s_uidx[ci];
        int fl = 3;
#pragma unroll
        for (int i = 0; i < D; i++) {
          en[i] = (nl * D + i) * ndp + (int)((pk >> (8 * i)) & 255u);
          const int f = s_eflag[en[i]];
          fl &= f;
          const int ni = f >> 8;
          n = ni > n ? ni : n;
          const int4 q = *(const int4 *)(s_eq + en[i] * 4);
          fold(h, q.x);
          if (K >= 2) fold(h, q.y);
          if (K >= 3) fold(h, q.z);
          if (K >= 4) fold(h, q.w);
        }
        same_pos = (fl & 2) != 0;
        emit = (fl & 1) && (h != s_hcur[nl]);  // env_map.h:158: `tn == curr` is a hash comparison
      }
      int tot;
      const int e = E + block_scan(emit, &tot, s_wsum);
      if (p < P && ci == 0) s_nbase[nl] = e;  // emitted before this node's first pair
      E += tot;
      __syncthreads();
      if (emit) {
        const int jpos = e - s_nbase[nl];
        const int ns = same_pos ? 0 : n;  // unchanged position: not traversed (env_map.h:163); tcnt[0] == 0
        unsigned int info = (unsigned)nl | ((unsigned)ns << 17);
#pragma unroll
        for (int i = 0; i < D; i++) info |= (unsigned)(en[i] - (nl * D + i) * ndp) << (5 + 4 * i);
        s_einfo[e] = info;
        s_fb[e] = 0xffffffffu;
        if (ns) atomicOr(&s_nmask[nl], 1ull << ns);
        atomicAdd(&s_ncnt[nl], 1);
        const int64_t idx = (node0 + nl) * (int64_t)nU + jpos;
        const bool wr = !(A.dbg & 2);  // timing ablation only
        if (wr && A.l_action) A.l_action[idx] = ci;
        if (wr && A.l_hash) A.l_hash[idx] = h;
        if (wr && A.l_state) {
          double *o = A.l_state + idx;
          const int64_t ss = A.l_stride;
#pragma unroll
          for (int i = 0; i < D; i++) {
            const double *st = s_est + en[i] * 4;
            o[(0 * D + i) * ss] = st[0];
            o[(1 * D + i) * ss] = st[1];
            o[(2 * D + i) * ss] = st[2];
            o[(3 * D + i) * ss] = st[3];
          }
          o[(4 * D) * ss] = 0.0;  // Waypoint::yaw of a control without yaw (primitive.h:322)
          o[(4 * D + 1) * ss] = s_node[nl * F + 4 * D + 1] + A.dt;  // env_map.h:161
        }
      }
    }
    __syncthreads();

    // ---- phase T2: cell tables of the sample counts in use
    if (tid < 64) {  // rows (node, n): one wave lists them
      const int nl = tid;
      const unsigned long long m = (nl < nn) ? s_nmask[nl] : 0ull;
      const int c = __popcll(m);
      int incl = c;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if (tid >= d) incl += o;
      }
      int r = incl - c;
      unsigned long long mm = m;
      while (mm) {
        const int n = __ffsll((long long)mm) - 1;
        mm &= mm - 1;
        s_rowinfo[r++] = (unsigned short)((nl << 6) | n);
      }
      if (tid == 63) *s_rows = incl;
    }
    __syncthreads();
    {
      const int R = *s_rows;
      const int per_row = D * ndp * tts;
      const int items = (A.dbg & 1) ? 0 : R * per_row;
      for (int x = tid; x < items; x += kBT) {
        const int r = x / per_row;
        const int y = x - r * per_row;
        const int aj = y / tts, k = y - aj * tts;  // aj = axis*ndp + value
        const int ax = aj / ndp, jv = aj - ax * ndp;
        const int ri = s_rowinfo[r];
        const int nl = ri >> 6, n = ri & 63;
        if (jv >= nd[ax] || k >= (int)s_tc[n]) continue;
        const double *nd_ = s_node + nl * F;
        Ax<K> q;
        q.init(nd_[0 * D + ax], (K >= 2) ? nd_[1 * D + ax] : 0.0, (K >= 3) ? nd_[2 * D + ax] : 0.0,
               (K >= 4) ? nd_[3 * D + ax] : 0.0, s_uval[aj]);
        const double t = s_tt[n * tts + k];
        // map_util.h:103-108: cell = round((pos - origin) / res - 0.5), then bounds.
        const double qd = div_by(q.template pos<false>(t) - org[ax], A.res, A.Rres);
        const double sh = qd - 0.5;
        // sh > -0.5  <=>  the rounded cell is >= 0; then qd > 0 and (qd - 0.5 being exact for
        // qd >= 0.5) round-half-away(sh) == trunc(qd).  Otherwise the cell is negative: outside.
        const int c = (int)qd;
        const bool in = (sh > -0.5) && (c < dims[ax]);
        s_cell[((nl * D + ax) * ndp + jv) * CS + (n - 5) * tts + k] = (short)(in ? c : -1);
      }
    }
    __syncthreads();

    // ---- phases W + B, in chunks of emitted pairs bounded by the work-list capacity
    const int CE = A.wl_cap / tts;
    for (int e0 = 0; e0 < E && !(A.dbg & 1); e0 += CE) {
      const int e1 = (e0 + CE < E) ? e0 + CE : E;
      if (tid == 0) *s_wcount = 0;
      __syncthreads();
      for (int base = e0; base < e1; base += kBT) {
        const int e = base + tid;
        const int c = (e < e1) ? (int)s_tc[s_einfo[e] >> 17] : 0;
        const int lane = tid & 63;
        int incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int o = __shfl_up(incl, d, 64);
          if (lane >= d) incl += o;
        }
        const int wave_total = __shfl(incl, 63, 64);
        int wbase = 0;
        if (lane == 63 && wave_total > 0) wbase = atomicAdd(s_wcount, wave_total);
        wbase = __shfl(wbase, 63, 64);
        const int off = wbase + incl - c;
        const int le = e - e0;  // < CE <= 1024
        for (int k = 0; k < c; k++) s_wl[off + k] = (unsigned short)((le << 6) | k);
      }
      __syncthreads();
      const int W = *s_wcount;
      for (int w0 = tid; w0 < W; w0 += kBT * kUB) {
        int ee[kUB], kk[kUB], midx[kUB];
        bool act[kUB], inside[kUB];
#pragma unroll
        for (int q = 0; q < kUB; q++) {
          const int w = w0 + q * kBT;
          act[q] = w < W;
          const int ent = s_wl[act[q] ? w : W - 1];
          ee[q] = e0 + (ent >> 6);
          kk[q] = ent & 63;
        }
        unsigned int info[kUB];
#pragma unroll
        for (int q = 0; q < kUB; q++) info[q] = s_einfo[ee[q]];
#pragma unroll
        for (int q = 0; q < kUB; q++) {
          const int nl = info[q] & 31;
          const int n = info[q] >> 17;
          const int tail = (n - 5) * tts + kk[q];
          int cell[D];
#pragma unroll
          for (int i = 0; i < D; i++) {
            const int jv = (info[q] >> (5 + 4 * i)) & 15;
            cell[i] = s_cell[((nl * D + i) * ndp + jv) * CS + tail];
          }
          int any = cell[0] | cell[1];
          int idx = cell[0] + dims[0] * cell[1];
          if (D == 3) { any |= cell[2]; idx += dims[0] * dims[1] * cell[2]; }
          inside[q] = any >= 0;
          midx[q] = (inside[q] && !(A.dbg & 16)) ? idx : 0;  // dbg 16: timing ablation, all lanes read cell 0
        }
        int mval[kUB];
        unsigned int rword[kUB];
#pragma unroll
        for (int q = 0; q < kUB; q++) {
          mval[q] = A.map[(unsigned)midx[q]];
          rword[q] = (A.region != nullptr) ? A.region[(unsigned)midx[q] >> 5] : 0xffffffffu;
        }
#pragma unroll
        for (int q = 0; q < kUB; q++) {
          const bool blocked = !inside[q] || !((rword[q] >> (midx[q] & 31)) & 1u) || mval[q] == 100;
          if (act[q] && blocked) atomicMin(&s_fb[ee[q]], (unsigned int)kk[q]);
        }
      }
      __syncthreads();
    }

    // ---- phase C: costs (and iteration counts) of the emitted successors
    for (int e = tid; e < E && !(A.dbg & 4); e += kBT) {
      const unsigned int info = s_einfo[e];
      const int nl = info & 31;
      const int n = info >> 17;
      const unsigned int fb = s_fb[e];
      const bool blocked = (fb != 0xffffffffu);
      // iterations the reference executes: up to and including the first blocked sample
      const int iters = blocked ? (int)fb + 1 : (int)s_tc[n];
      double J = 0;
#pragma unroll
      for (int i = 0; i < D; i++) J += s_eJ[(nl * D + i) * ndp + ((info >> (5 + 4 * i)) & 15)];
      const double cost = blocked ? INFINITY : 0.0 + (J + A.w * A.dt);
      const int64_t idx = (node0 + nl) * (int64_t)nU + (e - s_nbase[nl]);
      if (A.l_cost) A.l_cost[idx] = cost;
      if (A.l_iters) A.l_iters[idx] = iters;
    }
    for (int nl = tid; nl < nn; nl += kBT)
      if (A.l_count) A.l_count[node0 + nl] = s_ncnt[nl];
  }  // tile loop
}

template <int D, int K>
hipError_t launch_grid_inst(const GridArgs &a, hipStream_t stream) {
  if (a.n_nodes == 0) return hipSuccess;
  const int64_t n_tiles = (a.n_nodes + a.npb - 1) / a.npb;
  const int64_t blocks = n_tiles < (int64_t)a.grid_limit ? n_tiles : (int64_t)a.grid_limit;
  const size_t lds = grid_lds_bytes(D, a.npb, a.nU, a.ndp, a.n_max, a.wl_cap);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void *)expand_grid_kernel<D, K>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((expand_grid_kernel<D, K>), dim3((unsigned)blocks), dim3(kBT), lds, stream, a);
  return hipGetLastError();
}

}  // namespace

size_t grid_lds_bytes(int dim, int npb, int nU, int ndp, int n_max, int wl_cap) {
  return (size_t)GridLds(dim, npb, nU, ndp, n_max, wl_cap).total;
}

hipError_t launch_expand_grid(int dim, int control, const GridArgs &a, hipStream_t s) {
  if (dim == 2) {
    switch (control) {
      case 0x01: return launch_grid_inst<2, 1>(a, s);
      case 0x03: return launch_grid_inst<2, 2>(a, s);
      case 0x07: return launch_grid_inst<2, 3>(a, s);
      case 0x0f: return launch_grid_inst<2, 4>(a, s);
    }
  } else if (dim == 3) {
    switch (control) {
      case 0x01: return launch_grid_inst<3, 1>(a, s);
      case 0x03: return launch_grid_inst<3, 2>(a, s);
      case 0x07: return launch_grid_inst<3, 3>(a, s);
      case 0x0f: return launch_grid_inst<3, 4>(a, s);
    }
  }
  return hipErrorInvalidValue;
}

}  // namespace mplx
