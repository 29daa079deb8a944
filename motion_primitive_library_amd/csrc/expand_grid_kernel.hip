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
// What the first version of this kernel taught (profiles/README.md): with the
// arithmetic gone the time was barrier / latency bound -- eight waves marching
// through ten phases in lock step.  Here a wave never waits for another wave:
// all phases are wave-synchronous (LDS is in-order per wave), a workgroup is
// only a container for kWPB independent waves that share the read-only tables.
//
// Per node (one wave):
//   T1  axis entries (axis, value): limits, n_axis, end state, lattice integers,
//       u*u*T; prefix tables over the first D-1 axes (partial hash, flags); the
//       address LUT of the node's neighbourhood in the blocked-bit map
//   A   64 pairs per step: valid = AND of flags, n = max n_axis, hash = prefix
//       hash folded with the last axis, emit = valid && hash != hash(node)
//       (env_map.h:158); emitted pairs are appended, in order, to an LDS queue
//   D   whenever 64 pairs are queued (dense lanes): action / hash / Waypoint
//       written to the node's list; the sample loop of traverse_primitive, lane
//       = pair, k sequential, four samples in flight: the cell offsets of
//       p(t_k) come from per-(axis entry, n) rows kept in a small LDS row cache
//       (built on demand; t_k = the reference's accumulated `t += T/n` times,
//       env_map.h:97-99), the LUT turns three offsets into one word address +
//       bit of the blocked-bit map; cost = J + w*dt or +inf (env_map.h:162-169)
// The blocked-bit map (built on the device from the map and the search region
// whenever they change) stores 1 bit per cell, blocked = occupied or outside the
// region (env_map.h:104-119), in bricks of 8x8x8 cells (2D: 32x16) = one 64-byte
// line, so the 64 lanes of a sample step -- 64 controls of ONE node at the same
// t_k, i.e. positions within +-u_max t^2/2 of each other -- touch a handful of
// lines instead of 64, and a 512^3 map is 16 MiB instead of 128.
//
// Bit-exactness: n = max(5, ceil(max_v*T/res)) with max_v = max over axes equals
// the max over axes of the per-axis counts because *, / by a positive constant
// and ceil are monotone; everything else is the same per-axis arithmetic as the
// other kernels, evaluated once instead of once per pair.
//
// Scope: controls without yaw, no potential map, v_max > 0 (or VEL), Dim 2/3,
// K = 1..4, n_max <= 61, padded map <= 2^29 cells.
#include "mplx_internal.h"
#include "mplx_device_common.h"

namespace mplx {

// LDS carve-up, shared by host (size) and device (offsets).
struct GridLds {
  // shared by the workgroup (read-only after set-up)
  int o_tt, o_uval, o_uidx, o_tc, o_wave0;
  // per wave, relative to the wave's block
  int w_hcur, w_node, w_est, w_eJ, w_hp, w_eq, w_eflag, w_fp, w_lut, w_queue, w_misc, w_rowmap, w_cell, wave_bytes;
  int total;
  int F, EN, PN, LUTN, tts;
  __host__ __device__ GridLds(int D, int waves, int nU, int ndp, int n_max, int rmax) {
    F = 4 * D + 2;
    EN = D * ndp;
    PN = (D == 3) ? ndp * ndp : ndp;
    LUTN = 2 * (n_max + 2) + 1;
    tts = n_max + 1;
    int b = 0;
    o_tt = b; b += (n_max + 1) * tts * 8;
    o_uval = b; b += EN * 8;
    o_uidx = b; b += nU * 4;
    o_tc = b; b += 64;
    b = (b + 15) & ~15;
    o_wave0 = b;
    int w = 0;
    w_hcur = w; w += 8;
    w_node = w; w += F * 8;
    w_est = w; w += EN * 4 * 8;
    w_eJ = w; w += EN * 8;
    w_hp = w; w += PN * 8;
    w = (w + 15) & ~15;
    w_eq = w; w += EN * 4 * 4;
    w_eflag = w; w += EN * 4;
    w_fp = w; w += PN * 4;
    w_lut = w; w += D * LUTN * 4;
    w_queue = w; w += 128 * 4;
    w_misc = w; w += 8 * 4;
    w_rowmap = w; w += 64;
    w_cell = w; w += EN * rmax * tts;
    wave_bytes = (w + 15) & ~15;
    total = o_wave0 + waves * wave_bytes;
  }
};

namespace {

using namespace dev;

constexpr int kWPB = 4;         // waves (= nodes in flight) per workgroup
constexpr int kBT = 64 * kWPB;
constexpr int kTabStride = 64;  // row stride of the global time table (launch_make_tables)
constexpr int kUB = 8;          // samples in flight per lane
constexpr unsigned kOut = 1u << 29;

// Orders this wave's LDS traffic: LDS executes a wave's instructions in order, so
// only the compiler has to be kept from moving accesses across the point.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

template <int D, int K>
__global__ __launch_bounds__(kBT) void expand_grid_kernel(const GridArgs A) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int F = 4 * D + 2;
  const int nU = A.nU, ndp = A.ndp, RM = A.rmax;
  const GridLds L(D, kWPB, nU, ndp, A.n_max, RM);
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const double *s_tt = (const double *)(smem + L.o_tt);
  const double *s_uval = (const double *)(smem + L.o_uval);
  const unsigned int *s_uidx = (const unsigned int *)(smem + L.o_uidx);
  const unsigned char *s_tc = smem + L.o_tc;
  unsigned char *wb = smem + L.o_wave0 + wv * L.wave_bytes;
  uint64_t *s_hcur = (uint64_t *)(wb + L.w_hcur);
  double *s_node = (double *)(wb + L.w_node);
  double *s_est = (double *)(wb + L.w_est);
  double *s_eJ = (double *)(wb + L.w_eJ);
  uint64_t *s_hp = (uint64_t *)(wb + L.w_hp);
  int *s_eq = (int *)(wb + L.w_eq);
  int *s_eflag = (int *)(wb + L.w_eflag);
  int *s_fp = (int *)(wb + L.w_fp);
  unsigned int *s_lut = (unsigned int *)(wb + L.w_lut);
  unsigned int *s_queue = (unsigned int *)(wb + L.w_queue);
  int *s_misc = (int *)(wb + L.w_misc);  // [0..2] base cell of the node per axis
  unsigned char *s_rowmap = wb + L.w_rowmap;
  unsigned char *s_cell = wb + L.w_cell;

  const int tts = L.tts, EN = L.EN, PN = L.PN, LUTN = L.LUTN;
  const int half = A.n_max + 2;  // cell-offset code = offset + half
  const double T = A.dt;
  const double org[3] = {A.org0, A.org1, A.org2};
  const int dims[3] = {A.dim0, A.dim1, A.dim2};
  const int nd[3] = {A.nd0, A.nd1, A.nd2};

  // ---- once per (persistent) workgroup: shared read-only tables
  {
    double *tt = (double *)(smem + L.o_tt);
    for (int i = threadIdx.x; i < (A.n_max + 1) * tts; i += kBT) {
      const int n = i / tts, k = i - n * tts;
      tt[i] = A.ttab[n * kTabStride + k];
    }
    double *uv = (double *)(smem + L.o_uval);
    for (int i = threadIdx.x; i < EN; i += kBT) {
      const int ax = i / ndp, j = i - ax * ndp;
      uv[i] = A.uvals[ax * 16 + j];
    }
    unsigned int *ui = (unsigned int *)(smem + L.o_uidx);
    for (int i = threadIdx.x; i < nU; i += kBT) ui[i] = A.uidx[i];
    if (threadIdx.x < 64) smem[L.o_tc + threadIdx.x] = A.tcnt[threadIdx.x];
  }
  __syncthreads();  // the only workgroup barrier

  const int64_t wave_id = (int64_t)blockIdx.x * kWPB + wv;
  const int64_t wave_stride = (int64_t)gridDim.x * kWPB;
  double nxt = 0.0;  // lanes < F: one field of the next node (prefetched)
  if (wave_id < A.n_nodes && lane < F) nxt = A.nodes[(int64_t)lane * A.node_stride + wave_id];

  for (int64_t node = wave_id; node < A.n_nodes; node += wave_stride) {
    // ---- phase 0: node state into LDS, prefetch of the next node
    wave_sync();
    if (lane < F) s_node[lane] = nxt;
    if (node + wave_stride < A.n_nodes && lane < F)
      nxt = A.nodes[(int64_t)lane * A.node_stride + node + wave_stride];
    s_rowmap[lane] = 0xff;
    wave_sync();

    // ---- phase T1: axis entries; node hash (lane 63)
    if (lane < EN) {
      const int ax = lane / ndp, jv = lane - ax * ndp;
      int flag = 0;
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
        s_est[lane * 4 + 0] = np_;
        s_est[lane * 4 + 1] = nv_;
        s_est[lane * 4 + 2] = na_;
        s_est[lane * 4 + 3] = nj_;
        s_eq[lane * 4 + 0] = quantise(np_, 0.01, A.R001);
        s_eq[lane * 4 + 1] = (K >= 2) ? quantise(nv_, 0.1, A.R01) : 0;
        s_eq[lane * 4 + 2] = (K >= 3) ? quantise(na_, 0.1, A.R01) : 0;
        s_eq[lane * 4 + 3] = (K >= 4) ? quantise(nj_, 0.1, A.R01) : 0;
        s_eJ[lane] = u * u * T;  // Primitive::J of a forward primitive (see expand_kernel.hip)
        flag = (valid ? 1 : 0) | ((p == np_) ? 2 : 0) | (n << 8);
        if (jv == 0) {
          // reference cell of the node on this axis (any integer would do: offsets are relative to it)
          const double qd = div_by(p - org[ax], A.res, A.Rres);
          s_misc[ax] = (qd - 0.5 > -0.5) ? (int)qd : -1;
        }
      }
      s_eflag[lane] = flag;
    } else if (lane == 63) {
      double p[D], v[D], a[D], j[D];
#pragma unroll
      for (int i = 0; i < D; i++) {
        p[i] = s_node[0 * D + i];
        v[i] = (K >= 2) ? s_node[1 * D + i] : 0.0;
        a[i] = (K >= 3) ? s_node[2 * D + i] : 0.0;
        j[i] = (K >= 4) ? s_node[3 * D + i] : 0.0;
      }
      s_hcur[0] = lattice_hash<D, K>(p, v, a, j, A.R001, A.R01);
    }
    wave_sync();

    // ---- prefix tables over the first D-1 axes; address LUT of the neighbourhood
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
        int f = s_eflag[e0];
        {
          const int4 q = *(const int4 *)(s_eq + e0 * 4);
          fold(h, q.x);
          if (K >= 2) fold(h, q.y);
          if (K >= 3) fold(h, q.z);
          if (K >= 4) fold(h, q.w);
        }
        if (D == 3) {
          const int4 q = *(const int4 *)(s_eq + e1 * 4);
          fold(h, q.x);
          if (K >= 2) fold(h, q.y);
          if (K >= 3) fold(h, q.z);
          if (K >= 4) fold(h, q.w);
          const int f1 = s_eflag[e1];
          const int n0 = f >> 8, n1 = f1 >> 8;
          f = (f & f1 & 3) | ((n0 > n1 ? n0 : n1) << 8);
        }
        s_hp[x] = h;
        s_fp[x] = f;
      }
    }
    for (int x = lane; x < D * LUTN; x += 64) {
      const int ax = x / LUTN, d = x - ax * LUTN;
      const int c = s_misc[ax] + d - half;
      unsigned int w = kOut;
      if (c >= 0 && c < dims[ax]) {
        if (D == 3) {
          if (ax == 0) w = (unsigned)(((c >> 3) * 16) << 5) | (unsigned)(c & 7);
          else if (ax == 1) w = (unsigned)((((c >> 3) * A.nbx * 16) + ((c >> 2) & 1)) << 5) | (unsigned)((c & 3) << 3);
          else w = (unsigned)((((c >> 3) * A.nbx * A.nby * 16) + ((c & 7) << 1)) << 5);
        } else {
          if (ax == 0) w = (unsigned)(((c >> 5) * 16) << 5) | (unsigned)(c & 31);
          else w = (unsigned)((((c >> 4) * A.nbx * 16) + (c & 15)) << 5);
        }
      }
      s_lut[x] = w;
    }
    wave_sync();

    const uint64_t hcur = s_hcur[0];
    const double node_t = s_node[4 * D + 1];
    int qn = 0;       // queued emitted pairs (uniform)
    int drained = 0;  // successors already written for this node (uniform)
    int nrows = 0;    // rows in the cell-row cache (uniform)

    for (int base = 0; base < nU; base += 64) {
      // ---- phase A: 64 pairs
      {
        const int ci = base + lane;
        bool emit = false;
        unsigned int info = 0;
        if (ci < nU) {
          const unsigned int pk = s_uidx[ci];
          const int j0 = pk & 255, j1 = (pk >> 8) & 255, j2 = (pk >> 16) & 255;
          const int px = (D == 3) ? j0 * ndp + j1 : j0;
          const int eL = (D - 1) * ndp + ((D == 3) ? j2 : j1);
          uint64_t h = s_hp[px];
          const int4 q = *(const int4 *)(s_eq + eL * 4);
          fold(h, q.x);
          if (K >= 2) fold(h, q.y);
          if (K >= 3) fold(h, q.z);
          if (K >= 4) fold(h, q.w);
          const int f0 = s_fp[px], f1 = s_eflag[eL];
          const int fl = f0 & f1;
          const int n0 = f0 >> 8, n1 = f1 >> 8;
          const int n = (fl & 2) ? 0 : (n0 > n1 ? n0 : n1);  // unchanged position: not traversed (env_map.h:163)
          emit = (fl & 1) && (h != hcur);                    // env_map.h:158: `tn == curr` is a hash comparison
          info = (pk & 15u) | (((pk >> 8) & 15u) << 4) | (((pk >> 16) & 15u) << 8) | ((unsigned)n << 12) |
                 ((unsigned)ci << 18);
        }
        const unsigned long long m = __ballot(emit);
        if (emit) s_queue[qn + __popcll(m & ((1ull << lane) - 1ull))] = info;
        qn += __popcll(m);
      }
      const bool last = base + 64 >= nU;
      // ---- phase D: drain the queue, 64 dense lanes at a time
      while (qn >= 64 || (last && qn > 0)) {
        wave_sync();
        const int cnt_d = qn < 64 ? qn : 64;
        const bool act = lane < cnt_d;
        const unsigned int info = s_queue[act ? lane : 0];
        const int j0 = info & 15, j1 = (info >> 4) & 15, j2 = (info >> 8) & 15;
        const int n = act ? (int)((info >> 12) & 63u) : 0;
        const int ci = info >> 18;
        int en[3] = {j0, ndp + j1, 2 * ndp + j2};
        const int64_t idx = node * (int64_t)nU + drained + lane;
        if (act && !(A.dbg & 2)) {
          const int px = (D == 3) ? j0 * ndp + j1 : j0;
          uint64_t h = s_hp[px];
          const int4 q = *(const int4 *)(s_eq + en[D - 1] * 4);
          fold(h, q.x);
          if (K >= 2) fold(h, q.y);
          if (K >= 3) fold(h, q.z);
          if (K >= 4) fold(h, q.w);
          if (A.l_action) A.l_action[idx] = ci;
          if (A.l_hash) A.l_hash[idx] = h;
          if (A.l_state) {
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
            o[(4 * D) * ss] = 0.0;                // Waypoint::yaw of a control without yaw (primitive.h:322)
            o[(4 * D + 1) * ss] = node_t + A.dt;  // env_map.h:161
          }
        }
        // ---- the sample loop of traverse_primitive (env_map.h:97-120)
        const int cntl = (int)s_tc[n];  // iterations of `for (t = 0; t < T; t += T/n)`; tc[0] == 0
        int fb = -1;                    // first blocked sample
        unsigned long long pend = (A.dbg & 1) ? 0ull : __ballot(act && n != 0);
        while (pend) {
          const bool inp = (pend >> lane) & 1ull;
          int r = inp ? (int)s_rowmap[n] : 0;
          unsigned long long miss = __ballot(inp && r == 0xff);
          if (miss == pend && nrows == RM) {  // nothing usable and no room: start the cache over
            s_rowmap[lane] = 0xff;
            nrows = 0;
            wave_sync();
          }
          while (miss && nrows < RM) {
            const int src = __ffsll((long long)miss) - 1;
            const int nn = __builtin_amdgcn_readlane(n, src);
            // build the row of sample count nn: cell-offset codes of every axis entry at t_0 .. t_{cnt-1}
            const int cn = (int)s_tc[nn];
            const float inv_cn = 1.0f / (float)cn;
            for (int x = lane; x < EN * cn; x += 64) {
              const int aj = (int)(((float)x + 0.5f) * inv_cn);  // exact: x < 2^12
              const int k = x - aj * cn;
              const int ax = aj / ndp, jv = aj - ax * ndp;
              if (jv >= nd[ax]) continue;
              Ax<K> q;
              q.init(s_node[0 * D + ax], (K >= 2) ? s_node[1 * D + ax] : 0.0, (K >= 3) ? s_node[2 * D + ax] : 0.0,
                     (K >= 4) ? s_node[3 * D + ax] : 0.0, s_uval[aj]);
              const double t = s_tt[nn * tts + k];
              // map_util.h:103-108: cell = round((pos - origin) / res - 0.5), then bounds.
              const double qd = div_by(q.template pos<false>(t) - org[ax], A.res, A.Rres);
              const double sh = qd - 0.5;
              // sh > -0.5  <=>  the rounded cell is >= 0; then qd > 0 and (qd - 0.5 being exact for
              // qd >= 0.5) round-half-away(sh) == trunc(qd).  Otherwise the cell is negative: outside.
              int code = 0xff;
              if (sh > -0.5) {
                int d = (int)qd - s_misc[ax] + half;
                d = d < 0 ? 0 : (d > LUTN - 1 ? LUTN - 1 : d);  // out of range only for entries no valid pair uses
                code = d;
              }
              s_cell[(aj * RM + nrows) * tts + k] = (unsigned char)code;
            }
            if (lane == 0) s_rowmap[nn] = (unsigned char)nrows;
            const unsigned long long same = __ballot(inp && n == nn);
            if (n == nn) r = nrows;
            nrows++;
            miss &= ~same;
            wave_sync();
          }
          const unsigned long long ready = pend & ~miss;
          const bool rdy = (ready >> lane) & 1ull;
          {
            int ptr[D];
#pragma unroll
            for (int i = 0; i < D; i++) ptr[i] = (en[i] * RM + r) * tts;
            const int cl = rdy ? cntl : 0;
            bool done = !rdy;
            for (int k0 = 0; __ballot(!done) != 0ull; k0 += kUB) {
              unsigned int w[kUB], word[kUB];
              bool out[kUB];
#pragma unroll
              for (int q = 0; q < kUB; q++) {
                int k = k0 + q;
                k = k < cl ? k : (cl > 0 ? cl - 1 : 0);
                unsigned int any = 0, sum = 0;
#pragma unroll
                for (int i = 0; i < D; i++) {
                  const unsigned int e = done ? 0u : (unsigned)s_cell[ptr[i] + k];
                  any |= e;
                  const unsigned int ec = e < (unsigned)LUTN ? e : (unsigned)(LUTN - 1);
                  sum += s_lut[i * LUTN + ec];
                }
                out[q] = (any & 0x80u) || sum >= kOut;
                w[q] = sum;
              }
#pragma unroll
              for (int q = 0; q < kUB; q++) word[q] = A.blk[(out[q] || done) ? 0u : (w[q] >> 5)];
#pragma unroll
              for (int q = 0; q < kUB; q++) {
                const bool blocked = out[q] || ((word[q] >> (w[q] & 31u)) & 1u);
                if (!done && k0 + q < cl && blocked) { fb = k0 + q; done = true; }
              }
              if (k0 + kUB >= cl) done = true;
            }
          }
          pend = miss;
        }
        // ---- cost (env_map.h:162-169) and iteration count
        if (act && !(A.dbg & 4)) {
          const bool blocked = fb >= 0;
          double J = 0;
#pragma unroll
          for (int i = 0; i < D; i++) J += s_eJ[en[i]];
          const double cost = blocked ? INFINITY : 0.0 + (J + A.w * A.dt);
          if (A.l_cost) A.l_cost[idx] = cost;
          if (A.l_iters) A.l_iters[idx] = blocked ? fb + 1 : cntl;
        }
        // ---- pop 64 entries
        drained += cnt_d;
        const unsigned int mv = (lane + 64 < qn) ? s_queue[lane + 64] : 0u;
        wave_sync();
        if (lane + 64 < qn) s_queue[lane] = mv;
        qn -= cnt_d;
      }
    }
    if (lane == 0 && A.l_count) A.l_count[node] = drained;
  }
}

// Blocked-bit map: 1 bit per cell, 1 = occupied (map == 100), outside the search
// region, or padding; bricks of 8x8x8 cells (2D: 32x16) = 16 dwords.
template <int D>
__global__ void build_blocked_bits_kernel(const int8_t *map, const uint32_t *region, int dim0, int dim1, int dim2,
                                          int nbx, int nby, int64_t n_dwords, uint32_t *out) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_dwords) return;
  const int64_t b = g >> 4;
  const int dw = (int)(g & 15);
  uint32_t bits = 0;
  if (D == 3) {
    const int bx = (int)(b % nbx), by = (int)((b / nbx) % nby), bz = (int)(b / ((int64_t)nbx * nby));
    const int z = bz * 8 + (dw >> 1);
    for (int bit = 0; bit < 32; bit++) {
      const int y = by * 8 + (dw & 1) * 4 + (bit >> 3), x = bx * 8 + (bit & 7);
      bool blocked = true;
      if (x < dim0 && y < dim1 && z < dim2) {
        const int64_t idx = x + (int64_t)dim0 * y + (int64_t)dim0 * dim1 * z;
        blocked = map[idx] == 100 || (region != nullptr && !((region[idx >> 5] >> (idx & 31)) & 1u));
      }
      bits |= (blocked ? 1u : 0u) << bit;
    }
  } else {
    const int bx = (int)(b % nbx), by = (int)(b / nbx);
    const int y = by * 16 + dw;
    for (int bit = 0; bit < 32; bit++) {
      const int x = bx * 32 + bit;
      bool blocked = true;
      if (x < dim0 && y < dim1) {
        const int64_t idx = x + (int64_t)dim0 * y;
        blocked = map[idx] == 100 || (region != nullptr && !((region[idx >> 5] >> (idx & 31)) & 1u));
      }
      bits |= (blocked ? 1u : 0u) << bit;
    }
  }
  out[g] = bits;
}

template <int D, int K>
hipError_t launch_grid_inst(const GridArgs &a, hipStream_t stream) {
  if (a.n_nodes == 0) return hipSuccess;
  const int64_t n_wg = (a.n_nodes + kWPB - 1) / kWPB;
  const int64_t blocks = n_wg < (int64_t)a.grid_limit ? n_wg : (int64_t)a.grid_limit;
  const size_t lds = grid_lds_bytes(D, a.nU, a.ndp, a.n_max, a.rmax);
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

size_t grid_lds_bytes(int dim, int nU, int ndp, int n_max, int rmax) {
  return (size_t)GridLds(dim, kWPB, nU, ndp, n_max, rmax).total;
}
int grid_waves_per_block() { return kWPB; }

void blocked_bits_geometry(int dim, const int32_t *mdim, int *nbx, int *nby, int64_t *n_dwords) {
  if (dim == 3) {
    *nbx = (mdim[0] + 7) / 8;
    *nby = (mdim[1] + 7) / 8;
    *n_dwords = (int64_t)(*nbx) * (*nby) * ((mdim[2] + 7) / 8) * 16;
  } else {
    *nbx = (mdim[0] + 31) / 32;
    *nby = (mdim[1] + 15) / 16;
    *n_dwords = (int64_t)(*nbx) * (*nby) * 16;
  }
}

hipError_t launch_build_blocked_bits(int dim, const int8_t *map, const uint32_t *region, const int32_t *mdim,
                                     uint32_t *out, hipStream_t stream) {
  int nbx, nby;
  int64_t n;
  blocked_bits_geometry(dim, mdim, &nbx, &nby, &n);
  const unsigned blocks = (unsigned)((n + 255) / 256);
  if (dim == 3)
    hipLaunchKernelGGL(build_blocked_bits_kernel<3>, dim3(blocks), dim3(256), 0, stream, map, region, mdim[0], mdim[1],
                       mdim[2], nbx, nby, n, out);
  else
    hipLaunchKernelGGL(build_blocked_bits_kernel<2>, dim3(blocks), dim3(256), 0, stream, map, region, mdim[0], mdim[1],
                       1, nbx, nby, n, out);
  return hipGetLastError();
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
