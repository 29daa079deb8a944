// expand_lex_kernel.hip -- the factorised successor expansion for the control tables the reference's own programs
// build: the NESTED-LOOP (lexicographic) enumeration of per-axis values (test/test_planner_2d.cpp:49-53, every planner
// test), without yaw, on an occupancy map.  Same function, same results as expand_grid_kernel.hip
//   MPL::env_map<Dim>::get_succ, reference include/mpl_planner/env/env_map.h:147-172 with traverse_primitive :90-132
// and the same organisation (one WAVEFRONT owns one node; per-axis "entries" in LDS; rows of cell codes; the reach box
// staged as bits; see that file's header) -- rebuilt in round 4 around what the round-3 counters said about it: the
// kernel is VALU-issue bound wherever it is not bound by its own stores, a sixth of its static VALU instructions were
// SGPR spills (`v_readlane / v_writelane`, 4 cycles each), and a third of a C4 node's instructions enumerated pairs
// twice (phase A: validity + hash + ordered compaction into a list; phase D: the list, 64 at a time, hash again).
//
//  * No pair phase, no list.  With a lexicographic table the emitted successors of a node are, in ascending control
//    order, exactly the combinations of the entries that pass the limits (`valid lists` per axis) minus the ones whose
//    lattice hash equals the node's own (env_map.h:158 -- at most the "stay where you are" control, and only for a node
//    at rest).  Phase D therefore enumerates the combinations directly, 64 per step; a step's dropped lanes (ballot)
//    shift the list positions of everything behind them, their masks are kept per step for the later passes of a node
//    that needs several.  The set of sample counts in use -- the maxima over the product of the per-axis sets -- is the
//    union of the per-axis sets from the largest per-axis minimum upwards: a handful of scalar instructions.
//  * Entries sit at `axis * 16 + value index`: one DPP row per axis.  Per-axis reductions (the reach box of the free-box
//    query, the sets of sample counts, the valid lists) are ROW reductions or 16-bit fields of one ballot, all axes at
//    once; no division by the table stride anywhere.
//  * Every small per-wave table lives at a COMPILE-TIME offset from one base (sized for 16 values per axis), so an LDS
//    access is one address instruction + an immediate; the old layout (sized at run time from nU, ndp, n_max ...) kept
//    fifteen base pointers alive in SGPRs across the node loop.  No yaw, no potential map, no direct-evaluation path
//    (only SNP needs one), no gather mode: what is left fits the scalar register file.
//  * Sample loop: the staged word of a box row is ROTATED so that the bit of cell code ex is bit (ex & 31): a sample is
//    three byte look-ups, one multiply-add, one address add, one word look-up, `v_bfe_u32`, `v_lshl_or_b32`.
//
// Everything that decides a RESULT is the arithmetic of expand_grid_kernel.hip, expression for expression
// (mplx_device_common.h: Ax<K>, div_by, quantise, fold; -ffp-contract=off): the accumulated sample times, the
// half-away rounding of the cell, the `0.0 +` of the emitted state, J per axis in order.  tests/test_gpu_fullsize.py
// compares every pair of C2, C3 and C4 with the reference build through this kernel (route "grid", kernel "lex"),
// tests/test_gpu_lex.py the two kernels with each other on odd worlds.
//
// Scope: Dim 2/3, VEL / ACC / JRK, lexicographic control table with <= 32 values per axis (instantiations with tables
// of 8 / 16 / 32 entries per axis) and <= 8192 controls, v_max > 0 (or VEL), n_max <= 61, occupancy map (with or
// without a search region).  Everything else is expand_grid_kernel.hip's.
#include "mplx_internal.h"
#include "mplx_device_common.h"

namespace mplx {

using namespace dev;

constexpr int kLexWPB = 4;  // waves (= nodes in flight) per workgroup
constexpr int kLexBT = 64 * kLexWPB;
constexpr int kLexTabStride = 64;  // row stride of the global time table (launch_make_tables)
constexpr int kLexUB8 = 8;         // samples per step of the sample loop (4 for JRK: see the kernel)

// shared tables at compile-time offsets (given the table stride TS)
constexpr int kShUval = 0;                                                             // double[3][TS]
__host__ __device__ constexpr int lex_sh_tc(int TS) { return TS == 32 ? 768 : 384; }   // uchar[64]
__host__ __device__ constexpr int lex_sh_tt(int TS) { return lex_sh_tc(TS) + 64; }     // double[(n_max + 1)(n_max + 2) / 2]: row nn at nn (nn + 1) / 2
__host__ __device__ constexpr int lex_ts(int ndp) { return ndp <= 8 ? 8 : (ndp <= 16 ? 16 : 32); }

// bytes of the fixed-size per-wave tables = offset of the first run-time sized one (the prefix-hash table)
// (TS = table stride: entries reserved per axis, 8, 16 or 32 -- the LANE layout is one 16-lane DPP row per axis for 8 and
// 16, two rows per axis for 32, where a 3D node's 96 entries take two rounds of phase T1)
__host__ __device__ constexpr int lex_fixed_bytes(int D, int K, int TS) {
  const int F = 4 * D + 2, KQ = K == 3 ? 4 : K;
  int w = (F * 8 + 15) & ~15;                                                                                  // node
  w += 128 + D * TS * 4 + D * TS * 4 * KQ + D * TS * 8 * (K - 1) + (K == 3 ? D * TS * 8 : 0) + D * TS + D * TS;  // misc .. vlc
  w = (w + 15) & ~15;
  return (w + 128 + 7) & ~7;                                                                                   // rowmap
}
// the accumulated sample times of count nn (tc[nn] <= nn + 1 of them) start at entry nn (nn + 1) / 2 of the shared table
__host__ __device__ constexpr int lex_tt_entries(int n_max) { return (n_max + 1) * (n_max + 2) / 2; }

// per-wave tables at compile-time offsets from the wave's block
template <int D, int K, int TS>
struct LexW {
  static constexpr int F = 4 * D + 2;
  static constexpr int KQ = K == 3 ? 4 : K;
  static constexpr int NODE = 0;                                   // double[F]
  static constexpr int MISC = (F * 8 + 15) & ~15;                  // int[32]
  static constexpr int EFLAG = MISC + 128;                         // int[D][TS]
  static constexpr int EQ = EFLAG + D * TS * 4;                    // int[D][TS][KQ]
  static constexpr int EST = EQ + D * TS * 4 * KQ;                 // double[D][TS][K - 1]
  static constexpr int UQ = EST + D * TS * 8 * (K - 1);            // double[D][TS] (K = 3)
  static constexpr int VL = UQ + (K == 3 ? D * TS * 8 : 0);        // uchar[D][TS]: values inside the limits, in order
  static constexpr int VLC = VL + D * TS;                          // uchar[D][TS]: ... whose row the current count needs
  static constexpr int ROWMAP = (VLC + D * TS + 15) & ~15;         // ushort[64]
  static constexpr int HP = (ROWMAP + 128 + 7) & ~7;               // uint64[PN]; then (run-time sized) the dropped lanes
                                                                   // per step of phase D, the cell rows, the box
  static_assert(HP == lex_fixed_bytes(D, K, TS), "LexLds sizes the workgroup's LDS from lex_fixed_bytes");
};
enum { LM_BASE = 0, LM_NODEQ = 4 };  // misc words: cell of the node per axis [3]; lattice integers of the node [D][4]

// run-time part of the carve-up, shared by host (size) and device (offsets)
struct LexLds {
  int tts, rowcap, o_wave0, w_drop, w_cell, w_box, wave_bytes, total;
  __host__ __device__ LexLds(int D, int K, int waves, int ndp, int nU, int n_max, int rmax, int boxcap) {
    tts = n_max + 1;
    rowcap = rmax * tts;
    o_wave0 = (lex_sh_tt(lex_ts(ndp)) + lex_tt_entries(n_max) * 8 + 15) & ~15;
    int w = lex_fixed_bytes(D, K, lex_ts(ndp));
    const int PN = (D == 3) ? ndp * ndp : ndp;
    w += PN * 8;
    w_drop = w;
    w += ((nU + 63) >> 6) * 8;  // uint64 per step of phase D: the lanes whose successor equals the node (dropped)
    w_cell = w;
    w += D * ndp * rowcap + 8;  // + 8: the sample loop reads up to 7 codes past a row
    w = (w + 3) & ~3;
    w_box = w;
    w += boxcap * 4;
    wave_bytes = (w + 15) & ~15;
    total = o_wave0 + waves * wave_bytes;
  }
};

namespace {

template <typename T>
__device__ __forceinline__ void lex_st(T v, T *p) {
  if constexpr (sizeof(T) == 8) asm volatile("global_store_dwordx2 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dword %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
}

__device__ __forceinline__ void lex_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void lex_prio(int p) {
  if (p == 0) __builtin_amdgcn_s_setprio(0);
  else if (p == 1) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(3);
}

// min / max / or over each ROW of 16 lanes (all lanes active): the result sits in lane 15 of the row
template <bool MAX>
__device__ __forceinline__ int row_reduce_minmax(int v) {
  const int id = MAX ? (int)0x80000000 : 0x7fffffff;
  auto op = [](int a, int b) { return MAX ? (a > b ? a : b) : (a < b ? a : b); };
  int x = v;
  x = op(x, __builtin_amdgcn_update_dpp(id, v, 0x111, 0xf, 0xf, false));  // row_shr:1
  x = op(x, __builtin_amdgcn_update_dpp(id, v, 0x112, 0xf, 0xf, false));  // row_shr:2
  x = op(x, __builtin_amdgcn_update_dpp(id, v, 0x113, 0xf, 0xf, false));  // row_shr:3
  x = op(x, __builtin_amdgcn_update_dpp(id, x, 0x114, 0xf, 0xe, false));  // row_shr:4, banks 1-3
  x = op(x, __builtin_amdgcn_update_dpp(id, x, 0x118, 0xf, 0xc, false));  // row_shr:8, banks 2-3
  return x;
}
__device__ __forceinline__ unsigned int row_reduce_or(unsigned int v) {
  int x = (int)v;
  x |= __builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
  x |= __builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
  x |= __builtin_amdgcn_update_dpp(0, (int)v, 0x113, 0xf, 0xf, false);
  x |= __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xe, false);
  x |= __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xc, false);
  return (unsigned int)x;
}

template <int K>
__device__ __forceinline__ void lex_fold_entry(uint64_t &h, const int *eq, int e) {
  if (K == 1) {
    fold(h, eq[e]);
  } else if (K == 2) {
    const int2 q = *(const int2 *)(eq + e * 2);
    fold(h, q.x);
    fold(h, q.y);
  } else {
    const int4 q = *(const int4 *)(eq + e * 4);
    fold(h, q.x);
    fold(h, q.y);
    fold(h, q.z);
  }
}

template <int D, int K, int TS>
__global__ __launch_bounds__(kLexBT) void expand_lex_kernel(const GridArgs A_kernarg) {
  extern __shared__ __align__(16) unsigned char smem[];
  // the argument block is read where it lies (scalar loads next to their uses), see expand_grid_kernel.hip
  typedef const GridArgs __attribute__((address_space(4))) *KernargPtr;
  KernargPtr Ak = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
  (void)A_kernarg;
#define A (*Ak)
  typedef LexW<D, K, TS> W;
  constexpr int F = W::F, KQ = W::KQ;
  constexpr int UB = (D == 3 && K == 3) ? 4 : kLexUB8;
  const int ndp = A.ndp, RM = A.rmax;
  const LexLds L(D, K, kLexWPB, ndp, A.nU, A.n_max, RM, A.boxcap);
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const double *s_uval = (const double *)(smem + kShUval);
  constexpr int kShTc = lex_sh_tc(TS), kShTt = lex_sh_tt(TS);
  const unsigned char *s_tc = smem + kShTc;
  const double *s_tt = (const double *)(smem + kShTt);
  unsigned char *wb = smem + L.o_wave0 + wv * L.wave_bytes;
  double *s_node = (double *)(wb + W::NODE);
  int *s_misc = (int *)(wb + W::MISC);
  int *s_eflag = (int *)(wb + W::EFLAG);
  int *s_eq = (int *)(wb + W::EQ);
  double *s_est = (double *)(wb + W::EST);
  double *s_uq = (double *)(wb + W::UQ);
  unsigned char *s_vl = wb + W::VL;
  unsigned char *s_vlc = wb + W::VLC;
  unsigned short *s_rowmap = (unsigned short *)(wb + W::ROWMAP);
  uint64_t *s_drop = (uint64_t *)(wb + L.w_drop);
  uint64_t *s_hp = (uint64_t *)(wb + W::HP);
  unsigned char *s_cell = wb + L.w_cell;
  unsigned int *s_box = (unsigned int *)(wb + L.w_box);

  const int tts = L.tts, rowcap = L.rowcap;
  const int half = A.n_max + 2;  // cell-offset code = offset from the node's cell + half
  const double T = A.dt;
  const int dims[3] = {A.dim0, A.dim1, A.dim2};
  const int nd[3] = {A.nd0, A.nd1, A.nd2};

  // ---- node assignment: static striding, or chunks claimed from 64 counters (GridArgs::work); see expand_grid_kernel.hip
  const int wave_id = (int)blockIdx.x * kLexWPB + wv;
  const int wave_stride = (int)gridDim.x * kLexWPB;
  const int NN = (int)A.n_nodes;
  const bool dyn = A.work != nullptr;
  const int ck = dyn ? A.work_chunk : 1;
  int dyn_beg = 0, dyn_len = 0, dyn_step = 1;
  unsigned int *ctr = nullptr;
  if (dyn) {
    if (blockIdx.x == 0 && threadIdx.x < kWorkCounters) A.work_zero[threadIdx.x * 32] = 0u;  // for the next launch
    const int n_chunks = (NN + ck - 1) / ck, n_dyn = n_chunks > wave_stride ? n_chunks - wave_stride : 0;
    const int nc = gridDim.x < (unsigned)kWorkCounters ? (int)gridDim.x : kWorkCounters;
    const int cx = (int)(blockIdx.x % nc);
    const int base = n_dyn / nc, rem = n_dyn % nc;
    dyn_len = base + (cx < rem ? 1 : 0);
    if (A.work_blocked) { dyn_beg = wave_stride + cx * base + (cx < rem ? cx : rem); dyn_step = 1; }
    else { dyn_beg = wave_stride + cx; dyn_step = nc; }
    ctr = A.work + cx * 32;
  }
  auto claim = [&]() -> int {
    unsigned int v = 0;
    if (lane == 0) v = atomicAdd(ctr, 1u);
    const int j = __builtin_amdgcn_readfirstlane((int)v);
    return (j >= 0 && j < dyn_len) ? (dyn_beg + j * dyn_step) * ck : NN;
  };
  const int it0 = wave_id * ck;
  int chunk_end = it0 + ck < NN ? it0 + ck : NN;
  double nxt = 0.0;  // lanes < F: one field of the next node (prefetched)
  if (it0 < NN && lane < F) nxt = A.nodes[(int64_t)lane * A.node_stride + it0];
  int next_chunk = NN;
  if (dyn && it0 < NN) next_chunk = claim();

  // ---- once per (persistent) workgroup: shared read-only tables
  {
    double *uv = (double *)(smem + kShUval);
    if (threadIdx.x < D * TS) uv[threadIdx.x] = A.uvals[(threadIdx.x / TS) * A.uval_stride + threadIdx.x % TS];
    if (threadIdx.x < 64) smem[kShTc + threadIdx.x] = A.tcnt[threadIdx.x];
    double *tt = (double *)(smem + kShTt);
    for (int i = threadIdx.x; i < (A.n_max + 1) * tts; i += kLexBT) {
      const int nn = i / tts, k = i - nn * tts;
      if (k <= nn) tt[((nn * (nn + 1)) >> 1) + k] = A.ttab[nn * kLexTabStride + k];  // (tc[nn] <= nn + 1)
    }
  }
  __syncthreads();  // the only workgroup barrier
  asm volatile("" ::"v"(nxt));

  // Lane layout of the axis entries: LS lanes per axis (one DPP row of 16, or two for tables of up to 32 values per axis);
  // entry e = lane + 64 r of round r is (axis e / LS, value index e % LS).  NR = 1 except for Dim 3 with 32 values per axis.
  constexpr int LS = TS == 32 ? 32 : 16;
  constexpr int NR = (D * LS + 63) / 64;
  const int jv_l = lane & (LS - 1);
  const unsigned int below_l = (1u << jv_l) - 1u;
  int ax_r[NR], ti_r[NR];
  bool ent_r[NR];
#pragma unroll
  for (int r = 0; r < NR; r++) {
    ax_r[r] = (lane + 64 * r) / LS;
    ent_r[r] = ax_r[r] < D && jv_l < (ax_r[r] == 0 ? nd[0] : (ax_r[r] == 1 ? nd[1] : nd[2]));
    ti_r[r] = ax_r[r] * TS + jv_l;  // the entry's place in the tables
  }
  // the field of axis i in the ballots of the rounds (LS = 16: 16-bit fields of one ballot; LS = 32: halves of two)
  auto amask = [&](const unsigned long long (&b)[NR], int i) -> unsigned int {
    if (LS == 16) return (unsigned int)(b[0] >> (16 * i)) & 0xffffu;
    return (unsigned int)(b[i >> 1] >> (32 * (i & 1)));
  };
  // ... and of this lane's own axis in round r
  auto amask_lane = [&](const unsigned long long (&b)[NR], int r) -> unsigned int {
    const unsigned int lo = (unsigned int)b[r], hi = (unsigned int)(b[r] >> 32);
    if (LS == 16) return ((lane < 16) ? lo : (lane < 32) ? (lo >> 16) : (lane < 48) ? hi : (hi >> 16)) & 0xffffu;
    return lane < 32 ? lo : hi;
  };
  // the per-axis result of a row reduction done in every round (lane 15 of each 16-lane row holds its row's)
  auto axis_or = [&](const unsigned int (&x)[NR], int i) -> unsigned int {
    if (LS == 16) return (unsigned int)__builtin_amdgcn_readlane((int)x[0], 16 * i + 15);
    return (unsigned int)__builtin_amdgcn_readlane((int)x[i >> 1], 32 * (i & 1) + 15) |
           (unsigned int)__builtin_amdgcn_readlane((int)x[i >> 1], 32 * (i & 1) + 31);
  };
  auto axis_min = [&](const int (&x)[NR], int i) -> int {
    if (LS == 16) return __builtin_amdgcn_readlane(x[0], 16 * i + 15);
    const int a = __builtin_amdgcn_readlane(x[i >> 1], 32 * (i & 1) + 15), b = __builtin_amdgcn_readlane(x[i >> 1], 32 * (i & 1) + 31);
    return a < b ? a : b;
  };
  auto axis_max = [&](const int (&x)[NR], int i) -> int {
    if (LS == 16) return __builtin_amdgcn_readlane(x[0], 16 * i + 15);
    const int a = __builtin_amdgcn_readlane(x[i >> 1], 32 * (i & 1) + 15), b = __builtin_amdgcn_readlane(x[i >> 1], 32 * (i & 1) + 31);
    return a > b ? a : b;
  };
  int it_next = 0;

  for (int it = it0; it < NN; it = it_next) {
    asm volatile("" : "+s"(Ak));
    if (!dyn) {
      it_next = it + wave_stride;
    } else if (it + 1 < chunk_end) {
      it_next = it + 1;
    } else {
      it_next = next_chunk;
      chunk_end = it_next + ck < NN ? it_next + ck : NN;
      next_chunk = it_next < NN ? claim() : NN;
    }
    const int64_t node = it;
    // ---- phase 0: node state into LDS, prefetch of the next node
    lex_prio(0);
    lex_sync();
    if (lane < F) s_node[lane] = nxt;
    if (it_next < NN && lane < F) nxt = A.nodes[(int64_t)lane * A.node_stride + it_next];
    lex_sync();

    // ---- phase T1: axis entries (lane layout above); the node's own lattice integers (lanes 48 .. where they are free)
    int flag[NR], rb_lo[NR], rb_hi[NR];  // rb: cells this entry's p(t) spans (free-box query)
#pragma unroll
    for (int r = 0; r < NR; r++) {
    flag[r] = 0;
    rb_lo[r] = 0x7fffffff;
    rb_hi[r] = (int)0x80000000;
    const int ti_l = ti_r[r];
    if (ent_r[r]) {
      const int ax = ax_r[r];
      const double p = s_node[0 * D + ax];
      const double v = (K >= 2) ? s_node[1 * D + ax] : 0.0;
      const double a = (K >= 3) ? s_node[2 * D + ax] : 0.0;
      const double u = s_uval[ti_l];
      const double org = ax == 0 ? A.org0 : (ax == 1 ? A.org1 : A.org2);
      Ax<K> q;
      q.init(p, v, a, 0.0, u);
      const double mv = q.max_vel(T);
      bool valid = true;
      if (K >= 2 && A.v_max > 0) valid = valid && !(mv > A.v_max);
      if (K >= 3 && A.a_max > 0) valid = valid && !(q.max_acc(T) > A.a_max);
      // env_map.h:95, one axis' share of n = max(5, (int)ceil(max_v * T / res))
      int n = (int)ceil(div_by(mv * T, A.res, A.Rres));
      n = n < 5 ? 5 : (n > A.n_max ? A.n_max : n);
      const double np_ = q.template pos<true>(T);
      const double nv_ = q.template vel<true>(T);
      const double na_ = q.template acc<true>(T);
      // fields of order < K - 1; order K - 1 is (0.0 + u*T) + x0, order K is 0.0 + u, higher ones are 0
      if (K >= 3) s_uq[ti_l] = q.top_quotient();
      if (K >= 2) s_est[ti_l * (K - 1) + 0] = np_;
      if (K >= 3) s_est[ti_l * (K - 1) + 1] = nv_;
      s_eq[ti_l * KQ + 0] = quantise(np_, 0.01, A.R001);
      if (K >= 2) s_eq[ti_l * KQ + 1] = quantise(nv_, 0.1, A.R01);
      if (K >= 3) s_eq[ti_l * KQ + 2] = quantise(na_, 0.1, A.R01);
      flag[r] = (valid ? 1 : 0) | ((p == np_) ? 2 : 0) | (n << 8);
      if (A.sat != nullptr && valid) {
        // range of p(t) over [0, T] of this entry, as cells with one cell of slack on both sides;
        // K = 1, 2: exact extrema; K = 3: |p - p0| <= max_vel * T
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
        rb_lo[r] = (int)floor(div_by(pmin - org, A.res, A.Rres)) - 1;
        rb_hi[r] = (int)floor(div_by(pmax - org, A.res, A.Rres)) + 1;
      }
      if (jv_l == 0) {
        // the node's own cell on this axis (map_util.h:103-108); -1 stands for every negative cell
        const double qd = div_by(p - org, A.res, A.Rres);
        s_misc[LM_BASE + ax] = (qd - 0.5 > -0.5) ? (int)qd : -1;
      }
    } else if (LS == 16 && r == 0 && lane >= 48 && lane < 48 + 4 * D) {
      const int i = (lane - 48) >> 2, f = (lane - 48) & 3;
      if (f < K) {
        const double x = s_node[f * D + i];
        s_misc[LM_NODEQ + i * 4 + f] = f == 0 ? quantise(x, 0.01, A.R001) : quantise(x, 0.1, A.R01);
      }
    }
    if (ax_r[r] < D && jv_l < TS) s_eflag[ti_l] = flag[r];
    }  // rounds of T1
    if (LS == 32 && lane < 4 * D) {  // (no lanes to spare beside the entries: the node's lattice integers in a step of their own)
      const int i = lane >> 2, f = lane & 3;
      if (f < K) {
        const double x = s_node[f * D + i];
        s_misc[LM_NODEQ + i * 4 + f] = f == 0 ? quantise(x, 0.01, A.R001) : quantise(x, 0.1, A.R01);
      }
    }
    // per axis, the values that pass the limits, in order: fields of the rounds' ballots
    unsigned long long vm[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) vm[r] = __ballot((flag[r] & 1) != 0);
    const int nv0 = __popc(amask(vm, 0)), nv1 = __popc(amask(vm, 1)), nv2 = (D == 3) ? __popc(amask(vm, 2)) : 1;
#pragma unroll
    for (int r = 0; r < NR; r++)
      if (flag[r] & 1) s_vl[ax_r[r] * TS + __popc(amask_lane(vm, r) & below_l)] = (unsigned char)jv_l;
    // the sets of sample counts per axis (bit n of a 64-bit word, as two halves) by row reduction
    unsigned long long nm = 0;
    {
      unsigned int r_lo[NR], r_hi[NR];
#pragma unroll
      for (int r = 0; r < NR; r++) {
        const int n_l = flag[r] >> 8;
        const unsigned int b_lo = ((flag[r] & 1) && n_l < 32) ? (1u << n_l) : 0u, b_hi = ((flag[r] & 1) && n_l >= 32) ? (1u << (n_l - 32)) : 0u;
        r_lo[r] = row_reduce_or(b_lo);
        r_hi[r] = row_reduce_or(b_hi);
      }
      unsigned long long uni = 0;
      int lmin = 0;
#pragma unroll
      for (int i = 0; i < D; i++) {
        const unsigned long long mi = (unsigned long long)axis_or(r_lo, i) | ((unsigned long long)axis_or(r_hi, i) << 32);
        uni |= mi;
        const int lo_i = mi ? __ffsll((long long)mi) - 1 : 64;
        lmin = lo_i > lmin ? lo_i : lmin;
      }
      // maxima over the product of the per-axis sets = the union from the largest per-axis minimum upwards
      nm = lmin < 64 ? (uni >> lmin) << lmin : 0ull;
    }
    lex_sync();

    // ---- prefix table over the first D-1 axes, valid combinations only
    const int nA = nv0 * nv1 * nv2;  // combinations of entries inside the limits = pairs to enumerate
    if (D == 3) {
      const int n01 = nv0 * nv1;
      const float r1 = __builtin_amdgcn_rcpf((float)(nv1 > 0 ? nv1 : 1));
      for (int x = lane; x < n01; x += 64) {
        const int a_ = (int)(((float)x + 0.5f) * r1);
        const int b_ = x - a_ * nv1;
        const int j0 = s_vl[a_], j1 = s_vl[TS + b_];  // (exact quotient: x < 2^10)
        uint64_t h = 0;
        lex_fold_entry<K>(h, s_eq, j0);
        lex_fold_entry<K>(h, s_eq, TS + j1);
        s_hp[__umul24(j0, ndp) + j1] = h;
      }
    } else {
      if (lane < nv0) {
        const int j0 = s_vl[lane];
        uint64_t h = 0;
        lex_fold_entry<K>(h, s_eq, j0);
        s_hp[j0] = h;
      }
    }
    // hash of the node (uniform: scalar arithmetic)
    uint64_t hcur = 0;
#pragma unroll
    for (int i = 0; i < D; i++) {
      const int4 q = *(const int4 *)(s_misc + LM_NODEQ + i * 4);
      fold(hcur, __builtin_amdgcn_readfirstlane(q.x));
      if (K >= 2) fold(hcur, __builtin_amdgcn_readfirstlane(q.y));
      if (K >= 3) fold(hcur, __builtin_amdgcn_readfirstlane(q.z));
    }
    int base_c[3];
#pragma unroll
    for (int i = 0; i < 3; i++) base_c[i] = (i < D) ? __builtin_amdgcn_readfirstlane(s_misc[LM_BASE + i]) : 0;
    // ---- free-box shortcut (summed-area table of the blocked-bit map): is the whole box the node can reach in T free?
    unsigned int sat_v = 0;
    bool sat_inside = false;
    // Dim 2: the blocked bits of the node's REACH box (one row of <= 32 cells per lane, known here) are requested together
    // with the free-box query and land in LDS when the rows of cell codes exist -- a frontier of one node per wave (C2,
    // the batches of a 2D search) is bound by its dependent round trips to memory: node -> query -> box; this takes the
    // third off the chain (and the four wave reductions of the sampled box with it).  Every sample of an entry inside the
    // limits lies in that box (exact extrema of p(t), one cell of slack; K <= 2: the query's precondition).
    bool eager = false;
    int e_lo[2] = {0, 0}, e_nb[2] = {1, 1};
    unsigned int e_a0 = 0, e_a1 = 0, e_shf = 0, e_mask = 0;
    bool e_in = false;
    if (A.sat != nullptr) {
      int r_lo[NR], r_hi[NR];
#pragma unroll
      for (int r = 0; r < NR; r++) {
        r_lo[r] = row_reduce_minmax<false>(rb_lo[r]);
        r_hi[r] = row_reduce_minmax<true>(rb_hi[r]);
      }
      int rlo[3] = {0, 0, 0}, rhi[3] = {0, 0, 0};
      bool inside = true;
#pragma unroll
      for (int i = 0; i < D; i++) {
        rlo[i] = axis_min(r_lo, i);
        rhi[i] = axis_max(r_hi, i);
        inside = inside && rhi[i] >= rlo[i] && rlo[i] >= 0 && rhi[i] < dims[i];
      }
      sat_inside = inside;
      if (D == 2 && base_c[0] >= 0 && base_c[1] >= 0 && rhi[0] >= rlo[0] && rhi[1] >= rlo[1]) {
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 2; i++) {
          e_lo[i] = rlo[i] - base_c[i] + half;
          const int hi = rhi[i] - base_c[i] + half;
          ok = ok && e_lo[i] >= 0 && hi <= 255;
          e_nb[i] = hi - e_lo[i] + 1;
        }
        eager = ok && e_nb[0] <= 32 && e_nb[1] <= 64 && e_nb[1] <= A.boxcap;
        if (eager) {
          const int xw = rlo[0], ay = rlo[1] + lane;
          const int vlo = xw < 0 ? -xw : 0;
          const int vhi = (dims[0] - xw) < 32 ? (dims[0] - xw) : 32;
          e_mask = (vhi >= 32 ? 0xffffffffu : ((1u << (vhi > 0 ? vhi : 0)) - 1u)) & ~((1u << vlo) - 1u);
          e_in = lane < e_nb[1] && vhi > vlo && ay >= 0 && ay < dims[1];
          const int64_t off = e_in ? (int64_t)ay * (int64_t)dims[0] + xw : 0;
          const int64_t wi = off >> 5;
          e_shf = (unsigned)(off & 31);
          const int64_t w0 = wi < 0 ? 0 : wi;
          const int64_t w1 = wi + 1 >= A.blk_words ? A.blk_words - 1 : wi + 1;
          e_a0 = A.blk[w0];
          e_a1 = A.blk[w1 < 0 ? 0 : w1];
        }
      }
      if (inside && lane < (1 << D)) {
        const int cx = (lane & 1) ? rhi[0] + 1 : rlo[0];
        const int cy = (lane & 2) ? rhi[1] + 1 : rlo[1];
        const int cz = (D == 3) ? ((lane & 4) ? rhi[2] + 1 : rlo[2]) : 1;
        const int64_t idx = cx + (int64_t)(dims[0] + 1) * (cy + (int64_t)(dims[1] + 1) * cz);
        sat_v = A.sat[idx];
      }
    }
    const double node_t = s_node[4 * D + 1];
    if (A.dbg & 1) nm = 0;  // timing ablation: no sampling
    asm volatile("" ::"v"(nxt));  // the next node's state has arrived: no wait behind this node's stores
    bool safe = false;
    if (A.sat != nullptr) {
      unsigned int term = (lane < (1 << D)) ? sat_v : 0u;
      if ((D - __popc((unsigned)lane & ((1u << D) - 1u))) & 1) term = 0u - term;
#pragma unroll
      for (int d = 1; d < (1 << D); d <<= 1) term += (unsigned int)__shfl_xor((int)term, d, 64);
      safe = sat_inside && __builtin_amdgcn_readfirstlane((int)term) == 0;
    }
    lex_sync();  // prefix table complete

    // enumeration constants: x -> (a, b, c) over (nv0, nv1, nv2), c fastest
    const int in1 = (D == 3) ? nv2 : 1, in0 = nv1 * in1;
    const float r_in0 = __builtin_amdgcn_rcpf((float)(in0 > 0 ? in0 : 1)), r_in1 = __builtin_amdgcn_rcpf((float)(in1 > 0 ? in1 : 1));
    int E = nA;  // successors of the node: nA minus the dropped ones (known after pass 0)

    lex_prio(1);
    // ---- rounds of up to RM sample counts
    for (int pass = 0; pass == 0 || nm != 0ull; pass++) {
      unsigned long long sub = 0;
      if (safe) {
        sub = ~0ull;  // nothing to sample: every successor is handled in this one pass
      } else {
        int used = 0;
        for (unsigned long long t = nm; t; t &= t - 1ull) {
          const int nn = __ffsll((long long)t) - 1;
          const int cn = (int)s_tc[nn];
          if (used + cn > rowcap && sub) break;
          sub |= 1ull << nn;
          used += cn;
        }
      }
      nm &= ~sub;
      lex_sync();
      {
        int off = 0;
        unsigned short mine_off = 0xffff;
        for (unsigned long long t = safe ? 0ull : sub; t; t &= t - 1ull) {
          const int nn = __ffsll((long long)t) - 1;
          if (nn == lane) mine_off = (unsigned short)off;
          off += (int)s_tc[nn];
        }
        if (lane <= A.n_max) s_rowmap[lane] = mine_off;
      }
      // rows: cell-offset codes of every needed axis entry at t_0 .. t_{cnt-1} of each sample count
      int lo_l[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi_l[3] = {-1, -1, -1};
      {
        int row = 0;
        for (unsigned long long t = safe ? 0ull : sub; t; t &= t - 1ull) {
          const int nn = __ffsll((long long)t) - 1;
          const int cn = (int)s_tc[nn];
          const float inv_cn = __builtin_amdgcn_rcpf((float)cn);
          const double *trow = s_tt + ((nn * (nn + 1)) >> 1);
          // the row of (entry, count nn) is only read by pairs whose count IS nn, i.e. by entries with n_entry <= nn
          unsigned long long fm[NR];
#pragma unroll
          for (int r = 0; r < NR; r++) fm[r] = __ballot((flag[r] & 1) && (flag[r] >> 8) <= nn);
          lex_sync();  // (the previous count's list has been read)
#pragma unroll
          for (int r = 0; r < NR; r++)
            if ((fm[r] >> lane) & 1ull) s_vlc[ax_r[r] * TS + __popc(amask_lane(fm, r) & below_l)] = (unsigned char)jv_l;
          lex_sync();
#pragma unroll
          for (int ax = 0; ax < D; ax++) {
            const double p0 = s_node[0 * D + ax];
            const double v0 = (K >= 2) ? s_node[1 * D + ax] : 0.0;
            const double a0 = (K >= 3) ? s_node[2 * D + ax] : 0.0;
            const double org = ax == 0 ? A.org0 : (ax == 1 ? A.org1 : A.org2);
            const int shift = half - base_c[ax];
            const int nv = __popc(amask(fm, ax));
            for (int x = lane; x < nv * cn; x += 64) {
              const int vi = (int)(((float)x + 0.5f) * inv_cn);  // exact: x < 2^12
              const int k = x - __umul24(vi, cn);
              const int jv = (int)s_vlc[ax * TS + vi];
              Ax<K> q;
              q.init(p0, v0, a0, 0.0, s_uval[ax * TS + jv]);
              // map_util.h:103-108: cell = round((pos - origin) / res - 0.5)
              const double qd = div_by(q.pos_q(trow[k], K >= 3 ? s_uq[ax * TS + jv] : 0.0) - org, A.res, A.Rres);
              const int c = (qd - 0.5 > -0.5) ? (int)qd : -1;
              const int code = c + shift;  // 0 < code < 2 * half for an entry inside the limits (exact maxima for K <= 3)
              s_cell[__umul24(ax * ndp + jv, rowcap) + row + k] = (unsigned char)code;
              lo_l[ax] = code < lo_l[ax] ? code : lo_l[ax];
              hi_l[ax] = code > hi_l[ax] ? code : hi_l[ax];
            }
          }
          row += cn;
        }
      }
      // the box of codes the valid entries reach
      int lo[3] = {0, 0, 0}, nb[3] = {1, 1, 1};
      bool have_box = !safe && sub != 0ull;
      if (D == 2 && eager) {
        lo[0] = e_lo[0]; lo[1] = e_lo[1];
        nb[0] = e_nb[0]; nb[1] = e_nb[1];
      } else if (have_box) {
#pragma unroll
        for (int i = 0; i < D; i++) {
          lo[i] = wave_reduce_minmax<false>(lo_l[i]);
          const int hi = wave_reduce_minmax<true>(hi_l[i]);
          if (hi < lo[i]) have_box = false;
          nb[i] = hi - lo[i] + 1;
        }
      }
      lex_sync();  // rows complete
      const int WX = (nb[0] + 31) >> 5;
      const int n_rows = nb[1] * nb[2];
      const bool fits = have_box && n_rows * WX <= A.boxcap;
      const int rot = (WX == 1) ? (lo[0] & 31) : 0;  // WX == 1: the word is rotated so that code ex sits at bit (ex & 31)
      if (D == 2 && eager) {
        if (fits && pass == 0 && lane < nb[1]) {  // (the reach box serves every pass of the node)
          unsigned int val = e_in ? (__builtin_amdgcn_alignbit(e_a1, e_a0, e_shf) | ~e_mask) : 0xffffffffu;
          val = __builtin_amdgcn_alignbit(val, val, (32 - rot) & 31);
          s_box[lane] = val;
        }
      } else if (fits) {
        const float inv_ny = __builtin_amdgcn_rcpf((float)nb[1]);
        const int ax0 = base_c[0] + lo[0] - half;
        constexpr int SU = 4;  // rows per lane with their loads in flight together
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
              a0[u] = (A.dbg & 8) ? 0u : A.blk[w0];
              a1[u] = (A.dbg & 8) ? 0u : A.blk[w1 < 0 ? 0 : w1];
            }
#pragma unroll
            for (int u = 0; u < SU; u++) {
              const int r = r0 + 64 * u;
              unsigned int val = in[u] ? (__builtin_amdgcn_alignbit(a1[u], a0[u], shf[u]) | ~mask) : 0xffffffffu;
              if (WX == 1) val = __builtin_amdgcn_alignbit(val, val, (32 - rot) & 31);  // rotate left by rot
              if (r < n_rows) s_box[r * WX + w] = val;
            }
          }
        }
      }
      lex_sync();

      // ---- phase D: the combinations, 64 at a time, in ascending control order
      const int rowc = lo[1] + nb[1] * lo[2];
      int ndrop = 0;  // dropped combinations in the steps before this one (uniform)
      for (int x0 = 0; x0 < nA; x0 += 64) {
        const int x = x0 + lane;
        const bool act = x < nA;
        int j0 = 0, j1 = 0, j2 = 0;
        {
          const int xx = act ? x : 0;
          const int a_ = (int)(((float)xx + 0.5f) * r_in0);  // exact: xx < 2^12
          const int ra = xx - a_ * in0;
          int b_ = ra, c_ = 0;
          if (D == 3) {
            b_ = (int)(((float)ra + 0.5f) * r_in1);
            c_ = ra - b_ * in1;
          }
          j0 = s_vl[a_];
          j1 = s_vl[TS + b_];
          if (D == 3) j2 = s_vl[2 * TS + c_];
        }
        const int ci = (D == 3) ? (int)__umul24(__umul24(j0, nd[1]) + j1, nd[2]) + j2 : (int)__umul24(j0, nd[1]) + j1;
        const int en[3] = {j0, TS + j1, 2 * TS + j2};
        const int px = (D == 3) ? (int)__umul24(j0, ndp) + j1 : j0;
        int n, fl;
        {
          const int f0 = s_eflag[en[0]], f1 = s_eflag[en[1]];
          fl = f0 & f1 & 3;
          n = max(f0 >> 8, f1 >> 8);
          if (D == 3) {
            const int f2 = s_eflag[en[2]];
            fl &= f2;
            n = max(n, f2 >> 8);
          }
          if (fl & 2) n = 0;  // unchanged position: not traversed (env_map.h:163)
        }
        // lattice hash of the successor (waypoint.h:93-125); env_map.h:158: `tn == curr` is a hash comparison
        uint64_t h = s_hp[px];
        lex_fold_entry<K>(h, s_eq, en[D - 1]);
        unsigned long long dm;
        if (pass == 0) {
          dm = __ballot(act && h == hcur);
          if (lane == 0) s_drop[x0 >> 6] = dm;
        } else {
          const uint64_t d_ = s_drop[x0 >> 6];
          dm = (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)d_) |
               ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(d_ >> 32)) << 32);
        }
        const bool dropped = (dm >> lane) & 1ull;
        const int e = x - ndrop - __popcll(dm & ((1ull << lane) - 1ull));  // list position
        ndrop += __popcll(dm);
        if (x0 + 64 >= nA) E = nA - ndrop;  // (last step: everything dropped is known)
        const bool emit = act && !dropped;
        const bool mine = emit && (n ? ((sub >> n) & 1ull) != 0ull : pass == 0);
        const int64_t idx = node * A.l_nstride + e;
        // line padding: the lanes just past the end of the list complete the last 128-byte line of every row
        // (unspecified values, inside the node's own region; see expand_grid_kernel.hip)
        const bool past = !act && pass == 0 && A.l_pad;
        const bool pad16 = past && e < ((E + 15) & ~15);
        const bool pad32 = past && e < ((E + 31) & ~31);
        lex_prio(3);
        if ((mine || pad32) && !(A.dbg & 2)) {
          if (A.l_action) lex_st(mine ? ci : -1, &A.l_action[idx]);
          if (A.l_hash && (mine || pad16)) lex_st(h, &A.l_hash[idx]);
          if (A.l_state && (mine || pad16)) {
            double *o = A.l_state + idx;
            const int64_t ss = A.l_stride;
#pragma unroll
            for (int i = 0; i < D; i++) {
              const double *st = s_est + en[i] * (K - 1);
              const double u = s_uval[en[i]];
              const double uK = 0.0 + u;                                     // field of order K
              const double top = (0.0 + u * T) + s_node[(K - 1) * D + i];    // field of order K - 1
              lex_st((double)((K >= 2) ? st[0] : top), &o[(0 * D + i) * ss]);
              lex_st((double)((K >= 3) ? st[1] : (K == 2 ? top : uK)), &o[(1 * D + i) * ss]);
              lex_st((double)((K == 3) ? top : (K == 2 ? uK : 0.0)), &o[(2 * D + i) * ss]);
              lex_st((double)((K == 3) ? uK : 0.0), &o[(3 * D + i) * ss]);
            }
            lex_st(0.0, &o[(4 * D) * ss]);                // Waypoint::yaw of a control without yaw (primitive.h:322)
            lex_st(node_t + A.dt, &o[(4 * D + 1) * ss]);  // env_map.h:161
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
              aa[i] = (K == 3) ? top : (K == 2 ? uK : 0.0);
            }
            double hv;
            unsigned int fv;
            MPLX_POST_GOAL(pg, A.post, D)
            post_eval<D>(pg, h, pp, vv, aa, 0.0, &hv, &fv);
            if (A.post.heur) lex_st(hv, &A.post.heur[idx]);
            if (A.post.flags && mine) A.post.flags[idx] = (uint8_t)fv;
          }
        }
        // ---- the sample loop of traverse_primitive (env_map.h:97-120)
        const bool smp = mine && n != 0;
        const int cntl = smp ? (int)s_tc[n] : 0;  // iterations of `for (t = 0; t < T; t += T/n)`
        int fb = -1;                              // first blocked sample
        {
          const int r = smp ? (int)s_rowmap[n] : 0;
          int ptr[3] = {0, 0, 0};
          ptr[0] = __umul24(j0, rowcap) + r;
          ptr[1] = __umul24(ndp + j1, rowcap) + r;
          if (D == 3) ptr[2] = __umul24(2 * ndp + j2, rowcap) + r;
          bool done = !smp || safe || (A.dbg & 32);
          if (fits) {
            if (WX == 1) {
              const unsigned int *bx = s_box - rowc;
              for (int k0 = 0; __ballot(!done) != 0ull; k0 += UB) {
                unsigned int m = 0;
#pragma unroll
                for (int q = 0; q < UB; q++) {
                  const int k = k0 + q;
                  const unsigned int ex = s_cell[ptr[0] + k];
                  const int ey = s_cell[ptr[1] + k];
                  const int ez = (D == 3) ? (int)s_cell[ptr[2] + k] : 0;
                  const unsigned int word = bx[(D == 3 ? (int)__umul24(nb[1], ez) : 0) + ey];
                  m |= __builtin_amdgcn_ubfe(word, ex, 1u) << q;
                }
                const int left = cntl - k0;
                if (left < UB) m &= (1u << (left > 0 ? left : 0)) - 1u;
                if (!done && m) { fb = k0 + __ffs((int)m) - 1; done = true; }
                if (left <= UB) done = true;
              }
            } else {
              for (int k0 = 0; __ballot(!done) != 0ull; k0 += UB) {
                unsigned int m = 0;
#pragma unroll
                for (int q = 0; q < UB; q++) {
                  const int k = k0 + q;
                  const int ex = s_cell[ptr[0] + k];
                  const int ey = s_cell[ptr[1] + k];
                  const int ez = (D == 3) ? (int)s_cell[ptr[2] + k] : 0;
                  const int dx = ex - lo[0];
                  const unsigned int word = s_box[__umul24((D == 3 ? (int)__umul24(nb[1], ez) : 0) + ey - rowc, WX) + (dx >> 5)];
                  m |= ((word >> (dx & 31)) & 1u) << q;
                }
                const int left = cntl - k0;
                if (left < UB) m &= (1u << (left > 0 ? left : 0)) - 1u;
                if (!done && m) { fb = k0 + __ffs((int)m) - 1; done = true; }
                if (left <= UB) done = true;
              }
            }
          } else {
            // a box too large for the LDS budget (per-step displacements far beyond the BASELINE configurations): the
            // samples read the blocked-bit map directly, UB look-ups in flight per step
            for (int k0 = 0; __ballot(!done) != 0ull; k0 += UB) {
              unsigned int wd[UB];
              int sh[UB];
#pragma unroll
              for (int q = 0; q < UB; q++) {
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
              for (int q = 0; q < UB; q++) {
                if (!done && k0 + q < cntl && (sh[q] < 0 || ((wd[q] >> sh[q]) & 1u))) { fb = k0 + q; done = true; }
              }
              if (k0 + UB >= cntl) done = true;
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
          const double cost = blocked ? INFINITY : 0.0 + (J + A.w * A.dt);
          if (A.l_cost && (mine || pad16)) lex_st(cost, &A.l_cost[idx]);
          if (A.l_iters) lex_st(blocked ? fb + 1 : cntl, &A.l_iters[idx]);
        }
        lex_prio(1);
      }
      if (pass == 0 && lane == 0 && A.l_count) A.l_count[node] = E;
    }
  }
  if (A.done.flag != nullptr) {
    // small synchronous batch: the host spins on a pinned word (DoneSignal, mplx_internal.h)
    __threadfence_system();
    if (lane == 0) {
      const unsigned int waves = gridDim.x * (unsigned int)kLexWPB;
      if (__hip_atomic_fetch_add(A.done.count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == waves - 1u) {
        __hip_atomic_store(A.done.count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(A.done.flag, A.done.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
#undef A
}

template <int D, int K, int TS>
hipError_t lex_inst_attr() {
  static bool attr_set[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 63;
  if (!attr_set[dev] || dev == 63) {
    hipError_t e = hipFuncSetAttribute((const void *)expand_lex_kernel<D, K, TS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  return hipSuccess;
}

template <int D, int K, int TS>
hipError_t launch_lex_ts(const GridArgs &a, hipStream_t stream) {
  if (a.n_nodes == 0) return hipSuccess;
  const int64_t n_wg = (a.n_nodes + kLexWPB - 1) / kLexWPB;
  const int64_t blocks = n_wg < (int64_t)a.grid_limit ? n_wg : (int64_t)a.grid_limit;
  const size_t lds = lex_lds_bytes(D, K, a.ndp, a.nU, a.n_max, a.rmax, a.boxcap);
  if (hipError_t e = lex_inst_attr<D, K, TS>()) return e;
  hipLaunchKernelGGL((expand_lex_kernel<D, K, TS>), dim3((unsigned)blocks), dim3(kLexBT), lds, stream, a);
  return hipGetLastError();
}
template <int D, int K>
hipError_t launch_lex_inst(const GridArgs &a, hipStream_t stream) {
  return a.ndp <= 8 ? launch_lex_ts<D, K, 8>(a, stream) : (a.ndp <= 16 ? launch_lex_ts<D, K, 16>(a, stream) : launch_lex_ts<D, K, 32>(a, stream));
}

template <int D, int K, int TS>
int lex_resident_ts(size_t lds) {
  if (lex_inst_attr<D, K, TS>() != hipSuccess) return 0;
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)expand_lex_kernel<D, K, TS>, kLexBT, lds) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return nb;
}
template <int D, int K>
int lex_resident_inst(size_t lds, int ndp) {
  return ndp <= 8 ? lex_resident_ts<D, K, 8>(lds) : (ndp <= 16 ? lex_resident_ts<D, K, 16>(lds) : lex_resident_ts<D, K, 32>(lds));
}

}  // namespace

size_t lex_lds_bytes(int dim, int order, int ndp, int nU, int n_max, int rmax, int boxcap) {
  return (size_t)LexLds(dim, order, kLexWPB, ndp, nU, n_max, rmax, boxcap).total;
}
int lex_waves_per_block() { return kLexWPB; }

// the configurations this kernel has an instantiation for (the host checks the rest of the scope: lexicographic
// table, no potential map, v_max > 0 ...)
bool lex_covers(int dim, int control) {
  return (dim == 2 || dim == 3) && (control == 0x01 || control == 0x03 || control == 0x07);
}

hipError_t launch_expand_lex(int dim, int control, const GridArgs &a, hipStream_t s) {
  if (dim == 2) {
    switch (control) {
      case 0x01: return launch_lex_inst<2, 1>(a, s);
      case 0x03: return launch_lex_inst<2, 2>(a, s);
      case 0x07: return launch_lex_inst<2, 3>(a, s);
    }
  } else if (dim == 3) {
    switch (control) {
      case 0x01: return launch_lex_inst<3, 1>(a, s);
      case 0x03: return launch_lex_inst<3, 2>(a, s);
      case 0x07: return launch_lex_inst<3, 3>(a, s);
    }
  }
  return hipErrorInvalidValue;
}

int lex_resident_blocks(int dim, int control, int ndp, size_t lds) {
  if (dim == 2) {
    switch (control) {
      case 0x01: return lex_resident_inst<2, 1>(lds, ndp);
      case 0x03: return lex_resident_inst<2, 2>(lds, ndp);
      case 0x07: return lex_resident_inst<2, 3>(lds, ndp);
    }
  } else if (dim == 3) {
    switch (control) {
      case 0x01: return lex_resident_inst<3, 1>(lds, ndp);
      case 0x03: return lex_resident_inst<3, 2>(lds, ndp);
      case 0x07: return lex_resident_inst<3, 3>(lds, ndp);
    }
  }
  return 0;
}

}  // namespace mplx
