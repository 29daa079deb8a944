// mplx_grid_common.h -- what the factorised list kernels share: the LDS carve-up (host and device), the list stores,
// the wave-level helpers and the small per-pair functions.  Included by expand_grid_kernel.hip (one node per wave) and
// expand_pair_kernel.hip (two nodes per wave); every definition is the one that used to live in expand_grid_kernel.hip.
#ifndef MPLX_GRID_COMMON_H
#define MPLX_GRID_COMMON_H

#include "mplx_internal.h"
#include "mplx_device_common.h"

namespace mplx {

#ifndef MPLX_GRID_TT_RESIDENT
#define MPLX_GRID_TT_RESIDENT 1
#endif

// LDS carve-up, shared by host (size) and device (offsets).
struct GridLds {
  // shared by the workgroup (read-only after set-up)
  int o_uval, o_uidx, o_tc, o_tt, o_wave0;
  // per wave, relative to the wave's block
  int w_node, w_est, w_hp, w_eq, w_eflag, w_box, w_misc, w_rowmap, w_list, w_cell, w_uq, wave_bytes;
  int o_uyaw, w_yaw, w_ycs, w_yq, w_hmask, w_vs, w_ycsr;  // yaw controls only
  int total;
  int F, EN, PN, PNC, tts, KQ;
  // ym & 3: 0 = no yaw, 1 = yaw, 2 = yaw with the per-sample heading cost (wyaw > 0); ym & 4: per-sample |vel| of a
  // potential map with gradient_weight != 0 (env_map.h:116); ndy = distinct yaw rates
  // ulex: the control table is the nested-loop enumeration of its per-axis values (GridArgs::ulex): the per-control
  // entry indices are arithmetic then, no table
  __host__ __device__ GridLds(int D, int K, int waves, int nU, int ndp, int n_max, int rmax, int boxcap, int ym,
                              int ndy, int ulex) {
    const int grad = ym & 4;  // bit 2: velocity rows of ALL axes (potential maps with gradient_weight != 0)
    const int lean = ym & 64; // bit 6: expand_pair_kernel.hip's layout -- no velocity rows (computed per sample), no box / time-row array
    const int tt_resident = MPLX_GRID_TT_RESIDENT;
    ym &= 3;
    F = 4 * D + 2;
    EN = D * ndp;
    PN = (D == 3) ? ndp * ndp : ndp;
    tts = n_max + 1;
    KQ = K == 3 ? 4 : K;
    int b = 0;
    o_uval = b; b += EN * 8;
    o_uidx = b; b += ulex ? 0 : ((nU + 1) & ~1) * 2;  // 4 bits per axis
    o_tc = b; b += 64;
    b = (b + 7) & ~7;
    // the accumulated sample times of every count n <= n_max, [n_max + 1][tts]: resident for the workgroup's life, so
    // that a pass needs no round trip to the global table between the pair phase and the rows
    o_tt = b; b += tt_resident ? (n_max + 1) * tts * 8 : 0;
    o_uyaw = b; b += ym ? 16 * 8 : 0;
    b = (b + 15) & ~15;
    o_wave0 = b;
    int w = 0;
    w_node = w; w += F * 8;
    w_est = w; w += EN * (K - 1) * 8;  // end-state fields of order < K - 1 (the rest follow from u and the node)
    w_hp = w; w += PN * 8;
    w = (w + 15) & ~15;
    w_eq = w; w += EN * KQ * 4;  // lattice integers of order < K (KQ = K rounded up to 1, 2 or 4 for aligned vector loads)
    w_eflag = w; w += EN * 4;
    w = (w + 7) & ~7;
    w_box = w; w += lean ? 0 : ((boxcap * 4 > rmax * tts * 8) ? boxcap * 4 : rmax * tts * 8);  // also the sample times while rows are built
    w = (w + 15) & ~15;
    w_misc = w; w += 49 * 4;  // M_* below (M_WORDS)
    w_rowmap = w; w += ((n_max + 2) & ~1) * 2;  // per sample count n <= n_max: offset of its row inside an entry's block this pass (0xffff: not this pass)
    // (lean layout: per combination of axis entries a word + a list position instead of the list of pairs)
    PNC = ndp * ndp * (D == 3 ? ndp : 1);
    // lean: [nU] f64 cost + [PNC] u32 word + [PNC] u16 first position + [nU] u16 list + [nU] u16 iterations
    w = lean ? ((w + 7) & ~7) : w;
    w_list = w; w += lean ? ((nU * 8 + PNC * 6 + ((nU + 1) & ~1) * 4 + 7) & ~7) : ((nU + 1) & ~1) * 2;
    w_cell = w; w += EN * rmax * tts + 8;  // + 8: the sample loop reads up to 7 codes past a row
    w = (w + 15) & ~15;
    w_uq = w; w += K >= 3 ? EN * 8 : 0;         // per entry: the top coefficient's quotient (u / 6, u / 24) for the rows
    w_yaw = w; w += ym ? 16 * 8 : 0;            // yaw(T) per yaw value
    w_ycs = w; w += ym ? 16 * 16 : 0;           // cos, sin of yaw(T)
    w_yq = w; w += ym ? 16 * 4 : 0;             // lattice integer of yaw(T)
    w_hmask = w; w += ym ? ndp * ndp * 2 : 0;   // per (x entry, y entry): yaw values passing the heading limit
    w = (w + 15) & ~15;
    w_vs = w; w += lean ? 0 : (grad ? D * ndp * rmax * tts * 8 : (ym == 2 ? 2 * ndp * rmax * tts * 8 : 0));  // velocity of the x / y (gradient cost: all) entries at the sample times
    w_ycsr = w; w += ym == 2 ? ndy * rmax * tts * 16 : 0;    // cos, sin of the yaw at the sample times
    wave_bytes = (w + 15) & ~15;
    total = o_wave0 + waves * wave_bytes;
  }
};

// GridLds's mode word of expand_pair_kernel.hip (yaw tables, with or without the per-sample heading rows; lean layout)
__host__ __device__ constexpr int pair_lds_mode(bool ycost) { return (ycost ? 2 : 1) | 64; }

namespace {

using namespace dev;

#ifndef MPLX_GRID_WPB
#define MPLX_GRID_WPB 4
#endif
#ifndef MPLX_GRID_UB
#define MPLX_GRID_UB 8
#endif
constexpr int kWPB = MPLX_GRID_WPB;  // waves (= nodes in flight) per workgroup
constexpr int kBT = 64 * kWPB;
constexpr int kTabStride = 64;  // row stride of the global time table (launch_make_tables)
constexpr int kUB = MPLX_GRID_UB;    // samples per step of the sample loop

// List stores: never re-read by this kernel, so they must not allocate in L2.  Measured on C4 (same box,
// profiles/micro/store_policy.sh): default policy 0.723 ms, sc1 0.725, sc0 sc1 0.729, nt (what
// __builtin_nontemporal_store emits) 0.636 - 0.641, sc1 nt 0.626, sc0 sc1 nt 0.628 -> agent-scope non-temporal.
// -DMPLX_ST_ASM="..." selects other bits, -DMPLX_ST_BUILTIN the builtin.
#if !defined(MPLX_ST_ASM) && !defined(MPLX_ST_BUILTIN)
#define MPLX_ST_ASM "sc1 nt"
#endif
template <typename T>
__device__ __forceinline__ void st_stream(T v, T *p) {
#ifdef MPLX_ST_ASM
  if constexpr (sizeof(T) == 8) asm volatile("global_store_dwordx2 %0, %1, off " MPLX_ST_ASM ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dword %0, %1, off " MPLX_ST_ASM ::"v"(p), "v"(v) : "memory");
#else
  __builtin_nontemporal_store(v, p);
#endif
}

// Wave priority by progress through a node: 0 while the tables are built (T1, A), 1 for rows and box staging, 3
// for a phase-D step (list stores + sample loop), back to 1 after it.  A wave that is about to retire its stores
// and move on wins issue slots over waves that are still setting up, which keeps the store path busier: -5.5 % on
// C4 (0.632 -> 0.597 ms on one box; flat priority 3 around the stores alone -2 %, around a whole phase-D step -4 %;
// profiles/micro/variants_run.sh).  -DMPLX_NO_PRIO compiles it out.
__device__ __forceinline__ void wave_prio(int p) {
#ifndef MPLX_NO_PRIO
  if (p == 0) __builtin_amdgcn_s_setprio(0);
  else if (p == 1) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(3);
#else
  (void)p;
#endif
}

// Orders this wave's LDS traffic: LDS executes a wave's instructions in order, so
// only the compiler has to be kept from moving accesses across the point.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Folds the lattice integers (order < K) of axis entry `e` into a hash, one aligned vector load.
template <int K>
__device__ __forceinline__ void fold_entry(uint64_t &h, const int *s_eq, int e) {
  if (K == 1) {
    fold(h, s_eq[e]);
  } else if (K == 2) {
    const int2 q = *(const int2 *)(s_eq + e * 2);
    fold(h, q.x);
    fold(h, q.y);
  } else {
    const int4 q = *(const int4 *)(s_eq + e * 4);
    fold(h, q.x);
    fold(h, q.y);
    fold(h, q.z);
    if (K >= 4) fold(h, q.w);
  }
}

// valid / same-position bits (AND) and sample count (max) of a pair from its axis entries' flags
template <int D>
__device__ __forceinline__ int pair_flags(const int *s_eflag, int ndp, int j0, int j1, int j2) {
  const int f0 = s_eflag[j0], f1 = s_eflag[ndp + j1];
  int fl = f0 & f1 & 3, n = max(f0 >> 8, f1 >> 8);
  if (D == 3) {
    const int f2 = s_eflag[2 * ndp + j2];
    fl &= f2;
    n = max(n, f2 >> 8);
  }
  return fl | (n << 8);
}

// misc words of a wave
enum { M_BASE = 0, M_NV = 6, M_NODEQ = 12, M_VL = 24, M_YQ = 36, M_VLC = 37, M_WORDS = 49 };  // NV: valid entries per axis; NODEQ: [D][4]; VL: [D][16] bytes; YQ: the node's yaw integer; VLC: [D][16] bytes, the entries whose row the current sample count needs

// reference include/mpl_basis/math.h:15-19
__device__ __forceinline__ double wrap_angle(double a) {
  while (a > M_PI) a -= 2.0 * M_PI;
  while (a < -M_PI) a += 2.0 * M_PI;
  return a;
}

}  // namespace
}  // namespace mplx
#endif
