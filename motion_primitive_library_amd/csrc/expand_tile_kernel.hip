// expand_tile_kernel.hip -- tiled successor expansion with per-node successor
// LISTS as output (the reference's own output shape) for gfx950 (MI355X).
//
// Same function as expand_kernel.hip (MPL::env_map<Dim>::get_succ, reference
// include/mpl_planner/env/env_map.h:147-172 with traverse_primitive :90-132),
// restructured around what the first profile showed (profiles/README.md):
//   * 57 % of the pairs of a realistic frontier are dynamically invalid and the
//     valid ones need 5..21 map samples each, so "one lane walks one pair"
//     leaves two thirds of the lanes idle.  Here a workgroup owns a tile of
//     whole nodes and the work is re-packed between phases:
//       A1  every (node, control) pair: dynamic limits only (a few flops);
//           the valid pairs are compacted, in order, into an LDS list;
//       A2  valid pairs only, dense lanes: successor state, lattice hash,
//           position in the node's output list; state / hash / action are
//           written straight away; each pair appends its samples (pair, k) to
//           an LDS work list;
//       B   work list, dense lanes: ONE map sample per lane.  The sample time
//           comes from a table of sequentially accumulated times (the
//           reference's `for (t = 0; t < T; t += T/n)` runs n or n+1 times,
//           env_map.h:97-99); a blocked sample does an LDS atomicMin on its
//           pair's first-blocked index;
//       C   valid pairs: cost (= J + w dt, or +inf) and the optional iteration
//           count are written to the pair's list slot.
//   * the dense slot layout wrote 129 B for every pair; lists write only the
//     emitted successors, contiguously per node (count[node] entries starting
//     at node*nU), which is also what B_alg counts.
//   * the three divisors on the path (map resolution, 0.01, 0.1) are constants
//     of a launch; their refined reciprocals are computed once on the device
//     with the same v_rcp_f64 + 2 Newton steps hipcc emits for `a / b`, and a
//     quotient is then mul + fma + fma -- the tail of hipcc's own division
//     sequence, hence bit-identical to `/` for operands in normal range (every
//     value on this path; checked against true division in the tests).
//
// Scope of this kernel: controls without yaw, no potential map (both need the
// per-sample costs summed in order; they stay on expand_kernel.hip), any search
// region, v_max > 0 (so the sample count per pair is bounded and the work list
// has a capacity), Dim 2/3, K = 1..4.  Everything is bit-exact; arithmetic rules
// as in expand_kernel.hip (-ffp-contract=off, explicit fma only inside the
// division tail).
//
// SERVICE MODE (template flag SVC; mplx_api.cpp, "service").  A search asks for a few nodes at a time and waits for
// the answer: one launch + hipStreamSynchronize is 11 - 12 us before the kernel has done anything
// (profiles/micro/mailbox_latency.hip), as much as the work itself.  In service mode the same kernel stays RESIDENT:
// its workgroups wait for requests on a doorbell word in pinned host memory, run the tile loop on the nodes the host
// put into the landing block, and publish `done` -- a round trip of ~4 us instead of ~12.
//   * workgroup 0 is the coordinator: its wave 0 polls the host doorbell ((seq << 32) | n_nodes), republishes it in
//     device memory (cmd) for the other workgroups, and decides alone when to leave (quit word, or no request for
//     svc_idle ticks of the 100 MHz clock: a resident kernel must not outlive a host that went away);
//   * every workgroup reports the request it finished in fin[g]; workgroup 0 waits for all of them and then stores
//     `done` = seq with system-scope release: every list of the request is in the landing block before the host sees it;
//   * ALL 64 lanes of wave 0 poll and publish together (same address, same value).  `if (tid == 0)` next to the
//     barriers of the request loop is lane divergence around a convergent operation: hipcc threaded the other lanes of
//     wave 0 into the next trip's s_barrier while lane 0 still had its store to do, and the workgroup hung.
#include "mplx_internal.h"
#include "mplx_device_common.h"

#include <math.h>

namespace mplx {
namespace {

constexpr int kBT = 512;        // threads per workgroup (8 waves; two or three workgroups per CU)
constexpr int kWaves = kBT / 64;
constexpr int kMaxN = 63;       // largest sample count n handled by the time table
constexpr int kTabStride = 64;  // time table row stride (k = 0 .. n)

using namespace dev;


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

// Node state access.  ONE (a single node per workgroup, |U| > 512): the node index
// is wave-uniform, the loads become scalar loads.  Otherwise the tile's nodes
// are staged in LDS once (phase 0) and read from there.
template <int D, int K, bool ONE>
struct NodeLoad {
  double p[D], v[D], a[D], j[D];
  __device__ __forceinline__ void load(const double *nodes, int64_t node_stride, const double *s_node, int64_t node0,
                                       int nl) {
    if (ONE) {
      const double *nd = nodes + node0;
      const int64_t st = node_stride;
#pragma unroll
      for (int i = 0; i < D; i++) {
        p[i] = nd[(0 * D + i) * st];
        v[i] = (K >= 2) ? nd[(1 * D + i) * st] : 0.0;
        a[i] = (K >= 3) ? nd[(2 * D + i) * st] : 0.0;
        j[i] = (K >= 4) ? nd[(3 * D + i) * st] : 0.0;
      }
    } else {
      const double *nd = s_node + nl * (4 * D + 2);
#pragma unroll
      for (int i = 0; i < D; i++) {
        p[i] = nd[0 * D + i];
        v[i] = (K >= 2) ? nd[1 * D + i] : 0.0;
        a[i] = (K >= 3) ? nd[2 * D + i] : 0.0;
        j[i] = (K >= 4) ? nd[3 * D + i] : 0.0;
      }
    }
  }
};

// pair index inside the tile -> (local node, control)
template <bool ONE>
__device__ __forceinline__ void split_pair(int p, int nU, float inv_nU, int *nl, int *ci) {
  if (ONE) { *nl = 0; *ci = p; return; }
  // exact for p < 2^20: (p + 0.5) / nU is never within 2^-20 of an integer
  const int q = (int)(((float)p + 0.5f) * inv_nU);
  *nl = q;
  *ci = p - q * nU;
}

constexpr int kUB = 4;  // samples in flight per lane in phase B

constexpr uint64_t kSvcExit = ~0ull;

template <int D, int K, bool ONE, bool SVC>
__global__ __launch_bounds__(kBT) void expand_tile_kernel(const TileArgs A_kernarg) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ uint64_t s_cmd;
  // The argument block is read where it lies, through the kernarg segment pointer (constant address space: scalar
  // loads), laundered once per tile so that the loads stay next to their uses -- preloaded, the block takes most of
  // the 102 SGPRs and a quarter of the kernel's VALU instructions were lane spills (see expand_grid_kernel.hip).
  typedef const TileArgs __attribute__((address_space(4))) *KernargPtr;
  KernargPtr Ak = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
  (void)A_kernarg;
#define A (*Ak)
  constexpr int F = 4 * D + 2;
  // ---- LDS carve-up; must match tile_lds_bytes()
  const int P_cap = A.tile_pairs;
  const int tts = A.n_max + 1;                                  // time-table row stride = max iterations
  uint64_t *s_hcur = (uint64_t *)smem;                          // [npb] hash of each node
  double *s_tt = (double *)(s_hcur + A.npb);                    // [(n_max+1) * tts] accumulated sample times
  double *s_node = s_tt + (A.n_max + 1) * tts;                  // [npb * F] node states (unused when ONE)
  unsigned int *s_fb = (unsigned int *)(s_node + (ONE ? 0 : A.npb * F));  // [P_cap] first blocked k
  int *s_misc = (int *)(s_fb + P_cap);                          // [32] counters + wave sums
  int *s_ncnt = s_misc + 32;                                    // [npb] emitted per node
  int *s_nbase = s_ncnt + A.npb;                                // [npb] emitted-prefix at node start
  unsigned short *s_vlist = (unsigned short *)(s_nbase + A.npb);   // [P_cap] valid index -> tile pair
  unsigned short *s_j = s_vlist + P_cap;                        // [P_cap] list position (0xffff: not emitted)
  unsigned short *s_wl = s_j + P_cap;                           // [wl_cap] (valid index << 6) | k
  unsigned char *s_n = (unsigned char *)(s_wl + A.wl_cap);      // [P_cap] sample count n, by valid index
  unsigned char *s_tc = s_n + P_cap;                            // [64] loop iteration count of each n
  // 8-byte aligned tail: the control table
  double *s_U = (double *)(smem + A.lds_u_offset);              // [nU * udim]
  int *s_wsum = s_misc + 8;                                     // [kWaves]
  int *s_wcount = s_misc + 0;                                   // work-list fill

  const int tid = threadIdx.x;
  const int nU = A.nU;
  const int udim = A.udim;
  const double T = A.dt;
  const float inv_nU = A.inv_nU;
  const double org[3] = {A.org0, A.org1, A.org2};
  const int dims[3] = {A.dim0, A.dim1, A.dim2};

  // ---- once per (persistent) workgroup: sample-time tables and controls into LDS
  if (tid < 64) s_tc[tid] = A.tcnt[tid];
  for (int i = tid; i < (A.n_max + 1) * tts; i += kBT) {
    const int n = i / tts, k = i - n * tts;
    s_tt[i] = A.ttab[n * kTabStride + k];
  }
  for (int i = tid; i < nU * udim; i += kBT) s_U[i] = A.U[i];

  const bool wave0 = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;
  uint32_t svc_seq = SVC ? (uint32_t)A.svc_seq0 : 0u;
  for (;;) {  // requests (service mode); one trip otherwise
  int64_t n_nodes = A.n_nodes;
  if (SVC) {
    if (wave0) {
      uint64_t v;
      if (blockIdx.x == 0) {
        const uint64_t t0 = wall_clock64();
        for (;;) {
          v = __hip_atomic_load(&A.svc_mb->doorbell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if ((uint32_t)(v >> 32) == svc_seq + 1u) break;
          if (__hip_atomic_load(&A.svc_mb->quit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u ||
              wall_clock64() - t0 > A.svc_idle) {
            v = kSvcExit;
            break;
          }
        }
        __hip_atomic_store(&A.svc_dev[0], v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (v == kSvcExit) __hip_atomic_store(&A.svc_mb->alive, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      } else {
        for (;;) {
          v = __hip_atomic_load(&A.svc_dev[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
          if (v == kSvcExit || (uint32_t)(v >> 32) == svc_seq + 1u) break;
          __builtin_amdgcn_s_sleep(2);
        }
      }
      // the host wrote the nodes (and may have replaced the map) since the last request: nothing read after this
      // point may be served from a cache (vector L1 of this CU and L2 by the fence; the scalar cache, which the ONE
      // variant reads its node through, on its own).  Once per workgroup, before the barrier that lets the other waves
      // go on: per wave it was 8 x 64 L2 invalidations a request.
      // A workgroup without a tile in this request reads and writes nothing: no fences (each one empties or writes
      // back the L2 under the workgroups that are at work).
      if (v != kSvcExit && (int64_t)blockIdx.x * A.npb < (int64_t)(uint32_t)v) {
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        __builtin_amdgcn_s_dcache_inv();
      }
      s_cmd = v;
    }
    __syncthreads();
    const uint64_t cmd = s_cmd;
    __syncthreads();
    if (cmd == kSvcExit) break;
    svc_seq = (uint32_t)(cmd >> 32);
    n_nodes = (int64_t)(uint32_t)cmd;
  }
  const int64_t n_tiles = (n_nodes + A.npb - 1) / A.npb;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
  asm volatile("" : "+s"(Ak));
  __syncthreads();  // LDS of the previous tile is free; tables visible
  const int64_t node0 = tile * A.npb;
  const int nn = ONE ? 1 : (int)((n_nodes - node0) < (int64_t)A.npb ? (n_nodes - node0) : (int64_t)A.npb);
  const int P = nn * nU;

  // ---- phase 0: node states into LDS, per-node hash, counters
  if (tid == 0) *s_wcount = 0;
  if (!ONE) {
    for (int i = tid; i < nn * F; i += kBT) {
      const int r = i / nn, nl = i - r * nn;  // consecutive lanes -> consecutive nodes of one row
      s_node[nl * F + r] = A.nodes[(int64_t)r * A.node_stride + node0 + nl];
    }
    __syncthreads();
  }
  for (int nl = tid; nl < nn; nl += kBT) {
    NodeLoad<D, K, ONE> nd;
    nd.load(A.nodes, A.node_stride, s_node, node0, nl);
    s_hcur[nl] = lattice_hash<D, K>(nd.p, nd.v, nd.a, nd.j, A.R001, A.R01);
    s_ncnt[nl] = 0;
    s_nbase[nl] = 0;
  }
  __syncthreads();

  // ---- phase A1: dynamic limits of every pair; ordered compaction of the valid ones
  int V = 0;  // valid pairs so far (uniform)
  for (int base = 0; base < P; base += kBT) {
    const int p = base + tid;
    bool valid = false;
    int n = 0;
    if (p < P) {
      int nl, ci;
      split_pair<ONE>(p, nU, inv_nU, &nl, &ci);
      NodeLoad<D, K, ONE> nd;
      nd.load(A.nodes, A.node_stride, s_node, node0, nl);
      const double *u = s_U + ci * udim;
      double max_v = 0;
      valid = true;
#pragma unroll
      for (int i = 0; i < D; i++) {
        Ax<K> ax;
        ax.init(0.0, nd.v[i], nd.a[i], nd.j[i], u[i]);
        const double mv = ax.max_vel(T);
        if (mv > max_v) max_v = mv;
        if (K >= 2 && A.v_max > 0) valid = valid && !(mv > A.v_max);
        if (K >= 3 && A.a_max > 0) valid = valid && !(ax.max_acc(T) > A.a_max);
        if (K >= 4 && A.j_max > 0) valid = valid && !(ax.max_jrk(T) > A.j_max);
      }
      // env_map.h:95: n = max(5, (int)ceil(max_v * T / res))
      n = (int)ceil(div_by(max_v * T, A.res, A.Rres));
      n = n < 5 ? 5 : n;
    }
    int tot;
    const int v = V + block_scan(valid, &tot, s_wsum);
    if (valid) {
      s_vlist[v] = (unsigned short)p;
      s_n[v] = (unsigned char)(n > A.n_max ? 255 : n);
    }
    V += tot;
  }
  __syncthreads();

  if (A.dbg & 8) continue;  // timing ablation only
  // ---- phase A2: successor state, hash, list position, immediate writes, work list
  int E = 0;  // emitted successors so far in the tile (uniform)
  for (int base = 0; base < V; base += kBT) {
    const int v = base + tid;
    bool emit = false;
    int nl = 0, ci = 0, cnt = 0;
    double npos[D], nvel[D], nacc[D], njrk[D];
    uint64_t h_next = 0;
    double ct = 0;
    if (v < V) {
      const int p = s_vlist[v];
      split_pair<ONE>(p, nU, inv_nU, &nl, &ci);
      NodeLoad<D, K, ONE> nd;
      nd.load(A.nodes, A.node_stride, s_node, node0, nl);
      ct = ONE ? A.nodes[(4 * D + 1) * A.node_stride + node0] : s_node[nl * F + 4 * D + 1];
      const double *u = s_U + ci * udim;
      bool same_pos = true;
#pragma unroll
      for (int i = 0; i < D; i++) {
        Ax<K> ax;
        ax.init(nd.p[i], nd.v[i], nd.a[i], nd.j[i], u[i]);
        npos[i] = ax.template pos<true>(T);
        nvel[i] = ax.template vel<true>(T);
        nacc[i] = ax.template acc<true>(T);
        njrk[i] = ax.template jrk<true>(T);
        same_pos = same_pos && (nd.p[i] == npos[i]);
      }
      h_next = lattice_hash<D, K>(npos, nvel, nacc, njrk, A.R001, A.R01);
      emit = (h_next != s_hcur[nl]);  // env_map.h:158: `tn == curr` is a hash comparison
      s_fb[v] = 0xffffffffu;
      if (same_pos) s_n[v] = 0;  // never traversed (env_map.h:163); tcnt[0] == 0
      const int n = same_pos ? 0 : (int)s_n[v];
      if (emit && n != 255) cnt = (int)s_tc[n];
    }
    // ordered position among the emitted successors of the tile
    int tot;
    const int e = E + block_scan(emit, &tot, s_wsum);
    // the first valid pair of a node records where that node's emitted run starts
    if (!ONE && v < V) {
      bool first_of_node = (v == 0);
      if (!first_of_node) {
        int pl, pc;
        split_pair<ONE>((int)s_vlist[v - 1], nU, inv_nU, &pl, &pc);
        first_of_node = (pl != nl);
      }
      if (first_of_node) s_nbase[nl] = e;
    }
    E += tot;
    if (!ONE) __syncthreads();
    if (v < V) {
      if (emit) {
        const int j = ONE ? e : e - s_nbase[nl];
        s_j[v] = (unsigned short)j;
        const int64_t idx = (node0 + nl) * A.l_nstride + j;
        const bool wr = !(A.dbg & 2);  // timing ablation only
        if (wr && A.l_action) __builtin_nontemporal_store(ci, &A.l_action[idx]);  // outputs are never re-read here: stream past L2
        if (wr && A.l_hash) __builtin_nontemporal_store(h_next, &A.l_hash[idx]);
        if (wr && A.l_state) {
          double *o = A.l_state + idx;
          const int64_t ss = A.l_stride;
#pragma unroll
          for (int i = 0; i < D; i++) {
            __builtin_nontemporal_store(npos[i], &o[(0 * D + i) * ss]);
            __builtin_nontemporal_store(nvel[i], &o[(1 * D + i) * ss]);
            __builtin_nontemporal_store(nacc[i], &o[(2 * D + i) * ss]);
            __builtin_nontemporal_store(njrk[i], &o[(3 * D + i) * ss]);
          }
          __builtin_nontemporal_store(0.0, &o[(4 * D) * ss]);  // Waypoint::yaw of a control without yaw (primitive.h:322)
          __builtin_nontemporal_store(ct + A.dt, &o[(4 * D + 1) * ss]);  // env_map.h:161
        }
        // what the search computes for the successor next (graph_search.h:84-88), while it is in registers
        if (wr && (A.post.heur || A.post.flags)) {
          double hv;
          unsigned int fv;
          MPLX_POST_GOAL(pg, A.post, D)
          post_eval<D>(pg, h_next, npos, nvel, nacc, 0.0, &hv, &fv);
          if (A.post.heur) __builtin_nontemporal_store(hv, &A.post.heur[idx]);
          if (A.post.flags) A.post.flags[idx] = (uint8_t)fv;
        }
        if (!ONE) atomicAdd(&s_ncnt[nl], 1);
      } else {
        s_j[v] = 0xffffu;
      }
    }
    // work list: cnt entries (v, k); unordered, one LDS atomic per wave
    {
      const int lane = tid & 63;
      const int c = cnt;
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
      for (int k = 0; k < c; k++) s_wl[off + k] = (unsigned short)((v << 6) | k);
    }
  }
  __syncthreads();
  if (ONE && tid == 0) s_ncnt[0] = E;

  // ---- phase B: one map sample per lane, kUB samples in flight per lane.
  // Written as straight-line stages over kUB independent samples (no per-sample
  // branches) so the loads of all of them are in flight together; lanes past the
  // end of the work list re-do its last entry and only skip the final atomic.
  const int W = *s_wcount;
  for (int w0 = tid; w0 < W && !(A.dbg & 1); w0 += kBT * kUB) {
    int vv[kUB], kk[kUB], cc[kUB], nl[kUB];
    bool act[kUB];
#pragma unroll
    for (int q = 0; q < kUB; q++) {
      const int w = w0 + q * kBT;
      act[q] = w < W;
      const int ent = s_wl[act[q] ? w : W - 1];
      vv[q] = ent >> 6;
      kk[q] = ent & 63;
    }
    double tq[kUB];
#pragma unroll
    for (int q = 0; q < kUB; q++) {
      split_pair<ONE>((int)s_vlist[vv[q]], nU, inv_nU, &nl[q], &cc[q]);
      tq[q] = s_tt[(int)s_n[vv[q]] * tts + kk[q]];
    }
    double uq[kUB][D];
#pragma unroll
    for (int q = 0; q < kUB; q++) {
      const double *u = s_U + cc[q] * udim;
#pragma unroll
      for (int i = 0; i < D; i++) uq[q][i] = u[i];
    }
    int midx[kUB];
    bool inside[kUB];
#pragma unroll
    for (int q = 0; q < kUB; q++) {
      NodeLoad<D, K, ONE> nd;
      nd.load(A.nodes, A.node_stride, s_node, node0, nl[q]);
      bool in = true;
      int cell[D];
#pragma unroll
      for (int i = 0; i < D; i++) {
        Ax<K> ax;
        ax.init(nd.p[i], nd.v[i], nd.a[i], nd.j[i], uq[q][i]);
        // map_util.h:103-108: cell = round((pos - origin) / res - 0.5), then bounds.
        const double qd = div_by(ax.template pos<false>(tq[q]) - org[i], A.res, A.Rres);
        const double sh = qd - 0.5;
        // sh > -0.5  <=>  the rounded cell is >= 0; then qd > 0 and (qd - 0.5 being
        // exact for qd >= 0.5) round-half-away(sh) == trunc(qd).  Otherwise the cell
        // is negative, i.e. outside -- its value is irrelevant.
        const int c = (int)qd;
        cell[i] = c;
        in = in && (sh > -0.5) && (c < dims[i]);
      }
      inside[q] = in;
      int idx = cell[0] + dims[0] * cell[1];
      if (D == 3) idx += dims[0] * dims[1] * cell[2];
      midx[q] = (in && !(A.dbg & 16)) ? idx : 0;  // dbg 16: timing ablation, all lanes read cell 0
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
      if (act[q] && blocked) atomicMin(&s_fb[vv[q]], (unsigned int)kk[q]);
    }
  }
  __syncthreads();

  // ---- phase C: costs (and iteration counts) of the emitted successors
  for (int v = tid; v < V && !(A.dbg & 4); v += kBT) {
    const int j = s_j[v];
    if (j == 0xffff) continue;
    const int p = s_vlist[v];
    int nl, ci;
    split_pair<ONE>(p, nU, inv_nU, &nl, &ci);
    const double *u = s_U + ci * udim;
    const int n = s_n[v];
    bool blocked;
    int iters;
    if (n == 255) {
      // sample count beyond the time table: walk this pair serially, exactly like
      // expand_kernel.hip does (not reachable when v_max bounds n)
      NodeLoad<D, K, ONE> nd;
      nd.load(A.nodes, A.node_stride, s_node, node0, nl);
      Ax<K> ax[D];
      double max_v = 0;
#pragma unroll
      for (int i = 0; i < D; i++) {
        ax[i].init(nd.p[i], nd.v[i], nd.a[i], nd.j[i], u[i]);
        const double mv = ax[i].max_vel(T);
        if (mv > max_v) max_v = mv;
      }
      int nb = (int)ceil(max_v * T / A.res);
      nb = nb < 5 ? 5 : nb;
      const double sdt = T / nb;
      blocked = false;
      iters = 0;
      for (double t = 0; t < T && !blocked; t += sdt) {
        iters++;
        bool outside = false;
        int cell[D];
#pragma unroll
        for (int i = 0; i < D; i++) {
          cell[i] = (int)round((ax[i].template pos<false>(t) - org[i]) / A.res - 0.5);
          outside = outside || cell[i] < 0 || cell[i] >= dims[i];
        }
        blocked = outside;
        if (!outside) {
          int64_t idx = cell[0] + (int64_t)dims[0] * cell[1];
          if (D == 3) idx += (int64_t)dims[0] * dims[1] * cell[2];
          if (A.region != nullptr && !((A.region[idx >> 5] >> (idx & 31)) & 1u)) blocked = true;
          else if (A.map[idx] == 100) blocked = true;
        }
      }
    } else {
      // n == 0 marks an unchanged position: not traversed (env_map.h:163), tcnt[0] == 0
      const unsigned int fb = s_fb[v];
      blocked = (fb != 0xffffffffu);
      // iterations the reference executes: up to and including the first blocked sample
      iters = blocked ? (int)fb + 1 : (int)s_tc[n];
    }
    double J = 0;
#pragma unroll
    for (int i = 0; i < D; i++) J += u[i] * u[i] * T;  // Primitive::J of a forward primitive (see expand_kernel.hip)
    const double cost = blocked ? INFINITY : 0.0 + (J + A.w * A.dt);
    const int64_t idx = (node0 + nl) * A.l_nstride + j;
    if (A.l_cost) __builtin_nontemporal_store(cost, &A.l_cost[idx]);
    if (A.l_iters) __builtin_nontemporal_store(iters, &A.l_iters[idx]);
  }
  // ---- successors per node
  for (int nl = tid; nl < nn; nl += kBT)
    if (A.l_count) A.l_count[node0 + nl] = s_ncnt[nl];
  }  // tile loop
  if (!SVC) break;
  // Every store of this workgroup must have been ACKNOWLEDGED before wave 0 writes back what the L2 still holds and
  // reports the request.  The barrier alone does not give that: on gfx9 a workgroup-scope release only waits for
  // lgkmcnt (hipcc emits `global_store ... s_waitcnt lgkmcnt(0) ; s_barrier`), so every wave drains its own vmcnt
  // first -- an s_waitcnt, no cache maintenance.  (A system-scope fence in every wave was 8 x 64 L2 write-backs a
  // request: + 7 us on a one-node request of the 729-control table.)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (wave0) {
    if ((int64_t)blockIdx.x * A.npb < n_nodes) __threadfence_system();
    if (blockIdx.x != 0) {
      __hip_atomic_store(&A.svc_dev[1 + blockIdx.x], (uint64_t)svc_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      const int lane = tid & 63;
      for (;;) {  // lane l watches workgroups l + 1, l + 65, ...
        bool all = true;
        for (unsigned g = 1 + lane; g < gridDim.x; g += 64)
          all = all && __hip_atomic_load(&A.svc_dev[1 + g], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == (uint64_t)svc_seq;
        if (__ballot(!all) == 0ull) break;
      }
      __hip_atomic_store(&A.svc_mb->done, (uint64_t)svc_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  __syncthreads();
  }  // request loop
  if (!SVC && A.done.flag != nullptr) {  // see DoneSignal (mplx_internal.h); no barrier follows: one lane may act alone
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores are acknowledged (the barrier waits for lgkmcnt only)
    __syncthreads();
    if (tid == 0) {
      __threadfence_system();
      if (__hip_atomic_fetch_add(A.done.count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u) {
        __hip_atomic_store(A.done.count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(A.done.flag, A.done.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}
#undef A

// Sequentially accumulated sample times: row n holds t_0 .. t_{cnt-1} of
// `for (t = 0; t < T; t += T/n)` and tcnt[n] the number of iterations.
__global__ void make_tables_kernel(double T, double res, double *ttab, unsigned char *tcnt, double *recips) {
  const int n = threadIdx.x;
  if (n == 0) {
    recips[0] = refined_rcp(res);
    recips[1] = refined_rcp(0.01);
    recips[2] = refined_rcp(0.1);
  }
  if (n < 1 || n > kMaxN) { if (n == 0) tcnt[0] = 0; return; }
  const double sdt = T / n;
  int k = 0;
  for (double t = 0; t < T && k < kTabStride; t += sdt) ttab[n * kTabStride + k++] = t;
  tcnt[n] = (unsigned char)k;
}

template <int D, int K, bool ONE, bool SVC>
hipError_t launch_tile_inst(const TileArgs &a, hipStream_t stream) {
  const int64_t n_tiles = (a.n_nodes + a.npb - 1) / a.npb;
  // service mode: a.n_nodes is the CAPACITY of a request, a.grid_limit the resident workgroups (all of them must fit
  // on the device at once: each waits for the others)
  const int64_t blocks = n_tiles < (int64_t)a.grid_limit ? n_tiles : (int64_t)a.grid_limit;
  const size_t lds = tile_lds_bytes(a.tile_pairs, a.npb, a.wl_cap, a.n_max, 4 * D + 2, a.nU * a.udim, nullptr);
  static bool attr_set[64] = {};  // per device: see expand_grid_kernel.hip
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 63;
  if (!attr_set[dev] || dev == 63) {
    hipError_t e = hipFuncSetAttribute((const void *)expand_tile_kernel<D, K, ONE, SVC>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024 - (SVC ? 64 : 0));  // (the service form has a static command word)
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL((expand_tile_kernel<D, K, ONE, SVC>), dim3((unsigned)blocks), dim3(kBT), lds, stream, a);
  return hipGetLastError();
}

template <int D, int K>
hipError_t launch_tile_one(const TileArgs &a, hipStream_t stream) {
  if (a.n_nodes == 0) return hipSuccess;
  if (a.svc_mb) {
    if (a.npb == 1) return launch_tile_inst<D, K, true, true>(a, stream);
    return launch_tile_inst<D, K, false, true>(a, stream);
  }
  if (a.npb == 1) return launch_tile_inst<D, K, true, false>(a, stream);
  return launch_tile_inst<D, K, false, false>(a, stream);
}

}  // namespace

size_t tile_lds_bytes(int tile_pairs, int npb, int wl_cap, int n_max, int n_fields, int u_doubles,
                      int *u_offset) {
  size_t b = 0;
  b += (size_t)npb * 8;                           // s_hcur
  b += (size_t)(n_max + 1) * (n_max + 1) * 8;     // s_tt
  b += (npb == 1) ? 0 : (size_t)npb * n_fields * 8;  // s_node
  b += (size_t)tile_pairs * 4;                    // s_fb
  b += 32 * 4;                                    // s_misc
  b += (size_t)npb * 4 * 2;                       // s_ncnt, s_nbase
  b += (size_t)tile_pairs * 2 * 2;                // s_vlist, s_j
  b += (size_t)wl_cap * 2;                        // s_wl
  b += (size_t)tile_pairs;                        // s_n
  b += 64;                                        // s_tc
  b = (b + 15) & ~(size_t)15;
  if (u_offset) *u_offset = (int)b;
  b += (size_t)u_doubles * 8;                     // s_U
  return (b + 15) & ~(size_t)15;
}

hipError_t launch_make_tables(double T, double res, double *ttab, unsigned char *tcnt, double *recips,
                              hipStream_t stream) {
  hipLaunchKernelGGL(make_tables_kernel, dim3(1), dim3(64), 0, stream, T, res, ttab, tcnt, recips);
  return hipGetLastError();
}

// Workgroups of the service form of (dim, control, a.npb) the whole device keeps resident at once with a's LDS size, as
// the runtime accounts registers, LDS granules and wave slots (0: it cannot tell).  The resident kernel's workgroups wait
// for each other, so svc_launch asks before it launches more than that.
template <int D, int K>
int tile_svc_resident(const TileArgs &a) {
  const size_t lds = tile_lds_bytes(a.tile_pairs, a.npb, a.wl_cap, a.n_max, 4 * D + 2, a.nU * a.udim, nullptr);
  const void *fn = a.npb == 1 ? (const void *)expand_tile_kernel<D, K, true, true> : (const void *)expand_tile_kernel<D, K, false, true>;
  int dev = 0, cus = 0, nb = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, kBT, lds) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return nb * cus;
}

int tile_service_resident_workgroups(int dim, int control, const TileArgs &a) {
  if (dim == 2) {
    switch (control) {
      case 0x01: return tile_svc_resident<2, 1>(a);
      case 0x03: return tile_svc_resident<2, 2>(a);
      case 0x07: return tile_svc_resident<2, 3>(a);
      case 0x0f: return tile_svc_resident<2, 4>(a);
    }
  } else if (dim == 3) {
    switch (control) {
      case 0x01: return tile_svc_resident<3, 1>(a);
      case 0x03: return tile_svc_resident<3, 2>(a);
      case 0x07: return tile_svc_resident<3, 3>(a);
      case 0x0f: return tile_svc_resident<3, 4>(a);
    }
  }
  return 0;
}

hipError_t launch_expand_tile(int dim, int control, const TileArgs &a, hipStream_t s) {
  if (dim == 2) {
    switch (control) {
      case 0x01: return launch_tile_one<2, 1>(a, s);
      case 0x03: return launch_tile_one<2, 2>(a, s);
      case 0x07: return launch_tile_one<2, 3>(a, s);
      case 0x0f: return launch_tile_one<2, 4>(a, s);
    }
  } else if (dim == 3) {
    switch (control) {
      case 0x01: return launch_tile_one<3, 1>(a, s);
      case 0x03: return launch_tile_one<3, 2>(a, s);
      case 0x07: return launch_tile_one<3, 3>(a, s);
      case 0x0f: return launch_tile_one<3, 4>(a, s);
    }
  }
  return hipErrorInvalidValue;
}

}  // namespace mplx
