// identity_kernel.hip -- node identity of a whole batch of successors at memory bandwidth.
//
// What the graph search does with every successor right after get_succ is a hash-map look-up keyed by the
// Waypoint, whose == compares lattice hashes (reference include/mpl_basis/waypoint.h:128-135,
// include/mpl_planner/common/graph_search.h:84-88).  Batched: for every emitted successor g of the lists,
//   canon[g] = the smallest list index g' whose successor has the same lattice hash
// -- the same contract as the open-addressing table of post_kernel.hip (one 64-bit CAS + one atomicMin per successor
// into ONE table in HBM: 20 M successors = 20 M random 16-byte read-modify-writes, 2.8 ms on C4, 5.6 x the expansion
// that produced them).  Here the random accesses happen in LDS instead:
//
//   1. radix partition of the (hash, index) pairs by the top bits of a mixed hash: at most two scatter passes
//      (64 coarse buckets, then up to 256 fine buckets inside each), every pass = per-tile histogram, prefix sum,
//      scatter staged through LDS so that a tile's share of a bucket leaves as one contiguous run;
//   2. one workgroup per fine bucket (~1 k pairs): open-addressing table in LDS (64-bit ds_cmpst for the key,
//      ds_min for the index), then canon[] of the bucket's pairs from the same table.
//
// Equal hashes always meet in the same bucket, so the result is the global table's, bit for bit; it does not depend
// on any ordering (min over a set).  A bucket with more distinct keys than its LDS table holds (hash skew, or far more
// successors than the partition was sized for) is processed in 2, 4, 8 ... rounds over further hash bits, so the
// kernel is correct for every input and only slower for adversarial ones.
//
// HBM traffic per successor: 8 B hash read twice (histogram + scatter), 12 B written + read per partition level,
// 12 B read + a scattered 4 B written by the tables: ~70 B against ~130 B of random sector traffic before.
#include "mplx_internal.h"

namespace mplx {
namespace {

constexpr uint64_t kEmpty = ~0ull;
constexpr int kTile = 4096;   // pairs (or list slots) per workgroup and pass
constexpr int kBT = 256;
constexpr int kPer = kTile / kBT;
constexpr int kSlots = 4096;  // LDS table of one fine bucket
constexpr int kFill = 3400;   // distinct keys a round may hold before the bucket is split further (IdentityArgs::fill)

__device__ __forceinline__ uint64_t mix(uint64_t h) {  // bucket / slot selection only; never leaves the device
  h ^= h >> 33;
  h *= 0xff51afd7ed558ccdULL;
  h ^= h >> 33;
  h *= 0xc4ceb9fe1a85ec53ULL;
  h ^= h >> 33;
  return h;
}

// value of the exclusive prefix sum at flat counter index i: within its 4096-block + the scanned block totals
__device__ __forceinline__ uint32_t scanned(const uint32_t *cnt, const uint32_t *tot, int64_t i) {
  return cnt[i] + tot[i >> 12];
}

// ---- prefix sums over the (bucket-major, tile-minor) counters
__global__ __launch_bounds__(kBT) void id_scan_blocks_kernel(uint32_t *cnt, uint32_t *tot) {
  __shared__ uint32_t part[kBT];
  uint32_t *p = cnt + (int64_t)blockIdx.x * kTile + threadIdx.x * kPer;
  uint32_t v[kPer], s = 0;
#pragma unroll
  for (int i = 0; i < kPer; i++) { v[i] = p[i]; s += v[i]; }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int d = 1; d < kBT; d <<= 1) {
    const uint32_t o = threadIdx.x >= (unsigned)d ? part[threadIdx.x - d] : 0u;
    __syncthreads();
    part[threadIdx.x] += o;
    __syncthreads();
  }
  uint32_t run = part[threadIdx.x] - s;
#pragma unroll
  for (int i = 0; i < kPer; i++) { p[i] = run; run += v[i]; }
  if (threadIdx.x == kBT - 1) tot[blockIdx.x] = part[kBT - 1];
}
// exclusive scan of the block totals in place, one workgroup; tot[n] = grand total
__global__ __launch_bounds__(1024) void id_scan_totals_kernel(uint32_t *tot, int n) {
  __shared__ uint32_t part[1024];
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int i = base + threadIdx.x;
    const uint32_t v = i < n ? tot[i] : 0u;
    part[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      const uint32_t o = threadIdx.x >= (unsigned)d ? part[threadIdx.x - d] : 0u;
      __syncthreads();
      part[threadIdx.x] += o;
      __syncthreads();
    }
    if (i < n) tot[i] = carry + part[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += part[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) tot[n] = carry;
}

// ---- one partition pass.  LEVEL 1 reads the lists (strided slots, only j < count[node] carry a successor; packed
// lists are one node), LEVEL 2 reads level 1's output segment by segment (a tile never straddles two coarse buckets).
struct TileSrc {
  int64_t lo, hi;      // slots (level 1) or pairs (level 2) of this tile
  int64_t ctr0;        // flat counter index of (digit 0, this tile)
  int64_t ctr_stride;  // ... and the distance between consecutive digits
  int shift, nb;       // digit = (mix(h) >> shift) & (nb - 1)
  bool ok;
};

template <int LEVEL>
__device__ __forceinline__ TileSrc tile_of(const IdentityArgs &A, const uint32_t *s_seg) {
  TileSrc t{};
  if (LEVEL == 1) {
    t.lo = (int64_t)blockIdx.x * kTile;
    t.hi = t.lo + kTile < A.n_slots ? t.lo + kTile : A.n_slots;
    t.ctr0 = blockIdx.x;
    t.ctr_stride = A.tiles1;
    t.shift = 64 - A.b1;
    t.nb = 1 << A.b1;
    t.ok = true;
  } else {
    const int nb1 = 1 << A.b1;
    const uint32_t *start = s_seg, *tpre = s_seg + nb1 + 1;
    t.ok = blockIdx.x < tpre[nb1];
    int c = 0;
    for (int k = 0; k < nb1; k++) c = (t.ok && blockIdx.x >= tpre[k]) ? k : c;  // tpre is non-decreasing: the last k with tpre[k] <= block
    const int tl = (int)(blockIdx.x - tpre[c]);
    t.lo = (int64_t)start[c] + (int64_t)tl * kTile;
    t.hi = t.lo + kTile < (int64_t)start[c + 1] ? t.lo + kTile : (int64_t)start[c + 1];
    t.nb = 1 << A.b2;
    t.shift = 64 - A.b1 - A.b2;
    t.ctr_stride = (int64_t)(tpre[c + 1] - tpre[c]);
    t.ctr0 = (int64_t)tpre[c] * t.nb + tl;
  }
  return t;
}

// pair `i` of the tile: hash, list index, valid?
template <int LEVEL>
__device__ __forceinline__ bool load_pair(const IdentityArgs &A, int64_t s, int64_t hi, uint64_t *h, uint32_t *g) {
  if (s >= hi) return false;
  if (LEVEL == 1) {
    if (A.n_nodes == 1) {
      if (s >= (int64_t)A.count[0]) return false;
    } else {
      const uint32_t node = (uint32_t)s / (uint32_t)A.nstride;  // n_slots < 2^31 (checked by the host)
      const uint32_t j = (uint32_t)s - node * (uint32_t)A.nstride;
      if ((int)j >= A.count[node]) return false;
    }
    *h = A.hash[s];
    *g = (uint32_t)s;
  } else {
    *h = A.hk[0][s];
    *g = A.gi[0][s];
  }
  return true;
}

__device__ __forceinline__ void load_seg(const IdentityArgs &A, uint32_t *s_seg) {
  const int n = 2 * ((1 << A.b1) + 1);
  for (int i = threadIdx.x; i < n; i += kBT) s_seg[i] = A.seg[i];
  __syncthreads();
}

template <int LEVEL>
__global__ __launch_bounds__(kBT) void id_hist_kernel(const IdentityArgs A) {
  __shared__ uint32_t s_seg[2 * 65 + 2];
  __shared__ uint32_t hist[256];
  if (LEVEL == 2) load_seg(A, s_seg);
  const TileSrc t = tile_of<LEVEL>(A, s_seg);
  if (!t.ok) return;  // (uniform)
  hist[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kPer; i++) {
    uint64_t h;
    uint32_t g;
    if (load_pair<LEVEL>(A, t.lo + i * kBT + threadIdx.x, t.hi, &h, &g))
      atomicAdd(&hist[(mix(h) >> t.shift) & (t.nb - 1)], 1u);
  }
  __syncthreads();
  if ((int)threadIdx.x < t.nb) A.cnt[LEVEL - 1][t.ctr0 + threadIdx.x * t.ctr_stride] = hist[threadIdx.x];
}

template <int LEVEL>
__global__ __launch_bounds__(kBT) void id_scatter_kernel(const IdentityArgs A) {
  __shared__ uint32_t s_seg[2 * 65 + 2];
  __shared__ uint32_t hist[256], lbase[256], gbase[256];
  __shared__ uint64_t st_h[kTile];
  __shared__ uint32_t st_g[kTile];
  if (LEVEL == 2) load_seg(A, s_seg);
  const TileSrc t = tile_of<LEVEL>(A, s_seg);
  if (!t.ok) return;
  hist[threadIdx.x] = 0;
  __syncthreads();
  uint64_t h[kPer];
  uint32_t g[kPer], rk[kPer];  // rk: rank inside the tile's share of the digit | digit << 16; ~0 = no pair
#pragma unroll
  for (int i = 0; i < kPer; i++) {
    rk[i] = 0xffffffffu;
    if (load_pair<LEVEL>(A, t.lo + i * kBT + threadIdx.x, t.hi, &h[i], &g[i])) {
      const uint32_t d = (uint32_t)((mix(h[i]) >> t.shift) & (uint64_t)(t.nb - 1));
      rk[i] = atomicAdd(&hist[d], 1u) | (d << 16);
    }
  }
  __syncthreads();
  {  // exclusive scan of the 256 digit counts; where the tile's share of every digit starts in the output
    const uint32_t v = hist[threadIdx.x];
    lbase[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < kBT; d <<= 1) {
      const uint32_t o = threadIdx.x >= (unsigned)d ? lbase[threadIdx.x - d] : 0u;
      __syncthreads();
      lbase[threadIdx.x] += o;
      __syncthreads();
    }
    const uint32_t excl = lbase[threadIdx.x] - v;
    __syncthreads();
    lbase[threadIdx.x] = excl;
    if ((int)threadIdx.x < t.nb)
      gbase[threadIdx.x] = scanned(A.cnt[LEVEL - 1], A.tot[LEVEL - 1], t.ctr0 + threadIdx.x * t.ctr_stride);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kPer; i++) {
    if (rk[i] != 0xffffffffu) {
      const uint32_t p = lbase[rk[i] >> 16] + (rk[i] & 0xffffu);
      st_h[p] = h[i];
      st_g[p] = g[i];
    }
  }
  __syncthreads();
  const uint32_t total = lbase[255] + hist[255];
  uint64_t *oh = A.hk[LEVEL == 1 ? 0 : 1];
  uint32_t *og = A.gi[LEVEL == 1 ? 0 : 1];
  for (uint32_t p = threadIdx.x; p < total; p += kBT) {  // consecutive lanes -> consecutive addresses inside a digit's run
    const uint64_t hh = st_h[p];
    const uint32_t d = (uint32_t)((mix(hh) >> t.shift) & (uint64_t)(t.nb - 1));
    const uint32_t o = gbase[d] + (p - lbase[d]);
    oh[o] = hh;
    og[o] = st_g[p];
  }
}

// the coarse buckets as segments of level 2: start[c] (c = 0 .. nb1), tile prefix tpre[c] (c = 0 .. nb1)
__global__ __launch_bounds__(128) void id_segments_kernel(const IdentityArgs A) {
  __shared__ uint32_t start[66], nt[66];
  const int nb1 = 1 << A.b1;
  const int c = threadIdx.x;
  if (c <= nb1) start[c] = scanned(A.cnt[0], A.tot[0], (int64_t)c * A.tiles1);  // c == nb1: one past the counters = the total
  __syncthreads();
  if (c < nb1) nt[c] = (start[c + 1] - start[c] + kTile - 1) / kTile;
  __syncthreads();
  if (c == 0) {
    uint32_t run = 0;
    for (int k = 0; k < nb1; k++) {
      A.seg[nb1 + 1 + k] = run;
      run += nt[k];
    }
    A.seg[nb1 + 1 + nb1] = run;
  }
  if (c <= nb1) A.seg[c] = start[c];
}

// ---- one workgroup per fine bucket: identity table in LDS
__global__ __launch_bounds__(kBT) void id_tables_kernel(const IdentityArgs A) {
  __shared__ unsigned long long keys[kSlots];
  __shared__ uint32_t vals[kSlots];
  __shared__ uint32_t nuniq, ovf, special;
  __shared__ uint32_t s_seg[2 * 65 + 2];
  int64_t lo, hi;
  const uint64_t *hk;
  const uint32_t *gi;
  if (A.b2 > 0) {
    load_seg(A, s_seg);
    const int nb2 = 1 << A.b2;
    const int c = blockIdx.x >> A.b2, d = blockIdx.x & (nb2 - 1);
    const uint32_t *tpre = s_seg + (1 << A.b1) + 1;
    const int64_t ntile = (int64_t)(tpre[c + 1] - tpre[c]);
    const int64_t i0 = (int64_t)tpre[c] * nb2 + d * ntile;
    lo = scanned(A.cnt[1], A.tot[1], i0);
    hi = scanned(A.cnt[1], A.tot[1], i0 + ntile);  // the next bucket's first counter (zeroed padding after the last)
    hk = A.hk[1];
    gi = A.gi[1];
  } else {
    lo = scanned(A.cnt[0], A.tot[0], (int64_t)blockIdx.x * A.tiles1);
    hi = scanned(A.cnt[0], A.tot[0], (int64_t)(blockIdx.x + 1) * A.tiles1);
    hk = A.hk[0];
    gi = A.gi[0];
  }
  if (hi <= lo) return;
  // rounds over further hash bits when the bucket holds more distinct keys than the table takes
  for (uint32_t R = 1;; R <<= 1) {
    bool split = false;
    for (uint32_t r = 0; r < R; r++) {
      for (int i = threadIdx.x; i < kSlots; i += kBT) { keys[i] = kEmpty; vals[i] = 0xffffffffu; }
      if (threadIdx.x == 0) { nuniq = 0; ovf = 0; special = 0xffffffffu; }
      __syncthreads();
      for (int64_t p = lo + threadIdx.x; p < hi; p += kBT) {
        const uint64_t h = hk[p];
        const uint64_t m = mix(h);
        if (((uint32_t)(m >> 12) & (R - 1u)) != r) continue;
        const uint32_t g = gi[p];
        if (h == kEmpty) { atomicMin(&special, g); continue; }  // the one hash the key field cannot hold
        uint32_t s = (uint32_t)m & (kSlots - 1);
        for (int probes = 0;; probes++) {
          const unsigned long long old = atomicCAS(&keys[s], (unsigned long long)kEmpty, (unsigned long long)h);
          if (old == kEmpty || old == h) {
            atomicMin(&vals[s], g);
            if (old == kEmpty && atomicAdd(&nuniq, 1u) >= (uint32_t)A.fill) ovf = 1;
            break;
          }
          if (probes >= 1024) { ovf = 1; break; }
          s = (s + 1) & (kSlots - 1);
        }
      }
      __syncthreads();
      const bool over = ovf != 0 && R < (1u << 20);  // (uniform)
      __syncthreads();                              // everyone has read `ovf` before the next round clears it
      if (over) { split = true; break; }            // split further and start the bucket over; canon writes are idempotent
      for (int64_t p = lo + threadIdx.x; p < hi; p += kBT) {
        const uint64_t h = hk[p];
        const uint64_t m = mix(h);
        if (((uint32_t)(m >> 12) & (R - 1u)) != r) continue;
        const uint32_t g = gi[p];
        uint32_t c;
        if (h == kEmpty) {
          c = special;
        } else {
          uint32_t s = (uint32_t)m & (kSlots - 1);
          int probes = 0;
          while (keys[s] != h && probes++ <= 1024) s = (s + 1) & (kSlots - 1);
          c = keys[s] == h ? vals[s] : g;  // (a key that was not inserted: only past the 2^20-round limit)
        }
        A.canon[g] = (int32_t)c;
      }
      __syncthreads();
    }
    if (!split) break;
  }
}

}  // namespace

// bits of the two partition levels for `n_slots` list slots: fine buckets of ~1 k pairs
void identity_plan(int64_t n_slots, int *b1, int *b2) {
  int bits = 0;
  while (bits < 14 && ((int64_t)1024 << bits) < n_slots) bits++;
  if (bits <= 8) { *b1 = bits; *b2 = 0; }
  else { *b1 = 6; *b2 = bits - 6; }
}

// sizes of the workspace for `n_slots` slots: counters (uint32) of level 1 / level 2 incl. padding, block totals
void identity_sizes(int64_t n_slots, int b1, int b2, int64_t *tiles1, int64_t *tiles2_cap, int64_t *ctr1, int64_t *ctr2) {
  *tiles1 = (n_slots + kTile - 1) / kTile;
  *tiles2_cap = b2 ? *tiles1 + (1 << b1) : 0;
  auto pad = [](int64_t n) { return ((n + 1 + kTile - 1) / kTile) * kTile; };  // + 1: the "one past" read; whole scan blocks
  *ctr1 = pad((int64_t)(1 << b1) * *tiles1);
  *ctr2 = b2 ? pad((int64_t)(1 << b2) * *tiles2_cap) : 0;
}

int identity_default_fill() { return kFill; }

hipError_t launch_identity(const IdentityArgs &a, int64_t ctr1, int64_t ctr2, hipStream_t s) {
  if (a.n_slots <= 0) return hipSuccess;
  hipError_t e;
  if ((e = hipMemsetAsync(a.cnt[0], 0, (size_t)ctr1 * 4, s)) != hipSuccess) return e;
  hipLaunchKernelGGL(id_hist_kernel<1>, dim3((unsigned)a.tiles1), dim3(kBT), 0, s, a);
  hipLaunchKernelGGL(id_scan_blocks_kernel, dim3((unsigned)(ctr1 / kTile)), dim3(kBT), 0, s, a.cnt[0], a.tot[0]);
  hipLaunchKernelGGL(id_scan_totals_kernel, dim3(1), dim3(1024), 0, s, a.tot[0], (int)(ctr1 / kTile));
  hipLaunchKernelGGL(id_scatter_kernel<1>, dim3((unsigned)a.tiles1), dim3(kBT), 0, s, a);
  unsigned buckets = 1u << a.b1;
  if (a.b2 > 0) {
    if ((e = hipMemsetAsync(a.cnt[1], 0, (size_t)ctr2 * 4, s)) != hipSuccess) return e;
    hipLaunchKernelGGL(id_segments_kernel, dim3(1), dim3(128), 0, s, a);
    hipLaunchKernelGGL(id_hist_kernel<2>, dim3((unsigned)a.tiles2_cap), dim3(kBT), 0, s, a);
    hipLaunchKernelGGL(id_scan_blocks_kernel, dim3((unsigned)(ctr2 / kTile)), dim3(kBT), 0, s, a.cnt[1], a.tot[1]);
    hipLaunchKernelGGL(id_scan_totals_kernel, dim3(1), dim3(1024), 0, s, a.tot[1], (int)(ctr2 / kTile));
    hipLaunchKernelGGL(id_scatter_kernel<2>, dim3((unsigned)a.tiles2_cap), dim3(kBT), 0, s, a);
    buckets <<= a.b2;
  }
  hipLaunchKernelGGL(id_tables_kernel, dim3(buckets), dim3(kBT), 0, s, a);
  return hipGetLastError();
}

}  // namespace mplx
