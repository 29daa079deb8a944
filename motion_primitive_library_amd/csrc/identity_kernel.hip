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
//   1. radix partition of the (key, index) pairs by the top bits of a mixed hash: at most two scatter passes (64 coarse
//      buckets, then up to 256 fine buckets inside each), staged through LDS so that a tile's share of a bucket leaves
//      as one contiguous run.  Two forms:
//        exact    per pass a per-tile histogram and a prefix sum place every run (round 3; no capacities: every input);
//        claimed  buckets of fixed capacity, a tile claims its run with one returning atomic per digit; no histogram
//                 passes, the key mixed once (round 4: the default; an overflowing bucket makes the caller fall back);
//   2. one workgroup per fine bucket (~1 k pairs): open-addressing table in LDS (64-bit ds_cmpst for the key,
//      ds_min for the index), then canon[] of the bucket's pairs from the same table.
//
// Equal hashes always meet in the same bucket, so the result is the global table's, bit for bit; it does not depend
// on any ordering (min over a set).  A bucket with more distinct keys than its LDS table holds (hash skew, or far more
// successors than the partition was sized for) is processed in 2, 4, 8 ... rounds over further hash bits, so the
// kernel is correct for every input and only slower for adversarial ones.
//
// HBM traffic per successor, exact form: 8 B hash read twice per level (histogram + scatter), 12 B written + read per
// level, 12 B read + a scattered 4 B written by the tables: ~76 B; claimed form: 8 B read + 4 B canon + 12 B written,
// 12 + 12 B at level 2, 12 B by the tables: ~60 B -- against ~130 B of random sector traffic of the table in HBM.
#include "mplx_internal.h"

namespace mplx {
namespace {

constexpr uint64_t kEmpty = ~0ull;
constexpr int kTile = 4096;   // pairs (or list slots) per workgroup and pass
constexpr int kBT = 256;
constexpr int kPer = kTile / kBT;
constexpr int kSlots = 2048;  // LDS table of one fine bucket: keys + indices 24 KB
constexpr int kSlotsKeys = 3328;  // ... and the keys-only table of the first sweep over the same LDS: 26 KB, still six workgroups per CU
constexpr int kFill = 1500;   // distinct keys a round may hold before the bucket is split further (IdentityArgs::fill)
constexpr int kMaxProbe = 192; // longest probe sequence of an insert before the round is declared overflowed
constexpr int kChunk = 2048;  // pairs a workgroup has in flight at once in the table kernel (8 per thread)

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
// ---- one partition pass.  LEVEL 1 reads the lists (strided slots, only j < count[node] carry a successor; packed
// lists are one node), LEVEL 2 reads level 1's output segment by segment (a tile never straddles two coarse buckets).
struct TileSrc {
  int64_t lo, hi;      // slots (level 1) or pairs (level 2) of this tile
  int64_t ctr0;        // flat counter index of (digit 0, this tile)
  int64_t ctr_stride;  // ... and the distance between consecutive digits
  int shift, nb;       // digit = (mix(h) >> shift) & (nb - 1)
  bool ok;
  uint32_t node0, r0;  // level 1: node of the tile's first slot and that slot's offset inside the node
  float inv;           // level 1: 1 / nstride
};

// Workgroup -> tile, XCD-aware: the hardware deals workgroups round-robin over the 8 XCDs, each with its own L2.
// Tiles that follow each other write ADJACENT runs of every bucket (a run is ~100 - 200 bytes: a partial 128-byte line
// at both ends), so consecutive tiles must meet in the same L2 for those lines to be completed there instead of
// leaving eight L2s as partial-line writes: XCD x takes the x-th contiguous eighth of the tiles.
__device__ __forceinline__ uint32_t xcd_tile(uint32_t n_tiles) {
  const uint32_t per = (n_tiles + 7u) >> 3;
  return (blockIdx.x & 7u) * per + (blockIdx.x >> 3);  // >= n_tiles for the surplus workgroups of the rounded-up grid
}

template <int LEVEL>
__device__ __forceinline__ TileSrc tile_of(const IdentityArgs &A, const uint32_t *s_seg) {
  TileSrc t{};
  if (LEVEL == 1) {
    const uint32_t tile = xcd_tile((uint32_t)A.tiles1);
    if (tile >= (uint32_t)A.tiles1 || (blockIdx.x >> 3) >= (((uint32_t)A.tiles1 + 7u) >> 3)) { t.ok = false; return t; }
    t.lo = (int64_t)tile * kTile;
    t.hi = t.lo + kTile < A.n_slots ? t.lo + kTile : A.n_slots;
    t.ctr0 = tile;
    t.ctr_stride = A.tiles1;
    t.shift = A.b1 ? 64 - A.b1 : 0;  // (one bucket: the digit is masked to 0 whatever the shift)
    t.nb = 1 << A.b1;
    t.ok = true;
    t.node0 = (uint32_t)(t.lo / A.nstride);  // (uniform: one scalar division per workgroup)
    t.r0 = (uint32_t)(t.lo - (int64_t)t.node0 * A.nstride);
    t.inv = 1.0f / (float)A.nstride;
  } else {
    const int nb1 = 1 << A.b1;
    const uint32_t *start = s_seg, *tpre = s_seg + nb1 + 1;
    const uint32_t n_tiles = tpre[nb1], tile = xcd_tile(n_tiles);
    t.ok = tile < n_tiles && (blockIdx.x >> 3) < ((n_tiles + 7u) >> 3);
    int c = 0;
    for (int k = 0; k < nb1; k++) c = (t.ok && tile >= tpre[k]) ? k : c;  // tpre is non-decreasing: the last k with tpre[k] <= tile
    const int tl = (int)(tile - tpre[c]);
    t.lo = (int64_t)start[c] + (int64_t)tl * kTile;
    t.hi = t.lo + kTile < (int64_t)start[c + 1] ? t.lo + kTile : (int64_t)start[c + 1];
    t.nb = 1 << A.b2;
    t.shift = 64 - A.b1 - A.b2;
    t.ctr_stride = (int64_t)(tpre[c + 1] - tpre[c]);
    t.ctr0 = (int64_t)tpre[c] * t.nb + tl;
  }
  return t;
}

// The tile's pairs, kPer per thread, ALL loads in flight together: two dependent batches at level 1 (the counts of the
// slots' nodes, then the hashes of the slots that carry a successor), one at level 2.  Every load is unconditional on a
// clamped address so that the compiler issues the whole batch before the first use (a conditional load per slot -- the
// first version -- compiled to kPer serial round trips).  h[i] is only meaningful where ok bit i is set.
template <int LEVEL>
__device__ __forceinline__ uint32_t load_tile(const IdentityArgs &A, const TileSrc &t, uint64_t (&h)[kPer]) {
  uint32_t ok = 0;
  if (LEVEL == 1) {
    if (A.n_nodes == 1) {  // packed lists: one long list
      const int64_t total = (int64_t)A.count[0];
#pragma unroll
      for (int i = 0; i < kPer; i++) ok |= ((t.lo + i * kBT + (int64_t)threadIdx.x) < (t.hi < total ? t.hi : total) ? 1u : 0u) << i;
    } else {
      int cn[kPer];
      uint32_t jj[kPer];
      const bool wide = A.nstride >= kBT;  // (uniform) a row of kBT slots crosses at most one node boundary
      uint32_t q_run = 0, j_run = 0;
#pragma unroll
      for (int i = 0; i < kPer; i++) {
        uint32_t q;
        if (i == 0 || !wide) {
          // node = slot / nstride without a division per slot: x = (offset of the tile inside its first node) + (slot
          // inside the tile) < nstride + 4096 < 2^24 is exact in f32, so the f32 quotient is off by at most one
          const uint32_t x = t.r0 + (uint32_t)(i * kBT + threadIdx.x);
          q = (uint32_t)((float)x * t.inv);
          if (q * (uint32_t)A.nstride > x) q--;
          else if ((q + 1) * (uint32_t)A.nstride <= x) q++;
          jj[i] = x - q * (uint32_t)A.nstride;
        } else {  // ... and without the quotient from the second row on: kBT slots further, at most one node further
          j_run += (uint32_t)kBT;
          const bool over = j_run >= (uint32_t)A.nstride;
          jj[i] = over ? j_run - (uint32_t)A.nstride : j_run;
          q = q_run + (over ? 1u : 0u);
        }
        q_run = q;
        j_run = jj[i];
        const int64_t node = (int64_t)t.node0 + q;
        cn[i] = A.count[node < A.n_nodes ? node : A.n_nodes - 1];
      }
#pragma unroll
      for (int i = 0; i < kPer; i++)
        ok |= ((t.lo + i * kBT + (int64_t)threadIdx.x < t.hi && (int)jj[i] < cn[i]) ? 1u : 0u) << i;
    }
#pragma unroll
    for (int i = 0; i < kPer; i++) h[i] = A.hash[((ok >> i) & 1u) ? t.lo + i * kBT + threadIdx.x : t.lo];
  } else {
#pragma unroll
    for (int i = 0; i < kPer; i++) {
      const int64_t p = t.lo + i * kBT + threadIdx.x;
      ok |= (p < t.hi ? 1u : 0u) << i;
      h[i] = A.hk[0][p < t.hi ? p : t.lo];
    }
  }
  return ok;
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
  uint64_t h[kPer];
  const uint32_t ok = load_tile<LEVEL>(A, t, h);
#pragma unroll
  for (int i = 0; i < kPer; i++) {
    if ((ok >> i) & 1u) {
      atomicAdd(&hist[(mix(h[i]) >> t.shift) & (t.nb - 1)], 1u);
      // every successor starts as its own first occurrence (coalesced, in list order); the table kernel then only
      // writes the duplicates -- on a frontier without revisits that is next to nothing instead of one scattered
      // 4-byte write per successor
      if (LEVEL == 1) A.canon[t.lo + i * kBT + threadIdx.x] = (int32_t)(t.lo + i * kBT + threadIdx.x);
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < t.nb) A.cnt[LEVEL - 1][t.ctr0 + threadIdx.x * t.ctr_stride] = hist[threadIdx.x];
}

// Scatter of one tile.  Staged through LDS so that consecutive lanes write consecutive addresses inside a digit's run.
// Level 1 (sparse tiles: ~40 % of the slots of padded lists carry a successor) stages only the pair's POSITION in the
// tile (2 bytes) and reads hash and index again on the way out: 8 KB of LDS instead of 48, eight resident workgroups per
// CU instead of three to overlap the kernel's dependent round trips (counts, hashes, bases, stores): 266 -> 154 us on
// C4.  Level 2 (dense tiles, 16 gathers per thread on the way out) stages the pairs themselves: 167 us against 309.
template <int LEVEL>
__global__ __launch_bounds__(kBT) void id_scatter_kernel(const IdentityArgs A) {
  __shared__ uint32_t s_seg[2 * 65 + 2];
  __shared__ uint32_t hist[256], lbase[256], gbase[256], wtot[4];
#ifndef MPLX_ID_L2_PAIRS
#define MPLX_ID_L2_PAIRS 1
#endif
  constexpr bool kStagePairs = LEVEL == 2 && MPLX_ID_L2_PAIRS;
  __shared__ unsigned short st_i[kStagePairs ? 1 : kTile];
  __shared__ uint64_t st_h[kStagePairs ? kTile : 1];
  __shared__ uint32_t st_g[kStagePairs ? kTile : 1];
  if (LEVEL == 2) load_seg(A, s_seg);
  const TileSrc t = tile_of<LEVEL>(A, s_seg);
  if (!t.ok) return;
  hist[threadIdx.x] = 0;
  __syncthreads();
  uint32_t rk[kPer];  // rank inside the tile's share of the digit | digit << 16
  uint64_t h[kPer];
  uint32_t g[kPer];
  const uint32_t ok = load_tile<LEVEL>(A, t, h);
  if (kStagePairs) {
#pragma unroll
    for (int i = 0; i < kPer; i++) {
      const int64_t p = t.lo + i * kBT + threadIdx.x;
      g[i] = A.gi[0][p < t.hi ? p : t.lo];
    }
  }
  const uint64_t *src_h = LEVEL == 1 ? A.hash : A.hk[0];
#pragma unroll
  for (int i = 0; i < kPer; i++) {
    rk[i] = 0;
    if ((ok >> i) & 1u) {
      const uint32_t d = (uint32_t)((mix(h[i]) >> t.shift) & (uint64_t)(t.nb - 1));
      rk[i] = atomicAdd(&hist[d], 1u) | (d << 16);
    }
  }
  __syncthreads();
  {  // exclusive scan of the 256 digit counts (one per thread): inside each wave by shuffles, the four wave totals
     // through LDS; where the tile's share of every digit starts in the output
    const uint32_t v = hist[threadIdx.x];
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64);
      if ((int)(threadIdx.x & 63) >= d) inc += o;
    }
    if ((threadIdx.x & 63) == 63) wtot[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t before = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) before += wtot[w];
    lbase[threadIdx.x] = before + inc - v;
    if ((int)threadIdx.x < t.nb)
      gbase[threadIdx.x] = scanned(A.cnt[LEVEL - 1], A.tot[LEVEL - 1], t.ctr0 + threadIdx.x * t.ctr_stride);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kPer; i++) {
    if ((ok >> i) & 1u) {
      const uint32_t p = lbase[rk[i] >> 16] + (rk[i] & 0xffffu);
      if (kStagePairs) { st_h[p] = h[i]; st_g[p] = g[i]; }
      else st_i[p] = (unsigned short)(i * kBT + threadIdx.x);
    }
  }
  __syncthreads();
  const uint32_t total = lbase[255] + hist[255];
  uint64_t *oh = A.hk[LEVEL == 1 ? 0 : 1];
  uint32_t *og = A.gi[LEVEL == 1 ? 0 : 1];
  if (kStagePairs) {
    for (uint32_t p = threadIdx.x; p < total; p += kBT) {  // consecutive lanes -> consecutive addresses inside a digit's run
      const uint64_t hh = st_h[p];
      const uint32_t d = (uint32_t)((mix(hh) >> t.shift) & (uint64_t)(t.nb - 1));
      const uint32_t o = gbase[d] + (p - lbase[d]);
      oh[o] = hh;
      og[o] = st_g[p];
    }
  } else {
    // the hashes are read again through the staged positions: all kPer re-reads of a thread in flight at once (a loop
    // with one dependent load per trip, the first version, spent a round trip to L2 per 256 pairs)
    uint64_t hh[kPer];
    uint32_t src[kPer];
#pragma unroll
    for (int i = 0; i < kPer; i++) {
      const uint32_t p = i * kBT + threadIdx.x;
      src[i] = p < total ? (uint32_t)st_i[p] : 0u;
      hh[i] = src_h[t.lo + src[i]];
    }
    uint32_t gg[LEVEL == 2 ? kPer : 1];
    if (LEVEL == 2) {
#pragma unroll
      for (int i = 0; i < kPer; i++) gg[i] = A.gi[0][t.lo + src[i]];
    }
#pragma unroll
    for (int i = 0; i < kPer; i++) {
      const uint32_t p = i * kBT + threadIdx.x;
      if (p < total) {
        const uint32_t d = (uint32_t)((mix(hh[i]) >> t.shift) & (uint64_t)(t.nb - 1));
        const uint32_t o = gbase[d] + (p - lbase[d]);
        oh[o] = hh[i];
        og[o] = LEVEL == 2 ? gg[LEVEL == 2 ? i : 0] : (uint32_t)(t.lo + src[i]);
      }
    }
  }
}

// ---- exclusive scan of the block totals of one level in place, ONE workgroup (tot[n] = grand total); at level 1 of
// two, in the same launch (a dependent single-workgroup dispatch costs ~5 us + the gap), the coarse buckets as segments
// of level 2: seg[c] = start[c] (c = 0 .. nb1), seg[nb1 + 1 + c] = tile prefix tpre[c]
__global__ __launch_bounds__(1024) void id_scan_totals_kernel(const IdentityArgs A, int level, int n) {
  __shared__ uint32_t part[1024];
  __shared__ uint32_t carry;
  __shared__ uint32_t start[66], nt[66];
  uint32_t *tot = A.tot[level - 1];
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
  __threadfence_block();
  __syncthreads();  // (this workgroup's own writes to tot[] are read back below by other threads of it)
  const int nb1 = 1 << A.b1;
  if (level == 1 && A.b2 > 0) {
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
}

// where every fine bucket starts in the partitioned pairs: range[b], b = 0 .. buckets (one thread per bucket), so that a
// table workgroup starts with ONE load instead of segment table + two scanned counters.  (Its own launch: 16 385 buckets
// through the ONE workgroup of the totals scan measured +10 us, 17 trips of dependent look-ups on one CU.)
__global__ __launch_bounds__(kBT) void id_ranges_kernel(const IdentityArgs A, uint32_t n_buckets) {
  const uint32_t b = blockIdx.x * kBT + threadIdx.x;
  if (b > n_buckets) return;
  uint32_t v;
  if (A.b2 > 0) {
    const int nb1 = 1 << A.b1, nb2 = 1 << A.b2;
    const uint32_t *tpre = A.seg + nb1 + 1;
    if (b == n_buckets) {
      v = scanned(A.cnt[1], A.tot[1], (int64_t)tpre[nb1] * nb2);  // one past the last counter in use (zeroed padding): the total
    } else {
      const int c = (int)(b >> A.b2), d = (int)(b & (uint32_t)(nb2 - 1));
      v = scanned(A.cnt[1], A.tot[1], (int64_t)tpre[c] * nb2 + (int64_t)d * (int64_t)(tpre[c + 1] - tpre[c]));
    }
  } else {
    v = scanned(A.cnt[0], A.tot[0], (int64_t)b * A.tiles1);
  }
  A.range[b] = v;
}

// ---- the claimed form: no histogram passes.  Level-1 buckets are split into 8 shards (workgroup index & 7 = the XCD the
// hardware's round-robin puts the workgroup on, and the eighth of the tiles xcd_tile() gives it): one cursor per (bucket,
// shard), 64 bytes apart -- a single device-scope word takes ~88 returning atomics per microsecond, 12 k tiles on 64
// words would queue for longer than the kernel runs.  A tile's pairs of one digit leave as one contiguous run at the
// claimed offset; a run that does not fit its bucket raises the flags and is dropped (the caller falls back).
constexpr int kShards = 8;
constexpr int kCurPad = 16;

__device__ __forceinline__ void raise_overflow(const IdentityArgs &A) {
  *(volatile uint32_t *)A.ovf = 1u;
  *(volatile int32_t *)A.ovf_host = 1;
}

// The pairs leave level 1 as (mix(hash), list index): the mix is a bijection, so equal keys stay equal and distinct ones
// distinct, and level 2 and the tables take their digits, slots and round bits from the stored key -- the two 64-bit
// multiplications of the mix (quarter-rate on this hardware; the kernels of the exact form spend most of their VALU
// time on them, twice per pass and successor, dead list slots included) happen ONCE per successor: rows of 64 slots
// without a successor skip them (58 % of C4's slots are padding), and the mixed key is staged through LDS (rounds of
// 2 048 pairs: one round unless the lists are nearly full) instead of being recomputed on the way out.
constexpr int kStage1 = 2048;

__global__ __launch_bounds__(kBT) void id_part1_kernel(const IdentityArgs A) {
  __shared__ uint32_t hist[64], lbase[64], gbase[64];
  __shared__ uint64_t st_m[kStage1];
  __shared__ unsigned short st_i[kStage1];
  const TileSrc t = tile_of<1>(A, nullptr);
  if (!t.ok) return;  // (uniform)
  if (threadIdx.x < 64) hist[threadIdx.x] = 0;
  __syncthreads();
  uint32_t rk[kPer];  // rank inside the tile's share of the digit | digit << 16
  uint64_t h[kPer];   // hash, then mixed key
  const uint32_t ok = load_tile<1>(A, t, h);
#pragma unroll
  for (int i = 0; i < kPer; i++) {
    rk[i] = 0;
    const bool live = (ok >> i) & 1u;
    if (__ballot(live) != 0ull) {  // (wave-uniform: a row of 64 padding slots costs nothing)
      if (live) {
        // every successor starts as its own first occurrence (coalesced, in list order); the table kernel only writes
        // the duplicates
        A.canon[t.lo + i * kBT + threadIdx.x] = (int32_t)(t.lo + i * kBT + threadIdx.x);
        h[i] = mix(h[i]);
        const uint32_t d = (uint32_t)((h[i] >> t.shift) & (uint64_t)(t.nb - 1));
        rk[i] = atomicAdd(&hist[d], 1u) | (d << 16);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < 64) {  // wave 0: where the tile's share of every digit starts in LDS and in its bucket
    const uint32_t v = hist[threadIdx.x];
    uint32_t gb = 0xffffffffu;
    if (v) {  // (issued before the scan: the claim's round trip overlaps it)
      const uint32_t seg = threadIdx.x * kShards + (blockIdx.x & (kShards - 1));
      const uint32_t o = atomicAdd(&A.cur1[seg * kCurPad], v);
      if ((int64_t)o + (int64_t)v <= A.subcap1) gb = (uint32_t)((int64_t)seg * A.subcap1) + o;
      else raise_overflow(A);
    }
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64);
      if ((int)threadIdx.x >= d) inc += o;
    }
    lbase[threadIdx.x] = inc - v;
    gbase[threadIdx.x] = gb;
  }
  __syncthreads();
  const uint32_t total = lbase[63] + hist[63];
  for (uint32_t base = 0; base < total; base += kStage1) {
#pragma unroll
    for (int i = 0; i < kPer; i++) {
      if ((ok >> i) & 1u) {
        const uint32_t p = lbase[rk[i] >> 16] + (rk[i] & 0xffffu) - base;
        if (p < (uint32_t)kStage1) {
          st_m[p] = h[i];
          st_i[p] = (unsigned short)(i * kBT + threadIdx.x);
        }
      }
    }
    __syncthreads();
    const uint32_t n = total - base < (uint32_t)kStage1 ? total - base : (uint32_t)kStage1;
    for (uint32_t q = threadIdx.x; q < n; q += kBT) {  // consecutive lanes -> consecutive addresses inside a digit's run
      const uint64_t m = st_m[q];
      const uint32_t d = (uint32_t)((m >> t.shift) & (uint64_t)(t.nb - 1));
      const uint32_t gb = gbase[d];
      if (gb != 0xffffffffu) {
        const uint32_t o = gb + (base + q - lbase[d]);
        A.hk[0][o] = m;
        A.gi[0][o] = (uint32_t)(t.lo + st_i[q]);
      }
    }
    __syncthreads();
  }
}

// the (bucket, shard) segments of level 1 as tiles of level 2: tile_seg[0] = tiles, [1 + c] = pairs of segment c,
// [1 + 512 + tile] = c << 20 | tile inside c.  One workgroup.
__global__ __launch_bounds__(512) void id_seg_kernel(const IdentityArgs A) {
  __shared__ uint32_t part[512];
  const int nseg = (1 << A.b1) * kShards;
  const int c = threadIdx.x;
  uint32_t cnt = 0;
  if (c < nseg) {
    const uint32_t v = A.cur1[c * kCurPad];
    cnt = (int64_t)v < A.subcap1 ? v : (uint32_t)A.subcap1;
  }
  const uint32_t nt = (cnt + kTile - 1) / kTile;
  part[c] = nt;
  __syncthreads();
  for (int d = 1; d < 512; d <<= 1) {
    const uint32_t o = c >= d ? part[c - d] : 0u;
    __syncthreads();
    part[c] += o;
    __syncthreads();
  }
  const uint32_t before = part[c] - nt;
  if (c == 511) A.tile_seg[0] = part[511];
  A.tile_seg[1 + c] = cnt;
  {  // the capacity the fine buckets are USED with: 3/2 of the mean bucket + 512 pairs, at most the allocated one (which
     // is sized for lists without padding).  Tighter buckets = the pairs of level 2 in 60 % of the address range: fewer
     // pages and DRAM rows under the table kernel's 16 k concurrent streams.  An overflow falls back like any other.
    __shared__ uint32_t live;
    if (c == 0) live = 0;
    __syncthreads();
    if (cnt) atomicAdd(&live, cnt);
    __syncthreads();
    if (c == 0) {
      const int64_t mean = (int64_t)(live >> (A.b1 + A.b2));
      int64_t cap = (mean + (mean >> 1) + 512 + 15) & ~(int64_t)15;
      A.ovf[1] = (uint32_t)(cap < A.cap2 ? cap : A.cap2);
    }
  }
  for (uint32_t k = 0; k < nt; k++) A.tile_seg[1 + 512 + before + k] = ((uint32_t)c << 20) | k;
}

__global__ __launch_bounds__(kBT) void id_part2_kernel(const IdentityArgs A) {
  __shared__ uint32_t hist[256], lbase[256], gbase[256], wtot[4];
  __shared__ uint64_t st_h[kTile];
  __shared__ uint32_t st_g[kTile];
  // The overflow flag may be RAISED by other workgroups of this very launch while this one starts, and every wave does
  // its own load: the decision is taken once per workgroup (thread 0 reads, all branch on the LDS copy), otherwise
  // the waves that stayed would scatter through hist / lbase / gbase entries the leavers never wrote.
  __shared__ uint32_t quit;
  if (threadIdx.x == 0) quit = A.ovf[0];
  __syncthreads();
  if (quit) return;
  const int64_t cap2 = (int64_t)A.ovf[1];  // (id_seg_kernel)
  const uint32_t n_tiles = A.tile_seg[0], tile = xcd_tile(n_tiles);
  if (tile >= n_tiles || (blockIdx.x >> 3) >= ((n_tiles + 7u) >> 3)) return;
  const uint32_t e = A.tile_seg[1 + 512 + tile];
  const uint32_t c = e >> 20, tl = e & 0xfffffu;
  const int64_t seg0 = (int64_t)c * A.subcap1;
  const int64_t lo = seg0 + (int64_t)tl * kTile;
  const int64_t end = seg0 + (int64_t)A.tile_seg[1 + c];
  const int64_t hi = lo + kTile < end ? lo + kTile : end;
  const int nb = 1 << A.b2, shift = 64 - A.b1 - A.b2;
  const uint32_t d1 = c / kShards;
  hist[threadIdx.x] = 0;
  __syncthreads();
  uint32_t rk[kPer], g[kPer];
  uint64_t h[kPer];
  uint32_t ok = 0;
#pragma unroll
  for (int i = 0; i < kPer; i++) {
    const int64_t p = lo + i * kBT + threadIdx.x;
    ok |= (p < hi ? 1u : 0u) << i;
    h[i] = A.hk[0][p < hi ? p : lo];
    g[i] = A.gi[0][p < hi ? p : lo];
  }
#pragma unroll
  for (int i = 0; i < kPer; i++) {
    rk[i] = 0;
    if ((ok >> i) & 1u) {
      const uint32_t d = (uint32_t)((h[i] >> shift) & (uint64_t)(nb - 1));  // (the pairs carry the mixed key)
      rk[i] = atomicAdd(&hist[d], 1u) | (d << 16);
    }
  }
  __syncthreads();
  {
    const uint32_t v = hist[threadIdx.x];
    uint32_t gb = 0xffffffffu;
    if (v) {  // (issued before the scan: the claim's round trip overlaps it)
      const uint32_t b = d1 * (uint32_t)nb + threadIdx.x;
      const uint32_t o = atomicAdd(&A.cur2[b], v);
      if ((int64_t)o + (int64_t)v <= cap2) gb = (uint32_t)((int64_t)b * cap2) + o;
      else raise_overflow(A);
    }
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64);
      if ((int)(threadIdx.x & 63) >= d) inc += o;
    }
    if ((threadIdx.x & 63) == 63) wtot[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t before = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) before += wtot[w];
    lbase[threadIdx.x] = before + inc - v;
    gbase[threadIdx.x] = gb;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kPer; i++) {
    if ((ok >> i) & 1u) {
      const uint32_t p = lbase[rk[i] >> 16] + (rk[i] & 0xffffu);
      st_h[p] = h[i];
      st_g[p] = g[i];
    }
  }
  __syncthreads();
  // (staging in two rounds of half a tile -- 27 KB of LDS, five workgroups per CU instead of three -- measured slower,
  // 105 -> 115 us: the kernel moves 490 MB at 4.7 TB/s and is not short of waves)
  const uint32_t total = lbase[255] + hist[255];
  for (uint32_t p = threadIdx.x; p < total; p += kBT) {  // consecutive lanes -> consecutive addresses inside a digit's run
    const uint64_t hh = st_h[p];
    const uint32_t d = (uint32_t)((hh >> shift) & (uint64_t)(nb - 1));
    const uint32_t gb = gbase[d];
    if (gb != 0xffffffffu) {
      const uint32_t o = gb + (p - lbase[d]);
      A.hk[1][o] = hh;
      A.gi[1][o] = st_g[p];
    }
  }
}

// ---- the identity tables: one fine bucket per workgroup through an open-addressing table in LDS
// the (up to) eight pairs of a thread: one chunk of a bucket.  (Eight scalars each, not arrays: hipcc kept an indexed
// array in scratch memory -- 200 MB of spill traffic per launch.)
struct Pairs8 {
  uint64_t h0, h1, h2, h3, h4, h5, h6, h7;
  uint32_t g0, g1, g2, g3, g4, g5, g6, g7;
  uint32_t have;  // bit i: pair i of the chunk exists
};
struct TableCounters { uint32_t nuniq, nseen, ovf, special; };
constexpr int kPT = kChunk / kBT;
static_assert(kPT == 8, "the eight pairs of a thread are written out below");

#define MPLX_ID_LOAD1(i_, base_)                                  \
  {                                                               \
    const int64_t p_ = (base_) + (i_) * kBT + threadIdx.x;        \
    P.have |= (p_ < hi ? 1u : 0u) << (i_);                        \
    P.h##i_ = hk[p_ < hi ? p_ : lo];                              \
    P.g##i_ = gi[p_ < hi ? p_ : lo];                              \
  }
#define MPLX_ID_LOAD(base_)                                                                           \
  do {                                                                                                \
    P.have = 0;                                                                                       \
    MPLX_ID_LOAD1(0, base_) MPLX_ID_LOAD1(1, base_) MPLX_ID_LOAD1(2, base_) MPLX_ID_LOAD1(3, base_)   \
    MPLX_ID_LOAD1(4, base_) MPLX_ID_LOAD1(5, base_) MPLX_ID_LOAD1(6, base_) MPLX_ID_LOAD1(7, base_)   \
  } while (0)

// Barrier between phases that only hand LDS contents from wave to wave.  __syncthreads() is also a fence: the compiler
// puts s_waitcnt vmcnt(0) in front of it, i.e. every global load in flight has to land first -- including the lines the
// persistent kernel pulls for its NEXT bucket, whose HBM round trip would then sit in front of this bucket's first
// barrier instead of behind its sweeps.  (The table kernels' global stores, canon[], are read by nobody in the launch.)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// One bucket [lo, hi) of the partitioned pairs through the workgroup's LDS table.  Every sweep over the bucket has
// kChunk pairs (8 per thread) in flight at once, and the typical bucket IS one such chunk (`single`): the caller has it
// in P already and both sweeps (insert, look-up) run from registers.  Rounds over further hash bits when the bucket
// holds more distinct keys than the table takes.  PREMIXED: the pairs carry mix(hash) as their key (claimed form).
// Ends with a barrier: the table may be reused at once.
template <bool PREMIXED>
__device__ __forceinline__ void table_bucket(const IdentityArgs &A, const int64_t lo, const int64_t hi, const uint64_t *hk, const uint32_t *gi,
                                             unsigned long long *keys, uint32_t *vals, TableCounters &C, Pairs8 &P) {
  constexpr bool premixed = PREMIXED;
  const bool single = hi - lo <= kChunk;
  if (single) {
    // A first sweep over keys alone, in a table of kSlotsKeys = 3 328 slots over the SAME LDS (26 KB = the 2 048 keys +
    // 2 048 indices of the general table below and 2 KB more): a bucket whose pairs all have different keys -- every bucket
    // of a frontier without revisits -- is done after it, canon[g] = g is what level 1 wrote, and at a load of 0.37
    // instead of 0.61 the slowest lane of a wave probes half as far (3 072 and 3 584 slots -- the latter five workgroups
    // per CU -- measured 1 - 6 % slower).  Any duplicate (or one of the two unstorable keys) sends
    // the bucket through the general sweeps: there the extra pass costs a few plain reads per pair.
    unsigned long long *big = keys;
    for (int i = threadIdx.x; i < kSlotsKeys; i += kBT) big[i] = kEmpty;
    if (threadIdx.x == 0) { C.nuniq = 0; C.nseen = 0; C.ovf = 0; C.special = 0xffffffffu; }
    lds_barrier();
    uint32_t fresh = 0;
    auto insert0 = [&](const uint64_t hh, const bool exists) {
      if (!exists) return;
      fresh += 0x10000u;
      if (hh == kEmpty) { C.ovf = 1; return; }
      const uint64_t m = premixed ? hh : mix(hh);
      uint32_t addr = (uint32_t)(((uint64_t)(uint32_t)m * (uint64_t)kSlotsKeys) >> 32) * 8u;
      uint32_t won, fail;
      unsigned long long k_, save_, tmp_, wm_;
      uint32_t cnt_;
      asm volatile(
          "s_mov_b64 %[save], exec\n\t"
          "v_mov_b32 %[won], 0\n\t"
          "v_mov_b32 %[fail], 0\n\t"
          "s_movk_i32 %[cnt], %[maxp]\n"
          "1:\n\t"
          "ds_read_b64 %[k], %[addr]\n\t"
          "s_waitcnt lgkmcnt(0)\n\t"
          "v_cmp_eq_u64 vcc, -1, %[k]\n\t"
          "s_mov_b64 %[wm], 0\n\t"
          "s_and_saveexec_b64 %[tmp], vcc\n\t"
          "s_cbranch_execz 2f\n\t"
          "ds_cmpst_rtn_b64 %[k], %[addr], %[emp], %[hh]\n\t"
          "s_waitcnt lgkmcnt(0)\n\t"
          "v_cmp_eq_u64 vcc, -1, %[k]\n\t"
          "s_mov_b64 %[wm], vcc\n\t"
          "s_nop 1\n\t"
          "v_cndmask_b32 %[won], %[won], 1, vcc\n"
          "2:\n\t"
          "s_mov_b64 exec, %[tmp]\n\t"
          "v_cmp_ne_u64 vcc, %[k], %[hh]\n\t"
          "s_andn2_b64 vcc, vcc, %[wm]\n\t"
          "s_and_b64 exec, exec, vcc\n\t"
          "s_cbranch_execz 3f\n\t"
          "v_add_u32 %[addr], 8, %[addr]\n\t"
          "v_cmp_eq_u32 vcc, %[size], %[addr]\n\t"
          "s_nop 1\n\t"
          "v_cndmask_b32 %[addr], %[addr], 0, vcc\n\t"
          "s_sub_u32 %[cnt], %[cnt], 1\n\t"
          "s_cmp_lg_u32 %[cnt], 0\n\t"
          "s_cbranch_scc1 1b\n\t"
          "v_mov_b32 %[fail], 1\n"
          "3:\n\t"
          "s_mov_b64 exec, %[save]\n\t"
          : [addr] "+v"(addr), [won] "=&v"(won), [fail] "=&v"(fail), [k] "=&v"(k_), [save] "=&s"(save_), [tmp] "=&s"(tmp_), [wm] "=&s"(wm_), [cnt] "=&s"(cnt_)
          : [hh] "v"(hh), [emp] "v"((unsigned long long)kEmpty), [size] "v"((uint32_t)(kSlotsKeys * 8)), [maxp] "n"(kMaxProbe + 1)
          : "vcc", "scc", "memory");
      fresh += won;
      if (fail) C.ovf = 1;
    };
    insert0(P.h0, P.have & 1u); insert0(P.h1, P.have & 2u); insert0(P.h2, P.have & 4u); insert0(P.h3, P.have & 8u);
    insert0(P.h4, P.have & 16u); insert0(P.h5, P.have & 32u); insert0(P.h6, P.have & 64u); insert0(P.h7, P.have & 128u);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) fresh += (uint32_t)__shfl_xor((int)fresh, d, 64);
    if ((threadIdx.x & 63) == 0 && fresh) {
      if (fresh & 0xffffu) atomicAdd(&C.nuniq, fresh & 0xffffu);
      atomicAdd(&C.nseen, fresh >> 16);
    }
    lds_barrier();
    const bool clean = C.ovf == 0 && C.nuniq == C.nseen;  // (uniform)
    lds_barrier();  // everyone has read the counters before the general sweeps clear them
    if (clean) return;
  }
  for (uint32_t R = 1;; R <<= 1) {
    bool split = false;
    for (uint32_t r = 0; r < R; r++) {
      for (int i = threadIdx.x; i < kSlots; i += kBT) { keys[i] = kEmpty; vals[i] = 0xffffffffu; }
      if (threadIdx.x == 0) { C.nuniq = 0; C.nseen = 0; C.ovf = 0; C.special = 0xffffffffu; }
      lds_barrier();
      for (int64_t base = lo; base < hi; base += kChunk) {
        if (C.ovf) break;  // (not uniform, no barrier inside this loop: the round is void anyway)
        if (!single) MPLX_ID_LOAD(base);
        uint32_t fresh = 0;  // keys this thread put into the table | pairs it looked at << 16 (counted per wave below: one LDS atomic each per wave and chunk)
        auto insert = [&](const uint64_t hh, const uint32_t gg, const bool exists) {
          const uint64_t m = premixed ? hh : mix(hh);
          const bool mine = exists && ((uint32_t)(m >> 12) & (R - 1u)) == r;
          fresh += mine ? 0x10000u : 0u;
          if (mine && hh == kEmpty) atomicMin(&C.special, gg);  // the one key the key field cannot hold
          if (mine && hh != kEmpty) {
            // The probe loop, written out: the compiler's versions of it (four source shapes were tried) run 190 - 300
            // instructions per 64 pairs, most of them exec-mask bookkeeping around the early exits, and the kernel is bound
            // by exactly that instruction stream (profiles/README.md, round 4).  Here: the lanes still looking are the
            // exec mask; a plain read first (a key that is already in the table -- every duplicate after the first --
            // costs no CAS), a 64-bit CAS where the slot was empty (LDS atomics on scattered slots are cheap: 20 cycles per
            // wave, profiles/micro/lds_atomic_rate.hip), lanes leave the mask when their slot holds their key.
            uint32_t addr = ((uint32_t)m & (kSlots - 1)) * 8u;  // byte address of the slot inside keys[]
            uint32_t won, fail;
            unsigned long long k_, save_, tmp_, wm_;
            uint32_t cnt_;
            asm volatile(
                "s_mov_b64 %[save], exec\n\t"
                "v_mov_b32 %[won], 0\n\t"
                "v_mov_b32 %[fail], 0\n\t"
                "s_movk_i32 %[cnt], %[maxp]\n"
                "1:\n\t"
                "ds_read_b64 %[k], %[addr]\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_cmp_eq_u64 vcc, -1, %[k]\n\t"
                "s_mov_b64 %[wm], 0\n\t"
                "s_and_saveexec_b64 %[tmp], vcc\n\t"
                "s_cbranch_execz 2f\n\t"
                "ds_cmpst_rtn_b64 %[k], %[addr], %[emp], %[hh]\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_cmp_eq_u64 vcc, -1, %[k]\n\t"
                "s_mov_b64 %[wm], vcc\n\t"
                "s_nop 1\n\t"
                "v_cndmask_b32 %[won], %[won], 1, vcc\n"
                "2:\n\t"
                "s_mov_b64 exec, %[tmp]\n\t"
                "v_cmp_ne_u64 vcc, %[k], %[hh]\n\t"
                "s_andn2_b64 vcc, vcc, %[wm]\n\t"
                "s_and_b64 exec, exec, vcc\n\t"
                "s_cbranch_execz 3f\n\t"
                "v_add_u32 %[addr], 8, %[addr]\n\t"
                "v_and_b32 %[addr], %[wrap], %[addr]\n\t"
                "s_sub_u32 %[cnt], %[cnt], 1\n\t"
                "s_cmp_lg_u32 %[cnt], 0\n\t"
                "s_cbranch_scc1 1b\n\t"
                "v_mov_b32 %[fail], 1\n"
                "3:\n\t"
                "s_mov_b64 exec, %[save]\n\t"
                : [addr] "+v"(addr), [won] "=&v"(won), [fail] "=&v"(fail), [k] "=&v"(k_), [save] "=&s"(save_), [tmp] "=&s"(tmp_), [wm] "=&s"(wm_), [cnt] "=&s"(cnt_)
                : [hh] "v"(hh), [emp] "v"((unsigned long long)kEmpty), [wrap] "v"((uint32_t)(kSlots * 8 - 1)), [maxp] "n"(kMaxProbe + 1)
                : "vcc", "scc", "memory");
            fresh += won;
            if (fail) {
              // a probe sequence this long means the table is (nearly) full: more distinct keys than `fill` are on their
              // way in; the round is repeated with the bucket split further
              C.ovf = 1;
            } else {
              const uint32_t s = addr >> 3;
              if (vals[s] > gg) atomicMin(&vals[s], gg);
            }
          }
        };
        insert(P.h0, P.g0, P.have & 1u); insert(P.h1, P.g1, P.have & 2u); insert(P.h2, P.g2, P.have & 4u); insert(P.h3, P.g3, P.have & 8u);
        insert(P.h4, P.g4, P.have & 16u); insert(P.h5, P.g5, P.have & 32u); insert(P.h6, P.g6, P.have & 64u); insert(P.h7, P.g7, P.have & 128u);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) fresh += (uint32_t)__shfl_xor((int)fresh, d, 64);
        if ((threadIdx.x & 63) == 0 && fresh) {  // (8 pairs per lane x 64 lanes: both halves stay below 2^16)
          if ((fresh & 0xffffu) && atomicAdd(&C.nuniq, fresh & 0xffffu) + (fresh & 0xffffu) > (uint32_t)A.fill) C.ovf = 1;
          atomicAdd(&C.nseen, fresh >> 16);
        }
      }
      lds_barrier();
      const bool over = C.ovf != 0 && R < (1u << 20);  // (uniform)
      // as many keys as pairs: every pair of the round is the only one with its hash, and canon[g] = g is what the
      // level-1 pass wrote -- nothing to look up (a frontier without revisits: practically every bucket)
      const bool all_first = C.nuniq == C.nseen && C.special == 0xffffffffu;
      lds_barrier();                              // everyone has read the counters before the next round clears them
      if (over) { split = true; break; }            // split further and start the bucket over; canon writes are idempotent
      if (all_first) continue;
      for (int64_t base = lo; base < hi; base += kChunk) {
        if (!single) MPLX_ID_LOAD(base);
        auto lookup = [&](const uint64_t hh, const uint32_t gg, const bool exists) {
          const uint64_t m = premixed ? hh : mix(hh);
          if (exists && ((uint32_t)(m >> 12) & (R - 1u)) == r) {
            uint32_t c;
            if (hh == kEmpty) {
              c = C.special;
            } else {
              uint32_t s = (uint32_t)m & (kSlots - 1);
              int probes = 0;
              while (keys[s] != hh && probes++ <= kMaxProbe) s = (s + 1) & (kSlots - 1);
              c = keys[s] == hh ? vals[s] : gg;  // (a key that was not inserted: only past the 2^20-round limit)
            }
            if (c != gg) A.canon[gg] = (int32_t)c;  // (first occurrences were written by the level-1 histogram pass)
          }
        };
        lookup(P.h0, P.g0, P.have & 1u); lookup(P.h1, P.g1, P.have & 2u); lookup(P.h2, P.g2, P.have & 4u); lookup(P.h3, P.g3, P.have & 8u);
        lookup(P.h4, P.g4, P.have & 16u); lookup(P.h5, P.g5, P.have & 32u); lookup(P.h6, P.g6, P.have & 64u); lookup(P.h7, P.g7, P.have & 128u);
      }
      lds_barrier();
    }
    if (!split) break;
  }
}

// ---- exact form: one workgroup per fine bucket, the bucket's range from the prefix sums
__global__ __launch_bounds__(kBT) __attribute__((amdgpu_waves_per_eu(6))) void id_tables_kernel(const IdentityArgs A) {
  __shared__ unsigned long long keys[kSlotsKeys];  // (general table: keys[0 .. kSlots), the indices behind them)
  uint32_t *vals = (uint32_t *)(keys + kSlots);
  static_assert(kSlots * 12 <= kSlotsKeys * 8, "keys + indices of the general table inside the keys-only one");
  __shared__ TableCounters C;
  const int64_t lo = A.range[blockIdx.x], hi = A.range[blockIdx.x + 1];
  if (hi <= lo) return;
  const uint64_t *hk = A.hk[A.b2 > 0 ? 1 : 0];
  const uint32_t *gi = A.gi[A.b2 > 0 ? 1 : 0];
  Pairs8 P{};
  if (hi - lo <= kChunk) MPLX_ID_LOAD(lo);
  table_bucket<false>(A, lo, hi, hk, gi, keys, vals, C, P);
}

// ---- claimed form: one workgroup per fine bucket of fixed capacity, its pairs counted by the bucket's cursor.
// (Persistent workgroups that fetch the next bucket's count and pull its lines into the L2 while this one is processed
// were tried in round 4: the loads alone then take 52 us for C4, but the kernel got slower, 157 -> 198 us -- it is bound
// by the instruction stream of the probe loops, ~25 passes of ~60 instructions per wave and bucket, not by its memory
// round trips; profiles/README.md.)
__global__ __launch_bounds__(kBT) __attribute__((amdgpu_waves_per_eu(6))) void id_tables_claimed_kernel(const IdentityArgs A) {
  __shared__ unsigned long long keys[kSlotsKeys];  // (general table: keys[0 .. kSlots), the indices behind them)
  uint32_t *vals = (uint32_t *)(keys + kSlots);
  static_assert(kSlots * 12 <= kSlotsKeys * 8, "keys + indices of the general table inside the keys-only one");
  __shared__ TableCounters C;
  if (A.ovf[0]) return;  // (uniform) a bucket overflowed: the caller runs the exact form
  const int64_t cap2 = (int64_t)A.ovf[1];  // the capacity the buckets were filled with (id_seg_kernel)
  const int64_t n = (int64_t)A.cur2[blockIdx.x];
  const int64_t lo = (int64_t)blockIdx.x * cap2, hi = lo + (n < cap2 ? n : cap2);
  if (hi <= lo) return;
  const uint64_t *hk = A.hk[1];
  const uint32_t *gi = A.gi[1];
  Pairs8 P{};
  if (hi - lo <= kChunk) MPLX_ID_LOAD(lo);
  table_bucket<true>(A, lo, hi, hk, gi, keys, vals, C, P);
}
#undef MPLX_ID_LOAD1
#undef MPLX_ID_LOAD

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

// capacities of the claimed form: a (bucket, shard) segment of level 1 takes its even share of ALL slots + 1/8 (lists
// are never more than full; the mixed hash spreads distinct keys evenly, so only heavy duplication of few keys overflows),
// a fine bucket its even share + 1/4 + 512
void identity_claimed_sizes(int64_t n_slots, int b1, int b2, int64_t *subcap1, int64_t *cap2, int64_t *pairs1, int64_t *pairs2,
                            int64_t *tiles2_max, int64_t *cur_words) {
  const int64_t nseg = ((int64_t)1 << b1) * kShards, buckets = (int64_t)1 << (b1 + b2);
  *subcap1 = ((n_slots / nseg) * 9 / 8 + kTile + 15) & ~(int64_t)15;
  *cap2 = (((n_slots >> (b1 + b2)) * 5) / 4 + 512 + 15) & ~(int64_t)15;
  *pairs1 = nseg * *subcap1;
  *pairs2 = buckets * *cap2;
  *tiles2_max = *pairs1 / kTile + nseg + 1;  // (every segment: its whole tiles + one partial)
  *cur_words = 512 * kCurPad + buckets + 64;  // level-1 cursors, level-2 cursors, the device flag
}

hipError_t launch_identity(const IdentityArgs &a, int64_t ctr1, int64_t ctr2, hipStream_t s) {
  if (a.n_slots <= 0) return hipSuccess;
  hipError_t e;
  auto up8 = [](int64_t tiles) { return (unsigned)(((tiles + 7) / 8) * 8); };  // whole rounds over the 8 XCDs (xcd_tile)
  // counters of both levels in one fill when they are adjacent (post_api.cpp carves them so)
  const bool one_fill = a.b2 > 0 && a.cnt[1] == a.cnt[0] + ctr1;
  if ((e = hipMemsetAsync(a.cnt[0], 0, (size_t)(ctr1 + (one_fill ? ctr2 : 0)) * 4, s)) != hipSuccess) return e;
  hipLaunchKernelGGL(id_hist_kernel<1>, dim3(up8(a.tiles1)), dim3(kBT), 0, s, a);
  hipLaunchKernelGGL(id_scan_blocks_kernel, dim3((unsigned)(ctr1 / kTile)), dim3(kBT), 0, s, a.cnt[0], a.tot[0]);
  hipLaunchKernelGGL(id_scan_totals_kernel, dim3(1), dim3(1024), 0, s, a, 1, (int)(ctr1 / kTile));  // + segments
  hipLaunchKernelGGL(id_scatter_kernel<1>, dim3(up8(a.tiles1)), dim3(kBT), 0, s, a);
  unsigned buckets = 1u << a.b1;
  if (a.b2 > 0) {
    if (!one_fill && (e = hipMemsetAsync(a.cnt[1], 0, (size_t)ctr2 * 4, s)) != hipSuccess) return e;
    hipLaunchKernelGGL(id_hist_kernel<2>, dim3(up8(a.tiles2_cap)), dim3(kBT), 0, s, a);
    hipLaunchKernelGGL(id_scan_blocks_kernel, dim3((unsigned)(ctr2 / kTile)), dim3(kBT), 0, s, a.cnt[1], a.tot[1]);
    hipLaunchKernelGGL(id_scan_totals_kernel, dim3(1), dim3(1024), 0, s, a, 2, (int)(ctr2 / kTile));
    hipLaunchKernelGGL(id_scatter_kernel<2>, dim3(up8(a.tiles2_cap)), dim3(kBT), 0, s, a);
    buckets <<= a.b2;
  }
  hipLaunchKernelGGL(id_ranges_kernel, dim3(buckets / kBT + 1), dim3(kBT), 0, s, a, buckets);
  hipLaunchKernelGGL(id_tables_kernel, dim3(buckets), dim3(kBT), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_identity_claimed(const IdentityArgs &a, int64_t cur_words, hipStream_t s) {
  if (a.n_slots <= 0) return hipSuccess;
  hipError_t e;
  auto up8 = [](int64_t tiles) { return (unsigned)(((tiles + 7) / 8) * 8); };
  if ((e = hipMemsetAsync(a.cur1, 0, (size_t)cur_words * 4, s)) != hipSuccess) return e;  // cursors of both levels + flag
  hipLaunchKernelGGL(id_part1_kernel, dim3(up8(a.tiles1)), dim3(kBT), 0, s, a);
  hipLaunchKernelGGL(id_seg_kernel, dim3(1), dim3(512), 0, s, a);
  hipLaunchKernelGGL(id_part2_kernel, dim3(up8(a.tiles2_max)), dim3(kBT), 0, s, a);
  hipLaunchKernelGGL(id_tables_claimed_kernel, dim3(1u << (a.b1 + a.b2)), dim3(kBT), 0, s, a);
  return hipGetLastError();
}

}  // namespace mplx
