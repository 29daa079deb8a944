// Internal declarations shared by the C-ABI layer (mplx_api.cpp) and the HIP
// kernels (expand_kernel.hip).  Not installed; the public surface is
// include/mplx.h.
#ifndef MPLX_INTERNAL_H
#define MPLX_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mplx {

// validate_yaw (reference include/mpl_basis/primitive.h:504-525) compares d = v_hat . (cos yaw, sin yaw) with
// cos(yaw_max), and the reference's cos / sin are the HOST libm's.  The device's (OCML) differ from glibc's in the
// last place on a few per cent of arguments, so a decision within rounding noise of the threshold could come out
// differently.  Pinning: the detection pass (amb != null) flags every node with a decision closer to the threshold
// than `margin` (a bound on what the two libraries' rounding can move d - cos(yaw_max); decisions farther away are
// the same under either library) and the host then re-expands exactly those nodes in an override pass (tab != null)
// in which every cos / sin of a heading-limit decision and cos(yaw_max) come from a table the HOST filled with its
// libm, so the decision is the reference's own arithmetic on the reference's own values.  The per-sample heading
// COST (env_map.h:121-129) keeps device trig: it is continuous and held to north_star's 1e-6.
struct YawPin {
  int32_t *amb;               // detection: [0] = number of flagged nodes, [1 .. cap] their indices; null = off
  int32_t *any_host;          // detection: a word of pinned host memory set to 1 by any flagging wave (plain store), so
                              // that the host learns "nothing flagged" from its own memory, without a copy
  int32_t amb_cap;
  double margin;
  double tie_yaw;             // detection: yaw_max when the "velocity along x, |yaw| == yaw_max" tie is exact under the host's
                              // libm too (mplx_device_common.h near_limit), NaN when it is not (then no decision is exempt)
  const int32_t *node_list;   // override pass: the nodes to re-expand (kernel node k -> node_list[k]); null = all
  const double *tab;          // override pass: per listed node the host's trig values (layout per kernel); null = off
  int32_t tab_stride;         // doubles per listed node
  double cos_lim;             // override pass: the host's cos(yaw_max)
};

// Everything one expansion launch needs, passed by value as the kernel
// argument block (lives in SGPRs / constant cache: wave-uniform).
struct ExpandArgs {
  // voxel / occupancy grid, row-major with x fastest (reference
  // include/mpl_collision/map_util.h:34-41)
  const int8_t *map;
  const int8_t *pot;       // potential cells or nullptr (env_map.h:113-118)
  const uint32_t *region;  // search region, 1 bit per cell, or nullptr
  int32_t dim0, dim1, dim2;
  double org0, org1, org2;
  double res;
  // env parameters (env_base.h:368-392, env_map.h:294-296)
  double dt, w, wyaw;
  double v_max, a_max, j_max, yaw_max;
  double pot_w, grad_w;
  // controls [nU][udim]
  const double *U;
  int32_t nU, udim;
  // frontier, field-major [4D+2][node_stride]
  const double *nodes;
  int64_t n_nodes, node_stride;
  // dense successor slots (any may be nullptr)
  uint8_t *status;
  double *cost;
  uint64_t *hash;
  double *state;
  int64_t state_stride;
  int32_t *iters;
  int32_t stream_out;  // 1: the slots are final outputs (non-temporal stores); 0: scratch that is re-read soon
  YawPin yaw;          // heading-limit decisions pinned to the host libm (see YawPin)
};

// Arguments of the tiled, list-producing kernel (expand_tile_kernel.hip).
// Completion of a small synchronous launch seen through a word the KERNEL writes into pinned host memory when its last
// wave (workgroup) is through, instead of through hipStreamSynchronize: 8.2 us per empty launch + completion against
// 12.8 us (profiles/micro/mailbox_latency.hip).  flag == nullptr: no signalling (every launch the caller does not wait
// for on the spot).
struct DoneSignal {
  uint64_t *flag;   // pinned host memory; receives `seq` (system-scope release) after every store of the launch
  uint32_t *count;  // device memory, 0 between launches: waves (workgroups) that have finished
  uint64_t seq;
};

// What the graph search computes for every successor right after get_succ (graph_search.h:84-88),
// fused into the list stores of the expansion kernels: default heuristic (env_base.h:46-64) and goal flags
// (env_map.h:25-37, bit 0; env_base.h:47, bit 1).  Null pointers = off.  Same arithmetic as post_kernel.hip.
struct PostFuse {
  double *heur;     // [n_nodes * l_nstride] or null
  uint8_t *flags;   // [n_nodes * l_nstride] or null
  double goal[14];  // the goal waypoint's rows (pos, vel, acc, jrk, yaw, t)
  uint64_t goal_hash;
  double w, v_max, tol_pos, tol_vel, tol_acc, tol_yaw;
};

// Mailbox of the resident (service) form of the tiled kernel, in pinned host memory; every word on its own line.
struct SvcMailbox {
  uint64_t doorbell;  // host -> device: (seq << 32) | n_nodes of the request in the landing block
  uint64_t pad0[7];
  uint64_t done;      // device -> host: seq of the last request whose lists are complete in the landing block
  uint64_t pad1[7];
  uint32_t quit;      // host -> device: leave now
  uint32_t pad2[15];
  uint32_t alive;     // device -> host: 0 once the coordinator has decided to leave (the stream then drains)
  uint32_t pad3[15];
};

struct TileArgs {
  const int8_t *map;
  const uint32_t *region;
  int32_t dim0, dim1, dim2;
  double org0, org1, org2;
  double res;
  double dt, w;
  double v_max, a_max, j_max;
  const double *U;
  int32_t nU, udim;
  float inv_nU;
  const double *nodes;
  int64_t n_nodes, node_stride;
  // tiling
  int32_t npb;         // whole nodes per workgroup
  int32_t tile_pairs;  // npb * nU  (<= 1024)
  int32_t wl_cap;      // work-list capacity in samples
  int32_t n_max;       // largest sample count n served by the work list (<= 63)
  int32_t dbg;         // timing ablations (env MPLX_TILE_DBG); 0 in production
  int32_t lds_u_offset;  // byte offset of the control table inside the dynamic LDS
  int32_t grid_limit;  // persistent workgroups to launch (CUs x workgroups per CU)
  // tables made by launch_make_tables
  const double *ttab;          // [64][64] accumulated sample times
  const unsigned char *tcnt;   // [64] loop iteration counts
  double Rres, R001, R01;      // refined reciprocals of res, 0.01, 0.1
  // per-node successor lists: node k owns entries [k*nU, k*nU + l_count[k])
  int32_t *l_count;
  int32_t *l_action;
  double *l_cost;
  uint64_t *l_hash;
  double *l_state;
  int64_t l_stride;
  int32_t *l_iters;
  int64_t l_nstride;  // entries reserved per node (>= nU)
  int32_t l_pad;      // 1: complete the last 128-byte line of every list row (node stride is a multiple of 32)
  PostFuse post;      // heuristic / goal flags per successor, same indexing as the lists
  // service mode (expand_tile_kernel.hip): null / 0 for an ordinary launch
  SvcMailbox *svc_mb;   // pinned host memory
  uint64_t *svc_dev;    // device memory, zeroed before the launch: [0] = command, [1 + g] = last request workgroup g finished
  uint64_t svc_seq0;    // seq of the last request served before this launch
  uint64_t svc_idle;    // ticks of the 100 MHz clock without a request after which the kernel leaves
  DoneSignal done;      // ordinary launches of small synchronous batches
};

size_t tile_lds_bytes(int tile_pairs, int npb, int wl_cap, int n_max, int n_fields, int u_doubles,
                      int *u_offset);
hipError_t launch_make_tables(double T, double res, double *ttab, unsigned char *tcnt, double *recips,
                              hipStream_t stream);
hipError_t launch_expand_tile(int dim, int control, const TileArgs &args, hipStream_t stream);
int tile_service_resident_workgroups(int dim, int control, const TileArgs &a);  // 0: unknown

// Arguments of the factorised list-producing kernel (expand_grid_kernel.hip):
// the control table is given per axis as its distinct values plus, per control,
// the packed indices of its entries (j0 | j1 << 8 | j2 << 16).
struct GridArgs {
  const uint32_t *blk;   // blocked-bit map, 1 bit per cell, x fastest (launch_build_blocked_bits)
  int64_t blk_words;
  const int8_t *pot;     // potential map (then blk / sat describe "potential > 0 or outside the region") or null
  const uint32_t *region; // search region bits (read by the potential path only; folded into blk otherwise)
  double pot_w, grad_w;  // env_map.h:115-116: potential_weight, gradient_weight
  const uint32_t *sat;   // summed-area table of blk, sizes dim+1 with a zero border (launch_build_sat), or null
  int32_t dim0, dim1, dim2;
  double org0, org1, org2;
  double res;
  double dt, w;
  double v_max, a_max, j_max;
  double yaw_max, wyaw;  // yaw controls only
  const double *uvals;   // [4][16] distinct control values of each axis; row 3 = yaw rates (expand_lex_kernel: [3][uval_stride])
  int32_t uval_stride;   // 16, or 32 for the wide tables only expand_lex_kernel.hip takes (17 .. 32 values on an axis)
  const uint32_t *uidx;  // [nU] packed per-axis value indices
  int32_t nd0, nd1, nd2; // number of distinct values per axis
  int32_t ndy;           // distinct yaw rates (yaw controls)
  int32_t ulex;          // 1: control i is the i-th combination of the per-axis values in lexicographic order
  int32_t ndp;           // table stride over values (max nd)
  int32_t nU;
  const double *nodes;
  int64_t n_nodes, node_stride;
  int32_t n_max;         // largest sample count n (<= 61)
  int32_t rmax;          // sample counts handled per round (rows of cell codes per axis entry)
  int32_t boxcap;        // dwords of LDS per wave for the staged blocked bits
  int32_t gather;        // 1: no box staging, the sample loops read the blocked-bit map directly (small control tables)
  int32_t lex;           // host side only: 1 = this launch goes to expand_lex_kernel.hip (same arguments, its own LDS carve-up)
  // Dynamic node assignment (work == null: static striding).  Nodes differ a lot in work (dead at t = 0, free box, one
  // or several passes), and with a static assignment the waves lived only 46 % (C5) - 83 % (C4) of the kernel's
  // duration (SQ_WAVE_CYCLES against SQ_BUSY_CYCLES).  Chunks of work_chunk nodes: chunk w < W (the waves launched)
  // belongs to wave w, the others are claimed from kWorkCounters counters (a single counter serialises at ~15 ns per
  // claim: measured 3 x slower than static), counter blockIdx % kWorkCounters owning an equal share of them; each
  // counter sits on its own 128-byte line.  `work` is zero when the launch begins; the launch zeroes `work_zero`, the
  // set the NEXT launch of the stream will use (ping-pong: no memset, no bookkeeping between launches).
  unsigned int *work;
  unsigned int *work_zero;
  int32_t work_chunk;
  int32_t work_blocked;  // 1: every counter owns one contiguous block of chunks; 0: the chunks are dealt round-robin
  int32_t dbg;           // timing ablations (env MPLX_TILE_DBG); 0 in production
  int32_t grid_limit;    // persistent workgroups to launch
  const double *ttab;    // tables of launch_make_tables
  const unsigned char *tcnt;
  double Rres, R001, R01;
  int32_t *l_count;
  int32_t *l_action;
  double *l_cost;
  uint64_t *l_hash;
  double *l_state;
  int64_t l_stride;
  int32_t *l_iters;
  int64_t l_nstride;  // entries reserved per node (>= nU)
  int32_t l_pad;      // 1: complete the last 128-byte line of every list row (node stride is a multiple of 32)
  PostFuse post;      // heuristic / goal flags per successor, same indexing as the lists
  YawPin yaw;         // heading-limit decisions pinned to the host libm (see YawPin); tab row: [c0, s0, cT[16], sT[16]]
  // Pre-screen of yaw controls (grid_prescreen_kernel): the nodes whose own heading passes validate_yaw at t = 0, in
  // frontier order, and their number (device memory, written by the pre-screen launch that precedes this one on the
  // stream).  Null: the kernel walks [0, n_nodes) and tests every node itself.
  const int32_t *live;
  const uint32_t *live_n;
  DoneSignal done;      // small synchronous batches: see DoneSignal
};
// expand_pair_kernel.hip: yaw controls on a potential map over a pre-screened frontier, two nodes per wave (same
// arguments as expand_grid_kernel; GridArgs::live / live_n must be set, grid_limit = workgroups to launch)
size_t pair_lds_bytes(int dim, int order, int nU, int ndp, int n_max, int rmax, bool ycost, int ndy);
int pair_waves_per_block();
int pair_nodes_per_wave();
int pair_max_yaw_rates();   // yaw rates a lane of it carries through its sample loop
bool pair_covers(int dim, int control);
hipError_t launch_expand_pair(int dim, int control, const GridArgs &a, hipStream_t s);
int pair_resident_blocks(int dim, int control, int ndy, size_t lds);
constexpr int kWorkCounters = 64;
// list rows are completed to whole 128-byte lines only for control tables of at least this many entries (mplx_api.cpp)
constexpr int kLinePadMinControls = 256;
// lane-per-node validate_yaw(t = 0) over a whole frontier (expand_grid_kernel.hip); fills live / live_n of `a`'s launch
// (live_n is zero when the launch begins; the launch zeroes live_zero, the counter of the NEXT pre-screen of the stream)
hipError_t launch_grid_prescreen(int dim, int control, const GridArgs &a, int32_t *live, uint32_t *live_n, uint32_t *live_zero,
                                 hipStream_t s);
// Packing of the used list prefixes for the copy back to the host (pack_kernel.hip).
constexpr int kPackRows = 24;
struct PackArgs {
  const void *src[kPackRows];   // device rows, [n_nodes * node_stride] elements each
  int64_t dst_off[kPackRows];   // byte offset of the row's packed block inside dst
  int32_t es[kPackRows];        // element size, 4 or 8
  int32_t n_rows;
  int64_t node_stride;
  const int32_t *count;         // [n_nodes]
  const int64_t *offs;          // [n_nodes] exclusive prefix sum of count
  int64_t node0, off0;          // first node of the chunk and its offs
  char *dst;
};
hipError_t launch_pack_rows(const PackArgs &args, int64_t n_nodes, hipStream_t stream);
// offs[0..n] = exclusive prefix sums of count[0..n) (offs[n] = total); device pointers
hipError_t launch_scan_counts(const int32_t *count, int64_t n, int64_t *offs, hipStream_t stream);
size_t grid_lds_bytes(int dim, int order, int nU, int ndp, int n_max, int rmax, int boxcap, int yaw_mode, int ndy,
                      int ulex);
int grid_waves_per_block();
// workgroups of the (dim, control, potential) instantiation resident per CU with `lds` bytes each; 0 = unknown
int grid_resident_blocks(int dim, int control, bool pot, size_t lds);
hipError_t launch_expand_grid(int dim, int control, const GridArgs &args, hipStream_t stream);
// expand_lex_kernel.hip: the same function for lexicographic control tables without yaw on an occupancy map (GridArgs
// with ulex == 1, pot == null, live == null, yaw unused); its own LDS carve-up
bool lex_covers(int dim, int control);
size_t lex_lds_bytes(int dim, int order, int ndp, int nU, int n_max, int rmax, int boxcap);
int lex_waves_per_block();
int lex_resident_blocks(int dim, int control, int ndp, size_t lds);
hipError_t launch_expand_lex(int dim, int control, const GridArgs &args, hipStream_t stream);
// Blocked-bit map: 1 bit per cell in map order, 1 = occupied or outside the search
// region; (n_cells + 31) / 32 dwords.
// Summed-area table of the blocked bits: (d0+1)(d1+1)(d2+1) uint32 (2D: (d0+1)(d1+1)*2).
// mplx_edit_map: cells and (blk != null) their blocked bits patched in place (map_prep_kernel.hip)
hipError_t launch_gather_cells(const int8_t *cells, const int64_t *idx, int64_t n, int8_t *out, hipStream_t s);
hipError_t launch_edit_map(const int64_t *idx, const int8_t *val, int64_t n, int64_t n_cells, int8_t *map, uint32_t *blk,
                           const uint32_t *region, hipStream_t s);
hipError_t launch_build_sat(int dim, const uint32_t *blk, const int32_t *mdim, uint32_t *sat, hipStream_t stream);
hipError_t launch_build_blocked_bits(const int8_t *map, const uint32_t *region, int64_t n_cells, int potential,
                                     uint32_t *out, hipStream_t stream);

// store_model_kernel.hip (diagnostic): the list stores of a launch alone, into the lists themselves
hipError_t launch_store_model(const int32_t *count, int64_t n_nodes, int64_t S, int32_t *action, double *cost, uint64_t *hash,
                              double *state, int64_t state_stride, int n_fields, int pad, int blocks, int mode, hipStream_t s);

// Batched edge re-validation (edge_kernel.hip).
struct EdgeArgs {
  const int8_t *map;
  const uint32_t *region;
  int32_t dim0, dim1, dim2;
  double org0, org1, org2;
  double res;
  double dt, w;
  const double *U;
  int32_t nU, udim;
  const double *parents;   // [4D+2][stride]
  const int32_t *action;   // [n_edges]
  int64_t n_edges, stride;
  uint8_t *free_out;       // outputs, any may be null
  double *cost;
  int32_t *cells;          // [n_edges][cell_cap]
  int32_t *cell_count;
  int32_t cell_cap;
  uint8_t *outside_out;    // [n_edges] some sample outside the map, or null
};
hipError_t launch_check_edges(int dim, int control, const EdgeArgs &args, hipStream_t s);

// Successor post-processing (post_kernel.hip): heuristic, goal tolerances, node identity.
struct PostArgs {
  const int32_t *count;    // [n_nodes]
  const uint64_t *hash;    // [n_nodes * nstride]
  const double *state;     // [4D+2][sstride]
  int64_t n_nodes, nstride, sstride;
  double goal[14];         // goal waypoint, 4D+2 doubles
  uint64_t goal_hash;
  double w, v_max, tol_pos, tol_vel, tol_acc, tol_yaw;
  double *heur;            // outputs, any may be null
  uint8_t *flags;
  int32_t *canon;
  struct Slot { uint64_t key; uint32_t val; uint32_t pad; };
  Slot *keys;              // identity table: cap + 1 slots, cap a power of two; key = ~0 empty, val = smallest index
  uint64_t cap;            // (keys == null and canon != null: canon was filled by launch_identity before this launch)
};
hipError_t launch_post_lists(int dim, const PostArgs &args, hipStream_t s);
hipError_t launch_post_clear_first_flags(const PostArgs &args, hipStream_t s);  // bit 2 of every emitted entry's flags back to 0

// Node identity by radix partition + per-bucket LDS tables (identity_kernel.hip): canon[g] = smallest list index with
// the same lattice hash, for every emitted successor g.  The lists are the strided form (packed lists: n_nodes = 1,
// nstride = capacity, count = &total).
struct IdentityArgs {
  const int32_t *count;
  const uint64_t *hash;
  int64_t n_nodes, nstride;
  int32_t *canon;
  uint64_t *hk[2];   // workspace: (hash, list index) pairs after partition level 1 / 2, n_slots each
  uint32_t *gi[2];
  uint32_t *cnt[2];  // per level: (digit, tile) counters -> exclusive prefix sums inside 4096-blocks
  uint32_t *tot[2];  // per level: scanned block totals, [blocks + 1]
  uint32_t *seg;     // level-1 buckets as segments of level 2: start[nb1 + 1], tile prefix[nb1 + 1]
  uint32_t *range;   // [buckets + 1]: where every fine bucket starts in the partitioned pairs
  int64_t n_slots, tiles1, tiles2_cap;
  int b1, b2;        // digit bits of the two levels (b2 = 0: one level)
  int fill;          // distinct keys one round of a bucket's LDS table takes (identity_default_fill(); tests lower it)
  // claimed form (two levels only): buckets of fixed capacity, a tile claims its run of a bucket with one returning
  // atomic on the bucket's cursor -- no histogram passes, no prefix sums.  A bucket that overflows sets *ovf_host (pinned)
  // and the caller runs the exact form above instead.
  int claimed;
  uint32_t *cur1;      // [64 x 8 shards] level-1 cursors, kCurPad words apart
  uint32_t *cur2;      // [buckets] level-2 cursors = pairs in every fine bucket
  uint32_t *tile_seg;  // level-2 tiles: segment << 20 | tile inside the segment; [n_tiles2] behind the count at [0]
  uint32_t *ovf;       // device copy of the overflow flag: the later launches of the same call return at once
  int32_t *ovf_host;
  int64_t subcap1, cap2, tiles2_max;
};
void identity_claimed_sizes(int64_t n_slots, int b1, int b2, int64_t *subcap1, int64_t *cap2, int64_t *pairs1, int64_t *pairs2,
                            int64_t *tiles2_max, int64_t *cur_words);
int identity_default_fill();
void identity_plan(int64_t n_slots, int *b1, int *b2);
void identity_sizes(int64_t n_slots, int b1, int b2, int64_t *tiles1, int64_t *tiles2_cap, int64_t *ctr1, int64_t *ctr2);
hipError_t launch_identity(const IdentityArgs &a, int64_t ctr1, int64_t ctr2, hipStream_t s);
hipError_t launch_identity_claimed(const IdentityArgs &a, int64_t cur_words, hipStream_t s);

// Map preprocessing (map_prep_kernel.hip).  d, c1, c2: 3 entries (unused axes 1 / [0,1)).
hipError_t launch_potential_passes(const int8_t *map, const int32_t *d, const int32_t *c1, const int32_t *c2, int rn,
                                   int hn, const int8_t *lut, int8_t h_max, unsigned short *tmp_a,
                                   unsigned short *tmp_b, int8_t *out, hipStream_t s);
hipError_t launch_region_boxes(const int *cells, int n_path_cells, int dim, const int32_t *d, const int32_t *rn,
                               uint32_t *bits, hipStream_t s);
hipError_t launch_unpack_region(const uint32_t *bits, int64_t n_cells, uint8_t *bytes, hipStream_t s);

// Dense slots of a chunk of nodes -> per-node successor lists (used for the
// configurations the tiled kernel does not cover).
struct CompactArgs {
  const uint8_t *status;
  const double *cost;
  const uint64_t *hash;
  const double *state;   // [F][chunk_slots]
  const int32_t *iters;
  int64_t chunk_slots;
  int32_t nU, n_fields;
  int64_t node_offset, n_nodes_chunk;
  int32_t *l_count;
  int32_t *l_action;
  double *l_cost;
  uint64_t *l_hash;
  double *l_state;
  int64_t l_stride;
  int32_t *l_iters;
  int64_t l_nstride;  // entries reserved per node (>= nU)
  int32_t l_pad;      // 1: complete the last 128-byte line of every list row (node stride is a multiple of 32)
};
hipError_t launch_compact_lists(const CompactArgs &args, hipStream_t stream);

// Launches the successor-expansion kernel specialised for (dim, control) on
// `stream`.  Returns hipSuccess or the launch error.
hipError_t launch_expand(int dim, int control, const ExpandArgs &args, hipStream_t stream);

// Element-wise math probe (see mplx_selftest_math in mplx.h).
hipError_t launch_math_probe(int op, const double *a, const double *b, double *out, int64_t n,
                             hipStream_t stream);

// Packs a byte-per-cell mask into 1 bit per cell (word = cell >> 5).
hipError_t launch_pack_region(const uint8_t *bytes, uint32_t *bits, int64_t n_cells,
                              hipStream_t stream);

}  // namespace mplx
#endif
