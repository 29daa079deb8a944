/*
 * mplx.h -- C ABI of the MI355X successor-expansion engine (libmplx.so).
 *
 * This is the drop-in boundary for ONE path of sikang/motion_primitive_library
 * (MPL v1.2): MPL::env_map<Dim>::get_succ and everything it evaluates
 *   (reference include/mpl_planner/env/env_map.h:147-172 and :90-132).
 * The reference has no FFI: the extension point is the C++ virtual
 *   env_base<Dim>::get_succ(curr, succ, succ_cost, action_idx)
 *   (include/mpl_planner/common/env_base.h:358-362),
 * installed by MapPlanner<Dim>::setMapUtil (src/mpl_planner/map_planner.cpp:14-18).
 * A reference-side env subclass binds the entry points below; the adapter a
 * maintainer would add is include/mplx_env_map.hpp and INTEGRATION.md shows it.
 *
 * All pointers are plain C pointers; no C++ or torch types cross this line.
 * Every function returns 0 on success and a negative mplx_status on failure
 * and never throws; mplx_last_error() gives the text.  There is NO CPU
 * fallback: without a usable gfx950 device mplx_create fails.
 *
 * A context is single-owner and not thread-safe (the reference's env is not
 * re-entrant either: env_base.h:402-404).  Each context owns one HIP stream;
 * every *_device call is asynchronous on that stream.
 */
#ifndef MPLX_H
#define MPLX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPLX_ABI_VERSION 9

typedef struct mplx_ctx mplx_ctx;

typedef enum {
  MPLX_OK = 0,
  MPLX_ERR_ARG = -1,     /* bad argument / call order                         */
  MPLX_ERR_HIP = -2,     /* HIP runtime error (text in mplx_last_error)       */
  MPLX_ERR_NO_DEVICE = -3,
  MPLX_ERR_STATE = -4,   /* map / controls / params not set before expand     */
  MPLX_ERR_NOMEM = -5    /* host allocation failed inside the library         */
} mplx_status;

/* Control::Control bit flags (reference include/mpl_basis/control.h:10-20). */
enum {
  MPLX_VEL = 0x01, MPLX_ACC = 0x03, MPLX_JRK = 0x07, MPLX_SNP = 0x0f,
  MPLX_VELxYAW = 0x11, MPLX_ACCxYAW = 0x13, MPLX_JRKxYAW = 0x17, MPLX_SNPxYAW = 0x1f
};

/* Per-slot outcome.  The reference emits a successor for FINITE and BLOCKED
 * (env_map.h:162-170: blocked edges are returned with cost = +inf) and skips
 * the other two (env_map.h:158-160).                                         */
enum {
  MPLX_SLOT_SKIP_SAME = 0, /* successor == current node by lattice hash       */
  MPLX_SLOT_FINITE = 1,
  MPLX_SLOT_BLOCKED = 2,
  MPLX_SLOT_SKIP_DYN = 3   /* validate_primitive rejected it                  */
};

/* The env parameter block: env_base.h:368-392 and env_map.h:294-296, set in
 * the reference through PlannerBase::setVmax/... (planner_base.h:179-229).
 * A limit <= 0 disables that check (primitive.h:485, :506).                  */
typedef struct {
  int32_t control;          /* control flag of every node of the search       */
  int32_t reserved;
  double dt;                /* env_base.h:392 dt_                             */
  double w;                 /* env_base.h:370 w_                              */
  double wyaw;              /* env_base.h:372 wyaw_                           */
  double v_max, a_max, j_max, yaw_max; /* env_base.h:382-388                  */
  double potential_weight;  /* env_map.h:294                                  */
  double gradient_weight;   /* env_map.h:296                                  */
} mplx_params;

/* Dense successor slots, slot = node * nU + control_index, all arrays of
 * length n_nodes * nU.  Any pointer may be NULL (that output is skipped).
 * `state` is field-major with row stride `state_stride` (>= n_slots, in
 * doubles): rows pos[0..D) vel[0..D) acc[0..D) jrk[0..D) yaw t  = 4D+2 rows,
 * i.e. the reference Waypoint<Dim> payload (waypoint.h:32-37).  The state and
 * hash of a skipped slot are still those of the candidate successor.
 * cost is +inf unless status == MPLX_SLOT_FINITE.                            */
typedef struct {
  uint8_t *status;
  double *cost;
  uint64_t *hash;          /* waypoint.h:93-125, classic boost::hash_combine  */
  double *state;
  int64_t state_stride;
  int32_t *iters;          /* diagnostic: executed sample-loop iterations     */
} mplx_succ;

/* ---- lifetime ----------------------------------------------------------- */
/* dim = 2 or 3 (OccMapUtil / VoxelMapUtil).  device = HIP device ordinal.    */
int mplx_create(int dim, int device, mplx_ctx **out);
void mplx_destroy(mplx_ctx *ctx);
/* Text of the last error of ctx (or of the last failed mplx_create if NULL). */
const char *mplx_last_error(const mplx_ctx *ctx);
int mplx_abi_version(void);

/* ---- environment set-up (host pointers; copied to HBM, stream-ordered) --- */
/* MapUtil<Dim>::setMap, map_util.h:84-90.  cells: int8, x fastest
 * (getIndex, map_util.h:34-41); dim/origin have `dim` entries.               */
int mplx_set_map(mplx_ctx *ctx, const int8_t *cells, const int32_t *dim, const double *origin,
                 double res);
/* ABI v8.  An edit of a few cells of the map that is already on the device (a sensor update between two plans:
 * MapPlanner::updateBlockedNodes / updateClearedNodes, map_planner.cpp:160-185, follow such an edit): cell_index[i]
 * (x + dim0 * (y + dim1 * z), MapUtil::getIndex) takes values[i].  The int8 cells and the blocked-bit map derived from
 * them are patched in place -- no 128-MiB upload and no rebuild of the derived structures per edit at 512^3; the
 * free-box table of the factorised kernels is rebuilt by the first launch of >= 4 096 nodes after the edit.  Same
 * results as mplx_set_map with the edited array (tests/test_map_prep.py).                                           */
int mplx_edit_map(mplx_ctx *ctx, const int64_t *cell_index, const int8_t *values, int64_t n);
/* ABI v9.  A cell named more than once in one mplx_edit_map call takes its LAST value (= mplx_set_map with the edited
 * array).  Host -> device bytes the context's map calls have moved since mplx_create (mplx_set_map, _set_potential,
 * _set_region: one byte per cell; mplx_edit_map: 9 bytes per edited cell): what a re-planning loop checks to see that an
 * edit of k cells cost k cells, not the map (tests/test_lpastar.py).                                                   */
int mplx_map_upload_bytes(mplx_ctx *ctx, uint64_t *bytes);
/* ABI v9.  The values of a few cells of a map that lives on the device: which = 0 the map, 1 the potential map;
 * out[i] = cells[cell_index[i]].  Synchronous (one small gather launch + copy).                                         */
int mplx_read_cells(mplx_ctx *ctx, int which, const int64_t *cell_index, int64_t n, int8_t *out);

/* env_map::set_potential_map, env_map.h:181-183.  NULL clears it.  Same size
 * as the map.                                                                */
int mplx_set_potential(mplx_ctx *ctx, const int8_t *cells_or_null);
/* env_base::set_search_region, env_base.h:301-303.  One byte per cell
 * (non-zero = inside), NULL clears it.                                       */
int mplx_set_region(mplx_ctx *ctx, const uint8_t *cells_or_null);
/* env_base setters, env_base.h:237-292.                                      */
int mplx_set_params(mplx_ctx *ctx, const mplx_params *p);
/* env_base::set_u, env_base.h:234.  U is [nU][udim] row-major; udim = D, or
 * D+1 when the control flag carries yaw (last entry = yaw rate).             */
int mplx_set_controls(mplx_ctx *ctx, const double *U, int32_t nU, int32_t udim);

/* ---- map preprocessing on the device (the producers of the potential map and
 *      of the search region) ------------------------------- */
/* MapPlanner<Dim>::updatePotentialMap(pos) with createMask
 * (src/mpl_planner/map_planner.cpp:246-283, 286-391): every cell > 0 inside the
 * update box becomes H_MAX = 100 and stamps the int8 cone mask
 *   H_MAX * pow((1 - hypot(n0,n1)/rn) [* (1 - |n2|/hn)], pow)
 * around itself with `max`.  As in the reference the result REPLACES the map and
 * is installed as the potential map.  radius: `dim` doubles
 * (setPotentialRadius); range_or_null: `dim` doubles (setPotentialMapRange; NULL
 * or all zero = whole map) centred on pos; pow: the reference's pow_ (1.0).
 * h_map_out_or_null receives the new map (for the caller's MapUtil).          */
int mplx_update_potential_map(mplx_ctx *ctx, const double *pos, const double *radius,
                              const double *range_or_null, double pow, int8_t *h_map_out_or_null);
/* MapPlanner<Dim>::setSearchRegion(path, dense) (map_planner.cpp:46-95): the
 * cells along the path (MapUtil::rayTrace, map_util.h:117-135, when dense == 0 --
 * the reference's flag is inverted and so is this one) dilated by
 * ceil(search_radius / res) cells per axis.  path: [n_points][dim].  Installs the
 * region like mplx_set_region; h_region_out_or_null receives one byte per cell. */
int mplx_set_search_region_path(mplx_ctx *ctx, const double *path, int32_t n_points, int32_t dense,
                                const double *search_radius, uint8_t *h_region_out_or_null);

/* ---- expansion ---------------------------------------------------------- */
/* Batched get_succ on device-resident buffers, asynchronous on the context
 * stream.  d_nodes is field-major [4D+2][node_stride] (same rows as `state`),
 * n_nodes <= node_stride.  All pointers in d_out are device pointers.        */
int mplx_expand_device(mplx_ctx *ctx, const double *d_nodes, int64_t n_nodes, int64_t node_stride,
                       const mplx_succ *d_out);
/* Same with host buffers: H2D, kernel, D2H, synchronised on return.          */
int mplx_expand(mplx_ctx *ctx, const double *h_nodes, int64_t n_nodes, int64_t node_stride,
                const mplx_succ *h_out);
/* Exactly env_base<Dim>::get_succ for one node (env_base.h:358-362): compact
 * lists in ascending control index.  node: 4D+2 doubles.  succ: [nU][4D+2]
 * doubles (waypoint-major), cost: [nU], action: [nU]; *n_succ receives the
 * count.  Blocked successors are returned with cost = +inf, as the reference
 * does (env_map.h:162-170).                                                  */
int mplx_get_succ(mplx_ctx *ctx, const double *node, double *succ, double *cost, int32_t *action,
                  int32_t *n_succ);

/* Per-node successor LISTS -- the reference's own output shape (succ,
 * succ_cost, action_idx of env_base.h:358-362) for a whole frontier.  Node k
 * owns the entries [k*S, k*S + count[k]) of every array, S = node_stride (or nU
 * when node_stride is 0), in ascending control index; blocked successors are included with cost = +inf (env_map.h:162-170),
 * skipped ones are not.  Only emitted successors are written, so this is the
 * bandwidth-lean output (SURVEY.md 8d counts exactly these bytes).  Any pointer
 * but `count` may be NULL.  `state` as in mplx_succ (row stride state_stride >=
 * n_nodes*S doubles).  A node_stride that is a multiple of 32 entries makes
 * every 64-successor store of the kernels a set of full 128-byte lines and lets
 * the kernel complete the last line of each list (entries past count[k] up to
 * the next multiple of 32 are then unspecified): ~20 % faster on C4 than S = 729. */
typedef struct {
  int32_t *count;          /* [n_nodes]                                       */
  int32_t *action;         /* [n_nodes*S] control index of each successor     */
  double *cost;            /* [n_nodes*S]                                     */
  uint64_t *hash;          /* [n_nodes*S]                                     */
  double *state;           /* [4D+2][state_stride]                            */
  int64_t state_stride;
  int32_t *iters;          /* diagnostic: executed sample-loop iterations     */
  int64_t node_stride;     /* S: entries reserved per node, >= nU; 0 = nU     */
  /* ABI v8 -- what the graph search computes for every successor right after
   * get_succ (graph_search.h:84-88), written by the expansion kernels while the
   * successor is still in registers; needs mplx_set_goal:     */
  double *heur;            /* [n_nodes*S] env_base::get_heur, default branch (env_base.h:46-64), or NULL */
  uint8_t *flags;          /* [n_nodes*S] bit 0 inside the goal tolerances (env_map.h:25-37, no ray trace),
                              bit 1 same lattice state as the goal (env_base.h:47), or NULL              */
} mplx_succ_lists;

/* Batched get_succ producing lists; device pointers, asynchronous on the
 * context stream.  Kernel: expand_grid_kernel.hip for control tables with at
 * most 16 distinct values per axis (incl. yaw rates; potential maps with
 * gradient_weight == 0), else expand_tile_kernel.hip, else the dense kernel
 * followed by an on-device compaction (mplx_set_lists_route forces one).      */
int mplx_expand_lists_device(mplx_ctx *ctx, const double *d_nodes, int64_t n_nodes, int64_t node_stride,
                             const mplx_succ_lists *d_out);
/* Same with host buffers, synchronised.  Only the used prefix of every list is
 * written to the host arrays (entries past count[k] are left untouched).  Up to
 * 8 MiB of lists the kernel reads the nodes from and writes the lists to a pinned
 * host block itself (a single get_succ: ~25 - 45 us end to end); larger batches
 * are packed on the device and pipelined through pinned buffers at the PCIe link
 * rate.                                                                        */
int mplx_expand_lists(mplx_ctx *ctx, const double *h_nodes, int64_t n_nodes, int64_t node_stride,
                      const mplx_succ_lists *h_out);

/* ---- successor post-processing on the device: what
 *      GraphSearch::Astar does with every successor right after get_succ
 *      (graph_search.h:84-88, :146), for a whole batch of lists in HBM ------ */
typedef struct {
  const double *goal;      /* host pointer: goal waypoint, 4D+2 doubles         */
  int32_t control;         /* Waypoint::control of the successors               */
  int32_t goal_control;    /* Waypoint::control of the goal: the `goal_node_ == state` test of
                              env_base.h:47 hashes each side with ITS OWN flags (waypoint.h:93-125);
                              0 = same as `control`                               */
  double w, v_max;         /* env_base.h:370, :382 (cal_heur, :58-64)           */
  double tol_pos, tol_vel, tol_acc, tol_yaw; /* env_base.h:374-380; vel / acc /
                              yaw tests are skipped when < 0 (env_map.h:29-36)  */
} mplx_goal_spec;

/* ABI v8.  env_base::set_goal (env_base.h:274-276) for the device: the goal the `heur` / `flags` rows of
 * mplx_succ_lists refer to -- the default heuristic w * |pos - goal.pos|_inf / v_max (w, v_max as given here: the
 * search's, not necessarily the expansion's) and the tolerance tests of is_goal.  Copied; NULL clears it.  A launch
 * that asks for either row without a goal fails with MPLX_ERR_STATE.  Bit-identical to mplx_post_lists_device on
 * the same lists (tests/test_gpu_post.py), without reading them back: +9 bytes per successor on the launch instead
 * of a second pass over hash and position rows.                                                                   */
int mplx_set_goal(mplx_ctx *ctx, const mplx_goal_spec *goal);

/* Outputs, device pointers of n_nodes*S entries each (S = node_stride of the
 * lists), any may be NULL; only the entries of emitted successors are written.
 *   heur   env_base::get_heur, default branch (env_base.h:46-64): 0 for the
 *          goal's own lattice state, else w * |pos - goal.pos|_inf / v_max
 *          (w * |.|_inf when v_max <= 0)
 *   flags  bit 0: inside the goal tolerances (env_map.h:25-37; the ray trace of
 *          :38-43 is left to the caller), bit 1: same lattice state as the
 *          goal, bit 2 (with canon): first successor of the batch with its hash
 *   canon  list index (node*S + j) of the first successor of the batch with the
 *          same lattice hash -- the search's node identity (waypoint.h:128-135) */
typedef struct {
  double *heur;
  uint8_t *flags;
  int32_t *canon;
} mplx_post;

/* d_lists: the lists as filled by mplx_expand_lists_device (count, hash and
 * state are read).  Asynchronous on the context stream -- except with canon on
 * batches of >= 256 k list slots (ABI v7): their node identity claims runs in
 * buckets of fixed capacity, and the call waits for the stream ONCE, after
 * queueing everything, to learn that no bucket overflowed (then it returns), or
 * else runs the capacity-free form behind it (mplx_last_identity_form) -- and
 * then takes that form directly, asynchronously, for the context's next 8 calls
 * with canon (ABI v8: a frontier that overflows does so call after call).      */
int mplx_post_lists_device(mplx_ctx *ctx, const mplx_succ_lists *d_lists, int64_t n_nodes,
                           const mplx_goal_spec *goal, const mplx_post *d_out);

/* ---- packed lists and the multi-GPU exchange ----------------
 * The frontier is block-partitioned by node over the GPUs (rank r expands nodes
 * [r*N/G, (r+1)*N/G) against its own replica of the map: every (node, control)
 * pair is independent, env_map.h:147-172 reads nothing but its arguments), so the
 * expansion itself needs NO collective.  When a consumer wants the complete
 * successor set on every GPU (an on-device dedup / open-list merge; north_star:
 * "an RCCL all-gather of successor lists over xGMI only when the open set exceeds
 * a single GPU's launch"), the used prefixes of the lists are first PACKED on the
 * device -- node k owns entries [offs[k], offs[k+1]) of every row, no padding --
 * and the packed rows are all-gathered with exact sizes.                       */
typedef struct {
  int32_t *count;          /* [n_nodes]      copy of the lists' counts           */
  int64_t *offs;           /* [n_nodes + 1]  exclusive prefix sums; offs[n_nodes] = total entries */
  int32_t *action;         /* [capacity]     any row pointer may be NULL         */
  double *cost;
  uint64_t *hash;
  double *state;           /* [4D+2][state_stride]                               */
  int64_t state_stride;    /* >= capacity                                        */
  int64_t capacity;        /* entries every non-NULL row can hold                */
} mplx_packed_lists;

/* d_lists (as filled by mplx_expand_lists_device) -> d_out, all device pointers,
 * asynchronous on the context stream.  A capacity >= n_nodes * nU always
 * suffices; a smaller one is checked against the actual total (one
 * synchronisation) and the call fails with MPLX_ERR_ARG when it is too small.
 * h_total_or_null, when given, receives offs[n_nodes] (synchronises).           */
int mplx_pack_lists_device(mplx_ctx *ctx, const mplx_succ_lists *d_lists, int64_t n_nodes,
                           const mplx_packed_lists *d_out, int64_t *h_total_or_null);

/* mplx_post_lists_device on PACKED lists (entry i of the outputs belongs to packed entry i; `canon` holds packed
 * indices): the consumer of the all-gather -- every rank runs it on the gathered set and so learns, without the
 * host, which successors of the whole frontier are first occurrences of their lattice state (the search's node
 * identity, waypoint.h:128-135), their heuristic (env_base.h:46-64) and goal flags (env_map.h:25-37): the on-device
 * open-list merge north_star names as the reason to gather at all.  d_packed needs offs, hash and state.
 * Like mplx_post_lists_device it waits for the context's stream ONCE when canon is asked for on >= 256 k entries
 * (the claimed identity pass learns that no bucket overflowed); after an overflow the next 8 calls of the context
 * with canon take the capacity-free form directly and stay asynchronous (ABI v8).                                */
int mplx_post_packed_device(mplx_ctx *ctx, const mplx_packed_lists *d_packed, int64_t n_nodes,
                            const mplx_goal_spec *goal, const mplx_post *d_out);

/* RCCL communicator of the context (one context = one GPU = one rank; one process
 * per GPU or several contexts in one process).  The 128-byte id is RCCL's
 * ncclUniqueId: rank 0 makes it, the application hands it to the other ranks by
 * whatever means it has (MPI, a socket, torch.distributed ...).  librccl.so.1 is
 * loaded on first use; libmplx.so does not link it.                             */
#define MPLX_COMM_ID_BYTES 128
int mplx_comm_unique_id(uint8_t *id_out /* [MPLX_COMM_ID_BYTES] */);
int mplx_comm_init(mplx_ctx *ctx, const uint8_t *id, int32_t rank, int32_t world);
int mplx_comm_destroy(mplx_ctx *ctx);
/* Replicates the map (and the potential map / search region, when set) of rank
 * `root` on every rank with ncclBroadcast over xGMI instead of one H2D copy per
 * rank; every rank must have called mplx_set_map with the same geometry.        */
int mplx_comm_broadcast_map(mplx_ctx *ctx, int32_t root);
/* All-gather of packed lists: rank r contributes d_local (n_local nodes, packed by
 * mplx_pack_lists_device); every rank receives the concatenation in rank order
 * (= ascending global node index for the block partition) in d_all, whose rows
 * must hold the global totals (n_nodes_total nodes; capacity >= sum of totals).
 * Exact-size exchange: one small ncclAllGather of (n_local, total) pairs, then an
 * all-pairs ncclSend / ncclRecv of the rows inside ONE group call -- direct
 * peer-to-peer copies, all xGMI links busy at once, no padding and no ring.  d_all->offs is
 * rebuilt from the gathered counts.  h_node_offs / h_entry_offs ([world + 1], may
 * be NULL) receive the per-rank node and entry offsets.  Synchronises.          */
int mplx_comm_allgather_lists(mplx_ctx *ctx, const mplx_packed_lists *d_local, int64_t n_local,
                              const mplx_packed_lists *d_all, int64_t *h_node_offs, int64_t *h_entry_offs);

/* The schedule of that exchange as a PURE function (host arithmetic only: no
 * device, no RCCL, no context) -- what mplx_comm_allgather_lists executes, and
 * what a host with another transport (MPI, shared memory, a test's in-memory
 * mailboxes) can execute itself.  `meta` is what the ranks all-gather first,
 * MPLX_COMM_META int64 words per rank:
 *   [0] n_local nodes   [1] packed entries   [2] mask of the rows the rank wants
 *   gathered (MPLX_ROWBIT_*)   [3] capacity of the rank's d_all   [4] status of
 *   the rank's own argument checks (MPLX_OK or an error code)   [5..7] zero.
 * Every rank computes the SAME verdict from the same meta, so the ranks fail
 * together instead of one returning while its peers wait in the group: the call
 * returns MPLX_ERR_ARG / MPLX_ERR_STATE (and writes no ops) when any rank
 * reported a status, the row masks differ, or the gathered entries exceed any
 * rank's capacity.  Otherwise it returns the number of ops of `rank` (also when
 * ops is NULL or `cap` too small: call again with room) and fills
 * ops[0 .. min(n, cap)): first the local copies of the rank's own block, then per
 * step d = 1 .. world-1 and per row one send to (rank + d) % world and one
 * receive from (rank - d) % world (a different peer pair on every rank in every
 * step: all xGMI links busy), zero-byte transfers omitted on BOTH sides (a
 * transport pairs the k-th send of a -> b with the k-th receive of b from a).
 * node_offs / entry_offs ([world + 1], may be NULL) receive the prefix sums.     */
#define MPLX_COMM_META 8
enum { MPLX_ROWBIT_ACTION = 1, MPLX_ROWBIT_COST = 2, MPLX_ROWBIT_HASH = 4, MPLX_ROWBIT_STATE = 8 };
enum { MPLX_COMM_COPY = 0, MPLX_COMM_SEND = 1, MPLX_COMM_RECV = 2 };
enum { MPLX_ROW_COUNT = 0, MPLX_ROW_ACTION = 1, MPLX_ROW_COST = 2, MPLX_ROW_HASH = 3, MPLX_ROW_STATE0 = 4 /* + field */ };
typedef struct {
  int32_t kind;     /* MPLX_COMM_COPY / _SEND / _RECV                                      */
  int32_t peer;     /* the other rank (the own rank for a copy)                            */
  int32_t row;      /* MPLX_ROW_*; a state field f is MPLX_ROW_STATE0 + f                  */
  int32_t elem;     /* bytes per element of the row (4 or 8)                               */
  int64_t src_off;  /* bytes into the rank's LOCAL row (send, copy); 0 for a receive       */
  int64_t dst_off;  /* bytes into the rank's GATHERED row (receive, copy); 0 for a send    */
  int64_t bytes;
} mplx_comm_op;
int64_t mplx_comm_schedule(int32_t world, int32_t rank, const int64_t *meta, int32_t n_fields,
                           mplx_comm_op *ops, int64_t cap, int64_t *node_offs, int64_t *entry_offs);

/* ---- batched re-validation of stored edges for incremental re-planning
 *       --------------------------------------------------- */
/* For every edge (parent waypoint, action id): env_base::forward_action
 * (env_base.h:228-231), then
 *   free_flag   env_map::is_free(Primitive), env_map.h:60-76 (uniform samples
 *               i * t/n, n = ceil(max_v * t / res); occupied / outside / outside
 *               the search region -> 0); an edge that does not move (n == 0) is
 *               reported not free (the reference's behaviour there is undefined)
 *   cost        calculate_intrinsic_cost (env_base.h:343-345) = J + w*dt for a
 *               free edge, +inf otherwise -- what StateSpace::decreaseCost
 *               installs (state_space.h:230-253)
 *   cells       MapPlanner::getLinkedNodes (map_planner.cpp:125-157): the cell
 *               indices the samples fall into, consecutive repeats removed;
 *               cell_count[e] entries of row e ([n_edges][cell_cap]; a count
 *               above cell_cap means the row was truncated)
 * All pointers are host pointers; any output may be NULL.                     */
typedef struct {
  uint8_t *free_flag;
  double *cost;
  int32_t *cells;
  int32_t *cell_count;
  int32_t cell_cap;
  uint8_t *outside;        /* [n_edges] non-zero when a sample of the edge fell outside the map (its `cells`
                              entry is then the reference's own out-of-range int index arithmetic) */
} mplx_edges_out;
/* parents: field-major [4D+2][stride] (stride >= n_edges), actions: [n_edges]. */
int mplx_check_edges(mplx_ctx *ctx, const double *h_parents, const int32_t *h_actions, int64_t n_edges,
                     int64_t stride, const mplx_edges_out *h_out);

/* ---- device memory + stream helpers (so any host language can keep the
 *      frontier and the successor slots resident in HBM) ------------------- */
/* Ordering rule for buffers a launch READ (the frontier): writes through this API
 * (mplx_memcpy_h2d, mplx_memset) are ordered after every earlier launch on the
 * context, including the deferred heading-limit re-check of yaw controls, which
 * re-reads the flagged nodes.  Writes the library cannot see (another stream,
 * another library writing into memory handed over as pointers) must be preceded
 * by mplx_synchronize -- as they must anyway not to race with the launch itself. */
int mplx_device_alloc(mplx_ctx *ctx, size_t bytes, void **dptr);
int mplx_device_free(mplx_ctx *ctx, void *dptr);
int mplx_memcpy_h2d(mplx_ctx *ctx, void *dst, const void *src, size_t bytes);
int mplx_memcpy_d2h(mplx_ctx *ctx, void *dst, const void *src, size_t bytes);
int mplx_memset(mplx_ctx *ctx, void *dst, int value, size_t bytes);
int mplx_synchronize(mplx_ctx *ctx);
/* HIP-event stopwatch on the context stream: begin ... (launches) ... end.
 * mplx_timer_end synchronises and returns elapsed milliseconds.              */
int mplx_timer_begin(mplx_ctx *ctx);
int mplx_timer_end(mplx_ctx *ctx, float *ms);


/* ---- host search around the device get_succ (the CALLER of the hot path) --
 * MapPlanner<Dim>::plan == PlannerBase::plan + GraphSearch::Astar
 * (reference include/mpl_planner/common/planner_base.h:275-325,
 *  include/mpl_planner/common/graph_search.h:39-182).  The search itself stays
 * on the host, as in the reference; only get_succ is served by the engine.
 * With batch > 1 the planner expands the popped node together with the best
 * unexpanded OPEN nodes in one launch and serves later pops from that cache;
 * get_succ is a pure function of the node, so the plan is unchanged.         */
typedef struct mplx_planner mplx_planner;

/* env_base<Dim>::get_succ as a C hook (env_base.h:358-362): lets any other
 * env implementation feed the search; same buffers as mplx_get_succ.         */
typedef int (*mplx_succ_fn)(void *user, const double *node, double *succ, double *cost,
                            int32_t *action, int32_t *n_succ);
/* Batched hook: nodes field-major [4D+2][n]; dense slots out (status, cost,
 * state [4D+2][n*nU]) exactly as mplx_expand fills them.                     */
typedef int (*mplx_batch_fn)(void *user, const double *nodes, int64_t n, uint8_t *status,
                             double *cost, double *state);

typedef struct {
  int32_t control;      /* control flag of start / goal (Waypoint::control)   */
  int32_t max_expand;   /* PlannerBase::setMaxNum, planner_base.h:251; <=0 off */
  int32_t batch;        /* nodes per launch; 1 = one-at-a-time like Astar     */
  int32_t goal_control; /* control flag of the goal waypoint when it differs from the start's (the
                           reference's test_distance_map_planner_2d_with_yaw: ACCxYAW start, ACC goal);
                           0 = same as `control`                                */
  double dt, w, v_max;  /* used by the heuristic (env_base.h:58-64)           */
  double epsilon;       /* PlannerBase::setEpsilon, planner_base.h:238        */
  double tol_pos, tol_vel, tol_acc, tol_yaw; /* setTol, planner_base.h:262    */
} mplx_planner_config;

typedef struct {
  int32_t ok;              /* plan() return value                             */
  int32_t expansions;      /* expand_iteration, graph_search.h:64             */
  int32_t closed, opened, nodes; /* getCloseSet / getOpenSet sizes, hm_ size  */
  int32_t device_launches; /* provider calls made                             */
  int64_t pairs;           /* node x control pairs the provider evaluated     */
  double cost;             /* traj_cost_ (goal g-value)                       */
  double total_time;       /* Trajectory::getTotalTime                        */
  double J[4];             /* Trajectory::J(VEL, ACC, JRK, SNP)               */
  int32_t segments;
  int32_t state_mismatches; /* diagnostic: with MPLX_PLAN_CHECK_STATES=1, successors whose host-evaluated
                              state differed from the device's (must be 0)       */
} mplx_plan_summary;

int mplx_planner_create(int dim, mplx_planner **out);
void mplx_planner_destroy(mplx_planner *p);
/* Product wiring: successors come from this engine context (mplx_get_succ /
 * mplx_expand).  The context must outlive the planner.                       */
int mplx_planner_attach_ctx(mplx_planner *p, mplx_ctx *ctx);
/* Alternative wiring for other env implementations (either hook may be NULL). */
int mplx_planner_set_provider(mplx_planner *p, mplx_succ_fn single, mplx_batch_fn batched, void *user);
/* Host copy of the map for the start / goal tests (env_map.h:25-51).         */
int mplx_planner_set_map(mplx_planner *p, const int8_t *cells, const int32_t *dim,
                         const double *origin, double res);
/* ABI v9.  A few cells of that host copy take new values (the planner's side of mplx_edit_map: O(edited cells), not a
 * second copy of the map per updateBlockedNodes / updateClearedNodes).  Last value wins for a repeated cell.          */
int mplx_planner_edit_map(mplx_planner *p, const int64_t *cell_index, const int8_t *values, int64_t n);
int mplx_planner_set_controls(mplx_planner *p, const double *U, int32_t nU, int32_t udim);
int mplx_planner_configure(mplx_planner *p, const mplx_planner_config *cfg);
/* start / goal: 4D+2 doubles each.  Returns 0 when the search ran (see
 * out->ok for success), negative on provider / argument errors.              */
int mplx_planner_plan(mplx_planner *p, const double *start, const double *goal, mplx_plan_summary *out);
/* Trajectory of the last successful plan: segment start states [segments][4D+2]
 * and the control index of each segment (env_base::forward_action).          */
int mplx_planner_trajectory(mplx_planner *p, double *nodes, int32_t *actions, int32_t cap_segments);
/* The state the last segment reaches (4D+2 doubles): with the segment start states, the way points of
 * Trajectory::getWaypoints (trajectory.h), which MapPlanner::iterativePlan turns into the next search region
 * (map_planner.cpp:404-410).                                                   */
int mplx_planner_trajectory_end(mplx_planner *p, double *node);
/* Closed-set positions of the last plan (PlannerBase::getCloseSet): fills up
 * to cap points of D doubles; *n receives the closed-set size.               */
int mplx_planner_closed_set(mplx_planner *p, double *pos, int32_t cap, int32_t *n);
/* Open-set states of the last plan (PlannerBase::getOpenSet returns their positions): fills up to
 * cap rows of 4D+2 doubles; *n receives the open-set size.                    */
int mplx_planner_open_set(mplx_planner *p, double *states, int32_t cap, int32_t *n);
const char *mplx_planner_last_error(const mplx_planner *p);
/* ABI v8.  Where the wall time of the last mplx_planner_plan went and what its relaxation loop did (the search is
 * host code around the device's get_succ: graph_search.h:63-143 spends its time per relaxed edge).               */
typedef struct {
  double total_ms;      /* the whole plan() call                                                                  */
  double provider_ms;   /* inside the successor provider: launches, completion, transfers                         */
  double fill_ms;       /* moving lists that outlive their launch out of the landing buffer                       */
  double pick_ms;       /* choosing the nodes of the next launch (best open nodes, children that ride along)      */
  double relax_ms;      /* everything else of the search loop: relaxation passes, node table, heap, goal tests    */
  double recover_ms;    /* recoverTraj                                                                            */
  int64_t relaxed;      /* finite edges relaxed (graph_search.h:97-99 records each as a predecessor)              */
  int64_t improved;     /* ... of which lowered the child's g (graph_search.h:108)                                */
  int64_t pushes;       /* heap pushes (the rest of `improved` were updates in place)                             */
  int64_t materialised; /* nodes whose 4D+2 state was ever built on the host (picked for a launch or on the path) */
  int64_t heur_from_device; /* new nodes whose heuristic came with their list from the expansion launch (the `heur`
                               row of mplx_succ_lists) instead of being evaluated by the search                     */
} mplx_plan_timing;
int mplx_planner_timing(const mplx_planner *p, mplx_plan_timing *out);
/* ABI v8.  PlannerBase::setPriorTrajectory (planner_base.h:249-252; env_map::set_prior_trajectory, env_map.h:189-226): the
 * last trajectory of `from` (another planner, possibly with another control order) guides this planner's search -- a
 * state at time t is drawn to where that trajectory is at t (env_base::get_heur, env_base.h:46-52).  Call it after this
 * planner's map, v_max, w and dt are set, as the reference's programs do.  Occupancy maps (with a potential map: the
 * _potential variant below).  from == NULL: no prior trajectory.                                                       */
int mplx_planner_set_prior_trajectory(mplx_planner *p, const mplx_planner *from);
/* ABI v9.  The same with a potential map installed (env_map::set_prior_trajectory / traverse_trajectory with
 * potential_map_, env_map.h:197-216, 241-249): the prior's remaining cost carries potential_weight * value +
 * gradient_weight * |vel| of every cell its samples pass through, and a prior that enters an obstacle's core (value >=
 * 100) costs +inf.  potential: the host copy of the potential map (same size as the map), or NULL to read the values
 * of the few hundred cells concerned from the potential map the attached context holds on the device (mplx_read_cells). */
int mplx_planner_set_prior_trajectory_potential(mplx_planner *p, const mplx_planner *from, const int8_t *potential,
                                                double potential_weight, double gradient_weight);

/* ---- ABI v8: Lifelong Planning A* (PlannerBase::setLPAstar, planner_base.h:170-176; GraphSearch::LPAstar,
 *      graph_search.h:194-365; StateSpace::updateNode / increaseCost / decreaseCost / getSubStateSpace,
 *      state_space.h:116-281) in the engine's own planner.  With it on, the state space outlives mplx_planner_plan:
 *      a map edit is translated into edge-cost changes through the voxel -> edge table and the next plan repairs
 *      only what became inconsistent.  The caller edits its map, hands the new cells to mplx_planner_set_map (and to
 *      the context: mplx_set_map), then names the edited cells:                                                     */
int mplx_planner_set_lpastar(mplx_planner *p, int on);          /* PlannerBase::setLPAstar                        */
int mplx_planner_reset(mplx_planner *p);                        /* PlannerBase::reset: forget the state space     */
/* MapPlanner::getLinkedNodes (map_planner.cpp:125-157): (re)builds the voxel -> edge table from every stored edge --
 * ONE mplx_check_edges call for all of them -- and returns the linked points (cell centres, D doubles each; up to
 * cap_points are written, *n_points receives their number), the number of cells and of (cell, edge) entries.       */
int mplx_planner_linked_nodes(mplx_planner *p, double *points, int64_t cap_points, int64_t *n_points, int64_t *n_cells,
                              int64_t *n_entries);
/* MapPlanner::updateBlockedNodes / updateClearedNodes (map_planner.cpp:160-185): cells = [n][D] integer cell
 * coordinates that became occupied / free.  Cleared: is_free + intrinsic cost of all affected blocked edges in one
 * mplx_check_edges call against the context's CURRENT map (set it first).                                          */
int mplx_planner_update_blocked_nodes(mplx_planner *p, const int32_t *cells, int64_t n);
int mplx_planner_update_cleared_nodes(mplx_planner *p, const int32_t *cells, int64_t n);
/* StateSpace::getSubStateSpace (state_space.h:116-195): re-root the search tree at way point `time_step` of the last
 * trajectory (a robot that has executed that many primitives).                                                     */
int mplx_planner_sub_state_space(mplx_planner *p, int32_t time_step);
/* Another implementation of the batched edge re-validation (tests: the CPU oracle), shape of mplx_check_edges.     */
typedef int (*mplx_edges_fn)(void *user, const double *parents, const int32_t *actions, int64_t n_edges, uint8_t *free_flag,
                             double *cost, int32_t *cells, int32_t *cell_count, int32_t cell_cap);
int mplx_planner_set_edge_provider(mplx_planner *p, mplx_edges_fn fn, void *user);
/* ABI v8.  on != 0: the plans of this planner call mplx_set_goal on the attached context and take the heuristic of new
 * nodes from the `heur` row the expansion launches write (SURVEY.md 8f-2 with its consumer); 0 (the default): the search
 * evaluates it itself from the successor's position.  Same search bit for bit either way (tests/test_gpu_plan.py); on
 * the 3D problems of the bench line the row costs 3 % (8 bytes per successor over PCIe for a value 7 % of them need). */
int mplx_planner_use_device_heuristic(mplx_planner *p, int on);

/* Yaw controls: validate_yaw (primitive.h:504-525) compares with cos(yaw_max) and the reference's cos / sin are the
 * host libm's, the device's differ from them in the last place on a few per cent of arguments.  The engine therefore
 * flags every node with a heading-limit decision within rounding noise (2^-46) of its threshold and re-expands exactly
 * those nodes with trig values the HOST computes with its libm, at the next synchronising call (mplx_synchronize,
 * mplx_timer_end, mplx_memcpy_d2h, any host-pointer entry point): the successor SET is the reference's for every input,
 * not only for the tested ones.  The results of an asynchronous *_device launch are final after such a call.
 * (Statistics of that pass: mplx_yaw_pin_stats in mplx_debug.h.)                                                    */
/* Which kernel serves mplx_expand_lists*: AUTO picks the fastest one that
 * covers the configuration (GRID: controls with <= 16 distinct values per axis,
 * no yaw, no potential, bounded velocity; TILE: any control table, otherwise the
 * same scope; DENSE: everything, dense kernel + on-device compaction).  All
 * three produce identical lists; forcing a route that does not cover the
 * configuration makes the expand call fail with MPLX_ERR_STATE.  Used by the
 * parity tests to cross-check the kernels.                                    */
enum { MPLX_ROUTE_AUTO = 0, MPLX_ROUTE_DENSE = 1, MPLX_ROUTE_TILE = 2, MPLX_ROUTE_GRID = 3 };
int mplx_set_lists_route(mplx_ctx *ctx, int route);
/* Route taken by the last mplx_expand_lists* call (MPLX_ROUTE_*).            */
int mplx_last_lists_route(const mplx_ctx *ctx);
/* ABI v6.  The GRID route has two kernels with identical results: the general factorised one, and one for the
 * control tables the reference's programs build -- the nested-loop (lexicographic) enumeration of per-axis values,
 * no yaw, occupancy map (expand_lex_kernel.hip; MPLX_GRID_LEX=0 sends those to the general kernel too).  Which one
 * the last mplx_expand_lists* call ran: MPLX_KERNEL_NONE when the route was not GRID.                              */
enum { MPLX_KERNEL_NONE = 0, MPLX_KERNEL_GRID = 1, MPLX_KERNEL_LEX = 2,
       /* ABI v9: expand_pair_kernel.hip -- yaw controls on a potential map over a pre-screened frontier, two nodes per
        * wave (MPLX_GRID_PAIR=0 keeps such launches on the general kernel); identical lists                          */
       MPLX_KERNEL_PAIR = 3 };
int mplx_last_grid_kernel(const mplx_ctx *ctx);
/* ABI v7.  Which form of the node-identity pass the last mplx_post_*_device call with canon ran (all produce the
 * same canon[]): the table in HBM (small batches), the claimed partition (buckets of fixed capacity, one host round
 * trip to learn that none overflowed), the exact partition (histograms + prefix sums; MPLX_POST_CLAIMED=0), or the
 * claimed one followed by the exact one because a bucket overflowed (heavy duplication of few lattice states).      */
enum { MPLX_IDENTITY_TABLE = 0, MPLX_IDENTITY_CLAIMED = 1, MPLX_IDENTITY_EXACT = 2, MPLX_IDENTITY_CLAIMED_THEN_EXACT = 3 };
int mplx_last_identity_form(const mplx_ctx *ctx);
/* The service: how mplx_expand_lists (and mplx_get_succ, which calls it) serves
 * the small synchronous batches of a search -- at most MPLX_SERVICE_MAX_NODES
 * (256) nodes, control tables without yaw, no potential map, bounded velocity.
 * From the second such call in a row the batch goes to a kernel that STAYS
 * RESIDENT between the calls and takes its requests from a mailbox in pinned
 * host memory: ~4 us per round trip instead of the ~12 us of launch +
 * synchronise (expand_tile_kernel.hip, "SERVICE MODE").  The lists are the same
 * bytes either way.  The resident kernel leaves on its own after
 * MPLX_SERVICE_IDLE_US (2000) without a request and is asked to leave by every
 * other call into the context, so no call ever sees it; while it waits it
 * occupies a few workgroups of the device.  mode: 1 = on (the default; env
 * MPLX_SERVICE=0 turns it off per process), 0 = off (ends a resident kernel),
 * -1 = leave as is.  stats (may be NULL): [0] batches served by a resident
 * kernel, [1] times one was launched, [2] times one failed to answer (the batch
 * is then run as a launch of its own and the service stays off for the
 * context), [3] 1 if one is resident now.                                     */
int mplx_service(mplx_ctx *ctx, int mode, int64_t stats[4]);
/* Fills name (up to cap bytes) with the device name and gcn arch.            */
int mplx_device_info(mplx_ctx *ctx, char *name, size_t cap, int32_t *compute_units);

#ifdef __cplusplus
}
#endif
#endif
