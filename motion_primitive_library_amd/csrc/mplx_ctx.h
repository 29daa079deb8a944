// mplx_ctx.h -- the context object behind the C ABI (include/mplx.h) and the small
// helpers every API translation unit uses.  Private to libmplx.so.
#ifndef MPLX_CTX_H
#define MPLX_CTX_H

#include "../../include/mplx.h"
#include "../../include/mplx_debug.h"
#include "mplx_internal.h"

#include <cstdarg>
#include <cstdio>
#include <exception>
#include <new>
#include <string>
#include <vector>

namespace mplx_detail {

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
};

std::string &create_error();  // text of the last failed mplx_create (thread local)

}  // namespace mplx_detail

struct mplx_ctx {
  int dim = 0;
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::string err;

  // environment (device copies)
  mplx_detail::DevBuf map, pot, region_bits, region_bytes, U;
  bool has_map = false, has_pot = false, has_region = false, has_params = false, has_U = false;
  // mplx_set_goal: the goal the heur / flags rows of mplx_succ_lists refer to (PostFuse without its output pointers)
  bool has_goal = false;
  mplx::PostFuse goal_fuse{};
  int32_t mdim[3] = {1, 1, 1};
  double origin[3] = {0, 0, 0};
  double res = 0;
  int64_t n_cells = 0;
  uint64_t map_upload_bytes = 0;  // host -> device bytes of mplx_set_map / _set_potential / _set_region / _edit_map so far
  mplx_params prm{};
  int32_t nU = 0, udim = 0;
  double u_absmax = 0;  // max |u| over the spatial control entries
  // per-axis factorisation of the control table (expand_grid_kernel.hip)
  mplx_detail::DevBuf uvals, uidx, blk, sat;
  bool sat_ok = false;   // summed-area table of blk is current
  mplx_detail::DevBuf e_parents, e_action, e_free, e_cost, e_cells, e_count;  // edge re-validation staging
  mplx_detail::DevBuf post_keys;                 // node-identity table (post_api.cpp)
  mplx_detail::DevBuf post_ws;                   // workspace of the partitioned identity pass (identity_kernel.hip)
  mplx_detail::DevBuf prep_lut, prep_a, prep_b;  // map preprocessing scratch (map_prep_api.cpp)
  bool blk_ok = false;   // blocked-bit map matches the current map + region
  bool u_factored = false;
  bool u_wide = false;   // 17 .. 32 distinct values on some axis: only the lexicographic kernel (expand_lex_kernel.hip) takes the table
  bool u_lex = false;    // the table is the full Cartesian product of its per-axis values in lexicographic order
  int32_t u_nd[4] = {0, 0, 0, 0};  // distinct control values per axis; [3] = yaw rates
  // Diagnostic knobs, read from the environment once per context (mplx_create): ablations and forced code paths
  // for tests and profiling scripts; all off / automatic in production.
  struct Tuning {
    int grid_rmax = 0, grid_boxcap = 0, grid_blocks = 0;  // MPLX_GRID_RMAX / _BOXCAP / _BLOCKS (0 = automatic)
    int grid_waves_per_cu = 0;                            // MPLX_GRID_WAVES_PER_CU: occupancy cap (0 = automatic)
    int grid_gather = -1, grid_sat = -1;                  // MPLX_GRID_GATHER / MPLX_GRID_SAT: force 0 / 1 (-1 = automatic)
    bool grid_lex = true;                                 // MPLX_GRID_LEX=0: lexicographic tables run expand_grid_kernel like every other table
    bool grid_static = false;                             // MPLX_GRID_STATIC: static node striding instead of the work counters
    int grid_chunk = 0;                                   // MPLX_GRID_CHUNK: nodes per claim (0 = automatic)
    int grid_blocked = 0;                                 // MPLX_GRID_BLOCKED: contiguous shares per counter (A/B)
    int dbg = 0;                                         // MPLX_TILE_DBG ablation bits
    bool no_pair = false;                                // MPLX_GRID_PAIR=0: pre-screened yaw + potential launches stay with expand_grid_kernel
    int pair_rmax = 0, pair_wg_per_cu = 0;               // MPLX_PAIR_RMAX / MPLX_PAIR_WG_PER_CU: rows per pass / resident workgroups of expand_pair_kernel
    int zero_copy = 1;                                   // MPLX_ZERO_COPY=0: small batches through a device arena instead
    int arena_kb = 0;                                    // MPLX_ARENA_KB: largest batch served by the one-copy path
    bool no_sat = false, no_lex = false, no_line_pad = false;  // MPLX_GRID_NOSAT / MPLX_GRID_NOLEX / MPLX_NO_LINE_PAD
    int prescreen_min = 0;     // MPLX_GRID_PRESCREEN_MIN: smallest frontier that gets the lane-per-node pre-screen (0 = automatic, -1 = never)
    bool yaw_pin = true;       // MPLX_YAW_PIN=0: raw device trig decisions (to measure what the pinning is for)
    double yaw_margin = 0;     // MPLX_YAW_MARGIN: detection band (tests widen it to drive many nodes through the fix pass)
    int done_flag = 1;         // MPLX_DONE_FLAG=0: small synchronous launches end with hipStreamSynchronize (DoneSignal)
    int service = 1;           // MPLX_SERVICE=0: every small batch is its own launch (mplx_service)
    int service_idle_us = 500;    // MPLX_SERVICE_IDLE_US: the resident kernel leaves after this long without a request
    int service_max_nodes = 256;  // MPLX_SERVICE_MAX_NODES: larger batches are launches of their own
  } tune;
  int lists_route = MPLX_ROUTE_AUTO;
  int last_route = MPLX_ROUTE_AUTO;
  bool last_grid_pair = false;  // ... or to expand_pair_kernel.hip
  bool last_grid_lex = false;  // the last factorised launch went to expand_lex_kernel.hip (mplx_debug_last_kernel)
  int n_cus = 256;

  // tables of the tiled kernel (sample times, loop counts, reciprocals)
  mplx_detail::DevBuf tables;
  bool tables_ok = false;
  double tab_dt = 0, tab_res = 0;
  double recips[3] = {0, 0, 0};
  // scratch for the dense -> lists route
  mplx_detail::DevBuf d_status, d_cost, d_hash, d_state, d_iters;
  // staging for the host-pointer entry points
  mplx_detail::DevBuf s_nodes, s_status, s_cost, s_hash, s_state, s_iters, s_count, s_action;
  // small host-pointer batches: one device arena + one pinned mirror (mplx_expand_lists)
  mplx_detail::DevBuf s_arena;
  void *h_arena = nullptr;
  size_t h_arena_cap = 0;
  // Completion of small synchronous launches through a word the kernel writes itself (DoneSignal, mplx_internal.h)
  uint64_t *done_host = nullptr;     // pinned
  mplx_detail::DevBuf done_count;    // 4 bytes, zero between launches
  uint64_t done_seq = 0;
  bool want_done = false;            // set by a caller that will wait for the launch on the spot (around lists_device)
  bool done_armed = false;           // ... and the launch that went in carries the signal
  int64_t done_waits = 0, done_timeouts = 0;
  // Resident form of the tiled kernel for the small synchronous batches of a search ("service", mplx_api.cpp and
  // expand_tile_kernel.hip): requests go through a mailbox in pinned memory instead of launch + synchronise.
  struct Service {
    bool running = false;   // a resident kernel has been launched and has not been seen to leave
    bool disabled = false;  // it failed to answer once: never again in this context
    int streak = 0;         // eligible batches since the last other API call (the second one starts the service)
    hipStream_t stream = nullptr;
    mplx::SvcMailbox *mb = nullptr;  // pinned
    char *block = nullptr;           // pinned landing block: node rows, then the list rows, sized for `cap` nodes
    size_t block_cap = 0;
    mplx_detail::DevBuf dev;         // command word + one word per workgroup
    uint32_t seq = 0;
    int64_t cap = 0, S = 0;          // signature of the resident kernel: capacity in nodes, list stride,
    unsigned rows = 0;               // ... and which list rows it writes (bit 0 action, 1 cost, 2 hash, 3 iters, 4 state)
    int workgroups = 0;
    int64_t requests = 0, launches = 0, failures = 0;  // statistics (mplx_service)
  } svc;
  // pipelined copy back of the lists (lists_copy_api.cpp): packed chunks on the device, pinned landing buffers
  mplx_detail::DevBuf pk_dev[2], pk_offs;
  void *pk_pin[2] = {nullptr, nullptr};
  size_t pk_pin_cap = 0;
  hipEvent_t pk_ev[2] = {nullptr, nullptr};
  std::vector<int64_t> pk_hoffs;
  void *pk_hb = nullptr;  // pinned block of expand_lists_packed: nodes, counts, offsets
  size_t pk_hb_cap = 0;
  // yaw pinning (YawPin, mplx_internal.h): launches whose heading-limit decisions have not been checked against
  // the host libm yet; resolved by resolve_pending() at the next synchronising call
  struct YawPending {
    int kind = 0;  // 0: factorised lists kernel, 1: dense kernel
    mplx::GridArgs g{};
    mplx::ExpandArgs e{};
  };
  std::vector<YawPending> yaw_pending;
  int32_t *yaw_any_host = nullptr;  // pinned word: some launch since the last resolve flagged a node
  int32_t *id_ovf_host = nullptr;   // pinned word: a bucket of the claimed identity pass overflowed (post_api.cpp)
  mplx_detail::DevBuf edit_buf;     // cell indices + values of mplx_edit_map
  bool sat_stale = false;           // the blocked bits were patched by mplx_edit_map: the summed-area table waits for a launch worth rebuilding it for
  int id_backoff = 0;               // calls left that skip the claimed identity form after an overflow (post_api.cpp)
  int last_identity_form = 0;       // 0 none / table in HBM, 1 claimed, 2 exact partition, 3 claimed then exact (overflow)
  mplx_detail::DevBuf yaw_ring, yaw_ids, yaw_tab;  // flagged nodes per pending launch; node list + trig table of a fix pass
  std::vector<double> h_U;          // host copy of the control table (the fix pass needs the yaw rates)
  double h_uyaw[16] = {0};          // ... and of its distinct yaw rates, in the factorisation's order
  int64_t yaw_flagged = 0, yaw_fix_passes = 0;  // statistics (mplx_yaw_pin_stats)
  mplx_detail::DevBuf live_list;          // pre-screen of yaw controls: surviving nodes (int32 each)
  mplx_detail::DevBuf live_ctr;           // ... and their number: two counters used alternately (live_parity), zero between launches
  int live_parity = 0;
  mplx_detail::DevBuf work_counter;       // dynamic node assignment of the factorised kernel (GridArgs::work)
  int work_parity = 0;                    // which of the two counter sets the next launch uses
  // workgroups of the factorised kernel resident per CU (grid_resident_blocks), cached per (control, potential, LDS)
  struct GridOcc { int control; bool pot; size_t lds; int nb; };
  mutable std::vector<GridOcc> grid_occ;
  // RCCL communicator of this context (comm_api.cpp); the library is loaded on first use
  void *comm = nullptr;
  int comm_rank = 0, comm_world = 1;
  mplx_detail::DevBuf comm_meta;  // [world + 1][MPLX_COMM_META] int64: the meta record of every rank, then the own one
  std::vector<uint8_t> h_status;
  std::vector<double> h_cost, h_state;
};

namespace mplx_detail {

inline int fail(mplx_ctx *c, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf; else create_error() = buf;
  return code;
}

#define HIP_TRY(c, expr)                                                                   \
  do {                                                                                     \
    hipError_t e__ = (expr);                                                               \
    if (e__ != hipSuccess)                                                                 \
      return mplx_detail::fail((c), MPLX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                  __FILE__, __LINE__);                                                     \
  } while (0)

// No exception may cross the C ABI (include/mplx.h: "never throws"): entry points that touch std containers wrap
// their bodies in these.
#define MPLX_GUARD_BEGIN try {
#define MPLX_GUARD_END(c)                                                                            \
  }                                                                                                  \
  catch (const std::bad_alloc &) { return mplx_detail::fail((c), MPLX_ERR_NOMEM, "out of host memory"); } \
  catch (const std::exception &e) { return mplx_detail::fail((c), MPLX_ERR_NOMEM, "unexpected exception: %s", e.what()); } \
  catch (...) { return mplx_detail::fail((c), MPLX_ERR_NOMEM, "unexpected exception"); }

int svc_stop(mplx_ctx *c);  // mplx_api.cpp
int svc_request(mplx_ctx *c, const double *h_nodes, int64_t n_nodes, int64_t node_stride, const mplx_succ_lists *h_out,
                bool *handled, mplx_succ_lists *view);

// Every entry point that touches the device comes through here first.  A resident service kernel reads the
// context's tables through the arguments it was launched with and writes into the context's landing block: it is
// asked to leave before anything else happens (only the small-batch path of mplx_expand_lists talks to it instead).
inline int bind_device(mplx_ctx *c) {
  HIP_TRY(c, hipSetDevice(c->device));
  c->svc.streak = 0;
  if (c->svc.running) return svc_stop(c);
  return MPLX_OK;
}

inline int ensure(mplx_ctx *c, DevBuf &b, size_t bytes) {
  if (bytes <= b.cap) return MPLX_OK;
  if (b.p) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
  }
  HIP_TRY(c, hipMalloc(&b.p, bytes));
  b.cap = bytes;
  return MPLX_OK;
}

inline void release(DevBuf &b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
}

// The per-node lists of a host frontier, left packed in the context's pinned landing buffer (node k owns entries
// [offs[k], offs[k+1]) of every row): for callers inside the library that consume the lists at once (the host search,
// planner_capi.cpp) and would only copy them again.  Valid until the next call on the context.
struct PackedLists {
  int64_t total = 0;
  const int32_t *count = nullptr;   // [n_nodes]
  const int64_t *offs = nullptr;    // [n_nodes + 1]
  const double *cost = nullptr;     // [total]
  const uint64_t *hash = nullptr;   // [total]
  const int32_t *action = nullptr;  // [total]
  const double *state = nullptr;    // [4D+2][total]
  const double *heur = nullptr;     // [total] default heuristic of every successor (want_heur; needs mplx_set_goal)
};
int expand_lists_packed(mplx_ctx *c, const double *h_nodes, int64_t n_nodes, int64_t node_stride, bool want_state,
                        PackedLists *out, bool want_heur = false);  // want_state = false: out->state == nullptr, the kernel skips the states
// mplx_api.cpp: readiness check and the route dispatch behind mplx_expand_lists*
int ctx_ready(mplx_ctx *c);
int lists_on_device(mplx_ctx *c, const double *d_nodes, int64_t n_nodes, int64_t node_stride, const mplx_succ_lists *d);

// mplx_api.cpp: synchronises the stream and, where a launch flagged heading-limit decisions within rounding noise of
// their threshold, re-expands those nodes with the host libm's trig values (YawPin, mplx_internal.h).  Every
// synchronising entry point and every call that changes what a pending launch read goes through it.
int resolve_pending(mplx_ctx *c, bool stream_is_idle = false);
// after a small-batch launch through lists_on_device with want_done set: waits for it (kernel-written word, or the
// stream); *idle = everything enqueued on the context's stream so far is complete
int wait_small_launch(mplx_ctx *c);

// lists_copy_api.cpp: device lists -> host lists, only the used prefixes, pipelined through pinned memory
int copy_lists_to_host(mplx_ctx *c, const mplx_succ_lists &d, const mplx_succ_lists *h_out, int64_t n_nodes);
void release_copy_buffers(mplx_ctx *c);

}  // namespace mplx_detail
#endif
