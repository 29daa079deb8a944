// comm_api.cpp -- the multi-GPU exchange of the C ABI (include/mplx.h, "packed lists and the multi-GPU
// exchange"): RCCL over xGMI, one communicator per context.
//
// The reference has no multi-device code at all; what is restated here is the ownership rule that
// makes sharding legal: env_map<Dim>::get_succ (include/mpl_planner/env/env_map.h:147-172) is a pure function of
// (node, U, map), so rank r expands its block of the frontier alone and the ONLY exchange is the optional
// all-gather of the successor lists for a consumer that needs the whole set.
//
// MI355X: xGMI is point to point (7 links per GPU, one per peer), so the gather is scheduled as direct all-pairs
// copies -- inside ONE ncclGroup every rank ncclSend()s its packed rows to each peer and ncclRecv()s that peer's
// rows straight into their final place, exact sizes; the 7 transfers of a rank run concurrently, each on its own
// link -- instead of a ring (ncclAllGather / ncclBroadcast: per-link bound, 7 sequential hops) and without
// padding (C4's lists are 57 % padding).
//
// librccl.so.1 (573 MB) is loaded with dlopen on first use: a single-GPU user never pays for it and libmplx.so
// has no link-time dependency on it.
#include "mplx_ctx.h"

#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

using namespace mplx_detail;

namespace {

// the slice of rccl.h this file needs (ABI-stable NCCL 2 surface)
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclInt8 = 0, ncclInt32 = 2, ncclInt64 = 4 };

struct Rccl {
  void *h = nullptr;
  int (*GetUniqueId)(ncclUniqueId *) = nullptr;
  int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Broadcast)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  std::string err;
};

Rccl &rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // A process that already has an RCCL (PyTorch-ROCm ships its own librccl.so) must keep using that one: two
    // RCCL builds in one process corrupt each other's state.  So first look for a loaded copy, then load ROCm's.
    // MPLX_RCCL_LIB: a library with the NCCL C API to use instead (a site build of RCCL; in the tests a shared-memory
    // stand-in that lets several processes on ONE GPU form a communicator, tests/fake_rccl/) -- takes precedence.
    if (const char *over = getenv("MPLX_RCCL_LIB")) {
      r.h = dlopen(over, RTLD_NOW | RTLD_LOCAL);
      if (!r.h) { r.err = std::string("cannot load MPLX_RCCL_LIB: ") + dlerror(); return; }
    }
    for (const char *name : {"librccl.so", "librccl.so.1"}) {
      if (r.h) break;
      r.h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
    }
    if (!r.h)
      for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (r.h) break;
      }
    if (!r.h) { r.err = std::string("cannot load librccl.so.1: ") + dlerror(); return; }
    auto sym = [&](const char *n) {
      void *p = dlsym(r.h, n);
      if (!p && r.err.empty()) r.err = std::string("librccl.so.1 lacks ") + n;
      return p;
    };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.Broadcast = (decltype(r.Broadcast))sym("ncclBroadcast");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
  });
  return r;
}

#define NCCL_TRY(c, expr)                                                                               \
  do {                                                                                                  \
    const int e__ = (expr);                                                                             \
    if (e__ != 0)                                                                                       \
      return fail((c), MPLX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, rccl().GetErrorString(e__), __FILE__, __LINE__); \
  } while (0)

int need_rccl(mplx_ctx *c) {
  Rccl &r = rccl();
  if (!r.err.empty()) return fail(c, MPLX_ERR_STATE, "%s", r.err.c_str());
  return MPLX_OK;
}

}  // namespace

extern "C" {

int mplx_comm_unique_id(uint8_t *id_out) {
  if (!id_out) return MPLX_ERR_ARG;
  if (int rc = need_rccl(nullptr)) return rc;
  ncclUniqueId id;
  const int e = rccl().GetUniqueId(&id);
  if (e != 0) return fail(nullptr, MPLX_ERR_HIP, "ncclGetUniqueId failed: %s", rccl().GetErrorString(e));
  static_assert(sizeof id == MPLX_COMM_ID_BYTES, "ncclUniqueId size");
  std::memcpy(id_out, &id, sizeof id);
  return MPLX_OK;
}

int mplx_comm_init(mplx_ctx *c, const uint8_t *id, int32_t rank, int32_t world) {
  if (!c) return MPLX_ERR_ARG;
  if (!id || world < 1 || rank < 0 || rank >= world) return fail(c, MPLX_ERR_ARG, "mplx_comm_init: bad arguments");
  if (c->comm) return fail(c, MPLX_ERR_STATE, "mplx_comm_init: the context already has a communicator");
  if (int rc = need_rccl(c)) return rc;
  if (int rc = bind_device(c)) return rc;
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof uid);
  ncclComm_t comm = nullptr;
  NCCL_TRY(c, rccl().CommInitRank(&comm, world, uid, rank));
  c->comm = comm;
  c->comm_rank = rank;
  c->comm_world = world;
  return MPLX_OK;
}

int mplx_comm_destroy(mplx_ctx *c) {
  if (!c) return MPLX_ERR_ARG;
  if (!c->comm) return MPLX_OK;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  const int e = rccl().CommDestroy((ncclComm_t)c->comm);
  c->comm = nullptr;
  c->comm_world = 1;
  c->comm_rank = 0;
  if (e != 0) return fail(c, MPLX_ERR_HIP, "ncclCommDestroy failed: %s", rccl().GetErrorString(e));
  return MPLX_OK;
}

int mplx_comm_broadcast_map(mplx_ctx *c, int32_t root) {
  if (!c) return MPLX_ERR_ARG;
  if (!c->comm) return fail(c, MPLX_ERR_STATE, "mplx_comm_broadcast_map: mplx_comm_init first");
  if (!c->has_map) return fail(c, MPLX_ERR_STATE, "mplx_comm_broadcast_map: every rank sets the map geometry first");
  if (root < 0 || root >= c->comm_world) return fail(c, MPLX_ERR_ARG, "mplx_comm_broadcast_map: bad root");
  if (int rc = bind_device(c)) return rc;
  ncclComm_t comm = (ncclComm_t)c->comm;
  // which optional layers the root has: one tiny broadcast so that every rank issues the same collectives
  if (int rc = ensure(c, c->comm_meta, (size_t)(c->comm_world > 1 ? c->comm_world : 1) * 16)) return rc;
  int64_t flags[2] = {c->has_pot ? 1 : 0, c->has_region ? 1 : 0};
  HIP_TRY(c, hipMemcpyAsync(c->comm_meta.p, flags, 16, hipMemcpyHostToDevice, c->stream));
  NCCL_TRY(c, rccl().Broadcast(c->comm_meta.p, c->comm_meta.p, 2, ncclInt64, root, comm, c->stream));
  HIP_TRY(c, hipMemcpyAsync(flags, c->comm_meta.p, 16, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  const size_t n = (size_t)c->n_cells, words = (size_t)((c->n_cells + 31) >> 5);
  if (flags[0]) if (int rc = ensure(c, c->pot, n)) return rc;
  if (flags[1]) if (int rc = ensure(c, c->region_bits, words * 4)) return rc;
  NCCL_TRY(c, rccl().GroupStart());
  {  // a call that fails inside the group must not leave it open
    int e = rccl().Broadcast(c->map.p, c->map.p, n, ncclInt8, root, comm, c->stream);
    if (e == 0 && flags[0]) e = rccl().Broadcast(c->pot.p, c->pot.p, n, ncclInt8, root, comm, c->stream);
    if (e == 0 && flags[1]) e = rccl().Broadcast(c->region_bits.p, c->region_bits.p, words, ncclInt32, root, comm, c->stream);
    const int e_end = rccl().GroupEnd();
    NCCL_TRY(c, e);
    NCCL_TRY(c, e_end);
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->has_pot = flags[0] != 0;
  c->has_region = flags[1] != 0;
  c->blk_ok = false;
  return MPLX_OK;
}

int64_t mplx_comm_schedule(int32_t world, int32_t rank, const int64_t *meta, int32_t n_fields, mplx_comm_op *ops,
                           int64_t cap, int64_t *node_offs, int64_t *entry_offs) {
  if (world < 1 || rank < 0 || rank >= world || !meta || n_fields < 0 || cap < 0) return MPLX_ERR_ARG;
  const int G = world, me = rank;
  auto M = [&](int r, int w) { return meta[(size_t)r * MPLX_COMM_META + w]; };
  // ---- the verdict: a function of `meta` alone, hence the same on every rank
  int64_t esum = 0;
  for (int r = 0; r < G; r++) {
    if (M(r, 4) != MPLX_OK) return MPLX_ERR_STATE;                   // some rank failed its own argument checks
    if (M(r, 0) < 0 || M(r, 1) < 0) return MPLX_ERR_ARG;
    if (M(r, 2) != M(0, 2)) return MPLX_ERR_ARG;                     // the ranks disagree on the rows to gather
    esum += M(r, 1);
  }
  for (int r = 0; r < G; r++)
    if (esum > M(r, 3)) return MPLX_ERR_ARG;                         // some rank's gathered side is too small
  if (node_offs || entry_offs) {
    int64_t n = 0, e = 0;
    for (int r = 0; r <= G; r++) {
      if (node_offs) node_offs[r] = n;
      if (entry_offs) entry_offs[r] = e;
      if (r < G) { n += M(r, 0); e += M(r, 1); }
    }
  }
  // ---- the rows, in the order they are issued
  struct Row { int32_t row, elem; bool per_node; };
  Row rows[4 + 64];
  int nrows = 0;
  const int64_t mask = M(0, 2);
  rows[nrows++] = {MPLX_ROW_COUNT, 4, true};
  if (mask & MPLX_ROWBIT_ACTION) rows[nrows++] = {MPLX_ROW_ACTION, 4, false};
  if (mask & MPLX_ROWBIT_COST) rows[nrows++] = {MPLX_ROW_COST, 8, false};
  if (mask & MPLX_ROWBIT_HASH) rows[nrows++] = {MPLX_ROW_HASH, 8, false};
  if (mask & MPLX_ROWBIT_STATE)
    for (int f = 0; f < n_fields && f < 64; f++) rows[nrows++] = {MPLX_ROW_STATE0 + f, 8, false};
  std::vector<int64_t> noff((size_t)G + 1, 0), eoff((size_t)G + 1, 0);
  for (int r = 0; r < G; r++) {
    noff[(size_t)r + 1] = noff[(size_t)r] + M(r, 0);
    eoff[(size_t)r + 1] = eoff[(size_t)r] + M(r, 1);
  }
  int64_t n_ops = 0;
  auto put = [&](int32_t kind, int32_t peer, const Row &rw, int64_t src, int64_t dst, int64_t bytes) {
    if (ops && n_ops < cap) ops[n_ops] = mplx_comm_op{kind, peer, rw.row, rw.elem, src, dst, bytes};
    n_ops++;
  };
  auto units = [&](int r, const Row &rw) { return rw.per_node ? M(r, 0) : M(r, 1); };
  auto start = [&](int r, const Row &rw) { return rw.per_node ? noff[(size_t)r] : eoff[(size_t)r]; };
  for (int i = 0; i < nrows; i++)  // the rank's own block: a local copy into its place
    if (units(me, rows[i])) put(MPLX_COMM_COPY, me, rows[i], 0, start(me, rows[i]) * rows[i].elem, units(me, rows[i]) * rows[i].elem);
  for (int d = 1; d < G; d++) {
    const int to = (me + d) % G, from = (me - d + G) % G;
    for (int i = 0; i < nrows; i++) {
      const Row &rw = rows[i];
      if (units(me, rw)) put(MPLX_COMM_SEND, to, rw, 0, 0, units(me, rw) * rw.elem);
      if (units(from, rw)) put(MPLX_COMM_RECV, from, rw, 0, start(from, rw) * rw.elem, units(from, rw) * rw.elem);
    }
  }
  return n_ops;
}

int mplx_comm_allgather_lists(mplx_ctx *c, const mplx_packed_lists *loc, int64_t n_local, const mplx_packed_lists *all,
                              int64_t *h_node_offs, int64_t *h_entry_offs) {
  if (!c) return MPLX_ERR_ARG;
  if (!c->comm) return fail(c, MPLX_ERR_STATE, "mplx_comm_allgather_lists: mplx_comm_init first");
  MPLX_GUARD_BEGIN
  if (int rc = bind_device(c)) return rc;
  if (int rc = resolve_pending(c)) return rc;
  ncclComm_t comm = (ncclComm_t)c->comm;
  const int G = c->comm_world, me = c->comm_rank, F = 4 * c->dim + 2;
  // ---- this rank's own argument checks do NOT return yet: a rank that left here alone would leave its peers waiting
  // in the collectives below.  Their verdict travels in the meta all-gather and every rank bails out together.
  int my_status = MPLX_OK;
  const char *my_msg = "";
  if (!loc || !all || n_local < 0 || !loc->offs || !loc->count || !all->offs || !all->count) {
    my_status = MPLX_ERR_ARG;
    my_msg = "both sides need count and offs";
  } else if ((all->action && !loc->action) || (all->cost && !loc->cost) || (all->hash && !loc->hash) || (all->state && !loc->state)) {
    my_status = MPLX_ERR_ARG;
    my_msg = "a gathered row is requested that the local side lacks";
  } else if (all->state && all->state_stride < all->capacity) {
    my_status = MPLX_ERR_ARG;
    my_msg = "state_stride < capacity";
  }
  const bool ok = my_status == MPLX_OK;
  // ---- meta of every rank: (n_local, entries, row mask, capacity, status)
  if (int rc = ensure(c, c->comm_meta, (size_t)(G + 1) * MPLX_COMM_META * 8)) return rc;
  int64_t *d_meta = (int64_t *)c->comm_meta.p;  // [G][META], then this rank's own record
  int64_t *d_mine = d_meta + (size_t)MPLX_COMM_META * G;
  int64_t mine[MPLX_COMM_META] = {0};
  mine[0] = ok ? n_local : 0;
  mine[2] = ok ? (all->action ? MPLX_ROWBIT_ACTION : 0) | (all->cost ? MPLX_ROWBIT_COST : 0) | (all->hash ? MPLX_ROWBIT_HASH : 0) |
                     (all->state ? MPLX_ROWBIT_STATE : 0) : 0;
  mine[3] = ok ? all->capacity : 0;
  mine[4] = my_status;
  HIP_TRY(c, hipMemcpyAsync(d_mine, mine, sizeof mine, hipMemcpyHostToDevice, c->stream));
  if (ok) HIP_TRY(c, hipMemcpyAsync(d_mine + 1, loc->offs + n_local, 8, hipMemcpyDeviceToDevice, c->stream));
  NCCL_TRY(c, rccl().AllGather(d_mine, d_meta, MPLX_COMM_META, ncclInt64, comm, c->stream));
  std::vector<int64_t> meta((size_t)MPLX_COMM_META * G);
  HIP_TRY(c, hipMemcpyAsync(meta.data(), d_meta, meta.size() * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (!ok) return fail(c, my_status, "mplx_comm_allgather_lists: %s", my_msg);
  std::vector<int64_t> noff((size_t)G + 1, 0), eoff((size_t)G + 1, 0);
  const int64_t n_ops = mplx_comm_schedule(G, me, meta.data(), F, nullptr, 0, noff.data(), eoff.data());
  if (n_ops < 0) {  // the same verdict on every rank (a function of the gathered meta alone)
    for (int r = 0; r < G; r++)
      if (meta[(size_t)r * MPLX_COMM_META + 4] != MPLX_OK)
        return fail(c, (int)n_ops, "mplx_comm_allgather_lists: rank %d failed its argument checks (code %lld)", r,
                    (long long)meta[(size_t)r * MPLX_COMM_META + 4]);
    for (int r = 0; r < G; r++)
      if (meta[(size_t)r * MPLX_COMM_META + 2] != meta[2])
        return fail(c, (int)n_ops, "mplx_comm_allgather_lists: rank %d gathers rows 0x%llx, rank 0 rows 0x%llx", r,
                    (unsigned long long)meta[(size_t)r * MPLX_COMM_META + 2], (unsigned long long)meta[2]);
    int64_t esum = 0;
    for (int r = 0; r < G; r++) esum += meta[(size_t)r * MPLX_COMM_META + 1];
    for (int r = 0; r < G; r++)
      if (esum > meta[(size_t)r * MPLX_COMM_META + 3])
        return fail(c, (int)n_ops, "mplx_comm_allgather_lists: %lld gathered entries exceed the capacity %lld of rank %d",
                    (long long)esum, (long long)meta[(size_t)r * MPLX_COMM_META + 3], r);
    return fail(c, (int)n_ops, "mplx_comm_allgather_lists: inconsistent sizes across the ranks");
  }
  if (meta[(size_t)me * MPLX_COMM_META] != n_local) return fail(c, MPLX_ERR_STATE, "mplx_comm_allgather_lists: rank order mismatch");
  if (h_node_offs) std::memcpy(h_node_offs, noff.data(), (size_t)(G + 1) * 8);
  if (h_entry_offs) std::memcpy(h_entry_offs, eoff.data(), (size_t)(G + 1) * 8);
  std::vector<mplx_comm_op> ops((size_t)n_ops);
  (void)mplx_comm_schedule(G, me, meta.data(), F, ops.data(), n_ops, nullptr, nullptr);
  // ---- execute: local copies, then every send / receive of the all-pairs exchange inside ONE group
  auto src_of = [&](const mplx_comm_op &o) -> const char * {
    switch (o.row) {
      case MPLX_ROW_COUNT: return (const char *)loc->count;
      case MPLX_ROW_ACTION: return (const char *)loc->action;
      case MPLX_ROW_COST: return (const char *)loc->cost;
      case MPLX_ROW_HASH: return (const char *)loc->hash;
      default: return (const char *)(loc->state + (size_t)(o.row - MPLX_ROW_STATE0) * loc->state_stride);
    }
  };
  auto dst_of = [&](const mplx_comm_op &o) -> char * {
    switch (o.row) {
      case MPLX_ROW_COUNT: return (char *)all->count;
      case MPLX_ROW_ACTION: return (char *)all->action;
      case MPLX_ROW_COST: return (char *)all->cost;
      case MPLX_ROW_HASH: return (char *)all->hash;
      default: return (char *)(all->state + (size_t)(o.row - MPLX_ROW_STATE0) * all->state_stride);
    }
  };
  size_t k = 0;
  for (; k < ops.size() && ops[k].kind == MPLX_COMM_COPY; k++)
    HIP_TRY(c, hipMemcpyAsync(dst_of(ops[k]) + ops[k].dst_off, src_of(ops[k]) + ops[k].src_off, (size_t)ops[k].bytes,
                              hipMemcpyDeviceToDevice, c->stream));
  if (k < ops.size()) {
    NCCL_TRY(c, rccl().GroupStart());
    int e = 0;  // (a call that fails inside the group must not leave it open)
    for (; k < ops.size() && e == 0; k++) {
      const mplx_comm_op &o = ops[k];
      if (o.kind == MPLX_COMM_SEND) e = rccl().Send(src_of(o) + o.src_off, (size_t)o.bytes, ncclInt8, o.peer, comm, c->stream);
      else e = rccl().Recv(dst_of(o) + o.dst_off, (size_t)o.bytes, ncclInt8, o.peer, comm, c->stream);
    }
    const int e_end = rccl().GroupEnd();
    NCCL_TRY(c, e);
    NCCL_TRY(c, e_end);
  }
  HIP_TRY(c, mplx::launch_scan_counts(all->count, noff[(size_t)G], all->offs, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return MPLX_OK;
  MPLX_GUARD_END(c)
}

}  // extern "C"
