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

#include <cstring>
#include <mutex>

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
    for (const char *name : {"librccl.so", "librccl.so.1"}) {
      r.h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
      if (r.h) break;
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

int mplx_comm_allgather_lists(mplx_ctx *c, const mplx_packed_lists *loc, int64_t n_local, const mplx_packed_lists *all,
                              int64_t *h_node_offs, int64_t *h_entry_offs) {
  if (!c) return MPLX_ERR_ARG;
  if (!c->comm) return fail(c, MPLX_ERR_STATE, "mplx_comm_allgather_lists: mplx_comm_init first");
  if (!loc || !all || n_local < 0 || !loc->offs || !loc->count || !all->offs || !all->count)
    return fail(c, MPLX_ERR_ARG, "mplx_comm_allgather_lists: both sides need count and offs");
  if ((all->action && !loc->action) || (all->cost && !loc->cost) || (all->hash && !loc->hash) || (all->state && !loc->state))
    return fail(c, MPLX_ERR_ARG, "mplx_comm_allgather_lists: a gathered row is requested that the local side lacks");
  MPLX_GUARD_BEGIN
  if (int rc = bind_device(c)) return rc;
  if (int rc = resolve_pending(c)) return rc;
  ncclComm_t comm = (ncclComm_t)c->comm;
  const int G = c->comm_world, me = c->comm_rank, F = 4 * c->dim + 2;
  // ---- (n_local, total) of every rank
  if (int rc = ensure(c, c->comm_meta, (size_t)(G + 1) * 16)) return rc;
  int64_t *d_meta = (int64_t *)c->comm_meta.p;  // [G][2], then one scratch pair
  int64_t *d_mine = d_meta + 2 * G;
  HIP_TRY(c, hipMemcpyAsync(d_mine, &n_local, 8, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(d_mine + 1, loc->offs + n_local, 8, hipMemcpyDeviceToDevice, c->stream));
  NCCL_TRY(c, rccl().AllGather(d_mine, d_meta, 2, ncclInt64, comm, c->stream));
  std::vector<int64_t> meta((size_t)2 * G);
  HIP_TRY(c, hipMemcpyAsync(meta.data(), d_meta, (size_t)G * 16, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  std::vector<int64_t> noff((size_t)G + 1, 0), eoff((size_t)G + 1, 0);
  for (int r = 0; r < G; r++) {
    noff[(size_t)r + 1] = noff[(size_t)r] + meta[(size_t)2 * r];
    eoff[(size_t)r + 1] = eoff[(size_t)r] + meta[(size_t)2 * r + 1];
  }
  if (meta[(size_t)2 * me] != n_local) return fail(c, MPLX_ERR_STATE, "mplx_comm_allgather_lists: rank order mismatch");
  if (eoff[(size_t)G] > all->capacity)
    return fail(c, MPLX_ERR_ARG, "mplx_comm_allgather_lists: %lld gathered entries exceed the capacity %lld",
                (long long)eoff[(size_t)G], (long long)all->capacity);
  if (all->state && all->state_stride < all->capacity)
    return fail(c, MPLX_ERR_ARG, "mplx_comm_allgather_lists: state_stride < capacity");
  if (h_node_offs) std::memcpy(h_node_offs, noff.data(), (size_t)(G + 1) * 8);
  if (h_entry_offs) std::memcpy(h_entry_offs, eoff.data(), (size_t)(G + 1) * 8);
  // ---- all-pairs exchange: my rows to every peer, every peer's rows into their final place, one group
  struct RowPair { const void *src; char *dst; int es_n; /* 0: per-entry row, 1: the per-node count row */ int es; };
  std::vector<RowPair> rows;
  rows.push_back({loc->count, (char *)all->count, 1, 4});
  if (all->action) rows.push_back({loc->action, (char *)all->action, 0, 4});
  if (all->cost) rows.push_back({loc->cost, (char *)all->cost, 0, 8});
  if (all->hash) rows.push_back({loc->hash, (char *)all->hash, 0, 8});
  if (all->state)
    for (int f = 0; f < F; f++)
      rows.push_back({loc->state + (size_t)f * loc->state_stride, (char *)(all->state + (size_t)f * all->state_stride), 0, 8});
  const size_t my_n = (size_t)meta[(size_t)2 * me], my_e = (size_t)meta[(size_t)2 * me + 1];
  for (const RowPair &rw : rows) {  // my own block: a device copy
    const size_t bytes = (rw.es_n ? my_n : my_e) * (size_t)rw.es;
    const size_t at = (size_t)(rw.es_n ? noff[(size_t)me] : eoff[(size_t)me]) * (size_t)rw.es;
    if (bytes) HIP_TRY(c, hipMemcpyAsync(rw.dst + at, rw.src, bytes, hipMemcpyDeviceToDevice, c->stream));
  }
  if (G > 1) {
    NCCL_TRY(c, rccl().GroupStart());
    int e = 0;  // (a call that fails inside the group must not leave it open)
    for (int d = 1; d < G && e == 0; d++) {
      const int to = (me + d) % G, from = (me - d + G) % G;  // a different peer pair per step on every rank
      const size_t fn = (size_t)meta[(size_t)2 * from], fe = (size_t)meta[(size_t)2 * from + 1];
      for (const RowPair &rw : rows) {
        const size_t sb = (rw.es_n ? my_n : my_e) * (size_t)rw.es;
        const size_t rb = (rw.es_n ? fn : fe) * (size_t)rw.es;
        const size_t at = (size_t)(rw.es_n ? noff[(size_t)from] : eoff[(size_t)from]) * (size_t)rw.es;
        if (e == 0 && sb) e = rccl().Send(rw.src, sb, ncclInt8, to, comm, c->stream);
        if (e == 0 && rb) e = rccl().Recv(rw.dst + at, rb, ncclInt8, from, comm, c->stream);
      }
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
