// fake_rccl.cpp -- TEST INFRASTRUCTURE, not product: the slice of the NCCL / RCCL C API that
// motion_primitive_library_amd/csrc/comm_api.cpp binds (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy,
// ncclAllGather, ncclBroadcast, ncclSend, ncclRecv, ncclGroupStart, ncclGroupEnd, ncclGetErrorString), implemented
// over a file-backed shared mapping (/tmp) so that SEVERAL PROCESSES ON ONE GPU can form a "communicator".  RCCL itself
// refuses two ranks on one device, and the boxes this repository is tested on have one GPU: with this library loaded
// through MPLX_RCCL_LIB the G > 1 branch of mplx_comm_allgather_lists / mplx_comm_broadcast_map runs for real --
// the meta all-gather, the collective verdict, the local copies and every ncclSend / ncclRecv of the all-pairs group,
// with the device pointers and byte offsets the C function computes (tests/test_gpu_comm.py).
//
// Semantics kept from NCCL: calls are enqueued "on a stream" (here: the stream is synchronised and the transfer done
// on the spot), the k-th send of a -> b pairs with the k-th receive of b from a, everything between GroupStart and
// GroupEnd is issued together (all sends first, then all receives, so no ordering between ranks can deadlock).
// Every data movement goes device -> shared host memory -> device with hipMemcpy.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <vector>

extern "C" {
typedef struct { char internal[128]; } ncclUniqueId;
struct FakeComm;
typedef FakeComm *ncclComm_t;
}

namespace {

constexpr int kMaxRanks = 4;
constexpr size_t kBoxBytes = (size_t)8 << 20;  // per ordered pair of ranks: room for every message of one group
constexpr size_t kGatherBytes = (size_t)1 << 20;

struct Shared {
  std::atomic<uint32_t> arrived;
  std::atomic<uint32_t> sense;
  std::atomic<uint32_t> attached;
  uint32_t pad[13];
  // then: gather area [kMaxRanks][kGatherBytes], mailboxes [kMaxRanks][kMaxRanks][kBoxBytes]
};

size_t shm_size() { return sizeof(Shared) + kMaxRanks * kGatherBytes + (size_t)kMaxRanks * kMaxRanks * kBoxBytes; }

struct Op { bool send; void *ptr; size_t bytes; int peer; hipStream_t stream; };

}  // namespace

struct FakeComm {
  int rank = 0, world = 1;
  Shared *sh = nullptr;
  char name[64] = {0};
  uint32_t my_sense = 0;
  std::vector<uint8_t> host;  // staging for device <-> shared memory
  char *gather(int r) { return (char *)(sh + 1) + (size_t)r * kGatherBytes; }
  char *box(int from, int to) { return (char *)(sh + 1) + kMaxRanks * kGatherBytes + ((size_t)from * kMaxRanks + to) * kBoxBytes; }
  void barrier() {
    my_sense ^= 1u;
    if (sh->arrived.fetch_add(1) + 1 == (uint32_t)world) {
      sh->arrived.store(0);
      sh->sense.store(my_sense);
    } else {
      while (sh->sense.load() != my_sense) usleep(50);
    }
  }
};

namespace {
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
thread_local FakeComm *g_comm = nullptr;

int d2h(void *dst, const void *src, size_t n, hipStream_t s) {
  if (hipStreamSynchronize(s) != hipSuccess) return 1;
  return hipMemcpy(dst, src, n, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
int h2d(void *dst, const void *src, size_t n) { return hipMemcpy(dst, src, n, hipMemcpyHostToDevice) == hipSuccess ? 0 : 1; }

int run_group(FakeComm *c, std::vector<Op> &ops) {
  // all sends of this rank: appended to the (me -> peer) mailbox in issue order, each as [bytes][payload]
  size_t used[kMaxRanks] = {0};
  for (const Op &o : ops)
    if (o.send) {
      char *b = c->box(c->rank, o.peer) + used[o.peer];
      if (used[o.peer] + 16 + o.bytes > kBoxBytes) { fprintf(stderr, "fake_rccl: mailbox overflow\n"); return 1; }
      uint64_t n = o.bytes;
      std::memcpy(b, &n, 8);
      if (d2h(b + 16, o.ptr, o.bytes, o.stream)) return 1;
      used[o.peer] += 16 + ((o.bytes + 15) & ~(size_t)15);
    }
  c->barrier();  // every rank's sends are in the mailboxes
  size_t got[kMaxRanks] = {0};
  for (const Op &o : ops)
    if (!o.send) {
      const char *b = c->box(o.peer, c->rank) + got[o.peer];
      uint64_t n = 0;
      std::memcpy(&n, b, 8);
      if (n != o.bytes) { fprintf(stderr, "fake_rccl: rank %d expects %zu bytes from %d, the matching send has %llu\n", c->rank, o.bytes, o.peer, (unsigned long long)n); return 1; }
      if (hipStreamSynchronize(o.stream) != hipSuccess || h2d(o.ptr, b + 16, o.bytes)) return 1;
      got[o.peer] += 16 + ((o.bytes + 15) & ~(size_t)15);
    }
  c->barrier();  // mailboxes may be reused
  return 0;
}
}  // namespace

extern "C" {

int ncclGetUniqueId(ncclUniqueId *id) {
  std::memset(id, 0, sizeof *id);
  snprintf(id->internal, sizeof id->internal, "/tmp/mplx_fake_rccl_%d_%ld", (int)getpid(), (long)time(nullptr));
  return 0;
}

int ncclCommInitRank(ncclComm_t *out, int world, ncclUniqueId id, int rank) {
  if (world < 1 || world > kMaxRanks || rank < 0 || rank >= world) return 4;
  FakeComm *c = new FakeComm();
  c->rank = rank;
  c->world = world;
  std::memcpy(c->name, id.internal, sizeof c->name - 1);
  int fd = -1;
  if (rank == 0) {
    char tmp[80];
    snprintf(tmp, sizeof tmp, "%s.init", c->name);
    fd = open(tmp, O_CREAT | O_TRUNC | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)shm_size()) != 0) return 2;  // sparse, zero-filled
    if (rename(tmp, c->name) != 0) return 2;                        // visible to the others only at full size
  } else {
    for (int tries = 0; tries < 40000 && fd < 0; tries++) {
      fd = open(c->name, O_RDWR);
      if (fd < 0) usleep(500);
    }
    if (fd < 0) return 2;
  }
  void *p = mmap(nullptr, shm_size(), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return 2;
  c->sh = (Shared *)p;  // (a fresh segment is zero-filled: counters start at 0)
  c->sh->attached.fetch_add(1);
  while (c->sh->attached.load() < (uint32_t)world) usleep(200);
  c->barrier();
  if (rank == 0) unlink(c->name);  // everybody has it mapped
  *out = c;
  return 0;
}

int ncclCommDestroy(ncclComm_t c) {
  if (!c) return 0;
  munmap(c->sh, shm_size());
  delete c;
  return 0;
}

static size_t dtype_bytes(int t) { return t == 0 ? 1 : (t == 2 ? 4 : 8); }  // ncclInt8, ncclInt32, ncclInt64 (all comm_api.cpp uses)

int ncclAllGather(const void *send, void *recv, size_t count, int dtype, ncclComm_t c, hipStream_t s) {
  const size_t n = count * dtype_bytes(dtype);
  if (n > kGatherBytes) return 1;
  if (d2h(c->gather(c->rank), send, n, s)) return 1;
  c->barrier();
  for (int r = 0; r < c->world; r++)
    if (h2d((char *)recv + (size_t)r * n, c->gather(r), n)) return 1;
  c->barrier();
  return 0;
}

int ncclBroadcast(const void *send, void *recv, size_t count, int dtype, int root, ncclComm_t c, hipStream_t s) {
  // inside a group (mplx_comm_broadcast_map issues up to three): executed on the spot, every rank issues them in the
  // same order.  Large buffers go through the (root -> rank) mailboxes in pieces.
  const size_t n = count * dtype_bytes(dtype);
  for (size_t off = 0; off < n; off += kBoxBytes) {
    const size_t m = n - off < kBoxBytes ? n - off : kBoxBytes;
    if (c->rank == root) {
      if (d2h(c->box(root, root), (const char *)send + off, m, s)) return 1;
    }
    c->barrier();
    if (c->rank != root) {
      if (hipStreamSynchronize(s) != hipSuccess || h2d((char *)recv + off, c->box(root, root), m)) return 1;
    } else if (recv != send) {
      if (hipMemcpy((char *)recv + off, (const char *)send + off, m, hipMemcpyDeviceToDevice) != hipSuccess) return 1;
    }
    c->barrier();
  }
  return 0;
}

int ncclGroupStart() {
  g_depth++;
  return 0;
}

int ncclSend(const void *p, size_t count, int dtype, int peer, ncclComm_t c, hipStream_t s) {
  g_comm = c;
  g_ops.push_back(Op{true, (void *)p, count * dtype_bytes(dtype), peer, s});
  if (g_depth == 0) { std::vector<Op> one; one.swap(g_ops); return run_group(c, one); }
  return 0;
}

int ncclRecv(void *p, size_t count, int dtype, int peer, ncclComm_t c, hipStream_t s) {
  g_comm = c;
  g_ops.push_back(Op{false, p, count * dtype_bytes(dtype), peer, s});
  if (g_depth == 0) { std::vector<Op> one; one.swap(g_ops); return run_group(c, one); }
  return 0;
}

int ncclGroupEnd() {
  if (g_depth > 0) g_depth--;
  if (g_depth == 0 && !g_ops.empty()) {
    std::vector<Op> ops;
    ops.swap(g_ops);
    return run_group(g_comm, ops);
  }
  return 0;
}

const char *ncclGetErrorString(int e) {
  switch (e) {
    case 0: return "success";
    case 1: return "fake_rccl: transfer failed";
    case 2: return "fake_rccl: shared memory segment";
    case 4: return "fake_rccl: invalid argument";
    default: return "fake_rccl: error";
  }
}

}  // extern "C"
