// lists_copy_api.cpp -- the host-pointer half of mplx_expand_lists: successor lists from HBM into the caller's
// (pageable) arrays.  The lists are the batched form of the three output vectors of the reference's get_succ
// (include/mpl_planner/env/env_map.h:147-172); node k uses count[k] of the node_stride entries reserved for it.
//
// A plain hipMemcpy of the padded arrays moves 6.5 GB for BASELINE config C4 at the ~11 GB/s of pageable memory
// (0.59 s for a 0.66 ms kernel).  Here only the used prefixes cross the link: chunks of whole nodes are packed on
// the device (pack_kernel.hip), copied into pinned landing buffers, and scattered into the caller's arrays by a
// few host threads while the next chunk is in flight (two buffers).
#include "mplx_ctx.h"

#include <atomic>
#include <cstring>
#include <thread>

namespace mplx_detail {

namespace {

struct Row {
  const void *dev;
  char *host;
  int es;
  int64_t host_node_stride;  // elements between consecutive nodes in the host row
};

constexpr size_t kChunkBytes = 32u << 20;

int ensure_pinned(mplx_ctx *c, size_t bytes) {
  if (bytes <= c->pk_pin_cap) return MPLX_OK;
  for (int i = 0; i < 2; i++) {
    if (c->pk_pin[i]) HIP_TRY(c, hipHostFree(c->pk_pin[i]));
    c->pk_pin[i] = nullptr;
  }
  c->pk_pin_cap = 0;
  for (int i = 0; i < 2; i++) HIP_TRY(c, hipHostMalloc(&c->pk_pin[i], bytes, hipHostMallocDefault));
  c->pk_pin_cap = bytes;
  return MPLX_OK;
}

struct Chunk {
  int64_t node0, node1, off0, entries;
};

// node range [a, b) of one chunk, from the landing buffer into the caller's rows
void scatter(const Row *rows, int n_rows, const int64_t *row_off, const char *pin, const int32_t *count,
             const int64_t *offs, const Chunk &ch, int64_t a, int64_t b) {
  for (int r = 0; r < n_rows; r++) {
    const int es = rows[r].es;
    const char *src = pin + row_off[r];
    for (int64_t k = a; k < b; k++) {
      const int cnt = count[k];
      if (cnt > 0)
        std::memcpy(rows[r].host + (size_t)k * rows[r].host_node_stride * es, src + (size_t)(offs[k] - ch.off0) * es,
                    (size_t)cnt * es);
    }
  }
}

}  // namespace

void release_copy_buffers(mplx_ctx *c) {
  for (int i = 0; i < 2; i++) {
    if (c->pk_pin[i]) (void)hipHostFree(c->pk_pin[i]);
    c->pk_pin[i] = nullptr;
    if (c->pk_ev[i]) (void)hipEventDestroy(c->pk_ev[i]);
    c->pk_ev[i] = nullptr;
    release(c->pk_dev[i]);
  }
  c->pk_pin_cap = 0;
  release(c->pk_offs);
  if (c->pk_hb) (void)hipHostFree(c->pk_hb);
  c->pk_hb = nullptr;
  c->pk_hb_cap = 0;
}

int copy_lists_to_host(mplx_ctx *c, const mplx_succ_lists &d, const mplx_succ_lists *h, int64_t n_nodes) {
  const int F = 4 * c->dim + 2;
  const int64_t S = d.node_stride ? d.node_stride : c->nU;
  // counts first: they size everything else
  HIP_TRY(c, hipMemcpyAsync(h->count, d.count, (size_t)n_nodes * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  Row rows[mplx::kPackRows];
  int n_rows = 0;
  // 8-byte rows first so that every packed block starts 8-byte aligned
  if (h->cost) rows[n_rows++] = {d.cost, (char *)h->cost, 8, S};
  if (h->hash) rows[n_rows++] = {d.hash, (char *)h->hash, 8, S};
  if (h->state)
    for (int f = 0; f < F; f++)
      rows[n_rows++] = {d.state + (size_t)f * d.state_stride, (char *)(h->state + (size_t)f * h->state_stride), 8, S};
  if (h->action) rows[n_rows++] = {d.action, (char *)h->action, 4, S};
  if (h->iters) rows[n_rows++] = {d.iters, (char *)h->iters, 4, S};
  if (n_rows == 0) return MPLX_OK;
  int bpe = 0;  // bytes per list entry over the requested rows
  for (int r = 0; r < n_rows; r++) bpe += rows[r].es;

  std::vector<int64_t> &offs = c->pk_hoffs;
  offs.resize((size_t)n_nodes + 1);
  int64_t total = 0;
  int max_cnt = 0;
  for (int64_t k = 0; k < n_nodes; k++) {
    offs[(size_t)k] = total;
    const int cnt = h->count[k];
    total += cnt;
    max_cnt = cnt > max_cnt ? cnt : max_cnt;
  }
  offs[(size_t)n_nodes] = total;
  if (total == 0) return MPLX_OK;
  size_t cap = kChunkBytes;
  if ((size_t)max_cnt * bpe > cap) cap = (size_t)max_cnt * bpe;
  if ((size_t)total * bpe < cap) cap = (size_t)total * bpe;
  cap = (cap + 255) & ~(size_t)255;
  const int64_t cap_entries = (int64_t)(cap / bpe);
  if (int rc = ensure_pinned(c, cap)) return rc;
  for (int i = 0; i < 2; i++) {
    if (int rc = ensure(c, c->pk_dev[i], cap)) return rc;
    if (!c->pk_ev[i]) HIP_TRY(c, hipEventCreateWithFlags(&c->pk_ev[i], hipEventDisableTiming));
  }
  if (int rc = ensure(c, c->pk_offs, (size_t)n_nodes * 8)) return rc;
  HIP_TRY(c, hipMemcpyAsync(c->pk_offs.p, offs.data(), (size_t)n_nodes * 8, hipMemcpyHostToDevice, c->stream));

  // chunks of whole nodes
  std::vector<Chunk> chunks;
  for (int64_t a = 0; a < n_nodes;) {
    int64_t b = a + 1;
    while (b < n_nodes && offs[(size_t)b + 1] - offs[(size_t)a] <= cap_entries) b++;
    chunks.push_back({a, b, offs[(size_t)a], offs[(size_t)b] - offs[(size_t)a]});
    a = b;
  }
  const int n_chunks = (int)chunks.size();
  std::vector<int64_t> row_off((size_t)n_chunks * n_rows);
  for (int i = 0; i < n_chunks; i++) {
    int64_t o = 0;
    for (int r = 0; r < n_rows; r++) {
      row_off[(size_t)i * n_rows + r] = o;
      o += chunks[(size_t)i].entries * rows[r].es;
    }
  }

  // helper threads for the scatter: one per ~8 MiB of payload, at most 15 (+ this thread)
  int n_thr = (int)(((size_t)total * bpe) >> 23);
  const int hw = (int)std::thread::hardware_concurrency();
  if (n_thr > 15) n_thr = 15;
  if (hw > 0 && n_thr > hw - 1) n_thr = hw - 1;
  if (n_thr < 0) n_thr = 0;
  int parts = n_thr + 1;
  std::atomic<int> ready{0};   // chunks whose data has landed
  std::atomic<int> done{0};    // (chunk, helper) pairs finished
  std::atomic<bool> cancel{false};
  const int32_t *count = h->count;
  auto share = [&](int i, int part) {
    const Chunk &ch = chunks[(size_t)i];
    const int64_t n = ch.node1 - ch.node0;
    const int64_t a = ch.node0 + n * part / parts, b = ch.node0 + n * (part + 1) / parts;
    scatter(rows, n_rows, &row_off[(size_t)i * n_rows], (const char *)c->pk_pin[i & 1], count, offs.data(), ch, a, b);
  };
  std::vector<std::thread> helpers;
  try {
    for (int t = 0; t < n_thr; t++)
      helpers.emplace_back([&, t] {
        for (int i = 0; i < n_chunks; i++) {
          while (ready.load(std::memory_order_acquire) <= i) std::this_thread::yield();
          if (cancel.load(std::memory_order_acquire)) return;
          share(i, t + 1);
          done.fetch_add(1, std::memory_order_release);
        }
      });
  } catch (...) {
    // no threads to be had: nothing may throw across the C ABI -- this thread scatters alone
    cancel.store(true, std::memory_order_release);
    ready.store(n_chunks + 1, std::memory_order_release);
    for (auto &th : helpers) th.join();
    helpers.clear();
    ready.store(0, std::memory_order_release);
    n_thr = 0;
    parts = 1;
  }
  auto join_all = [&] {
    ready.store(n_chunks + 1, std::memory_order_release);
    for (auto &th : helpers) th.join();
  };

  int rc = MPLX_OK;
  auto issue = [&](int i) -> hipError_t {
    const Chunk &ch = chunks[(size_t)i];
    mplx::PackArgs a{};
    for (int r = 0; r < n_rows; r++) {
      a.src[r] = rows[r].dev;
      a.dst_off[r] = row_off[(size_t)i * n_rows + r];
      a.es[r] = rows[r].es;
    }
    a.n_rows = n_rows;
    a.node_stride = S;
    a.count = d.count;
    a.offs = (const int64_t *)c->pk_offs.p;
    a.node0 = ch.node0;
    a.off0 = ch.off0;
    a.dst = (char *)c->pk_dev[i & 1].p;
    hipError_t e = mplx::launch_pack_rows(a, ch.node1 - ch.node0, c->stream);
    if (e != hipSuccess) return e;
    if (ch.entries > 0) {
      e = hipMemcpyAsync(c->pk_pin[i & 1], c->pk_dev[i & 1].p, (size_t)ch.entries * bpe, hipMemcpyDeviceToHost, c->stream);
      if (e != hipSuccess) return e;
    }
    return hipEventRecord(c->pk_ev[i & 1], c->stream);
  };
  // chunk i is issued while chunk i-1 is scattered; buffer (i & 1) is free again once chunk i-2 is scattered,
  // which the wait at the end of the previous iteration guarantees
  for (int i = 0; i <= n_chunks && rc == MPLX_OK; i++) {
    hipError_t e = hipSuccess;
    if (i < n_chunks) e = issue(i);
    if (e == hipSuccess && i >= 1) e = hipEventSynchronize(c->pk_ev[(i - 1) & 1]);
    if (e != hipSuccess) {
      rc = fail(c, MPLX_ERR_HIP, "copy of the successor lists failed: %s", hipGetErrorString(e));
      break;
    }
    if (i >= 1) {
      ready.store(i, std::memory_order_release);
      share(i - 1, 0);
      while (done.load(std::memory_order_acquire) < i * n_thr) std::this_thread::yield();
    }
  }
  if (rc != MPLX_OK) {
    // let the helpers run out (they only touch memory that stays valid) and report the error
    join_all();
    (void)hipStreamSynchronize(c->stream);
    return rc;
  }
  join_all();
  return MPLX_OK;
}

int expand_lists_packed(mplx_ctx *c, const double *h_nodes, int64_t n_nodes, int64_t node_stride, bool want_state,
                        PackedLists *out, bool want_heur) {
  if (!c || !out || !h_nodes || n_nodes <= 0 || node_stride < n_nodes) return fail(c, MPLX_ERR_ARG, "expand_lists_packed: bad arguments");
  if (int rc = ctx_ready(c)) return rc;
  const int F = 4 * c->dim + 2;
  const int64_t S = (c->nU + 31) & ~31;  // line-aligned node stride (see expand_grid_kernel.hip)
  const int64_t n_slots = n_nodes * S;
  {
    // the batches of a search, from the second in a row: the resident kernel (mplx_api.cpp, "service"); the view
    // describes the lists in its landing block
    mplx_succ_lists want{}, v{};
    int32_t dummy_i = 0;
    double dummy_d = 0;
    uint64_t dummy_h = 0;
    want.count = &dummy_i; want.action = &dummy_i; want.cost = &dummy_d; want.hash = &dummy_h;  // (names the rows only)
    if (want_state) want.state = &dummy_d;
    if (want_heur) want.heur = &dummy_d;
    want.node_stride = S;
    bool handled = false;
    if (int rc = svc_request(c, h_nodes, n_nodes, node_stride, &want, &handled, &v)) return rc;
    if (handled) {
      c->pk_hoffs.resize((size_t)n_nodes + 1);
      for (int64_t k = 0; k <= n_nodes; k++) c->pk_hoffs[(size_t)k] = k * S;
      *out = PackedLists{};
      out->total = want_state ? v.state_stride : n_slots;  // row stride of the state rows
      out->count = v.count;
      out->offs = c->pk_hoffs.data();
      out->cost = v.cost;
      out->hash = v.hash;
      out->action = v.action;
      out->state = want_state ? v.state : nullptr;
      out->heur = want_heur ? v.heur : nullptr;
      return MPLX_OK;
    }
    const int counted = c->svc.streak;  // (what svc_request made of it; bind_device resets it)
    if (int rc = bind_device(c)) return rc;
    c->svc.streak = counted;
  }
  if (c->tune.zero_copy && (size_t)n_slots * (size_t)((want_state ? F * 8 : 0) + 24 + (want_heur ? 8 : 0)) <= ((size_t)32 << 20)) {
    // Batches of a search: the kernel reads the nodes from and writes the lists into one pinned host block itself
    // (only the used entries cross PCIe, while the kernel runs): the call is the kernel and one synchronisation.
    // The view then describes the strided lists as they are: offs[k] = k * S, row stride n_nodes * S.
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_cnt = up((size_t)F * n_nodes * 8);
    const size_t o_off = o_cnt + up((size_t)n_nodes * 4);
    const size_t o_act = o_off + up((size_t)(n_nodes + 1) * 8);
    const size_t o_cost = o_act + up((size_t)n_slots * 4);
    const size_t o_hash = o_cost + up((size_t)n_slots * 8);
    const size_t o_heur = o_hash + up((size_t)n_slots * 8);
    const size_t o_state = o_heur + (want_heur ? up((size_t)n_slots * 8) : 0);
    const size_t bytes = o_state + (want_state ? up((size_t)F * n_slots * 8) : 0);
    if (bytes > c->pk_hb_cap) {
      if (c->pk_hb) HIP_TRY(c, hipHostFree(c->pk_hb));
      c->pk_hb = nullptr;
      c->pk_hb_cap = 0;
      HIP_TRY(c, hipHostMalloc(&c->pk_hb, bytes + bytes / 2, hipHostMallocCoherent));  // read by the host while the kernel may still run (DoneSignal)
      c->pk_hb_cap = bytes + bytes / 2;
    }
    char *hb = (char *)c->pk_hb;
    for (int f = 0; f < F; f++)
      std::memcpy(hb + (size_t)f * n_nodes * 8, h_nodes + (size_t)f * node_stride, (size_t)n_nodes * 8);
    mplx_succ_lists d{};
    d.count = (int32_t *)(hb + o_cnt);
    d.action = (int32_t *)(hb + o_act);
    d.cost = (double *)(hb + o_cost);
    d.hash = (uint64_t *)(hb + o_hash);
    if (want_state) { d.state = (double *)(hb + o_state); d.state_stride = n_slots; }
    if (want_heur) d.heur = (double *)(hb + o_heur);
    d.node_stride = S;
    c->want_done = true;  // the kernel tells the host itself when the lists are in the block (DoneSignal)
    const int rc_launch = lists_on_device(c, (const double *)hb, n_nodes, n_nodes, &d);
    c->want_done = false;
    if (rc_launch) return rc_launch;
    if (int rc = wait_small_launch(c)) return rc;
    if (int rc = resolve_pending(c, true)) return rc;  // yaw pinning: lists final before the search reads them
    int64_t *offs = (int64_t *)(hb + o_off);
    for (int64_t k = 0; k <= n_nodes; k++) offs[k] = k * S;
    *out = PackedLists{};
    out->total = n_slots;
    out->count = d.count;
    out->offs = offs;
    out->cost = d.cost;
    out->hash = d.hash;
    out->action = d.action;
    out->state = d.state;
    out->heur = d.heur;
    return MPLX_OK;
  }
  // (larger batches are packed on the device and copied: the heuristic row does not travel that way -- the search
  // evaluates it itself, host_planner.hpp, as it does for every provider without that row)
  // pinned host block: [nodes F x n][count n][offs n + 1]
  const size_t o_cnt = ((size_t)F * n_nodes * 8 + 255) & ~(size_t)255;
  const size_t o_off = (o_cnt + (size_t)n_nodes * 4 + 255) & ~(size_t)255;
  const size_t hb_bytes = o_off + (size_t)(n_nodes + 1) * 8;
  if (hb_bytes > c->pk_hb_cap) {
    if (c->pk_hb) HIP_TRY(c, hipHostFree(c->pk_hb));
    c->pk_hb = nullptr;
    c->pk_hb_cap = 0;
    HIP_TRY(c, hipHostMalloc(&c->pk_hb, hb_bytes * 2, hipHostMallocCoherent));
    c->pk_hb_cap = hb_bytes * 2;
  }
  char *hb = (char *)c->pk_hb;
  for (int f = 0; f < F; f++)
    std::memcpy(hb + (size_t)f * n_nodes * 8, h_nodes + (size_t)f * node_stride, (size_t)n_nodes * 8);
  if (int rc = ensure(c, c->s_nodes, (size_t)F * n_nodes * 8)) return rc;
  if (int rc = ensure(c, c->s_count, (size_t)n_nodes * 4)) return rc;
  if (int rc = ensure(c, c->s_action, (size_t)n_slots * 4)) return rc;
  if (int rc = ensure(c, c->s_cost, (size_t)n_slots * 8)) return rc;
  if (int rc = ensure(c, c->s_hash, (size_t)n_slots * 8)) return rc;
  if (want_state)
    if (int rc = ensure(c, c->s_state, (size_t)F * n_slots * 8)) return rc;
  if (int rc = ensure(c, c->pk_offs, (size_t)n_nodes * 8)) return rc;
  HIP_TRY(c, hipMemcpyAsync(c->s_nodes.p, hb, (size_t)F * n_nodes * 8, hipMemcpyHostToDevice, c->stream));
  mplx_succ_lists d{};
  d.count = (int32_t *)c->s_count.p;
  d.action = (int32_t *)c->s_action.p;
  d.cost = (double *)c->s_cost.p;
  d.hash = (uint64_t *)c->s_hash.p;
  if (want_state) { d.state = (double *)c->s_state.p; d.state_stride = n_slots; }
  d.node_stride = S;
  if (int rc = lists_on_device(c, (const double *)c->s_nodes.p, n_nodes, n_nodes, &d)) return rc;
  if (int rc = resolve_pending(c)) return rc;  // yaw pinning: lists final before they are packed
  int32_t *cnt = (int32_t *)(hb + o_cnt);
  int64_t *offs = (int64_t *)(hb + o_off);
  HIP_TRY(c, hipMemcpyAsync(cnt, d.count, (size_t)n_nodes * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  int64_t total = 0;
  for (int64_t k = 0; k < n_nodes; k++) { offs[k] = total; total += cnt[k]; }
  offs[n_nodes] = total;
  *out = PackedLists{};
  out->total = total;
  out->count = cnt;
  out->offs = offs;
  if (total == 0) return MPLX_OK;
  const int bpe = 8 + 8 + (want_state ? 8 * F : 0) + 4;
  const size_t bytes = ((size_t)total * bpe + 255) & ~(size_t)255;
  if (int rc = ensure_pinned(c, bytes)) return rc;
  if (int rc = ensure(c, c->pk_dev[0], bytes)) return rc;
  HIP_TRY(c, hipMemcpyAsync(c->pk_offs.p, offs, (size_t)n_nodes * 8, hipMemcpyHostToDevice, c->stream));
  mplx::PackArgs a{};
  int r = 0;
  int64_t o = 0;
  auto row = [&](const void *src, int es) { a.src[r] = src; a.dst_off[r] = o; a.es[r] = es; o += total * es; r++; };
  row(d.cost, 8);
  row(d.hash, 8);
  if (want_state)
    for (int f = 0; f < F; f++) row(d.state + (size_t)f * n_slots, 8);
  row(d.action, 4);
  a.n_rows = r;
  a.node_stride = S;
  a.count = d.count;
  a.offs = (const int64_t *)c->pk_offs.p;
  a.node0 = 0;
  a.off0 = 0;
  a.dst = (char *)c->pk_dev[0].p;
  HIP_TRY(c, mplx::launch_pack_rows(a, n_nodes, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->pk_pin[0], c->pk_dev[0].p, (size_t)total * bpe, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  const char *pin = (const char *)c->pk_pin[0];
  out->cost = (const double *)(pin + a.dst_off[0]);
  out->hash = (const uint64_t *)(pin + a.dst_off[1]);
  out->state = want_state ? (const double *)(pin + a.dst_off[2]) : nullptr;
  out->action = (const int32_t *)(pin + a.dst_off[want_state ? 2 + F : 2]);
  return MPLX_OK;
}

}  // namespace mplx_detail
