// mplx_api.cpp -- the C ABI of libmplx.so (declared in include/mplx.h).
//
// Owns the per-context HIP state: one stream, the HBM-resident copies of the
// map / potential / search-region / control table, staging buffers for the
// host-pointer convenience calls, and a pair of events used as a stopwatch.
// There is deliberately no CPU implementation behind this ABI: every compute
// entry either runs the gfx950 kernels or returns an error.
#include "mplx_ctx.h"
#include "host_planner.hpp"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace mplx_detail;

namespace mplx_detail {
std::string &create_error() {
  thread_local std::string e;
  return e;
}
}  // namespace mplx_detail

namespace {

int yaw_slot(mplx_ctx *c, mplx::YawPin *y);  // yaw pinning, defined with the lists route below
int grid_work(mplx_ctx *c, mplx::GridArgs *a);
int launch_grid(mplx_ctx *c, mplx::GridArgs *a);

double mono_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

bool control_ok(int32_t control) {
  switch (control) {
    case MPLX_VEL: case MPLX_ACC: case MPLX_JRK: case MPLX_SNP:
    case MPLX_VELxYAW: case MPLX_ACCxYAW: case MPLX_JRKxYAW: case MPLX_SNPxYAW:
      return true;
    default:
      return false;
  }
}

int ready(mplx_ctx *c) {
  if (!c->has_map) return fail(c, MPLX_ERR_STATE, "mplx_set_map has not been called");
  if (!c->has_params) return fail(c, MPLX_ERR_STATE, "mplx_set_params has not been called");
  if (!c->has_U) return fail(c, MPLX_ERR_STATE, "mplx_set_controls has not been called");
  const int need = c->dim + ((c->prm.control & 0x10) ? 1 : 0);
  if (c->udim < need)
    return fail(c, MPLX_ERR_STATE, "controls have %d entries per row, control flag 0x%x needs %d",
                c->udim, c->prm.control, need);
  return MPLX_OK;
}

mplx::ExpandArgs make_args(mplx_ctx *c, const double *d_nodes, int64_t n_nodes, int64_t node_stride,
                           const mplx_succ *o) {
  mplx::ExpandArgs a{};
  a.map = (const int8_t *)c->map.p;
  a.pot = c->has_pot ? (const int8_t *)c->pot.p : nullptr;
  a.region = c->has_region ? (const uint32_t *)c->region_bits.p : nullptr;
  a.dim0 = c->mdim[0]; a.dim1 = c->mdim[1]; a.dim2 = c->mdim[2];
  a.org0 = c->origin[0]; a.org1 = c->origin[1]; a.org2 = c->origin[2];
  a.res = c->res;
  a.dt = c->prm.dt; a.w = c->prm.w; a.wyaw = c->prm.wyaw;
  a.v_max = c->prm.v_max; a.a_max = c->prm.a_max; a.j_max = c->prm.j_max; a.yaw_max = c->prm.yaw_max;
  a.pot_w = c->prm.potential_weight; a.grad_w = c->prm.gradient_weight;
  a.U = (const double *)c->U.p;
  a.nU = c->nU; a.udim = c->udim;
  a.nodes = d_nodes; a.n_nodes = n_nodes; a.node_stride = node_stride;
  a.status = o->status; a.cost = o->cost; a.hash = o->hash; a.state = o->state;
  a.state_stride = o->state_stride; a.iters = o->iters;
  return a;
}

}  // namespace

extern "C" {

int mplx_abi_version(void) { return MPLX_ABI_VERSION; }

const char *mplx_last_error(const mplx_ctx *ctx) {
  return ctx ? ctx->err.c_str() : create_error().c_str();
}

int mplx_create(int dim, int device, mplx_ctx **out) {
  if (!out) return fail(nullptr, MPLX_ERR_ARG, "mplx_create: out is NULL");
  *out = nullptr;
  if (dim != 2 && dim != 3) return fail(nullptr, MPLX_ERR_ARG, "mplx_create: dim must be 2 or 3, got %d", dim);
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return fail(nullptr, MPLX_ERR_NO_DEVICE,
                "mplx_create: no HIP device available (%s); this engine has no CPU fallback",
                e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
  if (device < 0 || device >= count)
    return fail(nullptr, MPLX_ERR_ARG, "mplx_create: device %d out of range [0,%d)", device, count);
  mplx_ctx *c = new (std::nothrow) mplx_ctx();
  if (!c) return fail(nullptr, MPLX_ERR_ARG, "mplx_create: out of host memory");
  c->dim = dim;
  c->device = device;
  e = hipSetDevice(device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreate(&c->ev0);
  if (e == hipSuccess) e = hipEventCreate(&c->ev1);
  if (e != hipSuccess) {
    fail(nullptr, MPLX_ERR_HIP, "mplx_create: HIP set-up failed: %s", hipGetErrorString(e));
    mplx_destroy(c);
    return MPLX_ERR_HIP;
  }
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
      c->n_cus = prop.multiProcessorCount;
  }
  {
    auto env_int = [](const char *name) { const char *e = getenv(name); return e ? atoi(e) : 0; };
    c->tune.grid_rmax = env_int("MPLX_GRID_RMAX");
    c->tune.grid_boxcap = env_int("MPLX_GRID_BOXCAP");
    c->tune.grid_blocks = env_int("MPLX_GRID_BLOCKS");
    c->tune.grid_waves_per_cu = env_int("MPLX_GRID_WAVES_PER_CU");
    c->tune.grid_static = getenv("MPLX_GRID_STATIC") != nullptr;
    if (getenv("MPLX_GRID_LEX")) c->tune.grid_lex = env_int("MPLX_GRID_LEX") != 0;
    c->tune.grid_chunk = env_int("MPLX_GRID_CHUNK");
    c->tune.grid_blocked = env_int("MPLX_GRID_BLOCKED");
    c->tune.grid_gather = getenv("MPLX_GRID_GATHER") ? env_int("MPLX_GRID_GATHER") : -1;
    c->tune.grid_sat = getenv("MPLX_GRID_SAT") ? env_int("MPLX_GRID_SAT") : -1;
    c->tune.dbg = env_int("MPLX_TILE_DBG");
    c->tune.no_pair = getenv("MPLX_GRID_PAIR") && env_int("MPLX_GRID_PAIR") == 0;
    c->tune.pair_rmax = env_int("MPLX_PAIR_RMAX");
    c->tune.pair_wg_per_cu = env_int("MPLX_PAIR_WG_PER_CU");
    c->tune.arena_kb = env_int("MPLX_ARENA_KB");
    c->tune.zero_copy = getenv("MPLX_ZERO_COPY") ? env_int("MPLX_ZERO_COPY") : 1;
    c->tune.no_sat = getenv("MPLX_GRID_NOSAT") != nullptr;
    c->tune.no_lex = getenv("MPLX_GRID_NOLEX") != nullptr;
    c->tune.no_line_pad = getenv("MPLX_NO_LINE_PAD") != nullptr;
    c->tune.prescreen_min = env_int("MPLX_GRID_PRESCREEN_MIN");
    c->tune.yaw_pin = !(getenv("MPLX_YAW_PIN") && atoi(getenv("MPLX_YAW_PIN")) == 0);
    c->tune.yaw_margin = getenv("MPLX_YAW_MARGIN") ? atof(getenv("MPLX_YAW_MARGIN")) : 0.0;
    if (getenv("MPLX_SERVICE")) c->tune.service = env_int("MPLX_SERVICE");
    if (getenv("MPLX_DONE_FLAG")) c->tune.done_flag = env_int("MPLX_DONE_FLAG");
    if (env_int("MPLX_SERVICE_IDLE_US") > 0) c->tune.service_idle_us = env_int("MPLX_SERVICE_IDLE_US");
    if (env_int("MPLX_SERVICE_MAX_NODES") > 0) c->tune.service_max_nodes = env_int("MPLX_SERVICE_MAX_NODES");
  }
  if (c->tune.done_flag) {
    e = hipHostMalloc((void **)&c->done_host, 64, hipHostMallocCoherent);
    if (e == hipSuccess) { *c->done_host = 0; e = hipMalloc(&c->done_count.p, 64); }
    if (e == hipSuccess) { c->done_count.cap = 64; e = hipMemset(c->done_count.p, 0, 64); }
    if (e != hipSuccess) {
      fail(nullptr, MPLX_ERR_HIP, "mplx_create: HIP set-up of the completion word failed: %s", hipGetErrorString(e));
      mplx_destroy(c);
      return MPLX_ERR_HIP;
    }
  }
  if (c->tune.service) {
    // what the first resident kernel needs, now rather than inside the first search (stream, mailbox, command words and
    // a landing block for small control tables: ~1 ms of runtime calls otherwise paid by the first plan())
    mplx_ctx::Service &sv = c->svc;
    e = hipStreamCreateWithFlags(&sv.stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipHostMalloc((void **)&sv.mb, sizeof(mplx::SvcMailbox), hipHostMallocCoherent);
    if (e == hipSuccess) {
      std::memset(sv.mb, 0, sizeof(mplx::SvcMailbox));
      e = hipHostMalloc((void **)&sv.block, (size_t)1 << 20, hipHostMallocCoherent);
    }
    if (e == hipSuccess) {
      sv.block_cap = (size_t)1 << 20;
      e = hipMalloc(&sv.dev.p, 1025 * 8);
    }
    if (e == hipSuccess) sv.dev.cap = 1025 * 8;
    if (e != hipSuccess) {
      fail(nullptr, MPLX_ERR_HIP, "mplx_create: HIP set-up of the service failed: %s", hipGetErrorString(e));
      mplx_destroy(c);
      return MPLX_ERR_HIP;
    }
  }
  *out = c;
  return MPLX_OK;
}

void mplx_destroy(mplx_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)mplx_detail::svc_stop(c);
  if (c->svc.stream) (void)hipStreamDestroy(c->svc.stream);
  if (c->svc.mb) (void)hipHostFree(c->svc.mb);
  if (c->svc.block) (void)hipHostFree(c->svc.block);
  release(c->svc.dev);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->done_host) (void)hipHostFree(c->done_host);
  release(c->done_count);
  for (DevBuf *b : {&c->map, &c->pot, &c->region_bits, &c->region_bytes, &c->U, &c->s_nodes, &c->s_status,
                    &c->s_cost, &c->s_hash, &c->s_state, &c->s_iters, &c->s_count, &c->s_action, &c->tables,
                    &c->d_status, &c->d_cost, &c->d_hash, &c->d_state, &c->d_iters, &c->uvals, &c->uidx, &c->blk, &c->sat, &c->prep_lut, &c->prep_a, &c->prep_b, &c->post_keys, &c->post_ws, &c->live_list, &c->live_ctr, &c->edit_buf, &c->e_parents, &c->e_action, &c->e_free, &c->e_cost, &c->e_cells, &c->e_count})
    release(*b);
  (void)mplx_comm_destroy(c);
  release(c->comm_meta);
  c->yaw_pending.clear();
  for (DevBuf *b : {&c->yaw_ring, &c->yaw_ids, &c->yaw_tab, &c->work_counter}) release(*b);
  if (c->yaw_any_host) (void)hipHostFree(c->yaw_any_host);
  if (c->id_ovf_host) (void)hipHostFree(c->id_ovf_host);
  mplx_detail::release_copy_buffers(c);
  release(c->s_arena);
  if (c->h_arena) (void)hipHostFree(c->h_arena);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int mplx_set_map(mplx_ctx *c, const int8_t *cells, const int32_t *dim, const double *origin, double res) {
  if (!c) return MPLX_ERR_ARG;
  if (!cells || !dim || !origin) return fail(c, MPLX_ERR_ARG, "mplx_set_map: NULL argument");
  if (!(res > 0)) return fail(c, MPLX_ERR_ARG, "mplx_set_map: resolution must be > 0");
  int64_t n = 1;
  for (int i = 0; i < c->dim; i++) {
    if (dim[i] <= 0) return fail(c, MPLX_ERR_ARG, "mplx_set_map: dim[%d] = %d", i, dim[i]);
    n *= dim[i];
  }
  if (n > 0x7fffffffLL)
    return fail(c, MPLX_ERR_ARG, "mplx_set_map: %lld cells exceed the reference's int cell index",
                (long long)n);
  if (int rc = bind_device(c)) return rc;
  if (int rc = resolve_pending(c)) return rc;  // a pending launch read the old map
  {  // a different grid (size, shape, origin or resolution) invalidates potential and region
    bool same = n == c->n_cells && res == c->res;
    for (int i = 0; i < c->dim; i++) same = same && dim[i] == c->mdim[i] && origin[i] == c->origin[i];
    if (!same) {
      c->has_pot = false;
      c->has_region = false;
    }
  }
  if (int rc = ensure(c, c->map, (size_t)n)) return rc;
  HIP_TRY(c, hipMemcpyAsync(c->map.p, cells, (size_t)n, hipMemcpyHostToDevice, c->stream));
  c->map_upload_bytes += (uint64_t)n;
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // caller may free `cells` on return
  for (int i = 0; i < 3; i++) {
    c->mdim[i] = i < c->dim ? dim[i] : 1;
    c->origin[i] = i < c->dim ? origin[i] : 0.0;
  }
  c->res = res;
  c->n_cells = n;
  c->has_map = true;
  c->blk_ok = false;
  return MPLX_OK;
}

int mplx_set_potential(mplx_ctx *c, const int8_t *cells) {
  if (!c) return MPLX_ERR_ARG;
  if (int rc = resolve_pending(c)) return rc;
  if (!cells) { c->blk_ok = c->blk_ok && !c->has_pot; c->has_pot = false; return MPLX_OK; }
  if (!c->has_map) return fail(c, MPLX_ERR_STATE, "mplx_set_potential: set the map first");
  if (int rc = bind_device(c)) return rc;
  if (int rc = ensure(c, c->pot, (size_t)c->n_cells)) return rc;
  HIP_TRY(c, hipMemcpyAsync(c->pot.p, cells, (size_t)c->n_cells, hipMemcpyHostToDevice, c->stream));
  c->map_upload_bytes += (uint64_t)c->n_cells;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->has_pot = true;
  c->blk_ok = false;
  return MPLX_OK;
}

int mplx_set_region(mplx_ctx *c, const uint8_t *cells) {
  if (!c) return MPLX_ERR_ARG;
  if (int rc = resolve_pending(c)) return rc;
  if (!cells) { c->blk_ok = c->blk_ok && !c->has_region; c->has_region = false; return MPLX_OK; }
  if (!c->has_map) return fail(c, MPLX_ERR_STATE, "mplx_set_region: set the map first");
  if (int rc = bind_device(c)) return rc;
  const size_t words = (size_t)((c->n_cells + 31) >> 5);
  if (int rc = ensure(c, c->region_bytes, (size_t)c->n_cells)) return rc;
  if (int rc = ensure(c, c->region_bits, words * 4)) return rc;
  HIP_TRY(c, hipMemcpyAsync(c->region_bytes.p, cells, (size_t)c->n_cells, hipMemcpyHostToDevice, c->stream));
  c->map_upload_bytes += (uint64_t)c->n_cells;
  HIP_TRY(c, mplx::launch_pack_region((const uint8_t *)c->region_bytes.p, (uint32_t *)c->region_bits.p,
                                      c->n_cells, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->has_region = true;
  c->blk_ok = false;
  return MPLX_OK;
}

int mplx_set_params(mplx_ctx *c, const mplx_params *p) {
  if (!c) return MPLX_ERR_ARG;
  if (!p) return fail(c, MPLX_ERR_ARG, "mplx_set_params: NULL");
  if (!control_ok(p->control)) return fail(c, MPLX_ERR_ARG, "mplx_set_params: unknown control flag 0x%x", p->control);
  if (!(p->dt > 0)) return fail(c, MPLX_ERR_ARG, "mplx_set_params: dt must be > 0");
  if (int rc = resolve_pending(c)) return rc;
  if (int rc = mplx_detail::svc_stop(c)) return rc;  // a resident kernel carries the old parameters in its arguments
  c->prm = *p;
  c->has_params = true;
  return MPLX_OK;
}

int mplx_set_goal(mplx_ctx *c, const mplx_goal_spec *g) {
  if (!c) return MPLX_ERR_ARG;
  if (int rc = resolve_pending(c)) return rc;
  mplx::PostFuse f{};
  bool has = false;
  if (g) {
    if (!g->goal) return fail(c, MPLX_ERR_ARG, "mplx_set_goal: goal waypoint is NULL");
    if (!control_ok(g->control) || (g->goal_control && !control_ok(g->goal_control)))
      return fail(c, MPLX_ERR_ARG, "mplx_set_goal: unknown control flag");
    const int F = 4 * c->dim + 2;
    for (int i = 0; i < F; i++) f.goal[i] = g->goal[i];
    // env_base.h:47 compares the goal with a state by hash, each side hashed with its own flags (waypoint.h:93-125)
    f.goal_hash = mplx::host::lattice_hash(c->dim, g->goal_control ? g->goal_control : g->control, g->goal);
    f.w = g->w; f.v_max = g->v_max;
    f.tol_pos = g->tol_pos; f.tol_vel = g->tol_vel; f.tol_acc = g->tol_acc; f.tol_yaw = g->tol_yaw;
    has = true;
  }
  if (has != c->has_goal || std::memcmp(&f, &c->goal_fuse, sizeof(f)) != 0) {
    if (int rc = mplx_detail::svc_stop(c)) return rc;  // a resident kernel carries the old goal in its arguments
    c->goal_fuse = f;
    c->has_goal = has;
  }
  return MPLX_OK;
}

int mplx_set_controls(mplx_ctx *c, const double *U, int32_t nU, int32_t udim) {
  if (!c) return MPLX_ERR_ARG;
  if (!U || nU <= 0 || udim < c->dim || udim > c->dim + 1)
    return fail(c, MPLX_ERR_ARG, "mplx_set_controls: need U != NULL, nU > 0, udim in {%d,%d}", c->dim, c->dim + 1);
  MPLX_GUARD_BEGIN
  if (int rc = bind_device(c)) return rc;
  if (int rc = resolve_pending(c)) return rc;
  const size_t bytes = (size_t)nU * udim * sizeof(double);
  if (int rc = ensure(c, c->U, bytes)) return rc;
  HIP_TRY(c, hipMemcpyAsync(c->U.p, U, bytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->nU = nU;
  c->udim = udim;
  c->h_U.assign(U, U + (size_t)nU * udim);
  c->u_absmax = 0;
  for (int32_t i = 0; i < nU; i++)
    for (int k = 0; k < c->dim; k++) {
      const double a = std::fabs(U[(size_t)i * udim + k]);
      if (a > c->u_absmax) c->u_absmax = a;
    }
  // distinct values per spatial axis (compared bit-for-bit so that signed zeros stay apart): up to 16 per axis for the
  // factorised kernels at large, up to 32 for the lexicographic one (`wide`: expand_lex_kernel.hip alone takes it)
  c->u_factored = true;
  c->u_wide = false;
  double vals32[4][32] = {};
  std::vector<uint32_t> packed((size_t)nU, 0u);
  c->u_nd[3] = 0;
  for (int k = 0; k < udim && c->u_factored; k++) {
    const int slot = k < c->dim ? k : 3;  // the yaw rate (column dim of a yaw control table) is the fourth factor
    int n = 0;
    for (int32_t i = 0; i < nU; i++) {
      const double x = U[(size_t)i * udim + k];
      int j = 0;
      for (; j < n; j++)
        if (std::memcmp(&vals32[slot][j], &x, sizeof x) == 0) break;
      if (j == n) {
        if (n == 32) { c->u_factored = false; break; }
        vals32[slot][n++] = x;
      }
      packed[(size_t)i] |= (uint32_t)j << (8 * slot);
    }
    c->u_nd[slot] = n;
    if (n > 16) c->u_wide = true;
  }
  double vals[4][16] = {};
  for (int a4 = 0; a4 < 4; a4++) std::memcpy(vals[a4], vals32[a4], sizeof vals[a4]);
  c->u_lex = false;
  if (c->u_factored) {
    for (int k = c->dim; k < 3; k++) c->u_nd[k] = 0;
    // nested-loop tables (first axis slowest, yaw rate fastest): control i <-> the i-th index combination
    int64_t prod = 1;
    for (int k = 0; k < c->dim; k++) prod *= c->u_nd[k];
    const int ny = udim > c->dim ? c->u_nd[3] : 1;
    prod *= ny;
    if (prod == nU) {
      c->u_lex = true;
      for (int32_t i = 0; i < nU && c->u_lex; i++) {
        int32_t r = i;
        uint32_t want = 0;
        if (udim > c->dim) { want |= (uint32_t)(r % ny) << 24; r /= ny; }
        for (int k = c->dim - 1; k >= 0; k--) { want |= (uint32_t)(r % c->u_nd[k]) << (8 * k); r /= c->u_nd[k]; }
        c->u_lex = want == packed[(size_t)i];
      }
    }
    // a wide table is only of use to the lexicographic kernel: no yaw column, nested-loop order
    if (c->u_wide && (!c->u_lex || udim != c->dim)) c->u_factored = false;
  }
  if (c->u_factored) {
    std::memcpy(c->h_uyaw, vals[3], sizeof c->h_uyaw);
    if (int rc = ensure(c, c->uvals, sizeof vals + sizeof vals32)) return rc;  // [4][16], then [4][32]
    if (int rc = ensure(c, c->uidx, (size_t)nU * 4)) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->uvals.p, vals, sizeof vals, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync((char *)c->uvals.p + sizeof vals, vals32, sizeof vals32, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->uidx.p, packed.data(), (size_t)nU * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  c->has_U = true;
  return MPLX_OK;
  MPLX_GUARD_END(c)
}

int mplx_expand_device(mplx_ctx *c, const double *d_nodes, int64_t n_nodes, int64_t node_stride,
                       const mplx_succ *d_out) {
  if (!c) return MPLX_ERR_ARG;
  if (!d_out || n_nodes < 0 || node_stride < n_nodes || (!d_nodes && n_nodes > 0))
    return fail(c, MPLX_ERR_ARG, "mplx_expand_device: bad arguments");
  if (int rc = ready(c)) return rc;
  if (n_nodes == 0) return MPLX_OK;
  const int64_t n_slots = n_nodes * c->nU;
  if (d_out->state && d_out->state_stride < n_slots)
    return fail(c, MPLX_ERR_ARG, "mplx_expand_device: state_stride %lld < n_slots %lld",
                (long long)d_out->state_stride, (long long)n_slots);
  if (int rc = bind_device(c)) return rc;
  mplx::ExpandArgs a = make_args(c, d_nodes, n_nodes, node_stride, d_out);
  a.stream_out = 1;
  if (int rc = yaw_slot(c, &a.yaw)) return rc;
  HIP_TRY(c, mplx::launch_expand(c->dim, c->prm.control, a, c->stream));
  if (a.yaw.amb) {
    mplx_ctx::YawPending p;
    p.kind = 1;
    p.e = a;
    c->yaw_pending.push_back(p);
  }
  return MPLX_OK;
}

int mplx_expand(mplx_ctx *c, const double *h_nodes, int64_t n_nodes, int64_t node_stride,
                const mplx_succ *h_out) {
  if (!c) return MPLX_ERR_ARG;
  if (!h_out || n_nodes < 0 || node_stride < n_nodes || (!h_nodes && n_nodes > 0))
    return fail(c, MPLX_ERR_ARG, "mplx_expand: bad arguments");
  if (int rc = ready(c)) return rc;
  if (n_nodes == 0) return MPLX_OK;
  if (int rc = bind_device(c)) return rc;
  const int F = 4 * c->dim + 2;
  const int64_t n_slots = n_nodes * c->nU;
  if (h_out->state && h_out->state_stride < n_slots)
    return fail(c, MPLX_ERR_ARG, "mplx_expand: state_stride < n_slots");
  // frontier: pack rows to stride n_nodes on the device
  if (int rc = ensure(c, c->s_nodes, (size_t)F * n_nodes * sizeof(double))) return rc;
  HIP_TRY(c, hipMemcpy2DAsync(c->s_nodes.p, (size_t)n_nodes * sizeof(double), h_nodes,
                              (size_t)node_stride * sizeof(double), (size_t)n_nodes * sizeof(double), F,
                              hipMemcpyHostToDevice, c->stream));
  mplx_succ d{};
  if (h_out->status) { if (int rc = ensure(c, c->s_status, (size_t)n_slots)) return rc; d.status = (uint8_t *)c->s_status.p; }
  if (h_out->cost) { if (int rc = ensure(c, c->s_cost, (size_t)n_slots * 8)) return rc; d.cost = (double *)c->s_cost.p; }
  if (h_out->hash) { if (int rc = ensure(c, c->s_hash, (size_t)n_slots * 8)) return rc; d.hash = (uint64_t *)c->s_hash.p; }
  if (h_out->iters) { if (int rc = ensure(c, c->s_iters, (size_t)n_slots * 4)) return rc; d.iters = (int32_t *)c->s_iters.p; }
  if (h_out->state) {
    if (int rc = ensure(c, c->s_state, (size_t)F * n_slots * 8)) return rc;
    d.state = (double *)c->s_state.p;
    d.state_stride = n_slots;
  }
  mplx::ExpandArgs a = make_args(c, (const double *)c->s_nodes.p, n_nodes, n_nodes, &d);
  a.stream_out = 1;
  if (int rc = yaw_slot(c, &a.yaw)) return rc;
  HIP_TRY(c, mplx::launch_expand(c->dim, c->prm.control, a, c->stream));
  if (a.yaw.amb) {
    mplx_ctx::YawPending p;
    p.kind = 1;
    p.e = a;
    c->yaw_pending.push_back(p);
    if (int rc = mplx_detail::resolve_pending(c)) return rc;
  }
  if (h_out->status) HIP_TRY(c, hipMemcpyAsync(h_out->status, d.status, (size_t)n_slots, hipMemcpyDeviceToHost, c->stream));
  if (h_out->cost) HIP_TRY(c, hipMemcpyAsync(h_out->cost, d.cost, (size_t)n_slots * 8, hipMemcpyDeviceToHost, c->stream));
  if (h_out->hash) HIP_TRY(c, hipMemcpyAsync(h_out->hash, d.hash, (size_t)n_slots * 8, hipMemcpyDeviceToHost, c->stream));
  if (h_out->iters) HIP_TRY(c, hipMemcpyAsync(h_out->iters, d.iters, (size_t)n_slots * 4, hipMemcpyDeviceToHost, c->stream));
  if (h_out->state)
    HIP_TRY(c, hipMemcpy2DAsync(h_out->state, (size_t)h_out->state_stride * 8, d.state, (size_t)n_slots * 8,
                                (size_t)n_slots * 8, F, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return MPLX_OK;
}

}  // extern "C"

// ---------------------------------------------------------------- lists route
namespace {

struct TilePlan {
  bool ok = false;
  int npb = 1, tile_pairs = 0, wl_cap = 0, n_max = 0, u_offset = 0, grid = 0;
};

// Decide whether the tiled kernel covers the current configuration and how to
// tile it.  The work-list capacity needs a bound on the per-pair sample count:
// a valid pair has max_v <= v_max (primitive.h:483-496), and for plain VEL
// control max_v = max |u|.
TilePlan plan_tile(const mplx_ctx *c) {
  TilePlan t;
  const mplx_params &p = c->prm;
  if (p.control & 0x10) return t;              // yaw: per-sample costs, dense kernel
  if (c->has_pot) return t;                    // potential: per-sample costs, dense kernel
  if (c->nU > 1024 || c->nU < 1) return t;
  double vbound;
  if ((p.control & 0x0f) == MPLX_VEL) vbound = c->u_absmax;
  else if (p.v_max > 0) vbound = p.v_max;
  else return t;                               // unbounded sample count
  const double nf = std::ceil(vbound * p.dt / c->res) + 1.0;  // +1: slack for the last rounding
  if (!(nf <= 63.0)) return t;
  int n_max = (int)nf;
  if (n_max < 5) n_max = 5;
  const int cnt_max = n_max + 1;               // the loop runs n or n+1 times
  int npb = 1024 / c->nU;
  if (npb < 1) npb = 1;
  if (npb > 32) npb = 32;
  for (; npb >= 1; npb--) {
    const int tp = npb * c->nU;
    int uoff = 0;
    const size_t lds = mplx::tile_lds_bytes(tp, npb, tp * cnt_max, n_max, 4 * c->dim + 2, c->nU * c->udim, &uoff);
    if (lds <= 80 * 1024 || npb == 1) {   // at least two 512-thread workgroups per CU (160 KiB LDS)
      if (lds > 160 * 1024 - 64) return t;  // (- 64: the service form's static command word)
      t.ok = true;
      t.npb = npb;
      t.tile_pairs = tp;
      t.wl_cap = tp * cnt_max;
      t.n_max = n_max;
      t.u_offset = uoff;
      t.grid = c->n_cus * (lds <= 53 * 1024 ? 3 : lds <= 80 * 1024 ? 2 : 1);
      return t;
    }
  }
  return t;
}

struct GridPlan {
  bool ok = false;
  int ndp = 1, n_max = 0, rmax = 0, boxcap = 0, grid = 0, order = 0;
  bool gather = false, use_sat = true;
  bool lex = false;  // expand_lex_kernel.hip serves it (lexicographic table, no yaw, occupancy map)
};

// Does the factorised kernel cover the current configuration, and how is it sized?
GridPlan plan_grid(const mplx_ctx *c) {
  GridPlan g;
  const mplx_params &p = c->prm;
  const bool yaw = (p.control & 0x10) != 0;
  if (yaw && (c->udim != c->dim + 1 || c->u_nd[3] < 1)) return g;
  if (!c->u_factored || c->nU < 1) return g;
  // (control tables of more than 1 024 entries or more than 16 values on an axis: the lexicographic kernel alone, up to 8 192)
  const bool lex_only = c->u_wide || c->nU > 1024;
  if (lex_only && (c->nU > 8192 || !c->u_lex || c->tune.no_lex || !c->tune.grid_lex || yaw || c->has_pot ||
                   !mplx::lex_covers(c->dim, p.control)))
    return g;
  double vbound;
  if ((p.control & 0x0f) == MPLX_VEL) vbound = c->u_absmax;
  else if (p.v_max > 0) vbound = p.v_max;
  else return g;
  const double nf = std::ceil(vbound * p.dt / c->res) + 1.0;
  if (!(nf <= 61.0)) return g;
  int n_max = (int)nf;
  if (n_max < 5) n_max = 5;
  int ndp = 1;
  for (int i = 0; i < c->dim; i++) ndp = c->u_nd[i] > ndp ? c->u_nd[i] : ndp;
  // LDS per wave: `rmax` rows of cell codes per axis entry and `boxcap` dwords of staged blocked bits.
  // A box of (n_max + 3)^(D-1) rows covers every node whose per-axis velocities keep their sign.
  const int ctl = p.control & 0x0f;
  const int order = ctl == MPLX_VEL ? 1 : ctl == MPLX_ACC ? 2 : ctl == MPLX_JRK ? 3 : 4;
  // (SNP: the rare primitives whose cell codes leave their range are sampled by direct evaluation, which covers
  // potential maps and the heading cost too since round 3)
  // GridLds's mode word: bits 0-1 the yaw tables, bit 2 the velocity rows of every axis (gradient cost of a potential map)
  const int ym = (yaw ? (p.wyaw > 0 ? 2 : 1) : 0) | ((c->has_pot && p.gradient_weight != 0) ? 4 : 0), ndy = yaw ? c->u_nd[3] : 0;
  int rmax = 4, boxcap = (c->dim == 3) ? (n_max + 3) * (n_max + 3) : 4 * (n_max + 3);
  if (boxcap < 64) boxcap = 64;
  if (boxcap > 1024) boxcap = 1024;
  if (c->has_pot) boxcap = 64;  // potential maps are sampled from the int8 map itself: no staged bits (LDS buys occupancy)
  // Gather mode (the sample loops read the blocked-bit map directly instead of staging the reach box in LDS): fewer
  // look-ups than box rows for small control tables, yet slower in practice -- a look-up costs 35 instructions
  // against 17 from LDS.
  g.gather = false;  // measured slower than staging on C2 / C3 (profiles/README.md round 2): kept as a forced mode
  if (c->tune.grid_gather >= 0) g.gather = c->tune.grid_gather != 0;
  // Free-box query: exact reach boxes exist for K <= 2; for K = 3 the box is the conservative |p - p0| <= max_vel * T,
  // rarely free, and the query is one more dependent round trip per node (C3: -6 % without it)
  g.use_sat = (!g.gather || c->has_pot) && order <= 2;
  if (c->tune.grid_sat >= 0) g.use_sat = c->tune.grid_sat != 0;
  if (g.gather) boxcap = 64;
  if (c->tune.grid_rmax > 0) rmax = c->tune.grid_rmax;
  if (c->tune.grid_boxcap > 0) boxcap = c->tune.grid_boxcap;
  if (rmax < 1) rmax = 1;
  const int ulex = (c->u_lex && !c->tune.no_lex) ? 1 : 0;  // = GridArgs::ulex
  // the lexicographic kernel: same plan, its own LDS carve-up and occupancy
  g.lex = ulex && !yaw && !c->has_pot && !g.gather && c->tune.grid_lex && mplx::lex_covers(c->dim, p.control);
  if (lex_only && !g.lex) return GridPlan();
  auto lds_of = [&](int rm) -> size_t {
    return g.lex ? mplx::lex_lds_bytes(c->dim, order, ndp, c->nU, n_max, rm, boxcap)
                 : mplx::grid_lds_bytes(c->dim, order, c->nU, ndp, n_max, rm, boxcap, ym, ndy, ulex);
  };
  while (rmax > 1 && lds_of(rmax) > 80 * 1024) rmax--;
  // (both factorised kernels run 4 waves = 4 nodes in flight per workgroup; plan_grid, grid_work and the prescreen
  // threshold size launches with the one figure)
  const int wpb = g.lex ? mplx::lex_waves_per_block() : mplx::grid_waves_per_block();
  if (mplx::lex_waves_per_block() != mplx::grid_waves_per_block()) return GridPlan();
  // The launch is persistent: every workgroup must be RESIDENT (a workgroup that waits for a slot starts its first,
  // statically assigned node only after another one has drained the whole queue).  What fits is the runtime's answer
  // for this instantiation (registers, LDS granules), not LDS bytes alone.
  auto resident = [&](int rm, size_t *lds_out) -> int {
    const size_t lds = lds_of(rm);
    *lds_out = lds;
    if (lds > 160 * 1024) return 0;
    int nb = -1;
    // (the cache key tells the two kernels apart through its control word: bit 8 = the lexicographic kernel)
    // ... and the lexicographic kernel's instantiations (table sizes 8 / 16 / 32 by the values per axis) through bits 12+
    const int key = p.control | (g.lex ? 0x100 | (ndp << 12) : 0);
    for (const auto &e : c->grid_occ)
      if (e.control == key && e.pot == c->has_pot && e.lds == lds) nb = e.nb;
    if (nb < 0) {
      nb = g.lex ? mplx::lex_resident_blocks(c->dim, p.control, ndp, lds) : mplx::grid_resident_blocks(c->dim, p.control, c->has_pot, lds);
      if (c->grid_occ.size() >= 8) c->grid_occ.clear();
      c->grid_occ.push_back({key, c->has_pot, lds, nb});
      if (getenv("MPLX_GRID_VERBOSE"))
        fprintf(stderr, "mplx: %s kernel control 0x%x pot %d rows/pass %d: LDS %zu B per workgroup, %d workgroups resident per CU\n",
                g.lex ? "lex" : "grid", p.control, (int)c->has_pot, rm, lds, nb);
    }
    const int by_lds = (int)((160 * 1024) / lds);
    return (nb > 0 && nb < by_lds) ? nb : by_lds;
  };
  // 16 waves per CU (4 per SIMD): what the register allocation of every instantiation allows, and the measured
  // optimum where more would fit (profiles/README.md)
  // (the lexicographic kernel is leaner and latency-bound: C4 edges-only 0.353 / 0.302 / 0.278 ms at 12 / 16 / 20 waves per
  // CU, profiles/r04_lex_occupancy.txt -- it takes what its registers and LDS allow, up to 24)
  const int cap = c->tune.grid_waves_per_cu > 0 ? c->tune.grid_waves_per_cu : (g.lex ? 24 : 16);
  size_t lds = 0;
  int per_cu = resident(rmax, &lds);
  if (per_cu < 1) return g;
  // A slightly smaller box budget when that admits another workgroup (lexicographic kernel, Dim 3): the staged box of a
  // node is (n_max + 3)^2 words at most and far smaller for nearly every node (the rare larger one reads the blocked-bit
  // map directly).  C3: 1024 -> 800 words = 4 instead of 3 workgroups per CU, 71.5 -> 65.4 us.
  if (g.lex && c->dim == 3 && c->tune.grid_boxcap <= 0 && per_cu * wpb < cap) {
    const int keep = boxcap;
    boxcap = (boxcap * 25 / 32) & ~31;
    size_t lds_b = 0;
    const int per_cu_b = boxcap >= 256 ? resident(rmax, &lds_b) : 0;
    if (per_cu_b > per_cu) { per_cu = per_cu_b; lds = lds_b; }
    else boxcap = keep;
  }
  // One row less per pass when that is what lets another workgroup in (the heading-cost tables of ACCxYAW with
  // wyaw > 0: 45 KB per workgroup = 3 resident, 35 KB = 4; C5 0.108 -> 0.103 ms, a second pass is rare)
  if (c->tune.grid_rmax <= 0 && rmax == 4 && per_cu * wpb < cap) {
    size_t lds3 = 0;
    const int per_cu3 = resident(3, &lds3);
    if (per_cu3 > per_cu) { rmax = 3; per_cu = per_cu3; lds = lds3; }
  }
  if (per_cu * wpb > cap) per_cu = cap / wpb;
  if (per_cu < 1) per_cu = 1;
  g.ok = true;
  g.ndp = ndp;
  g.n_max = n_max;
  g.rmax = rmax;
  g.boxcap = boxcap;
  g.order = order;
  g.grid = c->n_cus * per_cu;
  if (c->tune.grid_blocks > 0) g.grid = c->tune.grid_blocks;
  return g;
}

// Dynamic node assignment of the factorised kernel (GridArgs::work): two sets of counters in the context, used
// alternately; a launch finds its set zero and zeroes the other one for the next launch.
int grid_work(mplx_ctx *c, mplx::GridArgs *a) {
  a->work = nullptr;
  a->work_zero = nullptr;
  a->work_chunk = 1;
  const int64_t wpb = mplx::grid_waves_per_block();
  const int64_t n_wg = (a->n_nodes + wpb - 1) / wpb;
  const int64_t W = (n_wg < (int64_t)a->grid_limit ? n_wg : (int64_t)a->grid_limit) * wpb;
  if (c->tune.grid_static || a->n_nodes <= W) return MPLX_OK;  // one node per wave at most: nothing to balance
  const size_t set_bytes = (size_t)mplx::kWorkCounters * 128;
  if (!c->work_counter.p) {
    if (int rc = ensure(c, c->work_counter, 2 * set_bytes)) return rc;
    HIP_TRY(c, hipMemsetAsync(c->work_counter.p, 0, 2 * set_bytes, c->stream));
    c->work_parity = 0;
  }
  a->work = (unsigned int *)((char *)c->work_counter.p + (c->work_parity ? set_bytes : 0));
  a->work_zero = (unsigned int *)((char *)c->work_counter.p + (c->work_parity ? 0 : set_bytes));
  // (the caller flips work_parity once the launch is enqueued: launch_grid below)
  // chunk: whole nodes per claim; 1 while a wave gets fewer than ~16 nodes (balance matters most), more beyond
  int64_t per_wave = a->n_nodes / W, ck = per_wave / 16;
  if (ck < 1) ck = 1;
  if (ck > 8) ck = 8;
  if (c->tune.grid_chunk > 0) ck = c->tune.grid_chunk;
  a->work_chunk = (int32_t)ck;
  a->work_blocked = c->tune.grid_blocked ? 1 : 0;
  return MPLX_OK;
}

// Enqueues the factorised kernel.  The counter sets change hands only when the launch went in: a launch that failed
// has not zeroed the other set, so the sets are dropped and made afresh (zeroed) on the next use.
int launch_grid(mplx_ctx *c, mplx::GridArgs *a) {
  if (int rc = grid_work(c, a)) return rc;
  const hipError_t e = a->lex ? mplx::launch_expand_lex(c->dim, c->prm.control, *a, c->stream)
                              : mplx::launch_expand_grid(c->dim, c->prm.control, *a, c->stream);
  c->last_grid_lex = a->lex != 0;
  c->last_grid_pair = false;
  if (e != hipSuccess) {
    if (a->work) {
      (void)hipStreamSynchronize(c->stream);
      release(c->work_counter);
    }
    return fail(c, MPLX_ERR_HIP, "expand_grid_kernel launch failed: %s", hipGetErrorString(e));
  }
  if (a->work) c->work_parity ^= 1;
  return MPLX_OK;
}

// ---------------------------------------------------------------- yaw pinning (YawPin, mplx_internal.h)
constexpr int kAmbCap = 1023;          // flagged nodes recorded per launch; beyond that the whole launch is re-checked
constexpr int kYawRing = 32;           // launches that may wait for their check
constexpr double kYawMargin = 0x1p-46; // |d - cos(yaw_max)| below this is "within rounding noise": both libraries are
                                       // within a few ulp (2^-53) of the true cos / sin, d is two products and a sum

bool yaw_pin_active(const mplx_ctx *c) {
  return c->tune.yaw_pin && (c->prm.control & 0x10) && c->prm.yaw_max > 0;
}

void host_sincos(double x, double *s, double *c);

// The detection block of the next launch: a slot of the ring (older launches are resolved first when it is full).
int yaw_slot(mplx_ctx *c, mplx::YawPin *y) {
  *y = mplx::YawPin{};
  if (!yaw_pin_active(c)) return MPLX_OK;
  if ((int)c->yaw_pending.size() >= kYawRing)
    if (int rc = mplx_detail::resolve_pending(c)) return rc;
  if (!c->yaw_ring.p) {
    const size_t bytes = (size_t)kYawRing * (1 + kAmbCap) * 4;
    if (int rc = ensure(c, c->yaw_ring, bytes)) return rc;
    HIP_TRY(c, hipMemsetAsync(c->yaw_ring.p, 0, bytes, c->stream));
  }
  if (!c->yaw_any_host) {
    HIP_TRY(c, hipHostMalloc((void **)&c->yaw_any_host, 64, hipHostMallocCoherent));
    *c->yaw_any_host = 0;
  }
  y->any_host = c->yaw_any_host;
  y->amb = (int32_t *)c->yaw_ring.p + c->yaw_pending.size() * (size_t)(1 + kAmbCap);
  y->amb_cap = kAmbCap;
  y->margin = c->tune.yaw_margin > 0 ? c->tune.yaw_margin : kYawMargin;
  {  // is the x-aligned tie exact under THIS host's libm (near_limit, mplx_device_common.h)?
    double sp, cp, sm, cm;
    host_sincos(c->prm.yaw_max, &sp, &cp);
    host_sincos(-c->prm.yaw_max, &sm, &cm);
    const double lim = std::cos(c->prm.yaw_max);
    y->tie_yaw = (cp == lim && cm == lim) ? c->prm.yaw_max : std::nan("");
  }
  return MPLX_OK;
}

// cos and sin of one heading the way the reference's binary gets them: primitive.h:519-520 calls cos(w.yaw) and
// sin(w.yaw) in one expression, and GCC (the reference's compiler, at -O1 and above) fuses such a pair into ONE glibc
// sincos() call.  glibc's sincos is not bit-identical to its separate sin() / cos() on every argument (measured: 2 of
// 40 000 threshold headings, profiles/README.md round 2), so the pinning asks sincos() too.  cos(yaw_max) stands
// alone in the reference (primitive.h:521) and stays a plain cos().
void host_sincos(double x, double *s, double *c) { ::sincos(x, s, c); }

double host_wrap(double a) {  // mpl_basis/math.h:15-19
  while (a > M_PI) a -= 2.0 * M_PI;
  while (a < -M_PI) a += 2.0 * M_PI;
  return a;
}

// Re-expands the nodes `ids` of a pending launch with every trig value of a heading-limit decision taken from the
// HOST libm -- the library the reference itself calls (primitive.h:504-525 -> std::cos / std::sin).
int yaw_fix_pass(mplx_ctx *c, const mplx_ctx::YawPending &p, const int32_t *ids, int64_t n) {
  const int D = c->dim;
  const double *nodes = p.kind == 0 ? p.g.nodes : p.e.nodes;
  const int64_t nstride = p.kind == 0 ? p.g.node_stride : p.e.node_stride;
  const double T = c->prm.dt;
  // the nodes' yaw (row 4D of the frontier; device memory or a pinned host block the kernel read in place)
  std::vector<double> yaw((size_t)n);
  if (n <= 256) {
    for (int64_t k = 0; k < n; k++)
      HIP_TRY(c, hipMemcpyAsync(&yaw[(size_t)k], nodes + (int64_t)(4 * D) * nstride + ids[k], 8, hipMemcpyDefault, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  } else {
    int32_t hi = 0;
    for (int64_t k = 0; k < n; k++) hi = ids[k] > hi ? ids[k] : hi;
    std::vector<double> row((size_t)hi + 1);
    HIP_TRY(c, hipMemcpyAsync(row.data(), nodes + (int64_t)(4 * D) * nstride, ((size_t)hi + 1) * 8, hipMemcpyDefault, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int64_t k = 0; k < n; k++) yaw[(size_t)k] = row[(size_t)ids[k]];
  }
  const int nrate = p.kind == 0 ? 16 : c->nU;
  const int stride = 2 + 2 * nrate;
  std::vector<double> tab((size_t)n * stride, 0.0);
  for (int64_t k = 0; k < n; k++) {
    double *t = &tab[(size_t)k * stride];
    const double cyaw = yaw[(size_t)k];
    const double y0 = host_wrap((0.0 + 0.0) + cyaw);  // the yaw polynomial at t = 0 (primitive.h:329, 128-145)
    host_sincos(y0, &t[1], &t[0]);
    if (p.kind == 0) {
      // factorised kernel: [c0, s0, cT[16], sT[16]] over the distinct yaw rates
      for (int j = 0; j < c->u_nd[3] && j < 16; j++) {
        const double yT = host_wrap((0.0 + c->h_uyaw[j] * T) + cyaw);
        host_sincos(yT, &t[2 + 16 + j], &t[2 + j]);
      }
    } else {
      // dense kernel: [c0, s0, {cT, sT} per control]
      for (int i = 0; i < c->nU; i++) {
        const double yT = host_wrap((0.0 + c->h_U[(size_t)i * c->udim + D] * T) + cyaw);
        host_sincos(yT, &t[2 + 2 * i + 1], &t[2 + 2 * i]);
      }
    }
  }
  if (int rc = ensure(c, c->yaw_ids, (size_t)n * 4)) return rc;
  if (int rc = ensure(c, c->yaw_tab, tab.size() * 8)) return rc;
  HIP_TRY(c, hipMemcpyAsync(c->yaw_ids.p, ids, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->yaw_tab.p, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, c->stream));
  mplx::YawPin y{};
  y.node_list = (const int32_t *)c->yaw_ids.p;
  y.tab = (const double *)c->yaw_tab.p;
  y.tab_stride = stride;
  y.cos_lim = std::cos(c->prm.yaw_max);
  if (p.kind == 0) {
    mplx::GridArgs a = p.g;
    a.n_nodes = n;
    a.yaw = y;
    if (int rc = launch_grid(c, &a)) return rc;
  } else {
    mplx::ExpandArgs a = p.e;
    a.n_nodes = n;
    a.yaw = y;
    HIP_TRY(c, mplx::launch_expand(c->dim, c->prm.control, a, c->stream));
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // `tab` and `ids` leave scope; yaw_ids / yaw_tab are reused
  c->yaw_fix_passes++;
  return MPLX_OK;
}

// (Re)build the summed-area table of the CURRENT blocked bits (after mplx_edit_map patched them).
int rebuild_sat(mplx_ctx *c) {
  c->sat_stale = false;
  const int d2p = c->dim == 3 ? c->mdim[2] + 1 : 2;
  const int64_t sat_n = (int64_t)(c->mdim[0] + 1) * (c->mdim[1] + 1) * d2p;
  if (c->tune.no_sat || sat_n * 4 > (16LL << 30)) return MPLX_OK;
  if (int rc = ensure(c, c->sat, (size_t)sat_n * 4)) return rc;
  HIP_TRY(c, mplx::launch_build_sat(c->dim, (const uint32_t *)c->blk.p, c->mdim, (uint32_t *)c->sat.p, c->stream));
  c->sat_ok = true;
  return MPLX_OK;
}

int ensure_blocked_bits(mplx_ctx *c) {
  if (c->blk_ok) return MPLX_OK;
  c->sat_stale = false;
  const int64_t words = (c->n_cells + 31) >> 5;
  if (int rc = ensure(c, c->blk, (size_t)words * 4)) return rc;
  HIP_TRY(c, mplx::launch_build_blocked_bits((const int8_t *)(c->has_pot ? c->pot.p : c->map.p),
                                             c->has_region ? (const uint32_t *)c->region_bits.p : nullptr, c->n_cells,
                                             c->has_pot ? 1 : 0, (uint32_t *)c->blk.p, c->stream));
  // summed-area table for the free-box shortcut of the grid kernel (skipped for maps where it would not
  // fit an unsigned count or 16 GiB; the kernel then samples every node)
  c->sat_ok = false;
  const int d2p = c->dim == 3 ? c->mdim[2] + 1 : 2;
  const int64_t sat_n = (int64_t)(c->mdim[0] + 1) * (c->mdim[1] + 1) * d2p;
  if (!c->tune.no_sat && sat_n * 4 <= (16LL << 30)) {
    if (int rc = ensure(c, c->sat, (size_t)sat_n * 4)) return rc;
    HIP_TRY(c, mplx::launch_build_sat(c->dim, (const uint32_t *)c->blk.p, c->mdim, (uint32_t *)c->sat.p, c->stream));
    c->sat_ok = true;
  }
  c->blk_ok = true;
  return MPLX_OK;
}

int ensure_tables(mplx_ctx *c) {
  if (c->tables_ok && c->tab_dt == c->prm.dt && c->tab_res == c->res) return MPLX_OK;
  const size_t bytes = 64 * 64 * 8 + 64 + 64;  // ttab, tcnt, 3 reciprocals (8-byte aligned tail)
  if (int rc = ensure(c, c->tables, bytes)) return rc;
  double *ttab = (double *)c->tables.p;
  unsigned char *tcnt = (unsigned char *)c->tables.p + 64 * 64 * 8;
  double *rec = (double *)((unsigned char *)c->tables.p + 64 * 64 * 8 + 64);
  HIP_TRY(c, hipMemsetAsync(c->tables.p, 0, bytes, c->stream));
  HIP_TRY(c, mplx::launch_make_tables(c->prm.dt, c->res, ttab, tcnt, rec, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->recips, rec, 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->tab_dt = c->prm.dt;
  c->tab_res = c->res;
  c->tables_ok = true;
  return MPLX_OK;
}

// the goal of mplx_set_goal with this launch's output rows (null rows: nothing is computed)
mplx::PostFuse post_of(const mplx_ctx *c, const mplx_succ_lists *o) {
  mplx::PostFuse f = c->goal_fuse;
  f.heur = o->heur;
  f.flags = o->flags;
  return f;
}

mplx::TileArgs tile_args(mplx_ctx *c, const TilePlan &tp, const double *d_nodes, int64_t n_nodes, int64_t node_stride,
                         const mplx_succ_lists *o) {
  mplx::TileArgs a{};
  a.map = (const int8_t *)c->map.p;
  a.region = c->has_region ? (const uint32_t *)c->region_bits.p : nullptr;
  a.dim0 = c->mdim[0]; a.dim1 = c->mdim[1]; a.dim2 = c->mdim[2];
  a.org0 = c->origin[0]; a.org1 = c->origin[1]; a.org2 = c->origin[2];
  a.res = c->res;
  a.dt = c->prm.dt; a.w = c->prm.w;
  a.v_max = c->prm.v_max; a.a_max = c->prm.a_max; a.j_max = c->prm.j_max;
  a.U = (const double *)c->U.p;
  a.nU = c->nU; a.udim = c->udim;
  a.inv_nU = 1.0f / (float)c->nU;
  a.nodes = d_nodes; a.n_nodes = n_nodes; a.node_stride = node_stride;
  a.npb = tp.npb; a.tile_pairs = tp.tile_pairs; a.wl_cap = tp.wl_cap; a.n_max = tp.n_max;
  a.lds_u_offset = tp.u_offset; a.grid_limit = tp.grid;
  a.dbg = c->tune.dbg;  // timing ablations, 0 in production
  a.ttab = (const double *)c->tables.p;
  a.tcnt = (const unsigned char *)c->tables.p + 64 * 64 * 8;
  a.Rres = c->recips[0]; a.R001 = c->recips[1]; a.R01 = c->recips[2];
  a.l_count = o->count; a.l_action = o->action; a.l_cost = o->cost; a.l_hash = o->hash;
  a.l_state = o->state; a.l_stride = o->state_stride; a.l_iters = o->iters;
  a.l_nstride = o->node_stride ? o->node_stride : c->nU;
  a.post = post_of(c, o);
  return a;
}

int lists_device(mplx_ctx *c, const double *d_nodes, int64_t n_nodes, int64_t node_stride,
                 const mplx_succ_lists *o) {
  const int F = 4 * c->dim + 2;
  const int route = c->lists_route;
  if ((o->heur || o->flags) && !c->has_goal)
    return fail(c, MPLX_ERR_STATE, "the heur / flags rows of the lists need a goal: call mplx_set_goal first");
  GridPlan gp = (route == MPLX_ROUTE_AUTO || route == MPLX_ROUTE_GRID) ? plan_grid(c) : GridPlan();
  if (route == MPLX_ROUTE_GRID && !gp.ok)
    return fail(c, MPLX_ERR_STATE, "lists route GRID does not cover this configuration");
  // A few hundred nodes with a large control table (the batches of a 3D search) are bound by the latency of
  // one node, and a node is a whole workgroup in the tiled kernel but a single wave in the factorised one:
  // 32 us against 54 us per launch for 16 - 256 nodes at |U| = 729 (profiles/micro/route_latency.py; no
  // difference for |U| <= 125).
  if (route == MPLX_ROUTE_AUTO && gp.ok && n_nodes <= 512 && c->nU >= 512 && plan_tile(c).ok) gp.ok = false;
  if (gp.ok && n_nodes >= 0x7fffffffLL - 4096) gp.ok = false;  // the factorised kernel counts nodes in 32 bits
  if (gp.ok) {
    if (int rc = ensure_tables(c)) return rc;
    mplx::GridArgs a{};
    if (int rc = ensure_blocked_bits(c)) return rc;
    // blocked bits patched by mplx_edit_map: the free-box shortcut is off until a launch of at least a few thousand
    // nodes makes its table (1.3 ms of scans at 512^3) worth rebuilding; a search's batches sample every node meanwhile
    if (c->sat_stale && n_nodes >= 4096)
      if (int rc = rebuild_sat(c)) return rc;
    a.blk = (const uint32_t *)c->blk.p;
    a.blk_words = (c->n_cells + 31) >> 5;
    a.pot = c->has_pot ? (const int8_t *)c->pot.p : nullptr;
    a.region = c->has_region ? (const uint32_t *)c->region_bits.p : nullptr;
    a.pot_w = c->prm.potential_weight;
    a.grad_w = c->has_pot ? c->prm.gradient_weight : 0.0;
    const bool yaw = (c->prm.control & 0x10) != 0;
    // the free-box shortcut skips the sample loops, which a per-sample heading cost (wyaw > 0) still needs
    a.sat = (c->sat_ok && gp.order <= 3 && !(yaw && c->prm.wyaw > 0) && !c->tune.no_sat && gp.use_sat)
                ? (const uint32_t *)c->sat.p : nullptr;
    a.gather = gp.gather ? 1 : 0;
    a.yaw_max = c->prm.yaw_max; a.wyaw = c->prm.wyaw; a.ndy = yaw ? c->u_nd[3] : 0;
    a.dim0 = c->mdim[0]; a.dim1 = c->mdim[1]; a.dim2 = c->mdim[2];
    a.org0 = c->origin[0]; a.org1 = c->origin[1]; a.org2 = c->origin[2];
    a.res = c->res;
    a.dt = c->prm.dt; a.w = c->prm.w;
    a.v_max = c->prm.v_max; a.a_max = c->prm.a_max; a.j_max = c->prm.j_max;
    a.uvals = (const double *)c->uvals.p + (c->u_wide ? 4 * 16 : 0);
    a.uval_stride = c->u_wide ? 32 : 16;
    a.uidx = (const uint32_t *)c->uidx.p;
    a.nd0 = c->u_nd[0]; a.nd1 = c->u_nd[1]; a.nd2 = c->u_nd[2];
    a.ndp = gp.ndp;
    a.ulex = (c->u_lex && !c->tune.no_lex) ? 1 : 0;
    a.nU = c->nU;
    a.nodes = d_nodes; a.n_nodes = n_nodes; a.node_stride = node_stride;
    a.n_max = gp.n_max; a.rmax = gp.rmax; a.boxcap = gp.boxcap; a.grid_limit = gp.grid;
    a.lex = gp.lex ? 1 : 0;
    a.dbg = c->tune.dbg;  // timing ablations, 0 in production
    a.ttab = (const double *)c->tables.p;
    a.tcnt = (const unsigned char *)c->tables.p + 64 * 64 * 8;
    a.Rres = c->recips[0]; a.R001 = c->recips[1]; a.R01 = c->recips[2];
    a.l_count = o->count; a.l_action = o->action; a.l_cost = o->cost; a.l_hash = o->hash;
    a.l_state = o->state; a.l_stride = o->state_stride; a.l_iters = o->iters;
    a.l_nstride = o->node_stride ? o->node_stride : c->nU;
    // Line padding (expand_grid_kernel.hip) pays where a launch is bound by its stores -- the large control tables (C4: 729,
    // 17^3).  With short lists it only adds bytes: C5 (81 controls, 24 successors per live node) writes 23.1 MB padded and
    // 18.8 MB unpadded in the same 46.5 us, C3 and C2 likewise (profiles/r06_line_pad_small_lists.txt).
    a.l_pad = (a.l_nstride % 32 == 0 && !c->tune.no_line_pad && c->nU >= mplx::kLinePadMinControls) ? 1 : 0;
    a.post = post_of(c, o);
    if (int rc = yaw_slot(c, &a.yaw)) return rc;
    // Yaw controls with a heading limit on a frontier of several nodes per wave: validate_yaw(t = 0) of every node
    // first, lane per node, and the main kernel walks the survivors only (grid_prescreen_kernel).  Small batches (a
    // search's) skip it: one more launch costs them more than the dead nodes do.
    const int64_t ps_min = c->tune.prescreen_min > 0 ? c->tune.prescreen_min : (int64_t)4 * gp.grid * mplx::grid_waves_per_block();
    if (yaw && gp.order >= 2 && c->prm.yaw_max > 0 && c->tune.prescreen_min >= 0 && n_nodes >= ps_min && n_nodes < 0x7fffffffLL) {
      if (int rc = ensure(c, c->live_list, (size_t)n_nodes * 4)) return rc;
      // the survivors' counter: two words used alternately, each on its own line; a pre-screen launch finds its word
      // zero and zeroes the other one for the next launch of the stream (no memset per launch: it was 6.6 % of C5's GPU
      // time in round 3).  A launch that fails leaves the pair in an unknown state: dropped and made afresh.
      if (!c->live_ctr.p) {
        if (int rc = ensure(c, c->live_ctr, 256)) return rc;
        HIP_TRY(c, hipMemsetAsync(c->live_ctr.p, 0, 256, c->stream));
        c->live_parity = 0;
      }
      int32_t *live = (int32_t *)c->live_list.p;
      uint32_t *live_n = (uint32_t *)c->live_ctr.p + (c->live_parity ? 32 : 0);
      uint32_t *live_zero = (uint32_t *)c->live_ctr.p + (c->live_parity ? 0 : 32);
      const hipError_t pe = mplx::launch_grid_prescreen(c->dim, c->prm.control, a, live, live_n, live_zero, c->stream);
      if (pe != hipSuccess) {
        (void)hipStreamSynchronize(c->stream);
        release(c->live_ctr);
        return fail(c, MPLX_ERR_HIP, "grid_prescreen_kernel launch failed: %s", hipGetErrorString(pe));
      }
      c->live_parity ^= 1;
      a.live = live;
      a.live_n = live_n;
    }
    c->done_armed = false;
    if (c->want_done && c->tune.done_flag && c->done_host) {
      a.done.flag = c->done_host;
      a.done.count = (uint32_t *)c->done_count.p;
      a.done.seq = ++c->done_seq;
    }
    // Yaw controls on a potential map over a pre-screened frontier (BASELINE config 5): two nodes per wave
    // (expand_pair_kernel.hip) -- the few thousand survivors then are ONE round of wave tasks instead of two.  Same lists.
    bool paired = false;
    if (a.live != nullptr && c->has_pot && a.ulex && !gp.lex && !c->tune.no_pair && a.yaw.tab == nullptr &&
        mplx::pair_covers(c->dim, c->prm.control) && c->dim * gp.ndp <= 16 && a.ndy <= mplx::pair_max_yaw_rates()) {
      mplx::GridArgs b = a;
      int rm = c->tune.pair_rmax > 0 ? c->tune.pair_rmax : 3, per_cu = 0;
      size_t lds = 0;
      for (; rm >= 1 && per_cu < 1; rm--) {  // (rows per pass down to what fits at all)
        lds = mplx::pair_lds_bytes(c->dim, gp.order, c->nU, gp.ndp, gp.n_max, rm, c->prm.wyaw > 0, a.ndy);
        if (lds > 160 * 1024) continue;
        const int key = c->prm.control | 0x200 | (a.ndy << 12);  // (the occupancy cache's control word: bit 9 = the pair kernel, bits 12+ its yaw rates)
        int nb = -1;
        for (const auto &e : c->grid_occ)
          if (e.control == key && e.pot == c->has_pot && e.lds == lds) nb = e.nb;
        if (nb < 0) {
          nb = mplx::pair_resident_blocks(c->dim, c->prm.control, a.ndy, lds);
          if (c->grid_occ.size() >= 8) c->grid_occ.clear();
          c->grid_occ.push_back({key, c->has_pot, lds, nb});
          if (getenv("MPLX_GRID_VERBOSE"))
            fprintf(stderr, "mplx: pair kernel control 0x%x rows/pass %d: LDS %zu B per workgroup, %d workgroups resident per CU\n",
                    c->prm.control, rm, lds, nb);
        }
        const int by_lds = (int)((160 * 1024) / lds);
        per_cu = (nb > 0 && nb < by_lds) ? nb : by_lds;
        if (per_cu >= 1) { b.rmax = rm; break; }
      }
      if (per_cu >= 1) {
        const int cap = c->tune.pair_wg_per_cu > 0 ? c->tune.pair_wg_per_cu : 3;  // 3 waves per SIMD: what its registers allow
        if (per_cu > cap) per_cu = cap;
        b.grid_limit = c->n_cus * per_cu;
        b.work = nullptr;
        b.work_zero = nullptr;
        const hipError_t e = mplx::launch_expand_pair(c->dim, c->prm.control, b, c->stream);
        if (e != hipSuccess) return fail(c, MPLX_ERR_HIP, "expand_pair_kernel launch failed: %s", hipGetErrorString(e));
        c->last_grid_lex = false;
        c->last_grid_pair = true;
        paired = true;
      }
    }
    if (!paired)
      if (int rc = launch_grid(c, &a)) return rc;
    c->done_armed = a.done.flag != nullptr;
    if (a.yaw.amb) {
      mplx_ctx::YawPending p;
      p.kind = 0;
      p.g = a;
      p.g.live = nullptr;  // the fix pass walks its own node list
      p.g.live_n = nullptr;
      p.g.done = {};       // ... and is not the launch a caller waits for on the completion word
      c->yaw_pending.push_back(p);
    }
    c->last_route = MPLX_ROUTE_GRID;
    return MPLX_OK;
  }
  const TilePlan tp = (route == MPLX_ROUTE_AUTO || route == MPLX_ROUTE_TILE) ? plan_tile(c) : TilePlan();
  if (route == MPLX_ROUTE_TILE && !tp.ok)
    return fail(c, MPLX_ERR_STATE, "lists route TILE does not cover this configuration");
  if (tp.ok) {
    if (int rc = ensure_tables(c)) return rc;
    mplx::TileArgs a = tile_args(c, tp, d_nodes, n_nodes, node_stride, o);
    c->done_armed = false;
    if (c->want_done && c->tune.done_flag && c->done_host) {
      a.done.flag = c->done_host;
      a.done.count = (uint32_t *)c->done_count.p;
      a.done.seq = ++c->done_seq;
    }
    HIP_TRY(c, mplx::launch_expand_tile(c->dim, c->prm.control, a, c->stream));
    c->done_armed = a.done.flag != nullptr;
    c->last_route = MPLX_ROUTE_TILE;
    return MPLX_OK;
  }
  // dense kernel into scratch, chunk by chunk, then ordered compaction on the device
  c->done_armed = false;
  const int64_t max_chunk_slots = (int64_t)(256u << 20) / (F * 8 + 21);  // ~256 MiB of scratch
  int64_t chunk_nodes = max_chunk_slots / c->nU;
  if (chunk_nodes < 1) chunk_nodes = 1;
  if (chunk_nodes > n_nodes) chunk_nodes = n_nodes;
  const int64_t cs = chunk_nodes * c->nU;
  if (int rc = ensure(c, c->d_status, (size_t)cs)) return rc;
  if (int rc = ensure(c, c->d_cost, (size_t)cs * 8)) return rc;
  if (int rc = ensure(c, c->d_hash, (size_t)cs * 8)) return rc;
  if (int rc = ensure(c, c->d_state, (size_t)cs * 8 * F)) return rc;
  if (int rc = ensure(c, c->d_iters, (size_t)cs * 4)) return rc;
  for (int64_t k0 = 0; k0 < n_nodes; k0 += chunk_nodes) {
    const int64_t nk = (n_nodes - k0) < chunk_nodes ? (n_nodes - k0) : chunk_nodes;
    mplx_succ d{};
    d.status = (uint8_t *)c->d_status.p;
    d.cost = (double *)c->d_cost.p;
    d.hash = (uint64_t *)c->d_hash.p;
    d.state = o->state ? (double *)c->d_state.p : nullptr;
    d.state_stride = cs;
    d.iters = o->iters ? (int32_t *)c->d_iters.p : nullptr;
    mplx::ExpandArgs a = make_args(c, d_nodes + k0, nk, node_stride, &d);
    if (int rc = yaw_slot(c, &a.yaw)) return rc;
    HIP_TRY(c, mplx::launch_expand(c->dim, c->prm.control, a, c->stream));
    if (a.yaw.amb) {  // the scratch slots must be final before they are compacted: check this chunk now
      mplx_ctx::YawPending p;
      p.kind = 1;
      p.e = a;
      c->yaw_pending.push_back(p);
      if (int rc = mplx_detail::resolve_pending(c)) return rc;
    }
    mplx::CompactArgs ca{};
    ca.status = d.status; ca.cost = d.cost; ca.hash = d.hash; ca.state = d.state; ca.iters = d.iters;
    ca.chunk_slots = cs; ca.nU = c->nU; ca.n_fields = F;
    ca.node_offset = k0; ca.n_nodes_chunk = nk;
    ca.l_count = o->count; ca.l_action = o->action; ca.l_cost = o->cost; ca.l_hash = o->hash;
    ca.l_state = o->state; ca.l_stride = o->state_stride; ca.l_iters = o->iters;
    ca.l_nstride = o->node_stride ? o->node_stride : c->nU;
    HIP_TRY(c, mplx::launch_compact_lists(ca, c->stream));
  }
  if (o->heur || o->flags) {
    // the lane-per-pair kernel + compaction keeps no successor in registers at its list stores: the rows are made by
    // the stand-alone pass over the finished lists (post_kernel.hip; same values), which reads hash and state rows
    if (!o->hash || !o->state)
      return fail(c, MPLX_ERR_STATE, "heur / flags rows on the DENSE lists route need the hash and state rows as well");
    mplx::PostArgs pa{};
    pa.count = o->count;
    pa.hash = o->hash;
    pa.state = o->state;
    pa.sstride = o->state_stride;
    pa.n_nodes = n_nodes;
    pa.nstride = o->node_stride ? o->node_stride : c->nU;
    for (int i = 0; i < F; i++) pa.goal[i] = c->goal_fuse.goal[i];
    pa.goal_hash = c->goal_fuse.goal_hash;
    pa.w = c->goal_fuse.w; pa.v_max = c->goal_fuse.v_max;
    pa.tol_pos = c->goal_fuse.tol_pos; pa.tol_vel = c->goal_fuse.tol_vel; pa.tol_acc = c->goal_fuse.tol_acc; pa.tol_yaw = c->goal_fuse.tol_yaw;
    pa.heur = o->heur;
    pa.flags = o->flags;
    HIP_TRY(c, mplx::launch_post_lists(c->dim, pa, c->stream));
  }
  c->last_route = MPLX_ROUTE_DENSE;
  return MPLX_OK;
}

}  // namespace

namespace mplx_detail {
namespace {
// A failure after the pending list was swapped out: the ring still holds the counts and ids of launches that are no
// longer pending, and the next launches would start on those slots -- ids beyond their own n_nodes.  Leave no trace.
int resolve_failed(mplx_ctx *c, int rc) {
  c->yaw_pending.clear();
  if (c->yaw_any_host) *c->yaw_any_host = 0;
  if (c->yaw_ring.p) {
    (void)hipStreamSynchronize(c->stream);
    if (hipMemsetAsync(c->yaw_ring.p, 0, c->yaw_ring.cap, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)
      release(c->yaw_ring);  // re-allocated (and zeroed) by the next yaw_slot
  }
  return rc;
}
}  // namespace

int wait_small_launch(mplx_ctx *c) {
  if (!c->done_armed) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MPLX_OK;
  }
  c->done_armed = false;
  volatile uint64_t *f = c->done_host;
  const uint64_t want = c->done_seq;
  double t0 = 0;
  c->done_waits++;
  for (uint32_t spins = 1; *f != want; spins++) {
    __builtin_ia32_pause();
    if ((spins & 0xfffu) != 0) continue;
    const double now = mono_us();
    if (t0 == 0) t0 = now;
    if (now - t0 > 2e5) {  // 200 ms: the word did not come -- the stream decides, and this context stops asking for it
      c->done_timeouts++;
      c->tune.done_flag = 0;
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      (void)hipMemsetAsync(c->done_count.p, 0, 64, c->stream);
      return MPLX_OK;
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return MPLX_OK;
}

int resolve_pending(mplx_ctx *c, bool stream_is_idle) {
  if (c->yaw_pending.empty()) return MPLX_OK;
  MPLX_GUARD_BEGIN
  // several contexts of one process may sit on different GPUs: the fix pass allocates (yaw_ids, yaw_tab) and launches,
  // so the context's device must be the current one whatever entry point came through here
  if (stream_is_idle) {
    // (the caller has just seen the last launch of the stream finish: the common case below needs no runtime call)
    if (c->yaw_any_host && *(volatile int32_t *)c->yaw_any_host == 0) {
      c->yaw_pending.clear();
      return MPLX_OK;
    }
  }
  if (int rc = bind_device(c)) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (c->yaw_any_host && *(volatile int32_t *)c->yaw_any_host == 0) {
    // nothing was flagged by any launch since the last resolve (the common case: the kernels set this pinned word
    // themselves): the lists are final as they are, no copy, no second synchronisation
    c->yaw_pending.clear();
    return MPLX_OK;
  }
  if (c->yaw_any_host) *c->yaw_any_host = 0;
  const size_t np = c->yaw_pending.size(), slot = (size_t)(1 + kAmbCap);
  std::vector<int32_t> ring(np * slot);
  HIP_TRY(c, hipMemcpyAsync(ring.data(), c->yaw_ring.p, ring.size() * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  std::vector<mplx_ctx::YawPending> pend;
  pend.swap(c->yaw_pending);  // the fix passes launch without detection; nothing new becomes pending meanwhile
  bool any = false;
  for (size_t i = 0; i < np; i++) {
    const int32_t cnt = ring[i * slot];
    if (cnt <= 0) continue;
    any = true;
    const mplx_ctx::YawPending &p = pend[i];
    const int64_t n_all = p.kind == 0 ? p.g.n_nodes : p.e.n_nodes;
    std::vector<int32_t> ids;
    if (cnt <= kAmbCap) {
      ids.assign(ring.begin() + (long)(i * slot + 1), ring.begin() + (long)(i * slot + 1 + cnt));
      std::sort(ids.begin(), ids.end());
      ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    } else {  // more flagged nodes than the block records: re-check the whole launch
      ids.resize((size_t)n_all);
      for (int64_t k = 0; k < n_all; k++) ids[(size_t)k] = (int32_t)k;
    }
    c->yaw_flagged += (int64_t)ids.size();
    const int64_t chunk = 16384;
    for (int64_t k0 = 0; k0 < (int64_t)ids.size(); k0 += chunk) {
      const int64_t nk = std::min<int64_t>(chunk, (int64_t)ids.size() - k0);
      if (int rc = yaw_fix_pass(c, p, ids.data() + k0, nk)) return resolve_failed(c, rc);
    }
  }
  if (any && hipMemsetAsync(c->yaw_ring.p, 0, np * slot * 4, c->stream) != hipSuccess)
    return resolve_failed(c, fail(c, MPLX_ERR_HIP, "resolve_pending: clearing the detection ring failed"));
  return MPLX_OK;
  MPLX_GUARD_END(c)
}
// ---- the service: small synchronous batches through a resident kernel (expand_tile_kernel.hip, SERVICE MODE)
namespace {
struct ArenaLayout {  // one block: node rows, counts, then the list rows that were asked for, all sized for n_alloc nodes
  size_t o_count = 0, o_action = 0, o_cost = 0, o_hash = 0, o_iters = 0, o_heur = 0, o_flags = 0, o_state = 0, total = 0;
  int64_t n_alloc = 0, n_slots = 0;
};
enum : unsigned { kRowAction = 1, kRowCost = 2, kRowHash = 4, kRowIters = 8, kRowState = 16, kRowHeur = 32, kRowFlags = 64 };

unsigned rows_of(const mplx_succ_lists *o) {
  return (o->action ? kRowAction : 0u) | (o->cost ? kRowCost : 0u) | (o->hash ? kRowHash : 0u) |
         (o->iters ? kRowIters : 0u) | (o->state ? kRowState : 0u) | (o->heur ? kRowHeur : 0u) | (o->flags ? kRowFlags : 0u);
}

ArenaLayout arena_layout(int F, int64_t n_alloc, int64_t S, unsigned rows) {
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  ArenaLayout L;
  L.n_alloc = n_alloc;
  L.n_slots = n_alloc * S;
  L.o_count = up((size_t)F * n_alloc * 8);
  L.o_action = L.o_count + up((size_t)n_alloc * 4);
  L.o_cost = L.o_action + ((rows & kRowAction) ? up((size_t)L.n_slots * 4) : 0);
  L.o_hash = L.o_cost + ((rows & kRowCost) ? up((size_t)L.n_slots * 8) : 0);
  L.o_iters = L.o_hash + ((rows & kRowHash) ? up((size_t)L.n_slots * 8) : 0);
  L.o_heur = L.o_iters + ((rows & kRowIters) ? up((size_t)L.n_slots * 4) : 0);
  L.o_flags = L.o_heur + ((rows & kRowHeur) ? up((size_t)L.n_slots * 8) : 0);
  L.o_state = L.o_flags + ((rows & kRowFlags) ? up((size_t)L.n_slots) : 0);
  L.total = L.o_state + ((rows & kRowState) ? up((size_t)F * L.n_slots * 8) : 0);
  return L;
}

void arena_put_nodes(char *hb, const ArenaLayout &L, int F, const double *h_nodes, int64_t n_nodes, int64_t node_stride) {
  for (int f = 0; f < F; f++)
    std::memcpy(hb + (size_t)f * L.n_alloc * 8, h_nodes + (size_t)f * node_stride, (size_t)n_nodes * 8);
}

mplx_succ_lists arena_lists(char *b, const ArenaLayout &L, int64_t S, unsigned rows) {
  mplx_succ_lists d{};
  d.count = (int32_t *)(b + L.o_count);
  if (rows & kRowAction) d.action = (int32_t *)(b + L.o_action);
  if (rows & kRowCost) d.cost = (double *)(b + L.o_cost);
  if (rows & kRowHash) d.hash = (uint64_t *)(b + L.o_hash);
  if (rows & kRowIters) d.iters = (int32_t *)(b + L.o_iters);
  if (rows & kRowHeur) d.heur = (double *)(b + L.o_heur);
  if (rows & kRowFlags) d.flags = (uint8_t *)(b + L.o_flags);
  if (rows & kRowState) { d.state = (double *)(b + L.o_state); d.state_stride = L.n_slots; }
  d.node_stride = S;
  return d;
}

// the used prefix of every list, from the landing block into the caller's arrays
void arena_get_lists(const char *hb, const ArenaLayout &L, int F, int64_t S, int64_t n_nodes, const mplx_succ_lists *h_out) {
  const int32_t *cnt = (const int32_t *)(hb + L.o_count);
  std::memcpy(h_out->count, cnt, (size_t)n_nodes * 4);
  for (int64_t k = 0; k < n_nodes; k++) {
    const size_t m = (size_t)cnt[k], at = (size_t)k * (size_t)S;
    if (!m) continue;
    if (h_out->action) std::memcpy(h_out->action + at, hb + L.o_action + at * 4, m * 4);
    if (h_out->cost) std::memcpy(h_out->cost + at, hb + L.o_cost + at * 8, m * 8);
    if (h_out->hash) std::memcpy(h_out->hash + at, hb + L.o_hash + at * 8, m * 8);
    if (h_out->iters) std::memcpy(h_out->iters + at, hb + L.o_iters + at * 4, m * 4);
    if (h_out->heur) std::memcpy(h_out->heur + at, hb + L.o_heur + at * 8, m * 8);
    if (h_out->flags) std::memcpy(h_out->flags + at, hb + L.o_flags + at, m);
    if (h_out->state)
      for (int f = 0; f < F; f++)
        std::memcpy(h_out->state + (size_t)f * h_out->state_stride + at,
                    hb + L.o_state + ((size_t)f * L.n_slots + at) * 8, m * 8);
  }
}

// (Re)launch the resident kernel with the signature in c->svc (cap, S, rows); seq_served = the last request that
// has been answered.  The context's own stream is drained
// first: the resident kernel runs on a stream of its own and reads what earlier calls uploaded.
int svc_launch(mplx_ctx *c, const TilePlan &tp, uint32_t seq_served) {
  mplx_ctx::Service &sv = c->svc;
  const int F = 4 * c->dim + 2;
  HIP_TRY(c, hipSetDevice(c->device));
  if (!sv.stream) HIP_TRY(c, hipStreamCreateWithFlags(&sv.stream, hipStreamNonBlocking));
  if (!sv.mb) {
    HIP_TRY(c, hipHostMalloc((void **)&sv.mb, sizeof(mplx::SvcMailbox), hipHostMallocCoherent));
    std::memset(sv.mb, 0, sizeof(mplx::SvcMailbox));
  }
  if (int rc = ensure_tables(c)) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  const ArenaLayout L = arena_layout(F, sv.cap, sv.S, sv.rows);
  if (L.total > sv.block_cap) {
    if (sv.block) HIP_TRY(c, hipHostFree(sv.block));
    sv.block = nullptr;
    sv.block_cap = 0;
    HIP_TRY(c, hipHostMalloc((void **)&sv.block, L.total, hipHostMallocCoherent));
    sv.block_cap = L.total;
  }
  int64_t g = (sv.cap + tp.npb - 1) / tp.npb;
  if (g > tp.grid) g = tp.grid;  // every workgroup must be resident: each waits for the others
  if (g > 1024) g = 1024;
  sv.workgroups = (int)g;
  if (int rc = ensure(c, sv.dev, (size_t)(1 + g) * 8)) return rc;
  HIP_TRY(c, hipMemsetAsync(sv.dev.p, 0, (size_t)(1 + g) * 8, sv.stream));
  const mplx_succ_lists d = arena_lists(sv.block, L, sv.S, sv.rows);
  mplx::TileArgs a = tile_args(c, tp, (const double *)sv.block, sv.cap, sv.cap, &d);
  a.grid_limit = (int32_t)g;
  a.svc_mb = sv.mb;
  a.svc_dev = (uint64_t *)sv.dev.p;
  a.svc_seq0 = seq_served;  // the kernel waits for the request after this one
  a.svc_idle = (uint64_t)c->tune.service_idle_us * 100ull;  // ticks of the 100 MHz clock
  // every workgroup of the resident form waits for the others: the runtime's own occupancy figure has to cover the grid
  // (tp.grid is an LDS estimate; a register-limited instantiation or a smaller device would otherwise hang the handshake
  // until the 2 s give-up).  Fewer than asked for: this context serves its small batches with launches.
  const int resident = mplx::tile_service_resident_workgroups(c->dim, c->prm.control, a);
  if (resident > 0 && resident < (int)g) {
    sv.disabled = true;
    return MPLX_OK;
  }
  *(volatile uint32_t *)&sv.mb->quit = 0;
  *(volatile uint32_t *)&sv.mb->alive = 1;
  std::atomic_thread_fence(std::memory_order_seq_cst);
  HIP_TRY(c, mplx::launch_expand_tile(c->dim, c->prm.control, a, sv.stream));
  sv.running = true;
  sv.launches++;
  return MPLX_OK;
}
}  // namespace

int svc_stop(mplx_ctx *c) {
  mplx_ctx::Service &sv = c->svc;
  if (!sv.running) return MPLX_OK;
  sv.running = false;
  *(volatile uint32_t *)&sv.mb->quit = 1;  // the coordinator polls this word between requests
  std::atomic_thread_fence(std::memory_order_seq_cst);
  const hipError_t e = hipStreamSynchronize(sv.stream);
  *(volatile uint32_t *)&sv.mb->quit = 0;
  if (e != hipSuccess) {
    sv.disabled = true;
    return fail(c, MPLX_ERR_HIP, "the resident expansion kernel did not leave: %s", hipGetErrorString(e));
  }
  return MPLX_OK;
}

// One small batch through the resident kernel.  *handled = false: not this time (not eligible, or the service gave
// up) -- the caller runs the batch as a launch of its own.  h_out names the rows (and the list stride) wanted; the
// used prefixes are copied into its arrays, or, with `view`, left in the landing block and described there (row stride
// view->state_stride = capacity x list stride).
int svc_request(mplx_ctx *c, const double *h_nodes, int64_t n_nodes, int64_t node_stride, const mplx_succ_lists *h_out,
                bool *handled, mplx_succ_lists *view) {
  *handled = false;
  mplx_ctx::Service &sv = c->svc;
  if (!c->tune.service || sv.disabled) return MPLX_OK;
  const TilePlan tp = (c->lists_route == MPLX_ROUTE_AUTO || c->lists_route == MPLX_ROUTE_TILE) ? plan_tile(c) : TilePlan();
  // (more than 64 workgroups in the handshake cost more than they save: 256 nodes of the 729-control table, a
  // workgroup each, 88 us per request against 65 us as a launch; 64 nodes 30 against 41)
  constexpr int64_t kMaxWorkgroups = 64;
  if (!tp.ok || n_nodes > c->tune.service_max_nodes || (n_nodes + tp.npb - 1) / tp.npb > kMaxWorkgroups) {
    // not a search's batch: the row of such batches ends here
    sv.streak = 0;
    return MPLX_OK;
  }
  const int F = 4 * c->dim + 2;
  const int64_t S = h_out->node_stride ? h_out->node_stride : c->nU;
  const unsigned rows = rows_of(h_out);
  if (!sv.running || S != sv.S || (rows & ~sv.rows) != 0 || n_nodes > sv.cap) {
    if (!sv.running && ++sv.streak < 2) return MPLX_OK;  // a single call is not a search
    if (int rc = svc_stop(c)) return rc;
    int64_t cap = 64;
    while (cap < n_nodes) cap <<= 1;
    const size_t arena_max = c->tune.arena_kb > 0 ? (size_t)c->tune.arena_kb << 10 : (size_t)8 << 20;
    const unsigned want_rows = (S == sv.S) ? (rows | sv.rows) : rows;
    while (cap > n_nodes && (arena_layout(F, cap, S, want_rows).total > arena_max || (cap + tp.npb - 1) / tp.npb > kMaxWorkgroups)) cap >>= 1;
    if (cap < n_nodes) cap = n_nodes;
    if (arena_layout(F, cap, S, want_rows).total > arena_max) return MPLX_OK;
    sv.cap = cap;
    // (callers that alternate between two sets of rows -- a search's batches and single get_succ calls -- get the union,
    // not a restart per call)
    sv.rows = (S == sv.S) ? (rows | sv.rows) : rows;
    sv.S = S;
    if (sv.seq > 0xfffffff0u) {  // (the doorbell of the last request still carries the old number)
      sv.seq = 0;
      if (sv.mb) *(volatile uint64_t *)&sv.mb->doorbell = 0;
    }
    if (int rc = svc_launch(c, tp, sv.seq)) return rc;
    if (!sv.running) return MPLX_OK;  // (not resident on this device: see svc_launch)
  }
  const ArenaLayout L = arena_layout(F, sv.cap, sv.S, sv.rows);
  arena_put_nodes(sv.block, L, F, h_nodes, n_nodes, node_stride);
  const uint32_t seq = ++sv.seq;
  std::atomic_thread_fence(std::memory_order_release);
  *(volatile uint64_t *)&sv.mb->doorbell = ((uint64_t)seq << 32) | (uint64_t)(uint32_t)n_nodes;
  volatile uint64_t *done = &sv.mb->done;
  volatile uint32_t *alive = &sv.mb->alive;
  double t0 = 0;
  int relaunches = 0;
  for (uint32_t spins = 1; *done != (uint64_t)seq; spins++) {
    __builtin_ia32_pause();
    if ((spins & 0x3ffu) != 0) continue;
    const double now = mono_us();
    if (t0 == 0) t0 = now;
    if (*alive == 0 && *done != (uint64_t)seq && relaunches < 2) {
      // the kernel left (no request for service_idle_us) before it saw this one: the next one picks it up
      sv.running = false;
      relaunches++;
      if (int rc = svc_launch(c, tp, seq - 1)) return rc;
      if (!sv.running) return MPLX_OK;
    } else if (now - t0 > 2e6) {
      // no answer: give the batch to an ordinary launch and never try again in this context
      sv.failures++;
      (void)svc_stop(c);
      sv.disabled = true;
      return MPLX_OK;
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  if (view) *view = arena_lists(sv.block, L, sv.S, sv.rows);  // the lists where they landed (valid until the next call)
  else arena_get_lists(sv.block, L, F, sv.S, n_nodes, h_out);
  sv.requests++;
  c->last_route = MPLX_ROUTE_TILE;
  *handled = true;
  return MPLX_OK;
}

int ctx_ready(mplx_ctx *c) { return ready(c); }
int lists_on_device(mplx_ctx *c, const double *d_nodes, int64_t n_nodes, int64_t node_stride, const mplx_succ_lists *d) {
  return lists_device(c, d_nodes, n_nodes, node_stride, d);
}
}  // namespace mplx_detail

extern "C" {

int mplx_expand_lists_device(mplx_ctx *c, const double *d_nodes, int64_t n_nodes, int64_t node_stride,
                             const mplx_succ_lists *d_out) {
  if (!c) return MPLX_ERR_ARG;
  if (!d_out || !d_out->count || n_nodes < 0 || node_stride < n_nodes || (!d_nodes && n_nodes > 0))
    return fail(c, MPLX_ERR_ARG, "mplx_expand_lists_device: bad arguments");
  if (int rc = ready(c)) return rc;
  if (n_nodes == 0) return MPLX_OK;
  if (d_out->node_stride != 0 && d_out->node_stride < c->nU)
    return fail(c, MPLX_ERR_ARG, "mplx_expand_lists_device: node_stride %lld < nU %d", (long long)d_out->node_stride, c->nU);
  if (d_out->state && d_out->state_stride < n_nodes * (d_out->node_stride ? d_out->node_stride : c->nU))
    return fail(c, MPLX_ERR_ARG, "mplx_expand_lists_device: state_stride < n_nodes*node_stride");
  if (int rc = bind_device(c)) return rc;
  return lists_device(c, d_nodes, n_nodes, node_stride, d_out);
}

int mplx_expand_lists(mplx_ctx *c, const double *h_nodes, int64_t n_nodes, int64_t node_stride,
                      const mplx_succ_lists *h_out) {
  if (!c) return MPLX_ERR_ARG;
  if (!h_out || !h_out->count || n_nodes < 0 || node_stride < n_nodes || (!h_nodes && n_nodes > 0))
    return fail(c, MPLX_ERR_ARG, "mplx_expand_lists: bad arguments");
  if (int rc = ready(c)) return rc;
  if (n_nodes == 0) return MPLX_OK;
  const int F = 4 * c->dim + 2;
  if (h_out->node_stride != 0 && h_out->node_stride < c->nU)
    return fail(c, MPLX_ERR_ARG, "mplx_expand_lists: node_stride %lld < nU %d", (long long)h_out->node_stride, c->nU);
  const int64_t n_slots = n_nodes * (h_out->node_stride ? h_out->node_stride : c->nU);
  if (h_out->state && h_out->state_stride < n_slots)
    return fail(c, MPLX_ERR_ARG, "mplx_expand_lists: state_stride < n_nodes*node_stride");
  {
    // The batches of a search (a few nodes, the answer awaited before the next one is known) go through a kernel
    // that stays resident between them, from the second such call in a row: a mailbox round trip instead of launch +
    // synchronise (see expand_tile_kernel.hip, SERVICE MODE).  Any other call into the context ends it (bind_device).
    bool handled = false;
    if (int rc = mplx_detail::svc_request(c, h_nodes, n_nodes, node_stride, h_out, &handled, nullptr)) return rc;
    if (handled) return MPLX_OK;
    const int counted = c->svc.streak;  // (what svc_request made of it; bind_device resets it)
    if (int rc = bind_device(c)) return rc;
    c->svc.streak = counted;
  }
  {
    // Small batches (one get_succ, or the speculative batches of a search) are latency bound: nodes and every
    // output row live in ONE pinned host block that the kernel reads and writes itself over PCIe (only the used
    // list entries cross the link, while the kernel runs), so a call is the kernel and one synchronisation; the
    // used prefixes are then copied into the caller's arrays.  (A 2D 9-control get_succ: 25 us; with one pageable
    // copy per row 111 us, with one upload + one download through a device arena 28 us -- MPLX_ZERO_COPY=0.)
    const int64_t S = h_out->node_stride ? h_out->node_stride : c->nU;
    const unsigned rows = mplx_detail::rows_of(h_out);
    const mplx_detail::ArenaLayout L = mplx_detail::arena_layout(F, n_nodes, S, rows);
    const size_t total = L.total, o_count = L.o_count;
    const size_t arena_max = c->tune.arena_kb > 0 ? (size_t)c->tune.arena_kb << 10 : (size_t)8 << 20;
    if (total <= arena_max) {
      if (int rc = ensure(c, c->s_arena, total)) return rc;
      if (total > c->h_arena_cap) {
        if (c->h_arena) HIP_TRY(c, hipHostFree(c->h_arena));
        c->h_arena = nullptr;
        c->h_arena_cap = 0;
        HIP_TRY(c, hipHostMalloc(&c->h_arena, arena_max, hipHostMallocCoherent));  // read by the host while the kernel may still run (DoneSignal)
        c->h_arena_cap = arena_max;
      }
      char *hb = (char *)c->h_arena, *db = (char *)c->s_arena.p;
      mplx_detail::arena_put_nodes(hb, L, F, h_nodes, n_nodes, node_stride);
      const bool zero_copy = c->tune.zero_copy != 0;
      if (zero_copy) db = hb;  // the kernel reads the nodes from and writes the lists to the pinned host block itself
      else HIP_TRY(c, hipMemcpyAsync(db, hb, (size_t)F * n_nodes * 8, hipMemcpyHostToDevice, c->stream));
      const mplx_succ_lists d = mplx_detail::arena_lists(db, L, h_out->node_stride, rows);
      c->want_done = zero_copy;  // the kernel tells the host itself when the lists are in the block (DoneSignal)
      const int rc_launch = lists_device(c, (const double *)db, n_nodes, n_nodes, &d);
      c->want_done = false;
      if (rc_launch) return rc_launch;
      if (!zero_copy)
        HIP_TRY(c, hipMemcpyAsync(hb + o_count, db + o_count, total - o_count, hipMemcpyDeviceToHost, c->stream));
      if (int rc = mplx_detail::wait_small_launch(c)) return rc;
      if (!c->yaw_pending.empty()) {  // a fix pass of the yaw pinning rewrites lists on the device side
        if (int rc = resolve_pending(c, true)) return rc;
        if (!zero_copy) {
          HIP_TRY(c, hipMemcpyAsync(hb + o_count, db + o_count, total - o_count, hipMemcpyDeviceToHost, c->stream));
          HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
      }
      mplx_detail::arena_get_lists(hb, L, F, S, n_nodes, h_out);
      return MPLX_OK;
    }
  }
  if (h_out->heur || h_out->flags)
    return fail(c, MPLX_ERR_ARG, "mplx_expand_lists: the heur / flags rows come back through host pointers for batches of up to "
                                 "8 MiB of lists only (a search's); larger ones: mplx_expand_lists_device and a copy of the rows");
  if (int rc = ensure(c, c->s_nodes, (size_t)F * n_nodes * sizeof(double))) return rc;
  HIP_TRY(c, hipMemcpy2DAsync(c->s_nodes.p, (size_t)n_nodes * sizeof(double), h_nodes,
                              (size_t)node_stride * sizeof(double), (size_t)n_nodes * sizeof(double), F,
                              hipMemcpyHostToDevice, c->stream));
  mplx_succ_lists d{};
  if (int rc = ensure(c, c->s_count, (size_t)n_nodes * 4)) return rc;
  d.count = (int32_t *)c->s_count.p;
  if (h_out->action) { if (int rc = ensure(c, c->s_action, (size_t)n_slots * 4)) return rc; d.action = (int32_t *)c->s_action.p; }
  if (h_out->cost) { if (int rc = ensure(c, c->s_cost, (size_t)n_slots * 8)) return rc; d.cost = (double *)c->s_cost.p; }
  if (h_out->hash) { if (int rc = ensure(c, c->s_hash, (size_t)n_slots * 8)) return rc; d.hash = (uint64_t *)c->s_hash.p; }
  if (h_out->iters) { if (int rc = ensure(c, c->s_iters, (size_t)n_slots * 4)) return rc; d.iters = (int32_t *)c->s_iters.p; }
  if (h_out->state) {
    if (int rc = ensure(c, c->s_state, (size_t)F * n_slots * 8)) return rc;
    d.state = (double *)c->s_state.p;
    d.state_stride = n_slots;
  }
  d.node_stride = h_out->node_stride;
  if (int rc = lists_device(c, (const double *)c->s_nodes.p, n_nodes, n_nodes, &d)) return rc;
  // everything larger: only the used prefixes cross the link, packed on the device and pipelined through pinned
  // buffers (lists_copy_api.cpp)
  MPLX_GUARD_BEGIN
  if (int rc = resolve_pending(c)) return rc;
  return mplx_detail::copy_lists_to_host(c, d, h_out, n_nodes);
  MPLX_GUARD_END(c)
}

int mplx_get_succ(mplx_ctx *c, const double *node, double *succ, double *cost, int32_t *action,
                  int32_t *n_succ) {
  if (!c) return MPLX_ERR_ARG;
  if (!node || !succ || !cost || !action || !n_succ) return fail(c, MPLX_ERR_ARG, "mplx_get_succ: NULL argument");
  if (int rc = ready(c)) return rc;
  MPLX_GUARD_BEGIN
  const int F = 4 * c->dim + 2;
  const int nU = c->nU;
  c->h_state.resize((size_t)F * nU);
  int32_t count = 0;
  mplx_succ_lists o{};
  o.count = &count;
  o.action = action;
  o.cost = cost;
  o.state = c->h_state.data();
  o.state_stride = nU;
  if (int rc = mplx_expand_lists(c, node, 1, 1, &o)) return rc;
  for (int m = 0; m < count; m++)
    for (int f = 0; f < F; f++) succ[(size_t)m * F + f] = c->h_state[(size_t)f * nU + m];
  *n_succ = count;
  return MPLX_OK;
  MPLX_GUARD_END(c)
}

int mplx_device_alloc(mplx_ctx *c, size_t bytes, void **dptr) {
  if (!c) return MPLX_ERR_ARG;
  if (!dptr) return fail(c, MPLX_ERR_ARG, "mplx_device_alloc: NULL");
  *dptr = nullptr;
  if (int rc = bind_device(c)) return rc;
  HIP_TRY(c, hipMalloc(dptr, bytes ? bytes : 1));
  return MPLX_OK;
}

int mplx_device_free(mplx_ctx *c, void *dptr) {
  if (!c) return MPLX_ERR_ARG;
  if (!dptr) return MPLX_OK;
  if (int rc = bind_device(c)) return rc;
  if (int rc = resolve_pending(c)) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipFree(dptr));
  return MPLX_OK;
}

int mplx_memcpy_h2d(mplx_ctx *c, void *dst, const void *src, size_t bytes) {
  if (!c) return MPLX_ERR_ARG;
  if (int rc = bind_device(c)) return rc;
  // the write may land in a frontier a pending launch read: its yaw fix pass re-reads the nodes, so it runs first
  if (int rc = resolve_pending(c)) return rc;
  HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return MPLX_OK;
}

int mplx_memcpy_d2h(mplx_ctx *c, void *dst, const void *src, size_t bytes) {
  if (!c) return MPLX_ERR_ARG;
  if (int rc = bind_device(c)) return rc;
  if (int rc = resolve_pending(c)) return rc;
  HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return MPLX_OK;
}

int mplx_memset(mplx_ctx *c, void *dst, int value, size_t bytes) {
  if (!c) return MPLX_ERR_ARG;
  if (int rc = bind_device(c)) return rc;
  if (int rc = resolve_pending(c)) return rc;  // as mplx_memcpy_h2d
  HIP_TRY(c, hipMemsetAsync(dst, value, bytes, c->stream));
  return MPLX_OK;
}

int mplx_synchronize(mplx_ctx *c) {
  if (!c) return MPLX_ERR_ARG;
  if (int rc = bind_device(c)) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return resolve_pending(c);
}

int mplx_timer_begin(mplx_ctx *c) {
  if (!c) return MPLX_ERR_ARG;
  if (int rc = bind_device(c)) return rc;
  HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
  return MPLX_OK;
}

int mplx_timer_end(mplx_ctx *c, float *ms) {
  if (!c) return MPLX_ERR_ARG;
  if (!ms) return fail(c, MPLX_ERR_ARG, "mplx_timer_end: NULL");
  if (int rc = bind_device(c)) return rc;
  HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
  HIP_TRY(c, hipEventSynchronize(c->ev1));
  HIP_TRY(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
  return resolve_pending(c);  // (after the measurement: a fix pass of the yaw pinning is not part of the timed launches)
}

int mplx_selftest_math(mplx_ctx *c, int op, const double *a, const double *b, double *out, int64_t n) {
  if (!c) return MPLX_ERR_ARG;
  if (!a || !out || n < 0 || (op == 0 && !b) || op < 0 || op > 5)
    return fail(c, MPLX_ERR_ARG, "mplx_selftest_math: bad arguments");
  if (n == 0) return MPLX_OK;
  if (int rc = bind_device(c)) return rc;
  void *da = nullptr, *db = nullptr, *dout = nullptr;
  const size_t bytes = (size_t)n * sizeof(double);
  HIP_TRY(c, hipMalloc(&da, bytes));
  HIP_TRY(c, hipMalloc(&db, bytes));
  HIP_TRY(c, hipMalloc(&dout, bytes));
  HIP_TRY(c, hipMemcpyAsync(da, a, bytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(db, b ? b : a, bytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, mplx::launch_math_probe(op, (const double *)da, (const double *)db, (double *)dout, n, c->stream));
  HIP_TRY(c, hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  (void)hipFree(da);
  (void)hipFree(db);
  (void)hipFree(dout);
  return MPLX_OK;
}

int mplx_set_lists_route(mplx_ctx *c, int route) {
  if (!c) return MPLX_ERR_ARG;
  if (route < MPLX_ROUTE_AUTO || route > MPLX_ROUTE_GRID) return fail(c, MPLX_ERR_ARG, "mplx_set_lists_route: unknown route %d", route);
  c->lists_route = route;
  c->svc.streak = 0;
  return MPLX_OK;
}

int mplx_last_lists_route(const mplx_ctx *c) { return c ? c->last_route : MPLX_ERR_ARG; }
int mplx_last_grid_kernel(const mplx_ctx *c) {
  if (!c) return MPLX_ERR_ARG;
  if (c->last_route != MPLX_ROUTE_GRID) return MPLX_KERNEL_NONE;
  return c->last_grid_pair ? MPLX_KERNEL_PAIR : (c->last_grid_lex ? MPLX_KERNEL_LEX : MPLX_KERNEL_GRID);
}

int mplx_last_identity_form(const mplx_ctx *c) { return c ? c->last_identity_form : MPLX_ERR_ARG; }

int mplx_service(mplx_ctx *c, int mode, int64_t stats[4]) {
  if (!c) return MPLX_ERR_ARG;
  if (mode < -1 || mode > 1) return fail(c, MPLX_ERR_ARG, "mplx_service: mode must be -1, 0 or 1");
  if (mode >= 0) {
    c->tune.service = mode;
    c->svc.streak = 0;
    if (mode == 0) {
      if (int rc = mplx_detail::svc_stop(c)) return rc;
    }
  }
  if (stats) {
    stats[0] = c->svc.requests;
    stats[1] = c->svc.launches;
    stats[2] = c->svc.failures;
    stats[3] = c->svc.running ? 1 : 0;
  }
  return MPLX_OK;
}

int mplx_yaw_pin_stats(const mplx_ctx *c, int64_t *flagged_nodes, int64_t *fix_passes) {
  if (!c) return MPLX_ERR_ARG;
  if (flagged_nodes) *flagged_nodes = c->yaw_flagged;
  if (fix_passes) *fix_passes = c->yaw_fix_passes;
  return MPLX_OK;
}

int mplx_device_info(mplx_ctx *c, char *name, size_t cap, int32_t *compute_units) {
  if (!c) return MPLX_ERR_ARG;
  hipDeviceProp_t prop;
  HIP_TRY(c, hipGetDeviceProperties(&prop, c->device));
  if (name && cap) snprintf(name, cap, "%s (%s)", prop.name, prop.gcnArchName);
  if (compute_units) *compute_units = prop.multiProcessorCount;
  return MPLX_OK;
}

}  // extern "C"
