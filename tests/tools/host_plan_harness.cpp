// host_plan_harness.cpp -- TEST TOOL: the engine's host search (csrc/host_planner.hpp) driven by the CPU oracle as a
// "packed" successor provider, so that the search's own bookkeeping can be timed and checked where there is no GPU.
// The provider memoises lists per node state: from the second plan() of a process on it costs a look-up, and what
// is left of the wall time is the search.  Never part of the product (it links oracle/libmpl_oracle.so).
#include "../../motion_primitive_library_amd/csrc/host_planner.hpp"
#include "../../oracle/mpl_oracle.h"

#include <chrono>
#include <cstring>
#include <unordered_map>
#include <vector>

namespace {
using namespace mplx::host;

struct Memo {
  int32_t m;
  std::vector<double> cost;
  std::vector<uint64_t> hash;
  std::vector<int32_t> act;
};

struct Harness {
  const mpl_oracle_env *env = nullptr;
  Planner pl;
  int threads = 8;
  double provider_ms = 0;
  int64_t memo_hits = 0, memo_miss = 0;
  std::unordered_map<uint64_t, std::vector<std::pair<std::vector<double>, Memo>>> memo;  // key: lattice hash; exact state inside
  // landing buffer
  std::vector<int32_t> count, action;
  std::vector<int64_t> offs;
  std::vector<double> cost;
  std::vector<uint64_t> hash;
  std::vector<uint8_t> st;
  std::vector<double> dcost;
  std::vector<uint64_t> dhash;
};

int provider(void *user, const double *nodes, int64_t n, PackedView *out) {
  Harness *h = (Harness *)user;
  const auto t0 = std::chrono::steady_clock::now();
  const int nU = h->env->nU, f = 4 * h->env->dim + 2;
  const int64_t S = (nU + 31) & ~31;
  h->count.assign((size_t)n, 0);
  h->offs.resize((size_t)n + 1);
  h->action.resize((size_t)(n * S));
  h->cost.resize((size_t)(n * S));
  h->hash.resize((size_t)(n * S));
  // which nodes are not in the memo
  std::vector<int64_t> todo;
  std::vector<const Memo *> found((size_t)n, nullptr);
  std::vector<double> row((size_t)f);
  for (int64_t k = 0; k < n; k++) {
    for (int r = 0; r < f; r++) row[(size_t)r] = nodes[(size_t)r * n + k];
    const uint64_t key = mpl_oracle_hash(h->env->dim, h->env->control, row.data());
    auto it = h->memo.find(key);
    if (it != h->memo.end())
      for (auto &e : it->second)
        if (std::memcmp(e.first.data(), row.data(), sizeof(double) * (size_t)f) == 0) found[(size_t)k] = &e.second;
    if (!found[(size_t)k]) todo.push_back(k);
  }
  if (!todo.empty()) {
    const int64_t nt = (int64_t)todo.size();
    std::vector<double> sub((size_t)f * nt);
    for (int64_t j = 0; j < nt; j++)
      for (int r = 0; r < f; r++) sub[(size_t)r * nt + j] = nodes[(size_t)r * n + todo[(size_t)j]];
    h->st.resize((size_t)(nt * nU));
    h->dcost.resize((size_t)(nt * nU));
    h->dhash.resize((size_t)(nt * nU));
    mpl_oracle_out o{};
    o.status = h->st.data();
    o.cost = h->dcost.data();
    o.hash = h->dhash.data();
    if (int rc = mpl_oracle_expand(h->env, sub.data(), nt, &o, h->threads, nullptr)) return rc;
    for (int64_t j = 0; j < nt; j++) {
      Memo mm;
      for (int i = 0; i < nU; i++) {
        const size_t sl = (size_t)(j * nU + i);
        if (h->st[sl] != 1 && h->st[sl] != 2) continue;
        mm.cost.push_back(h->dcost[sl]);
        mm.hash.push_back(h->dhash[sl]);
        mm.act.push_back(i);
      }
      mm.m = (int32_t)mm.act.size();
      for (int r = 0; r < f; r++) row[(size_t)r] = sub[(size_t)r * nt + j];
      const uint64_t key = mpl_oracle_hash(h->env->dim, h->env->control, row.data());
      auto &vec = h->memo[key];
      vec.emplace_back(row, std::move(mm));
    }
    h->memo_miss += nt;
    // (pointers into memo vectors may have moved: look everything up again)
    for (int64_t k = 0; k < n; k++) {
      for (int r = 0; r < f; r++) row[(size_t)r] = nodes[(size_t)r * n + k];
      const uint64_t key = mpl_oracle_hash(h->env->dim, h->env->control, row.data());
      for (auto &e : h->memo[key])
        if (std::memcmp(e.first.data(), row.data(), sizeof(double) * (size_t)f) == 0) found[(size_t)k] = &e.second;
    }
  }
  h->memo_hits += n - (int64_t)todo.size();
  for (int64_t k = 0; k < n; k++) {
    const Memo &mm = *found[(size_t)k];
    h->count[(size_t)k] = mm.m;
    h->offs[(size_t)k] = k * S;
    std::memcpy(&h->cost[(size_t)(k * S)], mm.cost.data(), (size_t)mm.m * 8);
    std::memcpy(&h->hash[(size_t)(k * S)], mm.hash.data(), (size_t)mm.m * 8);
    std::memcpy(&h->action[(size_t)(k * S)], mm.act.data(), (size_t)mm.m * 4);
  }
  h->offs[(size_t)n] = n * S;
  out->total = n * S;
  out->count = h->count.data();
  out->offs = h->offs.data();
  out->cost = h->cost.data();
  out->hash = h->hash.data();
  out->action = h->action.data();
  out->state = nullptr;
  h->provider_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return 0;
}
}  // namespace

extern "C" {

struct hp_out {
  int32_t ok, expansions, closed, opened, nodes, launches, spec_hits, segments;
  int64_t pairs, memo_hits, memo_miss;
  double cost, wall_ms, provider_ms, total_time;
  double J[4];
  uint64_t closed_checksum;  // order-independent checksum of the closed set's keys
  uint64_t traj_checksum;
};

void *hp_create(const mpl_oracle_env *env, double eps, double tol_pos, int batch, int threads) {
  Harness *h = new Harness();
  h->env = env;
  h->threads = threads;
  Planner &pl = h->pl;
  pl.dim = env->dim;
  pl.control = env->control;
  pl.dt = env->dt;
  pl.w = env->w;
  pl.v_max = env->v_max;
  pl.eps = eps;
  pl.tol_pos = tol_pos;
  pl.batch = batch;
  pl.U.assign(env->U, env->U + (size_t)env->nU * env->udim);
  pl.nU = env->nU;
  pl.udim = env->udim;
  pl.grid.dim = env->dim;
  size_t n = 1;
  for (int i = 0; i < env->dim; i++) {
    pl.grid.n[i] = env->map_dim[i];
    pl.grid.origin[i] = env->origin[i];
    n *= (size_t)env->map_dim[i];
  }
  pl.grid.res = env->res;
  pl.grid.cells.assign(env->map, env->map + n);
  pl.packed = provider;
  pl.edges_only = true;
  pl.user = h;
  return h;
}

int hp_plan(void *hv, const double *start, const double *goal, hp_out *out) {
  Harness *h = (Harness *)hv;
  h->provider_ms = 0;
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = h->pl.plan(start, goal);
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  const PlanResult &r = h->pl.last;
  std::memset(out, 0, sizeof(*out));
  out->ok = r.ok;
  out->expansions = r.expansions;
  out->closed = r.closed;
  out->opened = r.opened;
  out->nodes = r.nodes;
  out->launches = r.device_launches;
  out->spec_hits = r.spec_hits;
  out->segments = (int32_t)r.traj_actions.size();
  out->pairs = r.pairs;
  out->memo_hits = h->memo_hits;
  out->memo_miss = h->memo_miss;
  out->cost = r.cost;
  out->wall_ms = ms;
  out->provider_ms = h->provider_ms;
  out->total_time = r.total_time;
  for (int i = 0; i < 4; i++) out->J[i] = r.J[i];
  out->closed_checksum = h->pl.closed_checksum();
  uint64_t tc = 0;
  for (size_t i = 0; i < r.traj_actions.size(); i++) tc = tc * 1000003ull + (uint64_t)r.traj_actions[i] + 1;
  for (double v : r.traj_nodes) { uint64_t b; std::memcpy(&b, &v, 8); tc = tc * 1000003ull + b; }
  out->traj_checksum = tc;
  return rc;
}

void hp_destroy(void *hv) { delete (Harness *)hv; }
}
