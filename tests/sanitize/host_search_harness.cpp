// host_search_harness.cpp -- the engine's HOST search (csrc/host_planner.hpp, csrc/host_lpastar.hpp) under
// AddressSanitizer / UBSan / _GLIBCXX_ASSERTIONS with the CPU oracle as successor provider (no GPU: sanitizers are for the
// CPU build only).  Built and run by tests/test_host_sanitizers.py:
//     g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -D_GLIBCXX_ASSERTIONS
//         tests/sanitize/host_search_harness.cpp oracle/mpl_oracle.cpp -pthread
// Scenarios (a random box world, 2D ACC, 9 controls): A* at batch 1 and 16; LPA*: plan, getLinkedNodes, a box of cells on
// the trajectory blocked + updateBlockedNodes, getSubStateSpace(k), plan from way point k (the sequence that corrupted
// memory before round 6), cleared + updateClearedNodes, plan; a search guided by a prior trajectory, without and with
// a potential map.  Exit code 0 = no sanitizer report and the plans agree where they must.
#include "../../motion_primitive_library_amd/csrc/host_planner.hpp"
#include "../../motion_primitive_library_amd/csrc/host_lpastar.hpp"
#include "../../oracle/mpl_oracle.h"

#include <cstdio>
#include <random>

using namespace mplx::host;

namespace {

struct World {
  int nx = 240, ny = 90;
  double res = 0.1;
  std::vector<int8_t> cells;
  std::vector<double> U;
  mpl_oracle_env env{};
};

void make_world(World &w, unsigned seed) {
  w.cells.assign((size_t)w.nx * w.ny, 0);
  std::mt19937 rng(seed);
  for (int b = 0; b < 34; b++) {
    const int bx = 12 + (int)(rng() % (unsigned)(w.nx - 30)), by = (int)(rng() % (unsigned)w.ny);
    const int sx = 2 + (int)(rng() % 7u), sy = 4 + (int)(rng() % 22u);
    for (int y = by; y < by + sy && y < w.ny; y++)
      for (int x = bx; x < bx + sx && x < w.nx; x++) w.cells[(size_t)y * w.nx + x] = 100;
  }
  for (double a : {-0.5, 0.0, 0.5})
    for (double b : {-0.5, 0.0, 0.5}) { w.U.push_back(a); w.U.push_back(b); }
  mpl_oracle_env &e = w.env;
  e.dim = 2;
  e.control = 0x03;
  e.dt = 1.0; e.w = 10.0; e.wyaw = 1.0;
  e.v_max = 1.0; e.a_max = 1.0; e.j_max = -1.0; e.yaw_max = -1.0;
  e.potential_weight = 0.1; e.gradient_weight = 0.0;
  e.map_dim[0] = w.nx; e.map_dim[1] = w.ny; e.map_dim[2] = 1;
  e.origin[0] = e.origin[1] = e.origin[2] = 0.0;
  e.res = w.res;
  e.map = w.cells.data();
  e.potential = nullptr;
  e.region = nullptr;
  e.U = w.U.data();
  e.nU = 9;
  e.udim = 2;
}

void configure(Planner &P, World &w, int batch) {
  P.dim = 2;
  P.control = 0x03;
  P.dt = 1.0; P.w = 10.0; P.v_max = 1.0; P.eps = 1.0;
  P.tol_pos = 0.5;
  P.batch = batch;
  P.U = w.U;
  P.nU = 9;
  P.udim = 2;
  P.grid.dim = 2;
  P.grid.n[0] = w.nx; P.grid.n[1] = w.ny; P.grid.n[2] = 1;
  P.grid.origin[0] = P.grid.origin[1] = P.grid.origin[2] = 0.0;
  P.grid.res = w.res;
  P.grid.cells = w.cells;
  P.single = mpl_oracle_get_succ;
  P.batched = mpl_oracle_batch;
  P.user = &w.env;
}

int edges_from_oracle(void *user, const double *parents, const int32_t *actions, int64_t n, uint8_t *free_flag, double *cost,
                      int32_t *cells, int32_t *cell_count, int32_t cell_cap) {
  return mpl_oracle_check_edges((const mpl_oracle_env *)user, parents, actions, n, free_flag, cost, cells, cell_count, cell_cap);
}

int pot_from_host(void *user, const int64_t *idx, int64_t n, int8_t *out) {
  const std::vector<int8_t> *p = (const std::vector<int8_t> *)user;
  for (int64_t i = 0; i < n; i++) out[i] = (*p)[(size_t)idx[i]];
  return 0;
}

#define CHECK(cond, msg) do { if (!(cond)) { std::fprintf(stderr, "harness: %s (line %d)\n", msg, __LINE__); return 1; } } while (0)

}  // namespace

int main() {
  World w;
  make_world(w, 20260101u);
  const double start[10] = {0.55, 4.55, 0, 0, 0, 0, 0, 0, 0, 0};
  const double goal[10] = {23.25, 4.45, 0, 0, 0, 0, 0, 0, 0, 0};

  // ---- A*, one node per provider call and sixteen
  double cost1 = 0;
  int closed1 = 0;
  for (int batch : {1, 16}) {
    Planner P;
    configure(P, w, batch);
    CHECK(P.plan(start, goal) == 0, "A* provider failed");
    CHECK(P.last.ok, "A* found no trajectory in the harness world");
    if (batch == 1) { cost1 = P.last.cost; closed1 = P.last.closed; }
    else CHECK(P.last.cost == cost1 && P.last.closed == closed1, "A* batch 16 differs from batch 1");
    std::printf("A* batch %d: cost %.3f, %d expansions, %d closed, %d launches\n", batch, P.last.cost, P.last.expansions, P.last.closed, P.last.device_launches);
  }

  // ---- LPA*: map edit combined with re-rooting, then clearing
  {
    Planner P;
    configure(P, w, 16);
    LpaPlanner L;
    L.cfg = &P;
    L.edges = edges_from_oracle;
    L.edges_user = &w.env;
    CHECK(L.plan(start, goal) == 0 && L.last.ok, "LPA* first plan");
    CHECK(L.last.cost == cost1, "LPA* first plan differs from A*");
    const std::vector<double> traj = L.last.traj_nodes;  // [segments][10]
    const int segs = (int)L.last.traj_actions.size();
    int64_t n_pts = 0;
    CHECK(L.linked_nodes(nullptr, &n_pts) == 0 && n_pts > 0, "getLinkedNodes");
    for (int half : {1, 3, 5}) {
      // block a box around the middle of the trajectory (free cells only, away from the new root and the goal)
      const double *mid = &traj[(size_t)(segs / 2) * 10];
      const int cx = (int)std::round(mid[0] / w.res - 0.5), cy = (int)std::round(mid[1] / w.res - 0.5);
      std::vector<int32_t> edit;
      for (int y = cy - half; y <= cy + half; y++)
        for (int x = cx - half; x <= cx + half; x++)
          if (x >= 0 && y >= 0 && x < w.nx && y < w.ny && w.cells[(size_t)y * w.nx + x] == 0) { edit.push_back(x); edit.push_back(y); }
      for (size_t i = 0; i < edit.size(); i += 2) {
        w.cells[(size_t)edit[i + 1] * w.nx + edit[i]] = 100;
        P.grid.cells[(size_t)edit[i + 1] * w.nx + edit[i]] = 100;
      }
      CHECK(L.update_blocked(edit.data(), (int64_t)edit.size() / 2) == 0, "updateBlockedNodes");
      const int k = 3;
      CHECK(L.sub_state_space(k) == 0, "getSubStateSpace");
      CHECK(L.plan(&traj[(size_t)k * 10], goal) == 0, "LPA* plan after edit + re-rooting");
      std::printf("LPA* box %d: edit %zu cells, re-rooted at %d: ok %d cost %.3f, %d expansions\n", 2 * half + 1, edit.size() / 2, k, (int)L.last.ok,
                  L.last.cost, L.last.expansions);
      CHECK(L.linked_nodes(nullptr, &n_pts) == 0, "getLinkedNodes after the re-plan");
      for (size_t i = 0; i < edit.size(); i += 2) {
        w.cells[(size_t)edit[i + 1] * w.nx + edit[i]] = 0;
        P.grid.cells[(size_t)edit[i + 1] * w.nx + edit[i]] = 0;
      }
      CHECK(L.update_cleared(edit.data(), (int64_t)edit.size() / 2) == 0, "updateClearedNodes");
      CHECK(L.plan(&traj[(size_t)k * 10], goal) == 0 && L.last.ok, "LPA* plan after clearing");
      // ... and from scratch for the next box
      L.reset();
      CHECK(L.plan(start, goal) == 0 && L.last.ok && L.last.cost == cost1, "LPA* plan after reset");
      CHECK(L.linked_nodes(nullptr, &n_pts) == 0, "getLinkedNodes after reset");
    }
  }

  // ---- prior trajectory: without and with a potential map (values from a host copy)
  {
    Planner first;
    configure(first, w, 16);
    CHECK(first.plan(start, goal) == 0 && first.last.ok, "prior: first plan");
    std::vector<int8_t> pot = w.cells;  // a crude potential field: a halo of 40 around every obstacle cell
    for (int y = 1; y + 1 < w.ny; y++)
      for (int x = 1; x + 1 < w.nx; x++)
        if (w.cells[(size_t)y * w.nx + x] == 0)
          for (int d = 0; d < 4; d++) {
            const int xx = x + (d == 0) - (d == 1), yy = y + (d == 2) - (d == 3);
            if (w.cells[(size_t)yy * w.nx + xx] == 100) pot[(size_t)y * w.nx + x] = 40;
          }
    for (int with_pot = 0; with_pot < 2; with_pot++) {
      Planner second;
      configure(second, w, 16);
      const int rc = second.set_prior_trajectory(first.last.traj_nodes.data(), first.last.traj_actions.data(), (int)first.last.traj_actions.size(),
                                                 first.control, first.U.data(), first.udim, first.dt, with_pot ? pot_from_host : nullptr,
                                                 with_pot ? (void *)&pot : nullptr, 0.5, 0.25);
      CHECK(rc == 0, "set_prior_trajectory");
      CHECK(second.plan(start, goal) == 0, "prior: guided plan");
      std::printf("prior trajectory%s: ok %d cost %.3f, %d expansions\n", with_pot ? " + potential" : "", (int)second.last.ok, second.last.cost,
                  second.last.expansions);
    }
  }
  std::printf("harness: ok\n");
  return 0;
}
