"""plan() wall time, GPU vs CPU (the second half of BASELINE.json's metric) on a 3D problem
large enough for batching to matter:

    python profiles/plan_bench.py [--edge 160] [--nu 9] [--eps 1.0]

  1. the reference's own MapPlanner<3>::plan on the host CPU (oracle/_ref/libmpl_ref_planner.so)
  2. the same reference planner with get_succ on the MI355X through include/mplx_env_map.hpp
  3. the engine's host A* with batched expansion (mplx_planner_*), several batch sizes
All must agree on success and trajectory cost.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import motion_primitive_library_amd as m  # noqa: E402
from oracle import oracle as O  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--edge", type=int, default=160)
ap.add_argument("--nu", type=int, default=9, help="control values per axis (|U| = nu^3)")
ap.add_argument("--eps", type=float, default=1.0)
ap.add_argument("--occ", type=float, default=0.08)
ap.add_argument("--skip-adapter", action="store_true")
args = ap.parse_args()

W = m.workloads
res, edge = 0.1, args.edge
grid = W.box_map([edge] * 3, res, args.occ, 4242, side_m=(0.5, 2.5))
flat = grid.ravel()
md, org = [edge] * 3, [0.0, 0.0, 0.0]
u_max = 2.0
vals = np.linspace(-u_max, u_max, args.nu)
U = W.grid_controls(vals, 3)


def free_near(p):
    c = np.array([int(x / res) for x in p])
    for r in range(0, 30):
        for d in np.ndindex(2 * r + 1, 2 * r + 1, 2 * r + 1):
            q = c + np.array(d) - r
            if np.all(q >= 0) and np.all(q < edge) and flat[q[0] + edge * (q[1] + edge * q[2])] == 0:
                return [(q[i] + 0.5) * res for i in range(3)]
    raise RuntimeError("no free cell")


start_p = free_near([1.0, 1.0, 1.0])
goal_p = free_near([edge * res - 1.0, edge * res - 1.2, edge * res - 1.5])
start = m.Waypoint(3, m.ACC, pos=start_p)
goal = m.Waypoint(3, m.ACC, pos=goal_p)
out = {"map": "%d^3 voxels, res %.2f, %.0f %% occupied" % (edge, res, 100 * np.mean(flat == 100)), "controls": int(U.shape[0]),
       "start": start_p, "goal": goal_p, "epsilon": args.eps, "v_max": 2.0, "a_max": u_max, "host_threads_used": 1}

oenv = O.Env(3, O.ACC, U, flat, md, org, res, v_max=2.0, a_max=u_max, dt=1.0)
t = time.time()
cpu = O.ref_plan(oenv, start.to_row(), goal.to_row(), use_gpu=False, epsilon=args.eps, reps=1)
out["reference_cpu"] = {"wall_ms": cpu["wall_ms"], "ok": cpu["ok"], "cost": cpu["cost"], "expansions": cpu["expansions"],
                        "closed": cpu["closed"]}
print("reference CPU plan: %.1f ms, %d expansions, cost %s" % (cpu["wall_ms"], cpu["expansions"], cpu["cost"]), file=sys.stderr)
if not args.skip_adapter:
    gpu = O.ref_plan(oenv, start.to_row(), goal.to_row(), use_gpu=True, epsilon=args.eps, reps=1)
    out["reference_planner_gpu_get_succ"] = {"wall_ms": gpu["wall_ms"], "ok": gpu["ok"], "cost": gpu["cost"],
                                             "expansions": gpu["expansions"]}
    assert gpu["ok"] == cpu["ok"] and gpu["cost"] == cpu["cost"] and gpu["expansions"] == cpu["expansions"]
    out["reference_planner_gpu_speculative"] = []
    for b in (16, 64, 256):
        r = O.ref_plan(oenv, start.to_row(), goal.to_row(), use_gpu=b, epsilon=args.eps, reps=1)
        out["reference_planner_gpu_speculative"].append({"batch": b, "wall_ms": r["wall_ms"], "launches": r["device_launches"],
                                                         "cost": r["cost"], "expansions": r["expansions"]})
        print("reference planner + adapter, speculative batch %d: %.1f ms, %d launches" % (b, r["wall_ms"], r["device_launches"]),
              file=sys.stderr)
        assert r["ok"] == cpu["ok"] and r["cost"] == cpu["cost"] and r["expansions"] == cpu["expansions"]

out["engine_batched"] = []
for batch in (1, 16, 64, 256, 1024):
    pl = m.MapPlanner(3, device=0)
    mu = m.MapUtil(3)
    mu.setMap(org, md, flat, res)
    pl.setMapUtil(mu)
    pl.setVmax(2.0)
    pl.setAmax(u_max)
    pl.setDt(1.0)
    pl.setU(U)
    pl.setEpsilon(args.eps)
    pl.setBatch(batch)
    pl.plan(start, goal)  # warm-up
    t0 = time.perf_counter()
    ok = pl.plan(start, goal)
    ms = (time.perf_counter() - t0) * 1e3
    s = pl.summary()
    pl.close()
    out["engine_batched"].append({"batch": batch, "wall_ms": ms, "ok": ok, "cost": s["cost"], "expansions": s["expansions"],
                                  "launches": s["device_launches"], "pairs": s["pairs"]})
    print("engine batch %d: %.1f ms, %d launches, %d pairs, cost %s" % (batch, ms, s["device_launches"], s["pairs"], s["cost"]),
          file=sys.stderr)
    assert ok == cpu["ok"] and s["cost"] == cpu["cost"]
best = min(out["engine_batched"], key=lambda r: r["wall_ms"])
out["speedup_best_vs_reference_cpu"] = cpu["wall_ms"] / best["wall_ms"]
print(json.dumps(out))
