"""TEST TOOL: times the engine's host search (csrc/host_planner.hpp) without a GPU -- the CPU oracle is the successor
provider (memoised, so from the second plan on the wall time is the search's own bookkeeping) -- and prints the
digests (cost, expansions, closed-set checksum, trajectory checksum) that must not change when the search is reworked.

    python tests/tools/host_plan_profile.py [--edge 120] [--batch 64] [--reps 3]
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from motion_primitive_library_amd import workloads as W  # noqa: E402
from oracle import oracle as O  # noqa: E402


class HpOut(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("ok", "expansions", "closed", "opened", "nodes", "launches", "spec_hits", "segments")] + \
               [(k, C.c_int64) for k in ("pairs", "memo_hits", "memo_miss")] + \
               [(k, C.c_double) for k in ("cost", "wall_ms", "provider_ms", "total_time")] + [("J", C.c_double * 4)] + \
               [("closed_checksum", C.c_uint64), ("traj_checksum", C.c_uint64)]


def build():
    so = os.path.join(HERE, "_host_plan_harness.so")
    src = os.path.join(HERE, "host_plan_harness.cpp")
    hdr = os.path.join(ROOT, "motion_primitive_library_amd", "csrc", "host_planner.hpp")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", so, src,
                        "-L" + os.path.join(ROOT, "oracle"), "-lmpl_oracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle")], check=True)
    lib = C.CDLL(so)
    lib.hp_create.restype = C.c_void_p
    lib.hp_create.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int]
    lib.hp_plan.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(HpOut)]
    lib.hp_destroy.argtypes = [C.c_void_p]
    return lib


def problem_3d(edge, nu=9, occ=0.08, v_max=2.0):
    """The 3D problem of profiles/plan_bench.py."""
    res = 0.1
    grid = W.box_map([edge] * 3, res, occ, 4242, side_m=(0.5, 2.5))
    flat = grid.ravel()
    U = W.grid_controls(np.linspace(-2.0, 2.0, nu), 3)

    def free_near(p):
        c = np.array([int(x / res) for x in p])
        for r in range(0, 30):
            for d in np.ndindex(2 * r + 1, 2 * r + 1, 2 * r + 1):
                q = c + np.array(d) - r
                if np.all(q >= 0) and np.all(q < edge) and flat[q[0] + edge * (q[1] + edge * q[2])] == 0:
                    return [(q[i] + 0.5) * res for i in range(3)]
        raise RuntimeError("no free cell")

    from motion_primitive_library_amd.env import Waypoint, ACC
    start = Waypoint(3, ACC, pos=free_near([1.0, 1.0, 1.0]))
    goal = Waypoint(3, ACC, pos=free_near([edge * res - 1.0, edge * res - 1.2, edge * res - 1.5]))
    env = O.Env(3, O.ACC, U, flat, [edge] * 3, [0.0, 0.0, 0.0], res, v_max=v_max, a_max=2.0, dt=1.0)
    return env, start.to_row(), goal.to_row()


def run(env, start, goal, batch=64, eps=1.0, reps=3, threads=8, tol_pos=0.5):
    lib = build()
    ce = env._c()
    h = lib.hp_create(C.addressof(ce), eps, tol_pos, batch, threads)
    s = np.ascontiguousarray(start, dtype=np.float64)
    g = np.ascontiguousarray(goal, dtype=np.float64)
    res = []
    for _ in range(reps):
        o = HpOut()
        rc = lib.hp_plan(h, s.ctypes.data, g.ctypes.data, C.byref(o))
        assert rc == 0, rc
        res.append({k: getattr(o, k) for k, _ in HpOut._fields_ if k != "J"} | {"J": list(o.J)})
    lib.hp_destroy(h)
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--edge", type=int, default=120)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--eps", type=float, default=1.0)
    a = ap.parse_args()
    env, s, g = problem_3d(a.edge)
    for r in run(env, s, g, a.batch, a.eps, a.reps):
        r["search_ms"] = r["wall_ms"] - r["provider_ms"]
        print(json.dumps(r))
