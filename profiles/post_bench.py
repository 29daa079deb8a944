"""Timing of the next-row kernels at BASELINE sizes (not the headline metric; recorded in profiles/README.md):
post-processing of C4's successor lists, map preprocessing at 256^3 / 512^3, edge re-validation."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C  # noqa: E402

import motion_primitive_library_amd as m  # noqa: E402
from motion_primitive_library_amd import _abi  # noqa: E402

out = {}
wl = m.workloads.make("C4")
env = m.EnvMap(3)
wl.apply(env)
fr = env.upload_frontier(wl.nodes)
lists = env.alloc_lists(wl.n_nodes, want_state=True)
env.expand_lists_resident(fr, lists)
env.synchronize()
n_emit = int(lists.count.download(np.int32, (wl.n_nodes,)).sum(dtype=np.int64))
goal = wl.nodes[:, 0].copy()
L = _abi.lib()
ns = lists.n_slots
heur = m.env.DeviceArray(env, ns * 8)
flags = m.env.DeviceArray(env, ns)
canon = m.env.DeviceArray(env, ns * 4)
g = _abi.GoalSpec()
g.goal, g.control, g.w, g.v_max = goal.ctypes.data, wl.control, 10.0, 2.0
g.tol_pos, g.tol_vel, g.tol_acc, g.tol_yaw = 0.5, -1.0, -1.0, -1.0
s = lists.c_struct()
for name, want_canon in (("heuristic + goal flags", False), ("heuristic + goal flags + node identity", True)):
    o = _abi.Post()
    o.heur, o.flags, o.canon = heur.ptr, flags.ptr, canon.ptr if want_canon else None
    for _ in range(2):
        _abi.check(env._ctx, L.mplx_post_lists_device(env._ctx, C.byref(s), wl.n_nodes, C.byref(g), C.byref(o)))
    env.synchronize()
    env.timer_begin()
    reps = 10
    for _ in range(reps):
        _abi.check(env._ctx, L.mplx_post_lists_device(env._ctx, C.byref(s), wl.n_nodes, C.byref(g), C.byref(o)))
    ms = env.timer_end() / reps
    out["post C4: " + name] = {"ms": ms, "successors": n_emit, "G successors/s": n_emit / ms / 1e6}
first = int(np.count_nonzero(flags.download(np.uint8, (ns,)) & 4))
out["post C4: unique lattice states"] = first
# the same on the frontier with realistic locality (open list of a search): how many successors are FIRST occurrences,
# i.e. what a fused "state only for first occurrences" store mode could save
wf = m.workloads.wavefront_frontier(wl, wl.n_nodes, 0)
fr2 = env.upload_frontier(wf)
env.expand_lists_resident(fr2, lists)
env.synchronize()
n_emit_wf = int(lists.count.download(np.int32, (wl.n_nodes,)).sum(dtype=np.int64))
o = _abi.Post()
o.heur, o.flags, o.canon = heur.ptr, flags.ptr, canon.ptr
_abi.check(env._ctx, L.mplx_memset(env._ctx, flags.ptr, 0, ns))
_abi.check(env._ctx, L.mplx_post_lists_device(env._ctx, C.byref(s), wl.n_nodes, C.byref(g), C.byref(o)))
env.synchronize()
first_wf = int(np.count_nonzero(flags.download(np.uint8, (ns,)) & 4))
out["post C4 wavefront frontier"] = {"successors": n_emit_wf, "unique lattice states": first_wf,
                                     "first-occurrence share": first_wf / max(n_emit_wf, 1)}
env.close()

for edge in (256, 512):
    grid = m.workloads.box_map([edge] * 3, 0.1, 0.15, 1005)
    env = m.EnvMap(3)
    env.setMap([0, 0, 0], [edge] * 3, grid, 0.1)
    t0 = time.perf_counter()
    env.updatePotentialMap([0, 0, 0], [1.0, 1.0, 1.0])
    t_first = time.perf_counter() - t0
    env.setMap([0, 0, 0], [edge] * 3, grid, 0.1)
    p = (C.c_double * 3)(0, 0, 0)
    r = (C.c_double * 3)(1.0, 1.0, 1.0)
    t0 = time.perf_counter()
    _abi.check(env._ctx, L.mplx_update_potential_map(env._ctx, p, r, None, 1.0, None))  # device only, no read-back
    t_dev = time.perf_counter() - t0
    path = np.array([[1.0, 1.0, 1.0], [edge * 0.1 - 1.0] * 3])
    sr = (C.c_double * 3)(0.5, 0.5, 0.5)
    t0 = time.perf_counter()
    _abi.check(env._ctx, L.mplx_set_search_region_path(env._ctx, path.ctypes.data, 2, 0, sr, None))
    t_reg = time.perf_counter() - t0
    out["map prep %d^3" % edge] = {"updatePotentialMap radius 1 m (10 cells), incl. read-back ms": t_first * 1e3,
                                   "updatePotentialMap device only ms": t_dev * 1e3, "setSearchRegion tunnel ms": t_reg * 1e3}
    env.close()
print(json.dumps(out, indent=1))
