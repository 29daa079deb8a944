"""Node identity (mplx_post_lists_device with canon) on C4's lists: the table in HBM (post_kernel.hip) against the radix
partition + LDS tables (identity_kernel.hip), random frontier (99.998 % first occurrences) and wavefront frontier (6.7 %).

    python profiles/micro/identity_bench.py [out.json]
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import motion_primitive_library_amd as m  # noqa: E402
from motion_primitive_library_amd import _abi  # noqa: E402

out = {}
wl = m.workloads.make("C4", n_nodes=int(os.environ.get("ID_BENCH_NODES", "65536")))
env = m.EnvMap(3)
wl.apply(env)
L = _abi.lib()
lists = env.alloc_lists(wl.n_nodes, want_state=True)
ns = lists.n_slots
heur = m.env.DeviceArray(env, ns * 8)
flags = m.env.DeviceArray(env, ns)
canon = m.env.DeviceArray(env, ns * 4)
goal = wl.nodes[:, 0].copy()
g = _abi.GoalSpec()
g.goal, g.control, g.w, g.v_max = goal.ctypes.data, wl.control, 10.0, 2.0
g.tol_pos, g.tol_vel, g.tol_acc, g.tol_yaw = 0.5, -1.0, -1.0, -1.0
s = lists.c_struct()


def run(want_canon, reps=10):
    o = _abi.Post()
    o.heur, o.flags, o.canon = heur.ptr, flags.ptr, canon.ptr if want_canon else None
    for _ in range(2):
        _abi.check(env._ctx, L.mplx_post_lists_device(env._ctx, C.byref(s), wl.n_nodes, C.byref(g), C.byref(o)))
    env.synchronize()
    env.timer_begin()
    for _ in range(reps):
        _abi.check(env._ctx, L.mplx_post_lists_device(env._ctx, C.byref(s), wl.n_nodes, C.byref(g), C.byref(o)))
    return env.timer_end() / reps


legs = (("random", wl.nodes), ("wavefront", None))
if os.environ.get("ID_BENCH_RANDOM_ONLY"):
    legs = legs[:1]
for label, nodes in legs:
    if nodes is None:
        nodes = m.workloads.wavefront_frontier(wl, wl.n_nodes, 0)
    fr = env.upload_frontier(nodes)
    env.expand_lists_resident(fr, lists)
    env.synchronize()
    n_emit = int(lists.count.download(np.int32, (wl.n_nodes,)).sum(dtype=np.int64))
    rec = {"successors": n_emit, "list_slots": ns}
    rec["heur_flags_only_ms"] = run(False)
    ref = None
    routes = (("table_in_hbm", "1000000000", "1"), ("partition_exact", "0", "0"), ("partition_claimed", "0", "1"))
    if os.environ.get("ID_BENCH_SKIP_TABLE"):
        routes = routes[1:]
    for route, env_min, claimed in routes:
        os.environ["MPLX_POST_PARTITION_MIN"] = env_min
        os.environ["MPLX_POST_CLAIMED"] = claimed
        ms = run(True)
        c = canon.download(np.int32, (ns,))
        cnt = lists.count.download(np.int32, (wl.n_nodes,))
        valid = (np.arange(lists.stride)[None, :] < cnt[:, None]).ravel()
        cv = c[valid]
        if ref is None:
            ref = cv
        rec[route] = {"ms": ms, "identity_ms": ms - rec["heur_flags_only_ms"], "G_successors_per_s": n_emit / ms / 1e6,
                      "form": env.last_identity_form(),
                      "first_occurrences": int(np.count_nonzero(cv == np.nonzero(valid)[0])),
                      "canon_equal_to_table_route": bool(np.array_equal(cv, ref))}
    out[label] = rec
    fr.free()
env.close()
print(json.dumps(out, indent=1))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
