"""The lists without the 112-byte successor states (mplx_succ_lists.state = NULL): what a search needs for every edge is
(action, cost, hash); the state matters only for lattice states it has not seen yet.  C4, resident launch and
host-pointer call.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import motion_primitive_library_amd as m  # noqa: E402

wl = m.workloads.make("C4")
env = m.EnvMap(wl.dim, 0)
wl.apply(env)
res = {"workload": "C4", "pairs": wl.n_pairs}
fr = env.upload_frontier(wl.nodes)
for name, want_state in (("full", True), ("edges_only", False)):
    lists = env.alloc_lists(wl.n_nodes, want_state=want_state, want_iters=False)
    for _ in range(3):
        env.expand_lists_resident(fr, lists)
    env.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        env.expand_lists_resident(fr, lists)
    env.synchronize()
    dt = (time.perf_counter() - t0) / 20
    res[name + "_resident_ms"] = round(dt * 1e3, 4)
    res[name + "_resident_pairs_per_s"] = wl.n_pairs / dt
    lists.free()
    out = env.expand_lists(wl.nodes, want_state=want_state, want_iters=False)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        out = env.expand_lists(wl.nodes, want_state=want_state, want_iters=False, out=out)
        ts.append(time.perf_counter() - t0)
    res[name + "_host_pointer_ms"] = round(min(ts) * 1e3, 2)
    del out
print(json.dumps(res))
env.close()
