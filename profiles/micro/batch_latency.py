"""Launch + synchronise latency of small HBM-resident batches (the batches of a search): C4's controls (|U| = 729) and
C2's (25) on the factorised kernel and on the workgroup-per-node kernel, 1 ... 1024 nodes.  Run on the GPU box."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import motion_primitive_library_amd as m
res = {}
for name in ("C4", "C2"):
    wl = m.workloads.make(name, scale=0.25, n_nodes=1024)
    env = m.EnvMap(wl.dim, 0)
    wl.apply(env)
    for route in ("grid", "tile", "auto"):
        env.set_lists_route(route)
        for n in (1, 64, 256, 1024):
            nodes = np.ascontiguousarray(wl.nodes[:, :n])
            fr = env.upload_frontier(nodes)
            lists = env.alloc_lists(n, want_state=True, want_iters=False)
            for _ in range(5):
                env.expand_lists_resident(fr, lists)
            env.synchronize()
            ts = []
            for _ in range(200):
                t0 = time.perf_counter()
                env.expand_lists_resident(fr, lists)
                env.synchronize()
                ts.append(time.perf_counter() - t0)
            res["%s %s/%d" % (name, route, n)] = round(float(np.median(ts)) * 1e6, 1)
            lists.free()
            fr.free()
    env.close()
print(json.dumps(res))
