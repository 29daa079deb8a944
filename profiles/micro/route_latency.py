"""Launch + synchronise latency of small resident batches by route (factorised / tiled / dense kernel) for small control
tables; run on the GPU box."""
import json, sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import motion_primitive_library_amd as m
res = {}
for name in ("2D9", "2D25", "3D125"):
    if name == "2D9":
        wl = m.workloads.make("C2", scale=0.25, n_nodes=1024); wl.U = m.workloads.grid_controls([-0.5, 0.0, 0.5], 2)
    elif name == "2D25":
        wl = m.workloads.make("C2", scale=0.25, n_nodes=1024)
    else:
        wl = m.workloads.make("C3", scale=0.5, n_nodes=1024)
    env = m.EnvMap(wl.dim, 0); wl.apply(env)
    for route in ("grid", "tile"):
        env.set_lists_route(route)
        for n in (1, 16, 64, 256, 1024):
            nodes = np.ascontiguousarray(wl.nodes[:, :n])
            fr = env.upload_frontier(nodes); lists = env.alloc_lists(n, want_state=True, want_iters=False)
            for _ in range(5): env.expand_lists_resident(fr, lists)
            env.synchronize(); ts = []
            for _ in range(200):
                t0 = time.perf_counter(); env.expand_lists_resident(fr, lists); env.synchronize(); ts.append(time.perf_counter() - t0)
            res["%s %s/%d" % (name, route, n)] = round(float(np.median(ts)) * 1e6, 1)
            lists.free(); fr.free()
    env.close()
print(json.dumps(res))
