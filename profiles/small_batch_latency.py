"""Latency of one resident launch for small frontiers (the batches a search sends): routes x frontier sizes on the
C4 environment (3D 512^3... scaled map, ACC, |U| = 729).  Wall time per launch + synchronize, median of 200."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import motion_primitive_library_amd as m  # noqa: E402

wl = m.workloads.make("C4", scale=0.25, n_nodes=4096)
env = m.EnvMap(wl.dim, 0)
wl.apply(env)
res = {}
for route in ("grid", "tile", "dense"):
    env.set_lists_route(route)
    for n in (1, 4, 16, 64, 256, 1024, 4096):
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
        res["%s/%d" % (route, n)] = round(float(np.median(ts)) * 1e6, 1)
        lists.free()
        fr.free()
print(json.dumps(res))
env.close()
