"""Wall time of one host-pointer mplx_expand_lists call for the batch sizes a search sends (caller's arrays reused):
2D |U| = 9 (the reference's test scenarios) and 3D |U| = 729 (plan_bench), n = 1 .. 256 nodes.  Median of 300."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import motion_primitive_library_amd as m  # noqa: E402

res = {}
for name, wl in (("2D |U|=9", None), ("3D |U|=729", m.workloads.make("C4", scale=0.25, n_nodes=256))):
    if wl is None:
        wl = m.workloads.make("C2", scale=0.25, n_nodes=256)
        wl.U = m.workloads.grid_controls([-0.5, 0.0, 0.5], 2)
    env = m.EnvMap(wl.dim, 0)
    wl.apply(env)
    for n in (1, 16, 64, 256):
        nodes = np.ascontiguousarray(wl.nodes[:, :n])
        out = env.expand_lists(nodes)
        for _ in range(5):
            env.expand_lists(nodes, out=out)
        ts = []
        for _ in range(300):
            t0 = time.perf_counter()
            env.expand_lists(nodes, out=out)
            ts.append(time.perf_counter() - t0)
        res["%s n=%d" % (name, n)] = round(float(np.median(ts)) * 1e6, 1)
    env.close()
print(json.dumps(res))
