"""C5 (and other workloads) kernel time per launch on ONE allocation for a list of environment settings, each in its own
context (the tuning variables are read at mplx_create):  python profiles/micro/c5_sweep.py C5 MPLX_GRID_RMAX=1 MPLX_GRID_RMAX=2 ...
("-" = the default environment).  Run on the GPU box."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import motion_primitive_library_amd as m  # noqa: E402

name = sys.argv[1]
settings = sys.argv[2:] or ["-"]
wl = m.workloads.make(name, potential_fn=m.workloads.device_potential_fn(0) if name == "C5" else None)
res = {}
for rnd in range(2):
    for st in settings:
        kv = [x.split("=") for x in st.split(",") if "=" in x]
        for k, v in kv:
            os.environ[k] = v
        env = m.EnvMap(wl.dim, 0)
        wl.apply(env)
        fr = env.upload_frontier(wl.nodes)
        lists = env.alloc_lists(wl.n_nodes, want_state=True)
        for _ in range(30):
            env.expand_lists_resident(fr, lists)
        env.synchronize()
        ts = []
        for _ in range(5):
            env.timer_begin()
            for _ in range(20):
                env.expand_lists_resident(fr, lists)
            ts.append(env.timer_end() / 20)
        res.setdefault(st, []).append(round(float(np.median(ts)) * 1e3, 2))
        lists.free()
        fr.free()
        env.close()
        for k, _ in kv:
            os.environ.pop(k, None)
print(json.dumps({"workload": name, "kernel_us": res}))
