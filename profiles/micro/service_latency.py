"""Round trip of one small synchronous batch (mplx_expand_lists, host pointers): a launch of its own against the
resident kernel (mplx_service), C1's table (2D, 9 controls) and the 3D search's (729 controls).
    python profiles/micro/service_latency.py [out.json]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import motion_primitive_library_amd as m  # noqa: E402

out = {}
for label, wl, sizes in (("2D_9_controls", m.workloads.make("C2", scale=0.25, n_nodes=512), (1, 16, 64)),
                         ("3D_729_controls", m.workloads.make("C4", scale=0.125, n_nodes=512), (1, 16, 64))):
    if label.startswith("2D"):
        wl.U = m.workloads.grid_controls([-1.0, 0.0, 1.0], 2)
    for n in sizes:
        rec = {}
        for mode in (0, 1):
            env = m.EnvMap(wl.dim)
            wl.apply(env)
            env.service(mode)
            b = np.ascontiguousarray(wl.nodes[:, :n])
            o = env.expand_lists(b, want_iters=False)
            for _ in range(50):
                env.expand_lists(b, want_iters=False, out=o)
            reps = 2000
            t0 = time.perf_counter()
            for _ in range(reps):
                env.expand_lists(b, want_iters=False, out=o)
            dt = (time.perf_counter() - t0) / reps * 1e6
            st = env.service()
            rec["resident" if mode else "launch"] = round(dt, 2)
            if mode:
                rec["served"] = st["requests"]
                rec["failures"] = st["failures"]
            env.close()
        out["%s_n%d" % (label, n)] = rec
        print(label, n, rec, flush=True)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
