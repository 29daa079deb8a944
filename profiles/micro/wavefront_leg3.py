"""Round 4: which part of bench.py's sequence makes its wavefront leg slow?  argv[1] selects what runs BEFORE the leg:
none | main (a context + the lists of the headline kept allocated, 6 probe allocations made and freed) | e2e | edges |
main+e2e+edges (the bench's own order).  The leg itself = bench.run_config on the wavefront frontier; then the same leg
again, and per-launch times of a third run."""
import json, os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import motion_primitive_library_amd as m
import bench

what = sys.argv[1]
wl = m.workloads.make("C4")
res = {"before": what}
keep = []
if "main" in what:
    env = m.EnvMap(wl.dim, 0)
    wl.apply(env)
    fr = env.upload_frontier(wl.nodes)
    tried = []
    for i in range(6):
        l = env.alloc_lists(wl.n_nodes, want_state=True, want_iters=False)
        tried.append(l)
        res.setdefault("probe_ms", []).append(round(bench.time_lists(env, fr, l, 20, 5), 4))
    for l in tried[:-1]:
        l.free()
    keep = [env, fr, tried[-1]]
if "e2e" in what:
    res["e2e"] = bench.extra_e2e(m, wl)["e2e_ms_per_step"]
if "edges" in what:
    env2 = m.EnvMap(wl.dim, 0)
    wl.apply(env2)
    fr2 = env2.upload_frontier(wl.nodes)
    l2 = env2.alloc_lists(wl.n_nodes, want_state=False, want_iters=False)
    res["edges_only"] = round(bench.time_lists(env2, fr2, l2, 20, 5), 4)
    l2.free(); fr2.free(); env2.close()
w2 = copy.copy(wl)
w2.nodes = m.workloads.wavefront_frontier(wl, wl.n_nodes, 0)
res["leg"] = round(bench.run_config(m, w2, 20, 5)["kernel_ms"], 4)
res["leg_again"] = round(bench.run_config(m, w2, 20, 5)["kernel_ms"], 4)
res["leg_random_frontier"] = round(bench.run_config(m, wl, 20, 5)["kernel_ms"], 4)
print(json.dumps(res))
