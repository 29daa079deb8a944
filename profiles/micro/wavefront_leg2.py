"""Round 4: the wavefront leg of bench.py after an idle gap / after the dense count_work launch (see wavefront_leg.py)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import motion_primitive_library_amd as m
import bench

wl = m.workloads.make("C4")
res = {}
wf = m.workloads.wavefront_frontier(wl, wl.n_nodes, 0)
env = m.EnvMap(wl.dim, 0)
wl.apply(env)
fr = env.upload_frontier(wf)
lists = env.alloc_lists(wl.n_nodes, want_state=True, want_iters=False)


def batch(k):
    env.synchronize()
    env.timer_begin()
    for _ in range(k):
        env.expand_lists_resident(fr, lists)
    return env.timer_end() / k


for _ in range(300):
    env.expand_lists_resident(fr, lists)
res["steady"] = [round(batch(20), 4) for _ in range(3)]
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.0:
    pass
res["after_idle_each_of_30"] = [round(batch(1), 4) for _ in range(30)]
res["count_work"] = bench.count_work(env, fr, wl.n_nodes)
res["after_count_work_5_20"] = [round(bench.time_lists(env, fr, lists, 20, 5), 4)]
res["then_each_of_30"] = [round(batch(1), 4) for _ in range(30)]
res["count_work2"] = bench.count_work(env, fr, wl.n_nodes)
res["after_count_work_each_of_30"] = [round(batch(1), 4) for _ in range(30)]
print(json.dumps(res))
