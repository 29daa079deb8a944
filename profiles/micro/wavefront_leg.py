"""Round 4: why does bench.py's wavefront LEG measure 0.75 - 0.97 ms when the same kernel on the same frontier takes
0.627 ms in profiles/micro/wavefront_bisect.py?  Reproduces the leg (planner search on the device, then a fresh context,
5 warm-up + 20 timed launches) and times every launch on its own, with and without a spin-up, under the env given."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import motion_primitive_library_amd as m

wl = m.workloads.make("C4")
res = {"env": {k: v for k, v in os.environ.items() if k.startswith("MPLX_")}}
wf = m.workloads.wavefront_frontier(wl, wl.n_nodes, 0)
t_gen = time.time()
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


res["setup_s"] = round(time.time() - t_gen, 2)
res["each_of_first_30"] = [round(batch(1), 4) for _ in range(30)]
res["leg_like_5_20"] = round(batch(20), 4)
for _ in range(300):
    env.expand_lists_resident(fr, lists)
res["after_300_more"] = [round(batch(20), 4) for _ in range(3)]
fr2 = env.upload_frontier(wl.nodes)
fr, fr_w = fr2, fr
res["random_same_alloc"] = [round(batch(20), 4) for _ in range(3)]
fr = fr_w
res["wavefront_again"] = [round(batch(20), 4) for _ in range(3)]
# a second allocation of the lists in the same process (the leg's lists are the 3rd-or-so allocation of the bench process)
l2 = env.alloc_lists(wl.n_nodes, want_state=True, want_iters=False)
lists, l1 = l2, lists
res["wavefront_second_alloc"] = [round(batch(20), 4) for _ in range(3)]
l1.free()
l3 = env.alloc_lists(wl.n_nodes, want_state=True, want_iters=False)
lists = l3
res["wavefront_third_alloc_after_free"] = [round(batch(20), 4) for _ in range(3)]
print(json.dumps(res))
