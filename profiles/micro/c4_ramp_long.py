"""C4 launches back to back for ~5 s right after set-up in a fresh process: kernel ms per launch in groups of 50 against
wall time -- does the GPU settle in one clock state, and when?  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import motion_primitive_library_amd as m

wl = m.workloads.make("C4")
env = m.EnvMap(wl.dim, 0)
wl.apply(env)
fr = env.upload_frontier(wl.nodes)
lists = env.alloc_lists(wl.nodes.shape[1], want_state=True, want_iters=False)
t0 = time.perf_counter()
series = []
while time.perf_counter() - t0 < 5.0:
    env.timer_begin()
    for _ in range(50):
        env.expand_lists_resident(fr, lists)
    series.append((round(time.perf_counter() - t0, 2), round(env.timer_end() / 50, 4)))
print(series[::4])
