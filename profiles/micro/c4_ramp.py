"""Clock ramp after idle: C4 launches timed in groups of 10 (HIP events) right after 6 s of GPU idle.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import motion_primitive_library_amd as m

wl = m.workloads.make("C4")
env = m.EnvMap(wl.dim, 0)
wl.apply(env)
fr = env.upload_frontier(wl.nodes)
lists = env.alloc_lists(wl.nodes.shape[1], want_state=True, want_iters=False)
for _ in range(3):
    env.expand_lists_resident(fr, lists)
env.synchronize()
for trial in range(2):
    time.sleep(6)
    t0 = time.perf_counter()
    series = []
    for g in range(60):
        env.timer_begin()
        for _ in range(10):
            env.expand_lists_resident(fr, lists)
        ms = env.timer_end() / 10
        series.append((round((time.perf_counter() - t0) * 1e3), round(ms, 4)))
    print("trial %d (wall ms, kernel ms per launch):" % trial, series[:12], "...", series[-6:])
