"""Why does the same C4 launch take 0.50 ms in one process and 0.565 ms in the next?  In ONE process: the lists are
allocated, timed (30 launches, HIP events) and freed several times, then timed repeatedly without re-allocating, then a
long soak.  Allocation-dependent -> placement (pages / channels); time-dependent -> clocks.  Run on the GPU box."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import motion_primitive_library_amd as m

wl = m.workloads.make("C4")
env = m.EnvMap(wl.dim, 0)
wl.apply(env)
fr = env.upload_frontier(wl.nodes)
N = wl.nodes.shape[1]

def timeit(lists, k=30):
    for _ in range(3):
        env.expand_lists_resident(fr, lists)
    env.synchronize()
    env.timer_begin()
    for _ in range(k):
        env.expand_lists_resident(fr, lists)
    return env.timer_end() / k

print("fresh allocation each time:")
for rep in range(6):
    lists = env.alloc_lists(N, want_state=True, want_iters=False)
    print("  alloc %d: %.4f ms  (state ptr %#x)" % (rep, timeit(lists), lists.state.ptr))
    lists.free()
lists = env.alloc_lists(N, want_state=True, want_iters=False)
print("same allocation, repeated:")
for rep in range(6):
    print("  run %d: %.4f ms" % (rep, timeit(lists)))
print("soak (2000 launches), then again:")
t0 = time.time()
print("  soak avg %.4f ms" % timeit(lists, 2000), "wall %.1f s" % (time.time() - t0))
for rep in range(3):
    print("  run %d: %.4f ms" % (rep, timeit(lists)))
time.sleep(5)
print("after 5 s idle: %.4f ms" % timeit(lists))
