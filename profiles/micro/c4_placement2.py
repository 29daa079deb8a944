"""Which buffer's placement decides the slow mode?  Keep the Lists object, re-allocate either only the state rows or only
the other rows (count / action / cost / hash) between timings.  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import motion_primitive_library_amd as m
from motion_primitive_library_amd.env import DeviceArray

wl = m.workloads.make("C4")
env = m.EnvMap(wl.dim, 0)
wl.apply(env)
fr = env.upload_frontier(wl.nodes)
N = wl.nodes.shape[1]

def expand_ms(lists, k=40):
    for _ in range(60):
        env.expand_lists_resident(fr, lists)
    env.synchronize()
    env.timer_begin()
    for _ in range(k):
        env.expand_lists_resident(fr, lists)
    return env.timer_end() / k

lists = env.alloc_lists(N, want_state=True, want_iters=False)
print("initial: %.4f ms" % expand_ms(lists))
for rep in range(16):
    if rep % 2 == 0:
        nb = lists.state.nbytes
        lists.state.free()
        lists.state = DeviceArray(env, nb)
        what = "state re-allocated"
    else:
        for name in ("action", "cost", "hash"):
            b = getattr(lists, name)
            nb = b.nbytes
            b.free()
            setattr(lists, name, DeviceArray(env, nb))
        what = "action/cost/hash re-allocated"
    print("%-30s %.4f ms  (state %#x cost %#x hash %#x action %#x)" % (what, expand_ms(lists), lists.state.ptr, lists.cost.ptr, lists.hash.ptr, lists.action.ptr))
