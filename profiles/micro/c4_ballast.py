"""Does the slow placement mode of C4 belong to a REGION of HBM?  A fresh process allocates X GB of ballast first, then
the lists, and times the expansion (argument: X in GB; run it once per X).  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import motion_primitive_library_amd as m
from motion_primitive_library_amd.env import DeviceArray

x = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
wl = m.workloads.make("C4")
env = m.EnvMap(wl.dim, 0)
ballast = [DeviceArray(env, 1 << 30) for _ in range(int(x))] if sys.argv[2:] == ["chunks"] else ([DeviceArray(env, int(x * (1 << 30)))] if x > 0 else [])
wl.apply(env)
fr = env.upload_frontier(wl.nodes)
N = wl.nodes.shape[1]

def expand_ms(lists, k=20):
    for _ in range(40):
        env.expand_lists_resident(fr, lists)
    env.synchronize()
    env.timer_begin()
    for _ in range(k):
        env.expand_lists_resident(fr, lists)
    return env.timer_end() / k

out = []
held = []
for i in range(4):
    lists = env.alloc_lists(N, want_state=True, want_iters=False)
    out.append(round(expand_ms(lists), 4))
    held.append(lists)
print("ballast %5.1f GB%s: successive list allocations (all held): %s" % (x, " in 1 GB chunks" if sys.argv[2:] == ["chunks"] else "", out))
