"""Round 4: is C4's slow placement mode a property of the BOX or of the allocation?  One fresh process: N allocations
of the C4 lists one after the other (each freed before the next), kernel time of each.  Run several processes per box
and on several boxes (every gpurun call is a fresh box); one JSON line per process -> profiles/r04_placement_per_box.txt."""
import json, os, socket, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import motion_primitive_library_amd as m

n_alloc = int(sys.argv[1]) if len(sys.argv) > 1 else 8
wl = m.workloads.make("C4")
env = m.EnvMap(wl.dim, 0)
wl.apply(env)
fr = env.upload_frontier(wl.nodes)
N = wl.nodes.shape[1]


def expand_ms(lists, k=20):
    for _ in range(100):
        env.expand_lists_resident(fr, lists)
    env.synchronize()
    env.timer_begin()
    for _ in range(k):
        env.expand_lists_resident(fr, lists)
    return env.timer_end() / k


ms = []
for rep in range(n_alloc):
    lists = env.alloc_lists(N, want_state=True, want_iters=False)
    ms.append(round(expand_ms(lists), 4))
    lists.free()
print(json.dumps({"host": socket.gethostname(), "pid": os.getpid(), "ms": ms}))
