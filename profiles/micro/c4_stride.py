"""Entries reserved per node (mplx_succ_lists::node_stride) against the slow placement mode of C4: 736 (= 729 rounded up
to 32, the default), 768, 800, 1024; each allocated / timed / freed several times in one process.  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import motion_primitive_library_amd as m

wl = m.workloads.make("C4")
env = m.EnvMap(wl.dim, 0)
wl.apply(env)
fr = env.upload_frontier(wl.nodes)
N = wl.nodes.shape[1]

def expand_ms(lists, k=20):
    for _ in range(30):
        env.expand_lists_resident(fr, lists)
    env.synchronize()
    env.timer_begin()
    for _ in range(k):
        env.expand_lists_resident(fr, lists)
    return env.timer_end() / k

strides = [int(a) for a in sys.argv[2:]] or [736, 768, 800, 1024]
res = {s: [] for s in strides}
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    for s in strides:
        lists = env.alloc_lists(N, want_state=True, want_iters=False, stride=s)
        res[s].append(round(expand_ms(lists), 4))
        lists.free()
for s in strides:
    slow = sum(1 for x in res[s] if x > 0.53)
    print("node_stride %5d: %s  slow %d of %d" % (s, res[s], slow, len(res[s])))
