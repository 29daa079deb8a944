"""Is the slow mode of a C4 allocation (0.565 against 0.495 ms) a property of the memory it got?  Per allocation: the
expansion's kernel time and the rate of a plain device memset over the same state rows.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import motion_primitive_library_amd as m
from motion_primitive_library_amd import _abi

wl = m.workloads.make("C4")
env = m.EnvMap(wl.dim, 0)
wl.apply(env)
fr = env.upload_frontier(wl.nodes)
N = wl.nodes.shape[1]
L = _abi.lib()

def expand_ms(lists, k=40):
    for _ in range(60):
        env.expand_lists_resident(fr, lists)
    env.synchronize()
    env.timer_begin()
    for _ in range(k):
        env.expand_lists_resident(fr, lists)
    return env.timer_end() / k

def memset_gbps(buf, k=5):
    _abi.check(env._ctx, L.mplx_memset(env._ctx, buf.ptr, 0, buf.nbytes))
    env.synchronize()
    env.timer_begin()
    for _ in range(k):
        _abi.check(env._ctx, L.mplx_memset(env._ctx, buf.ptr, 0, buf.nbytes))
    return buf.nbytes * k / (env.timer_end() * 1e-3) / 1e9

keep = []
for rep in range(12):
    lists = env.alloc_lists(N, want_state=True, want_iters=False)
    e = expand_ms(lists)
    print("alloc %2d: expand %.4f ms   memset of the state rows %.0f GB/s   state ptr %#x" % (rep, e, memset_gbps(lists.state), lists.state.ptr))
    lists.free()
