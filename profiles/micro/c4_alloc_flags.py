"""The state rows of C4's lists allocated with hipExtMallocWithFlags: default / physically contiguous / uncached --
expansion kernel time per allocation (placement differs per allocation).  Run on the GPU box."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import motion_primitive_library_amd as m

hip = C.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipFree.argtypes = [C.c_void_p]

class Raw:
    def __init__(self, nbytes, flags):
        p = C.c_void_p()
        rc = hip.hipExtMallocWithFlags(C.byref(p), nbytes, flags)
        if rc != 0:
            raise RuntimeError("hipExtMallocWithFlags(%d, %#x) -> %d" % (nbytes, flags, rc))
        self.ptr, self.nbytes = p.value, nbytes
    def free(self):
        hip.hipFree(C.c_void_p(self.ptr))

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

lists = env.alloc_lists(N, want_state=True, want_iters=False)
variants = [("default", 0), ("contiguous", 4), ("uncached", 3)]
res = {v: [] for v, _ in variants}
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    for name, fl in variants:
        nb = lists.state.nbytes
        lists.state.free()
        try:
            lists.state = Raw(nb, fl)
        except RuntimeError as e:
            res[name].append(str(e)[:60])
            lists.state = Raw(nb, 0)
            continue
        res[name].append(round(expand_ms(lists), 4))
for v, _ in variants:
    print("%-11s %s" % (v, res[v]))
