"""Round 4: per allocation of the C4 lists -- the real kernel's time (its placement mode) next to pure stores into the SAME
state-row allocation under three layouts (write_layout_lib.hip): F (shipped, field-major), F2 (16-byte stores, a lane owns
two successors), R7t (112-byte records, transposed), and a linear fill.  Does the slow mode exist for pure stores, and
does a layout remove it?  One JSON line per allocation."""
import ctypes as C, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import motion_primitive_library_amd as m

lib = C.CDLL(os.path.join(HERE, "write_layout_lib.so"))
lib.run_layout.restype = C.c_float
lib.run_layout.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
n_alloc = int(sys.argv[1]) if len(sys.argv) > 1 else 8
wl = m.workloads.make("C4")
env = m.EnvMap(wl.dim, 0)
wl.apply(env)
fr = env.upload_frontier(wl.nodes)
N = wl.nodes.shape[1]


def expand_ms(lists, k=20):
    for _ in range(60):
        env.expand_lists_resident(fr, lists)
    env.synchronize()
    env.timer_begin()
    for _ in range(k):
        env.expand_lists_resident(fr, lists)
    return env.timer_end() / k


for rep in range(n_alloc):
    lists = env.alloc_lists(N, want_state=True, want_iters=False)
    S = lists.stride
    rec = {"alloc": rep, "kernel_ms": round(expand_ms(lists), 4), "node_stride": S}
    env.synchronize()
    for name, layout, chunk in (("F", 0, 4), ("F_chunk1", 0, 1), ("F2", 1, 4), ("R7t", 2, 4), ("fill", 3, 4)):
        rec[name] = round(lib.run_layout(lists.state.ptr, N * S, N, S, layout, chunk, 20), 4)
    rec["kernel_ms_again"] = round(expand_ms(lists), 4)
    print(json.dumps(rec), flush=True)
    lists.free()
