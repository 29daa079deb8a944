"""A/B of two builds of libmplx.so in ONE process on ONE allocation of the output lists -- the only fair comparison on C4,
whose kernel time depends on where the lists landed (0.49 ... 0.57 ms per allocation, profiles/r03_c4_placement_counters.txt).

    python profiles/micro/ab_same_alloc.py OLD.so [NEW.so] [workloads, default C4,C3,C5,C2] [rounds]

The package is imported twice (once per library: _abi reads MPLX_LIB at import); frontier and lists are allocated by the
first copy (plain device memory) and both copies' contexts launch into them, alternating.  Run on the GPU box.
"""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "motion_primitive_library_amd")


def load(name, lib):
    os.environ["MPLX_LIB"] = os.path.abspath(lib)
    spec = importlib.util.spec_from_file_location(name, os.path.join(PKG, "__init__.py"), submodule_search_locations=[PKG])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    import ctypes
    probe = ctypes.CDLL(os.path.abspath(lib))
    mod._abi.SYMBOLS = [n for n in mod._abi.SYMBOLS if hasattr(probe, n)]  # an older build lacks the newer entry points
    mod._abi.lib()
    return mod


old_so = sys.argv[1]
new_so = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2].endswith(".so") else os.path.join(PKG, "csrc", "libmplx.so")
rest = [a for a in sys.argv[2:] if not a.endswith(".so")]
names = rest[0].split(",") if rest else ["C4", "C3", "C5", "C2"]
rounds = int(rest[1]) if len(rest) > 1 else 7
A, B = load("mplx_a", old_so), load("mplx_b", new_so)
out = {}
for wname in names:
    pot = A.workloads.device_potential_fn(0) if wname == "C5" else None
    wl = A.workloads.make(wname, potential_fn=pot)
    ea, eb = A.EnvMap(wl.dim, 0), B.EnvMap(wl.dim, 0)
    wl.apply(ea)
    wl.apply(eb)
    fr = ea.upload_frontier(wl.nodes)
    lists = ea.alloc_lists(wl.n_nodes, want_state=True, want_iters=False)

    def struct_b():
        s, t = lists.c_struct(), B._abi.SuccLists()
        for f, _ in B._abi.SuccLists._fields_:
            setattr(t, f, getattr(s, f))
        return t

    sb = struct_b()
    import ctypes as C

    def launch_a():
        ea.expand_lists_resident(fr, lists)

    def launch_b():
        eb._flush()
        B._abi.check(eb._ctx, B._abi.lib().mplx_expand_lists_device(eb._ctx, fr.ptr, wl.n_nodes, fr.n_nodes, C.byref(sb)))

    def timed(env, fn, k):
        for _ in range(10):
            fn()
        env.synchronize()
        env.timer_begin()
        for _ in range(k):
            fn()
        return env.timer_end() / k

    k = 20 if wname == "C4" else 100
    for _ in range(3):
        timed(ea, launch_a, k), timed(eb, launch_b, k)
    ta, tb = [], []
    for _ in range(rounds):
        ta.append(timed(ea, launch_a, k))
        tb.append(timed(eb, launch_b, k))
    med = lambda v: sorted(v)[len(v) // 2]
    out[wname] = {"old_ms": med(ta), "new_ms": med(tb), "new_over_old": med(tb) / med(ta),
                  "old_range": [min(ta), max(ta)], "new_range": [min(tb), max(tb)]}
    print(wname, json.dumps(out[wname]), flush=True)
    lists.free()
    fr.free()
    ea.close()
    eb.close()
