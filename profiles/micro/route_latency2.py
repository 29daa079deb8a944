"""Round 4: small resident batches by kernel -- expand_lex_kernel (grid route, MPLX_GRID_LEX=1), expand_grid_kernel
(MPLX_GRID_LEX=0), expand_tile_kernel -- for the control tables of the BASELINE configurations: per launch, back to back
on the stream (HIP events: kernel + launch gap) and launch + synchronise seen from the host."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import motion_primitive_library_amd as m

res = {}
for name in ("2D25", "3D125", "3D729"):
    wl = {"2D25": lambda: m.workloads.make("C2", scale=0.25, n_nodes=1024),
          "3D125": lambda: m.workloads.make("C3", scale=0.5, n_nodes=1024),
          "3D729": lambda: m.workloads.make("C4", scale=0.25, n_nodes=1024)}[name]()
    for label, lex, route in (("lex", "1", "grid"), ("grid", "0", "grid"), ("tile", "1", "tile")):
        os.environ["MPLX_GRID_LEX"] = lex
        env = m.EnvMap(wl.dim, 0)
        wl.apply(env)
        env.set_lists_route(route)
        for n in (1, 16, 64, 256, 1024):
            nodes = np.ascontiguousarray(wl.nodes[:, :n])
            fr = env.upload_frontier(nodes)
            lists = env.alloc_lists(n, want_state=False, want_iters=False)
            for _ in range(20):
                env.expand_lists_resident(fr, lists)
            env.synchronize()
            env.timer_begin()
            for _ in range(100):
                env.expand_lists_resident(fr, lists)
            ev = env.timer_end() / 100 * 1e3
            ts = []
            for _ in range(100):
                t0 = time.perf_counter()
                env.expand_lists_resident(fr, lists)
                env.synchronize()
                ts.append(time.perf_counter() - t0)
            res["%s %s/%d" % (name, label, n)] = [round(ev, 1), round(float(np.median(ts)) * 1e6, 1)]
            lists.free()
            fr.free()
        env.close()
print(json.dumps(res))
