import sys
sys.path.insert(0, "/root/repo")
import motion_primitive_library_amd as m
for name in ("C2", "C3"):
    wl = m.workloads.make(name)
    out = {}
    for route in ("grid", "tile", "dense"):
        env = m.EnvMap(wl.dim, 0); wl.apply(env); env.set_lists_route(route)
        fr = env.upload_frontier(wl.nodes); lists = env.alloc_lists(wl.n_nodes, want_state=True, want_iters=False)
        try:
            for _ in range(20): env.expand_lists_resident(fr, lists)
            env.synchronize(); env.timer_begin()
            for _ in range(100): env.expand_lists_resident(fr, lists)
            out[route] = round(env.timer_end() / 100 * 1e3, 2)
        except Exception as e:
            out[route] = str(e)[:60]
        env.close()
    print(name, out, flush=True)
