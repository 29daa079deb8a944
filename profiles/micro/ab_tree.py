"""python ab_tree.py <tree> <workload>: kernel time of the lists call with the tree's own library (3 allocations x 3)."""
import json, os, sys
tree = os.path.abspath(sys.argv[1]); sys.path.insert(0, tree)
import motion_primitive_library_amd as m
assert os.path.abspath(m.__file__).startswith(tree)
name = sys.argv[2]
edges = "--edges" in sys.argv
wl = m.workloads.make(name, potential_fn=m.workloads.device_potential_fn(0) if name == "C5" else None)
e = m.EnvMap(wl.dim, 0); wl.apply(e)
fr = e.upload_frontier(wl.nodes)
out = []
for al in range(3):
    lists = e.alloc_lists(wl.n_nodes, want_state=not edges, want_iters=False)
    for _ in range(100): e.expand_lists_resident(fr, lists)
    e.synchronize()
    r = []
    for rep in range(3):
        e.timer_begin()
        for _ in range(20): e.expand_lists_resident(fr, lists)
        r.append(round(e.timer_end() / 20 * 1e3, 2))
    out.append(r); lists.free()
print(json.dumps({"tree": sys.argv[1], "wl": name, "edges": edges, "us": out, "kernel": e.last_grid_kernel()}))
