"""A/B of builds-time-equal variants selected by environment variables (read when a context is created) on the SAME
allocation of the lists, alternating, for several allocations:

    python profiles/micro/env_ab.py <C2|C3|C4|C5> [--edges] [--wavefront] [--allocs N] VAR=val[,VAR2=val2] VAR=val ...

e.g.  env_ab.py C4 MPLX_GRID_LEX=0 MPLX_GRID_LEX=1      the general factorised kernel against expand_lex_kernel.hip"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import motion_primitive_library_amd as m

args = sys.argv[1:]
name = args.pop(0)
edges = "--edges" in args
wavefront = "--wavefront" in args
n_alloc = int(args[args.index("--allocs") + 1]) if "--allocs" in args else 3
variants = [a for a in args if "=" in a]
wl = m.workloads.make(name, potential_fn=m.workloads.device_potential_fn(0) if name == "C5" else None)
if wavefront:
    wl.nodes = m.workloads.wavefront_frontier(wl, wl.n_nodes, 0)
envs = {}
for v in variants:
    kv = dict(x.split("=") for x in v.split(","))
    os.environ.update(kv)
    e = m.EnvMap(wl.dim, 0)
    wl.apply(e)
    envs[v] = e
    for k in kv:
        os.environ.pop(k)
first = envs[variants[0]]
fr = first.upload_frontier(wl.nodes)


def timed(e, lists, k=20):
    for _ in range(5):
        e.expand_lists_resident(fr, lists)
    e.synchronize()
    e.timer_begin()
    for _ in range(k):
        e.expand_lists_resident(fr, lists)
    return e.timer_end() / k


for a in range(n_alloc):
    lists = first.alloc_lists(wl.n_nodes, want_state=not edges, want_iters=False)
    for _ in range(6):
        for v in variants:
            timed(envs[v], lists)
    rounds = [{v: timed(envs[v], lists) for v in variants} for _ in range(5)]
    med = {v: sorted(r[v] for r in rounds)[2] for v in variants}
    print(json.dumps({"workload": name, "edges_only": edges, "wavefront": wavefront, "alloc": a,
                      "kernel": {v: envs[v].last_grid_kernel() for v in variants},
                      "median_ms": {v: round(x, 5) for v, x in med.items()},
                      "ratio_to_first": {v: round(med[v] / med[variants[0]], 4) for v in variants}}), flush=True)
    lists.free()
