"""A/B of MPLX_TILE_DBG variants of the factorised kernel on the SAME allocation of the C4 lists, for several allocations
(both placement modes): python profiles/micro/dbg_ab.py <workload> <dbg,dbg,...> [allocations]."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import motion_primitive_library_amd as m

name = sys.argv[1]
dbgs = [int(x) for x in sys.argv[2].split(",")]
n_alloc = int(sys.argv[3]) if len(sys.argv) > 3 else 6
wl = m.workloads.make(name, potential_fn=m.workloads.device_potential_fn(0) if name == "C5" else None)
envs = {}
for d in dbgs:
    os.environ["MPLX_TILE_DBG"] = str(d)
    e = m.EnvMap(wl.dim, 0)
    wl.apply(e)
    envs[d] = e
os.environ.pop("MPLX_TILE_DBG")
first = envs[dbgs[0]]
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
    lists = first.alloc_lists(wl.n_nodes, want_state=True, want_iters=False)
    for _ in range(4):
        for d in dbgs:
            timed(envs[d], lists)
    rounds = [{d: timed(envs[d], lists) for d in dbgs} for _ in range(5)]
    med = {d: sorted(r[d] for r in rounds)[2] for d in dbgs}
    print(json.dumps({"alloc": a, "median_ms": {str(d): round(v, 4) for d, v in med.items()},
                      "ratio_to_first": {str(d): round(med[d] / med[dbgs[0]], 4) for d in dbgs}}), flush=True)
    lists.free()
