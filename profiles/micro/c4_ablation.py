"""What bounds the C4 launch of the end-of-round kernel?  Ablations of expand_grid_kernel on ONE box, ONE process and ONE
allocation of the output lists (so the placement mode is the same for every variant), alternating in rounds:

    full            the product launch (lists with state rows)
    no_state        edges only: count + action + cost + hash, the 14 state rows not written (a product mode)
    no_stores       no list stores at all except count (MPLX_TILE_DBG=6; timing only, outputs invalid)
    no_sampling     no rows / box staging / sample loops (MPLX_TILE_DBG=1; timing only)
    no_freebox      the summed-area-table shortcut off (MPLX_GRID_NOSAT=1; a valid launch)
    compute_only    no sampling and no stores (MPLX_TILE_DBG=7): T1 + pair phase + list building

    python profiles/micro/c4_ablation.py [out.json]          timing table
    python profiles/micro/c4_ablation.py --only NAME         20 launches of one variant (for rocprofv3 --pmc passes)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import motion_primitive_library_amd as m  # noqa: E402

VARIANTS = [("full", {}, True), ("no_state", {}, False), ("no_stores", {"MPLX_TILE_DBG": "6"}, True),
            ("no_sampling", {"MPLX_TILE_DBG": "1"}, True), ("no_freebox", {"MPLX_GRID_NOSAT": "1"}, True),
            ("compute_only", {"MPLX_TILE_DBG": "7"}, True)]
only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
wl = m.workloads.make("C4")
envs = {}
for name, knobs, _ in VARIANTS:
    if only and name != only:
        continue
    for k in ("MPLX_TILE_DBG", "MPLX_GRID_NOSAT"):
        os.environ.pop(k, None)
    os.environ.update(knobs)
    e = m.EnvMap(wl.dim, 0)  # the knobs are read when the context is created
    wl.apply(e)
    envs[name] = e
for k in ("MPLX_TILE_DBG", "MPLX_GRID_NOSAT"):
    os.environ.pop(k, None)
first = next(iter(envs.values()))
fr = first.upload_frontier(wl.nodes)
lists = first.alloc_lists(wl.n_nodes, want_state=True, want_iters=False)  # plain device memory: every context may write it
state = lists.state


def launch(name, k):
    e = envs[name]
    lists.state = state if dict((v[0], v[2]) for v in VARIANTS)[name] else None
    for _ in range(k):
        e.expand_lists_resident(fr, lists)
    lists.state = state


def timed(name, k=20):
    e = envs[name]
    launch(name, 5)
    e.synchronize()
    e.timer_begin()
    launch(name, k)
    return e.timer_end() / k


for _ in range(3):  # clocks up
    for name in envs:
        timed(name)
if only:
    print(only, "%.4f ms" % timed(only))
    sys.exit(0)
rounds = [{name: timed(name) for name in envs} for _ in range(7)]
out = {name: {"median_ms": sorted(r[name] for r in rounds)[len(rounds) // 2], "min_ms": min(r[name] for r in rounds),
              "max_ms": max(r[name] for r in rounds)} for name in envs}
full = out["full"]["median_ms"]
for name in out:
    out[name]["vs_full"] = out[name]["median_ms"] / full
print(json.dumps(out, indent=1))
if len(sys.argv) > 1 and not sys.argv[1].startswith("--"):
    json.dump(out, open(sys.argv[1], "w"), indent=1)
