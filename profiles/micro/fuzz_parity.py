"""Parity fuzz on the GPU box: many random "odd" worlds (tests/helpers.py::odd_world) with frontiers large enough for the
dynamic node assignment, small forced grids and chunk sizes, resident launches, every route -- against the oracle.

    python profiles/micro/fuzz_parity.py [first_seed] [n_seeds]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import motion_primitive_library_amd as m  # noqa: E402
from helpers import assert_lists_equal, check_fused_rows, engine_env, odd_world, oracle_env  # noqa: E402
from oracle import oracle as O  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    n_nodes = int(rng.choice([700, 1500, 2311, 5000]))
    os.environ["MPLX_GRID_BLOCKS"] = str(int(rng.choice([3, 8, 40, 256])))
    os.environ["MPLX_GRID_CHUNK"] = str(int(rng.choice([0, 1, 2, 5])))
    # round 4 (expand_lex_kernel): starved row / box budgets now and then (several passes per node, samples straight from the
    # blocked-bit map), nodes at rest (the dropped successor), tables of 17 .. 25 values per axis
    for var, choices in (("MPLX_GRID_RMAX", [0, 0, 0, 1, 2]), ("MPLX_GRID_BOXCAP", [0, 0, 0, 8, 64])):
        v = int(rng.choice(choices))
        if v:
            os.environ[var] = str(v)
        else:
            os.environ.pop(var, None)
    try:
        wl, control, pot = odd_world(m, seed, n_nodes, decorrelate=seed >= 20000)
        if seed % 3 == 0:
            wl.nodes[wl.dim:4 * wl.dim, ::int(rng.choice([3, 7, 50]))] = 0.0
        if seed % 5 == 0 and not (control & 0x10) and wl.dim == 2:
            nv = int(rng.choice([17, 21, 25]))
            lo, hi = wl.U[:, 0].min(), wl.U[:, 0].max()
            wl.U = m.workloads.grid_controls(list(np.linspace(lo, hi, nv)), 2)
        ref = O.expand(oracle_env(wl), wl.nodes, threads=os.cpu_count())
        env = engine_env(m, wl)
        rtol = 1e-6 if control & 0x10 else 0.0
        fr = env.upload_frontier(wl.nodes)
        lists = env.alloc_lists(n_nodes, want_state=True, want_iters=True, want_heur=True, want_flags=True)
        hb, fb = lists.heur, lists.flags
        for launch in range(2):
            # round 5: the second launch also writes the search's per-successor rows (mplx_set_goal: heuristic, goal flags)
            fused = launch == 1
            lists.heur, lists.flags = (hb, fb) if fused else (None, None)
            if fused:
                goal = np.ascontiguousarray(wl.nodes[:, int(rng.integers(0, n_nodes))])
                tols = (float(rng.uniform(0.05, 2.0)), float(rng.choice([-1.0, 0.4, 3.0])), float(rng.choice([-1.0, 2.5])),
                        float(rng.choice([-1.0, 0.7])))
                w_h, v_h = float(rng.uniform(0.5, 12.0)), float(rng.choice([-1.0, 0.0, 1.7]))
                env.set_goal(goal, w=w_h, v_max=v_h, tol_pos=tols[0], tol_vel=tols[1], tol_acc=tols[2], tol_yaw=tols[3])
            env.expand_lists_resident(fr, lists)
            env.synchronize()
            got = lists.download()
            assert_lists_equal(got, ref, n_nodes, wl.U.shape[0], cost_rtol=rtol,
                               what="seed %d launch %d route %s" % (seed, launch, env.last_lists_route()))
            if fused:
                check_fused_rows(got, goal, control, wl.dim, w_h, v_h, tols, what="seed %d" % seed)
        lists.heur, lists.flags = hb, fb
        route = env.last_lists_route() + ("/" + env.last_grid_kernel() if env.last_lists_route() == "grid" else "")
        lists.free()
        fr.free()
        env.close()
        print("seed %d ok: dim %d control 0x%x %d nodes route %s blocks %s chunk %s%s" % (
            seed, wl.dim, control, n_nodes, route, os.environ["MPLX_GRID_BLOCKS"], os.environ["MPLX_GRID_CHUNK"],
            " potential" if pot is not None else ""))
    except AssertionError as e:
        bad += 1
        print("seed %d FAILED: %s" % (seed, str(e)[:300]))
print("fuzz: %d of %d seeds failed" % (bad, count))
sys.exit(1 if bad else 0)
