"""GPU: a 40-world slice of the parity fuzz (profiles/micro/fuzz_parity.py ran 2 200 such worlds in round 2): random
"odd" worlds -- nothing round in resolution, duration, origin, control values, velocities or weights; dimension,
control flag and potential map chosen by the seed -- with frontiers large enough for the dynamic node assignment,
small forced grids and claim chunks, two resident launches each, EVERY list field against the reference build
(reference include/mpl_planner/env/env_map.h:147-172)."""
import os

import numpy as np
import pytest

from helpers import assert_lists_equal, check_fused_rows, engine_env, odd_world, oracle_env, require_reference_build
from oracle import oracle as O

pytestmark = pytest.mark.gpu

SEEDS = list(range(7000, 7040))


@pytest.mark.parametrize("block", range(4))
def test_fuzz_slice_against_the_reference(engine, monkeypatch, block):
    use_ref = require_reference_build()
    seen = set()
    for seed in SEEDS[block * 10:(block + 1) * 10]:
        rng = np.random.default_rng(seed)
        n_nodes = int(rng.choice([700, 1500, 2311, 5000]))
        monkeypatch.setenv("MPLX_GRID_BLOCKS", str(int(rng.choice([3, 8, 40, 256]))))
        monkeypatch.setenv("MPLX_GRID_CHUNK", str(int(rng.choice([0, 1, 2, 5]))))
        wl, control, pot = odd_world(engine, seed, n_nodes)
        ref = O.expand(oracle_env(wl), wl.nodes, threads=os.cpu_count() or 1, ref=use_ref)
        env = engine_env(engine, wl)
        rtol = 1e-6 if control & 0x10 else 0.0  # xYAW: per-sample heading cost uses device trig (north_star: 1e-6)
        fr = env.upload_frontier(wl.nodes)
        lists = env.alloc_lists(n_nodes, want_state=True, want_iters=True, want_heur=True, want_flags=True)
        hb, fb = lists.heur, lists.flags
        for launch in range(2):
            # the second launch also writes the search's per-successor rows (mplx_set_goal: heuristic, goal flags) --
            # the lists themselves must not depend on them
            fused = launch == 1
            lists.heur, lists.flags = (hb, fb) if fused else (None, None)
            if fused:
                D = wl.dim
                goal = np.ascontiguousarray(wl.nodes[:, int(rng.integers(0, n_nodes))])
                tols = (float(rng.uniform(0.05, 2.0)), float(rng.choice([-1.0, 0.4, 3.0])), float(rng.choice([-1.0, 2.5])),
                        float(rng.choice([-1.0, 0.7])))
                w_h, v_h = float(rng.uniform(0.5, 12.0)), float(rng.choice([-1.0, 0.0, 1.7]))
                env.set_goal(goal, w=w_h, v_max=v_h, tol_pos=tols[0], tol_vel=tols[1], tol_acc=tols[2], tol_yaw=tols[3])
            env.expand_lists_resident(fr, lists)
            env.synchronize()
            got = lists.download()
            assert_lists_equal(got, ref, n_nodes, wl.U.shape[0], cost_rtol=rtol,
                               what="fuzz seed %d launch %d route %s" % (seed, launch, env.last_lists_route()))
            if fused:
                check_fused_rows(got, goal, control, D, w_h, v_h, tols, what="fuzz seed %d" % seed)
        lists.heur, lists.flags = hb, fb
        seen.add((wl.dim, control, env.last_lists_route(), pot is not None))
        lists.free()
        fr.free()
        env.close()
    assert len(seen) >= 6, seen  # the slice really spreads over dimensions / controls / potential maps
