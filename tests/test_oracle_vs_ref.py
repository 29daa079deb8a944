"""CPU: the restatement (oracle/mpl_oracle.cpp) against the reference's own
headers compiled with stand-in Eigen/Boost (oracle/_ref/libmpl_ref.so), on
inputs larger and more varied than the committed fixture.  Runs where the
prebuilt _ref library exists (it can only be BUILT where /root/reference is)."""
import os

import numpy as np
import pytest

from helpers import assert_slots_equal, oracle_env
from oracle import oracle as O

REF_SO = os.path.join(O.HERE, "_ref", "libmpl_ref.so")
pytestmark = pytest.mark.skipif(not (os.path.exists(REF_SO) or os.path.isdir("/root/reference/include")),
                                reason="oracle/_ref not built and /root/reference absent")


@pytest.mark.parametrize("name,scale,n_nodes", [("C2", 0.25, 600), ("C3", 0.2, 300), ("C4", 0.125, 200),
                                                 ("C5", 0.2, 400)])
def test_port_equals_reference_on_baseline_configs(name, scale, n_nodes):
    import motion_primitive_library_amd.workloads as W
    wl = W.make(name, scale=scale, n_nodes=n_nodes)
    a = O.expand(oracle_env(wl), wl.nodes, threads=4)
    b = O.expand(oracle_env(wl), wl.nodes, threads=4, ref=True)
    assert_slots_equal(a, b, cost_rtol=0.0, what=name)
    assert a["stats"] == b["stats"]


def test_hash_heur_loopcount_match_reference():
    rng = np.random.default_rng(3)
    for dim in (2, 3):
        for control in (0x01, 0x03, 0x07, 0x0F, 0x13, 0x1F):
            for _ in range(50):
                wp = np.round(rng.uniform(-20, 20, 4 * dim + 2), 2)
                goal = np.round(rng.uniform(-20, 20, 4 * dim + 2), 2)
                assert O.lattice_hash(dim, control, wp) == O.lattice_hash(dim, control, wp, ref=True)
                assert O.heur(dim, control, 10.0, 1.5, wp, goal) == O.heur(dim, control, 10.0, 1.5, wp, goal, ref=True)
    for T in (1.0, 0.5, 0.3, 2.0):
        for n in range(5, 120):
            assert O.loop_count(T, n) == O.loop_count(T, n, ref=True)


def test_goal_tolerance_matches_reference_is_goal():
    """The norm tests of env_map::is_goal (env_map.h:25-37) against the reference's is_goal on a free map."""
    rng = np.random.default_rng(9)
    hits = 0
    for dim in (2, 3):
        for _ in range(400):
            goal = np.round(rng.uniform(-5, 5, 4 * dim + 2), 2)
            wp = goal + rng.choice([0.0, 0.1, 0.3, 0.5, 0.50000001, 1.0], size=goal.size) * rng.choice([-1, 1], size=goal.size)
            tp, tv, ta, ty = rng.choice([0.5, 0.3, 1.0]), rng.choice([-1.0, 0.0, 0.3]), rng.choice([-1.0, 0.5]), rng.choice([-1.0, 0.3])
            a = O.goal_tol(dim, wp, goal, tp, tv, ta, ty)
            assert a == O.goal_tol(dim, wp, goal, tp, tv, ta, ty, ref=True)
            hits += a
    assert 0 < hits < 800
