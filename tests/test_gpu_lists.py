"""GPU: the list-producing entry points (mplx_expand_lists*, the tiled kernel
expand_tile_kernel.hip and the dense+compaction route) against the oracle."""
import numpy as np
import pytest

from helpers import (assert_lists_equal, engine_env, engine_env_from_case, golden_cases, oracle_env,
                     oracle_env_from_case)
from test_gpu_parity import YAW_COST_RTOL, _small_world

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("control", [0x01, 0x03, 0x07, 0x0F, 0x11, 0x13, 0x17, 0x1F])
@pytest.mark.parametrize("variant", ["plain", "potential", "region", "nolimits"])
def test_lists_all_controls(engine, oracle_lib, dim, control, variant):
    """plain / region without yaw run the tiled kernel; potential, yaw and
    unbounded-velocity cases run the dense kernel + on-device compaction."""
    wl = _small_world(engine, dim, control, seed=500 * dim + control, potential=(variant == "potential"),
                      region=(variant == "region"), limits=(variant != "nolimits"), n_nodes=70)
    env = engine_env(engine, wl)
    got = env.expand_lists(wl.nodes)
    env.close()
    ref = oracle_lib.expand(oracle_env(wl), wl.nodes, threads=8)
    rtol = YAW_COST_RTOL if control & 0x10 else 0.0
    assert_lists_equal(got, ref, wl.n_nodes, wl.U.shape[0], cost_rtol=rtol,
                       what="lists dim%d ctrl0x%x %s" % (dim, control, variant))


@pytest.mark.parametrize("name,scale,n_nodes", [("C2", 0.25, 1500), ("C3", 0.25, 700), ("C4", 0.125, 300),
                                                 ("C5", 0.2, 512)])
def test_lists_baseline_configs_scaled(engine, oracle_lib, name, scale, n_nodes):
    wl = engine.workloads.make(name, scale=scale, n_nodes=n_nodes)
    env = engine_env(engine, wl)
    got = env.expand_lists(wl.nodes)
    env.close()
    ref = oracle_lib.expand(oracle_env(wl), wl.nodes, threads=8)
    rtol = YAW_COST_RTOL if wl.control & 0x10 else 0.0
    assert_lists_equal(got, ref, wl.n_nodes, wl.U.shape[0], cost_rtol=rtol, what="lists " + name)


_GOLDEN = list(golden_cases())


@pytest.mark.parametrize("name,case,exp", _GOLDEN, ids=[c[0] for c in _GOLDEN])
def test_lists_reproduce_reference_golden_vectors(engine, name, case, exp):
    env = engine_env_from_case(engine, case)
    got = env.expand_lists(case["nodes"])
    env.close()
    rtol = YAW_COST_RTOL if case["control"] & 0x10 else 0.0
    assert_lists_equal(got, exp, case["nodes"].shape[1], case["U"].shape[0], cost_rtol=rtol, what="lists " + name)


def test_lists_ragged_tiles_and_resident_buffers(engine, oracle_lib):
    """Frontier sizes around the nodes-per-workgroup tiling, and the HBM-resident call."""
    wl = engine.workloads.make("C2", scale=0.25, n_nodes=200)  # |U| = 25 -> 32 nodes per workgroup
    env = engine_env(engine, wl)
    for n in (1, 31, 32, 33, 64, 199, 200):
        sub = np.ascontiguousarray(wl.nodes[:, :n])
        got = env.expand_lists(sub)
        ref = oracle_lib.expand(oracle_env(wl), sub, threads=4)
        assert_lists_equal(got, ref, n, wl.U.shape[0], what="ragged n=%d" % n)
    fr = env.upload_frontier(wl.nodes)
    lists = env.alloc_lists(wl.n_nodes, want_state=True, want_iters=True)
    env.expand_lists_resident(fr, lists)
    env.synchronize()
    got = lists.download()
    ref = oracle_lib.expand(oracle_env(wl), wl.nodes, threads=4)
    assert_lists_equal(got, ref, wl.n_nodes, wl.U.shape[0], what="resident")
    lists.free()
    fr.free()
    env.close()


def test_hoisted_division_equals_true_division(engine):
    """The mul+fma+fma quotient of the tiled kernel must be the correctly rounded
    quotient: the cell index and sample count computed with it are checked against
    the dense kernel (true `/`) on a frontier that sits on cell faces."""
    W = engine.workloads
    grid = W.box_map([64, 64, 64], 0.1, 0.15, 77)
    U = W.grid_controls(np.arange(-2, 2.01, 0.5), 3)
    nodes = W.random_frontier(grid, [0, 0, 0], 0.1, 600, 78, 0x03, 2.0, 0.5)
    nodes[:3] = np.round(nodes[:3], 1)  # positions exactly on multiples of the resolution
    for res, org in ((0.1, [0, 0, 0]), (0.05, [-0.3, 0.1, 0.0]), (0.3, [0.05, 0, 0])):
        env = engine.EnvMap(3)
        env.setMap(org, [64, 64, 64], grid, res)
        env.set_control(0x03)
        env.set_u(U)
        env.set_v_max(2.0 if res >= 0.1 else 1.5)
        dense = env.expand(nodes)
        lists = env.expand_lists(nodes)
        env.close()
        assert_lists_equal(lists, dense, 600, U.shape[0], what="res %g" % res)
