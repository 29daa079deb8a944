"""GPU parity tests: the HIP engine (through the C ABI) against the CPU oracle
on identical (frontier, U, map) inputs.

Bar (BASELINE.json north_star): successor identity bit-exact -- status, lattice
hash, every field of the successor Waypoint, and the executed sample-loop
iteration count; edge cost within 1e-6 relative.  For every control without
yaw the engine is required to be bit-exact on the cost as well; the yaw
controls go through device cos/sin (OCML) vs the host libm, so their cost is
held to 1e-6 relative (tolerance stated at each assert).
"""
import numpy as np
import pytest

from helpers import assert_slots_equal, engine_env, oracle_env

pytestmark = pytest.mark.gpu

YAW_COST_RTOL = 1e-6  # north_star tolerance for edge costs


def _run_both(m, O, wl, threads=8, want_iters=True):
    env = engine_env(m, wl)
    got = env.expand(wl.nodes, want_state=True, want_iters=want_iters)
    env.close()
    ref = O.expand(oracle_env(wl), wl.nodes, threads=threads)
    return got, ref


def _small_world(m, dim, control, seed, n_nodes=96, edge=48, potential=False, region=False, limits=True,
                 res=0.1):
    """A small random environment exercising one control flag."""
    W = m.workloads
    rng = np.random.default_rng(seed)
    grid = W.box_map([edge] * dim, res, 0.15, seed, side_m=(0.3, 1.2))
    base = control & 0x0F
    vals = [-1.0, 0.0, 1.0] if dim == 3 else [-1.0, -0.5, 0.0, 0.5, 1.0]
    U = W.grid_controls(vals, dim, yaw_rates=[-0.5, 0.0, 0.5] if control & 0x10 else None)
    nodes = W.random_frontier(grid, [0.0] * dim, res, n_nodes, seed + 1, control, 1.5, 0.5, 1.0, 0.5, 1.0, 0.5)
    # push a few nodes to the map border / outside and add signed zeros
    nodes[0, :4] = [0.0, -0.03, edge * res - 0.01, edge * res + 0.2][: 4]
    if base >= 0x03:
        nodes[dim, 4:8] = -0.0
    params = {}
    if limits:
        params.update({"v_max": 1.5, "a_max": 1.0, "j_max": 1.5})
        if control & 0x10:
            params["yaw_max"] = 0.6
    pot = None
    reg = None
    if potential:
        pot = W.potential_field(grid, res, 0.4, 0.4 if dim == 3 else None)
        params.update({"potential_weight": 0.5, "gradient_weight": 0.25})
    if region:
        lo = [0.5] * dim
        hi = [edge * res - 0.5] * dim
        reg = W.tunnel_region([edge] * dim, [0.0] * dim, res, lo, hi, 1.0)
    return W.Workload("small", dim, control, pot if potential else grid, [0.0] * dim, res, U, nodes, params,
                      potential=pot, region=reg)


# ---------------------------------------------------------------- known answers
def test_appendix_b_3d_acc(engine, oracle_lib):
    """SURVEY.md Appendix B micro-case (3D ACC) through the C ABI."""
    m = engine
    U = np.array([[1, -1, 0.5], [2, 0, 0], [1.5, -0.5, 0], [1, -0.5, 0]], dtype=float)
    env = m.EnvMap(3)
    env.setMap([0, 0, 0], [40, 40, 40], np.zeros(40 ** 3, np.int8), 0.1)
    env.set_control(m.ACC)
    env.set_u(U)
    env.set_v_max(2.0)
    nodes = oracle_lib.make_nodes(3, [[1, 2, 2]], vel=[[0.5, 0, -0.5]], t=[3])
    r = env.expand(nodes)
    assert r["status"].tolist() == [1, 3, 1, 1]
    assert [hex(int(h)) for h in r["hash"][[0, 2, 3]]] == ["0x29b55209a71350a4", "0x29b5520a2bcc5303",
                                                           "0x29b55209a7654187"]
    assert r["cost"][[0, 2, 3]].tolist() == [12.25, 12.5, 11.25]
    assert r["iters"].tolist() == [16, 0, 20, 16]
    assert r["state"][:, 0].tolist() == [2.0, 1.5, 1.75, 1.5, -1.0, 0.0, 1.0, -1.0, 0.5, 0, 0, 0, 0, 4.0]
    # the reference-shaped single-node call
    succ, cost, act = env.get_succ(m.Waypoint(3, m.ACC, pos=[1, 2, 2], vel=[0.5, 0, -0.5], t=3))
    assert act == [0, 2, 3] and cost == [12.25, 12.5, 11.25]
    assert succ[1].pos.tolist() == [2.25, 1.75, 1.5] and succ[1].t == 4.0
    env.close()


def test_first_expansion_corridor_start(engine):
    """first expansion of test_planner_2d's start node
    (free space around it): action 4 skipped, costs 10.25 / 10.5, hashes."""
    m = engine
    U = m.workloads.grid_controls([-0.5, 0, 0.5], 2)
    env = m.EnvMap(2)
    env.setMap([0, -5], [799, 199], np.zeros(799 * 199, np.int8), 0.05)
    env.set_control(m.ACC)
    env.set_u(U)
    env.set_v_max(1.0)
    env.set_a_max(1.0)
    succ, cost, act = env.get_succ(m.Waypoint(2, m.ACC, pos=[2.5, -3.5]))
    assert act == [0, 1, 2, 3, 5, 6, 7, 8]
    assert cost == [10.5, 10.25, 10.5, 10.25, 10.25, 10.5, 10.25, 10.5]
    assert succ[0].pos.tolist() == [2.25, -3.75] and succ[0].vel.tolist() == [-0.5, -0.5]
    r = env.expand(np.array([[2.5], [-3.5], [0], [0], [0], [0], [0], [0], [0], [0]], dtype=float))
    want = [0x00028253dc64246a, 0x00028253dc6422e8, 0x00028253dc641beb, 0x00028253a2ec94b1, None,
            0x00028253a2ec9c5e, 0x00028253a3f54b0b, 0x00028253a3f46151, 0x00028253a3f47b3b]
    for i, h in enumerate(want):
        if h is not None:
            assert int(r["hash"][i]) == h
    assert int(r["hash"][4]) == 0x00028253a2ec9d98 and r["status"][4] == 0
    env.close()


# ------------------------------------------------------- every control, both dims
@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("control", [0x01, 0x03, 0x07, 0x0F])
@pytest.mark.parametrize("variant", ["plain", "potential", "region", "nolimits"])
def test_all_controls_bit_exact(engine, oracle_lib, dim, control, variant):
    wl = _small_world(engine, dim, control, seed=100 * dim + control, potential=(variant == "potential"),
                      region=(variant == "region"), limits=(variant != "nolimits"))
    got, ref = _run_both(engine, oracle_lib, wl)
    assert_slots_equal(got, ref, cost_rtol=0.0, what="dim%d ctrl0x%x %s" % (dim, control, variant))
    assert ref["stats"]["finite"] > 0 and ref["stats"]["emitted"] > ref["stats"]["finite"]


@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("control", [0x11, 0x13, 0x17, 0x1F])
@pytest.mark.parametrize("variant", ["plain", "potential"])
def test_yaw_controls(engine, oracle_lib, dim, control, variant):
    wl = _small_world(engine, dim, control, seed=300 * dim + control, potential=(variant == "potential"))
    got, ref = _run_both(engine, oracle_lib, wl)
    # identity bit-exact; cost within the north_star tolerance (device trig)
    assert_slots_equal(got, ref, cost_rtol=YAW_COST_RTOL, what="yaw dim%d ctrl0x%x %s" % (dim, control, variant))


# ------------------------------------------------- BASELINE configs at small scale
@pytest.mark.parametrize("name,scale,n_nodes", [("C2", 0.25, 1024), ("C3", 0.25, 512), ("C4", 0.125, 256),
                                                 ("C5", 0.2, 512)])
def test_baseline_configs_scaled(engine, oracle_lib, name, scale, n_nodes):
    wl = engine.workloads.make(name, scale=scale, n_nodes=n_nodes)
    got, ref = _run_both(engine, oracle_lib, wl)
    rtol = YAW_COST_RTOL if wl.control & 0x10 else 0.0
    assert_slots_equal(got, ref, cost_rtol=rtol, what=name)


# ------------------------------------------------------------------ edge cases
def test_empty_and_ragged_frontiers(engine, oracle_lib):
    wl = _small_world(engine, 3, 0x03, seed=7, n_nodes=130)
    env = engine_env(engine, wl)
    r0 = env.expand(wl.nodes[:, :0])
    assert r0["status"].size == 0
    for n in (1, 2, 63, 65, 130):
        sub = np.ascontiguousarray(wl.nodes[:, :n])
        got = env.expand(sub)
        ref = oracle_lib.expand(oracle_env(wl), sub)
        assert_slots_equal(got, ref, what="ragged n=%d" % n)
    env.close()


def test_single_control_and_get_succ_lists(engine, oracle_lib):
    wl = _small_world(engine, 2, 0x03, seed=11, n_nodes=40)
    env = engine_env(engine, wl)
    oenv = oracle_env(wl)
    ref = oracle_lib.expand(oenv, wl.nodes)
    nU = wl.U.shape[0]
    for k in range(0, 40, 7):
        wp = engine.Waypoint.from_row(2, wl.control, wl.nodes[:, k])
        succ, cost, act = env.get_succ(wp)
        st = ref["status"][k * nU:(k + 1) * nU]
        want_act = [i for i in range(nU) if st[i] in (1, 2)]
        assert act == want_act
        for j, i in enumerate(want_act):
            assert np.array_equal(succ[j].to_row(), ref["state"][:, k * nU + i])
            c = ref["cost"][k * nU + i]
            assert cost[j] == c or (np.isinf(cost[j]) and np.isinf(c))
    env.close()


def test_resident_buffers_match_host_path(engine, oracle_lib):
    wl = _small_world(engine, 3, 0x07, seed=21, n_nodes=200)
    env = engine_env(engine, wl)
    host = env.expand(wl.nodes, want_iters=True)
    fr = env.upload_frontier(wl.nodes)
    slots = env.alloc_slots(wl.n_nodes, want_state=True, want_iters=True)
    env.expand_resident(fr, slots)
    env.synchronize()
    dev = slots.download()
    for k in ("status", "cost", "hash", "state", "iters"):
        assert np.array_equal(host[k], dev[k], equal_nan=True), k
    slots.free()
    fr.free()
    env.close()


def test_device_math_matches_host_libm(engine):
    """The libm-class operations on the path: / sqrt round ceil must be
    correctly rounded (bit-equal to the host); cos/sin are reported in ULP."""
    env = engine.EnvMap(2)
    rng = np.random.default_rng(5)
    a = np.concatenate([rng.uniform(-60, 60, 200000), rng.uniform(-1e-3, 1e-3, 1000),
                        np.arange(-3000, 3000) * 0.01, np.arange(-600, 600) * 0.05])
    b = rng.choice([0.01, 0.1, 0.05, 0.2, 3.0, 7.0], size=a.size)
    assert np.array_equal(env.selftest_math(0, a, b), a / b)
    assert np.array_equal(env.selftest_math(1, np.abs(a)), np.sqrt(np.abs(a)))
    half = np.concatenate([a, np.arange(-2000, 2000) + 0.5, a / b - 0.5])
    want_round = np.where(half >= 0, np.floor(half + 0.5), np.ceil(half - 0.5))  # half away from zero
    frac_half = np.abs(half - np.trunc(half)) == 0.5
    want_round = np.where(frac_half, np.trunc(half) + np.sign(half), np.round(half))
    assert np.array_equal(env.selftest_math(4, half), want_round)
    assert np.array_equal(env.selftest_math(5, half), np.ceil(half))
    ang = np.concatenate([rng.uniform(-np.pi, np.pi, 200000), 0.5 * np.arange(-6, 7)])
    for op, f in ((2, np.cos), (3, np.sin)):
        d = env.selftest_math(op, ang)
        h = f(ang)
        ulp = np.abs(d - h) / np.spacing(np.abs(h))
        print("device %s vs host libm: %.4f%% differ, max %.2f ulp" % (f.__name__, 100 * np.mean(d != h), ulp.max()))
        assert ulp.max() <= 2.0
    env.close()


# ------------------------------------------- engine vs reference-made golden vectors
from helpers import engine_env_from_case, golden_cases  # noqa: E402

_GOLDEN = list(golden_cases())


@pytest.mark.parametrize("name,case,exp", _GOLDEN, ids=[c[0] for c in _GOLDEN])
def test_engine_reproduces_reference_golden_vectors(engine, name, case, exp):
    """The committed fixture was produced by the reference's own get_succ
    (tests/golden/make_golden.py); the HIP engine must reproduce it directly."""
    env = engine_env_from_case(engine, case)
    got = env.expand(case["nodes"])
    env.close()
    rtol = YAW_COST_RTOL if case["control"] & 0x10 else 0.0
    assert_slots_equal(got, exp, cost_rtol=rtol, what=name)
