"""GPU: expand_lex_kernel.hip -- the factorised kernel for lexicographic control tables without yaw on an occupancy
map (round 4) -- against the oracle and against expand_grid_kernel.hip (MPLX_GRID_LEX=0), which it must equal bit for bit:
count, action order, hash, cost, iteration count and the full successor state.

Covered on purpose: every (dimension, VEL / ACC / JRK) instantiation; nodes AT REST with a zero control in the table (the
one successor get_succ drops, env_map.h:158 -- its list position shifts everything behind it); search regions; a starved
row budget (several passes per node: the dropped lanes of pass 0 are replayed from LDS) and a starved box budget (samples
read from the blocked-bit map); odd worlds with nothing round in them; tiny grids with claimed chunks; the BASELINE
configurations at reduced size; the completion word of small synchronous launches."""
import numpy as np
import pytest

from helpers import assert_lists_equal, engine_env, odd_world, oracle_env
from test_gpu_parity import _small_world

pytestmark = pytest.mark.gpu


def _lists(engine, wl, expect_kernel, resident=False):
    env = engine_env(engine, wl)
    env.set_lists_route("grid")
    if resident:
        fr = env.upload_frontier(wl.nodes)
        lists = env.alloc_lists(wl.n_nodes, want_state=True, want_iters=True)
        env.expand_lists_resident(fr, lists)
        env.synchronize()
        got = lists.download()
        lists.free()
        fr.free()
    else:
        got = env.expand_lists(wl.nodes)
    assert env.last_lists_route() == "grid" and env.last_grid_kernel() == expect_kernel
    env.close()
    return got


def _same_lists(a, b, what=""):
    assert np.array_equal(a["count"], b["count"]), what
    live = (np.arange(a["stride"])[None, :] < a["count"][:, None]).ravel()
    for key in ("action", "hash", "iters", "cost"):
        assert np.array_equal(a[key][live], b[key][live]), (what, key)
    assert np.array_equal(a["state"][:, live].view(np.uint64), b["state"][:, live].view(np.uint64)), what


@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("control", [0x01, 0x03, 0x07])
@pytest.mark.parametrize("variant", ["plain", "region"])
def test_lex_kernel_against_the_oracle_and_the_general_kernel(engine, oracle_lib, monkeypatch, dim, control, variant):
    wl = _small_world(engine, dim, control, seed=7000 + 10 * dim + control, region=(variant == "region"), n_nodes=300)
    # a tenth of the nodes at rest (and, for JRK, without acceleration): the zero control reproduces the node
    rest = np.arange(0, wl.n_nodes, 10)
    wl.nodes[dim:4 * dim, rest] = 0.0
    ref = oracle_lib.expand(oracle_env(wl), wl.nodes, threads=8)
    got = _lists(engine, wl, "lex")
    assert_lists_equal(got, ref, wl.n_nodes, wl.U.shape[0], what="lex dim%d ctrl0x%x %s" % (dim, control, variant))
    st = ref["status"].reshape(wl.n_nodes, -1)
    assert (st[rest] == 0).any(), "no node reproduced itself: the dropped-successor path was not exercised"
    monkeypatch.setenv("MPLX_GRID_LEX", "0")
    old = _lists(engine, wl, "grid")
    _same_lists(got, old, "lex vs grid dim%d ctrl0x%x %s" % (dim, control, variant))


@pytest.mark.parametrize("rmax,boxcap", [("1", None), (None, "8"), ("2", "40")])
def test_lex_kernel_small_lds_budgets(engine, oracle_lib, monkeypatch, rmax, boxcap):
    if rmax:
        monkeypatch.setenv("MPLX_GRID_RMAX", rmax)
    if boxcap:
        monkeypatch.setenv("MPLX_GRID_BOXCAP", boxcap)
    for dim, control, region in ((3, 0x03, False), (2, 0x07, True), (3, 0x07, True), (2, 0x01, False)):
        wl = _small_world(engine, dim, control, seed=3100 + dim + control, region=region, n_nodes=120)
        wl.nodes[dim:4 * dim, ::7] = 0.0
        ref = oracle_lib.expand(oracle_env(wl), wl.nodes, threads=8)
        got = _lists(engine, wl, "lex")
        assert_lists_equal(got, ref, wl.n_nodes, wl.U.shape[0], what="lex rmax=%s boxcap=%s dim%d ctrl0x%x" % (
            rmax, boxcap, dim, control))


@pytest.mark.parametrize("seed", [0, 1, 2, 8, 9, 10, 16, 17, 18, 24, 26, 32, 34])
def test_lex_kernel_odd_worlds(engine, oracle_lib, seed):
    wl, control, pot = odd_world(engine, seed, 160)
    if control & 0x18 or pot is not None:
        pytest.skip("yaw / SNP / potential map: the general kernel's")
    ref = oracle_lib.expand(oracle_env(wl), wl.nodes, threads=8)
    got = _lists(engine, wl, "lex", resident=bool(seed & 1))
    assert_lists_equal(got, ref, wl.n_nodes, wl.U.shape[0], what="lex odd world %d" % seed)


@pytest.mark.parametrize("name,scale,n_nodes", [("C2", 0.25, 3000), ("C3", 0.25, 1500), ("C4", 0.125, 700)])
def test_lex_kernel_baseline_configs_scaled(engine, oracle_lib, monkeypatch, name, scale, n_nodes):
    wl = engine.workloads.make(name, scale=scale, n_nodes=n_nodes)
    ref = oracle_lib.expand(oracle_env(wl), wl.nodes, threads=8)
    got = _lists(engine, wl, "lex", resident=True)
    assert_lists_equal(got, ref, wl.n_nodes, wl.U.shape[0], what="lex " + name)
    # tiny grid + claimed chunks: every node exactly once whatever the dealing
    monkeypatch.setenv("MPLX_GRID_BLOCKS", "5")
    monkeypatch.setenv("MPLX_GRID_CHUNK", "3")
    again = _lists(engine, wl, "lex", resident=True)
    _same_lists(got, again, "lex %s on 5 workgroups, chunks of 3" % name)


def test_lex_kernel_is_not_taken_outside_its_scope(engine, monkeypatch):
    W = engine.workloads
    # yaw, potential map, SNP, a shuffled table: the general kernel
    wl = W.make("C5", scale=0.125, n_nodes=64)
    env = engine_env(engine, wl)
    env.expand_lists(wl.nodes)
    assert env.last_lists_route() == "grid" and env.last_grid_kernel() == "grid"
    env.close()
    wl = W.make("C2", scale=0.125, n_nodes=64)
    wl.U = np.ascontiguousarray(wl.U[np.random.default_rng(1).permutation(wl.U.shape[0])])
    env = engine_env(engine, wl)
    env.expand_lists(wl.nodes)
    assert env.last_lists_route() == "grid" and env.last_grid_kernel() == "grid"
    env.close()
    wl = W.make("C2", scale=0.125, n_nodes=64)
    env = engine_env(engine, wl)
    env.expand_lists(wl.nodes)
    assert env.last_grid_kernel() == "lex"
    env.set_lists_route("tile")
    env.expand_lists(wl.nodes)
    assert env.last_grid_kernel() == "none"
    env.close()


@pytest.mark.parametrize("case", ["n_max_62", "unbounded_velocity", "9261_controls", "8000_controls_in_scope"])
def test_lex_scope_exits_still_give_the_reference_lists(engine, oracle_lib, case):
    """The edges of the lexicographic kernel's scope (expand_lex_kernel.hip header; plan_grid in mplx_api.cpp): more than 61
    samples per primitive (v_max * dt / res > 60), no velocity bound (v_max <= 0: the sample count has no bound either)
    and more than 8 192 controls leave the factorised route altogether -- AUTO must pick a kernel that covers them and the
    lists must still be the reference's; 8 000 controls (20^3) are the largest cubic table still inside."""
    W = engine.workloads
    if case in ("n_max_62", "unbounded_velocity"):
        wl = _small_world(engine, 3, 0x03, seed=9100, n_nodes=64)
        wl.params["v_max"] = 6.15 if case == "n_max_62" else -1.0  # ceil(6.15 / 0.1) + 1 = 63 > 61
        if case == "n_max_62":
            wl.nodes[3:6, :8] = np.array([[6.0, -6.0, 5.5, 0.0, 6.1, -6.1, 3.0, 6.0]] * 3) * np.array([[1.0], [0.5], [-1.0]])
    else:
        k = 21 if case == "9261_controls" else 20
        wl = _small_world(engine, 3, 0x03, seed=9200, n_nodes=24)
        wl.U = W.grid_controls(np.linspace(-1.0, 1.0, k), 3)
        assert wl.U.shape[0] == k ** 3
    ref = oracle_lib.expand(oracle_env(wl), wl.nodes, threads=8)
    env = engine_env(engine, wl)
    got = env.expand_lists(wl.nodes)
    route, kernel = env.last_lists_route(), env.last_grid_kernel()
    env.close()
    if case == "8000_controls_in_scope":
        assert route == "grid" and kernel == "lex", (route, kernel)
    else:
        assert route in ("tile", "dense") and kernel == "none", (case, route, kernel)
    assert_lists_equal(got, ref, wl.n_nodes, wl.U.shape[0], what="scope exit %s (route %s)" % (case, route))
    st = ref["status"]
    assert (st == 1).any() and (st == 2).any()


def test_lex_kernel_single_node_calls_and_empty_lists(engine, oracle_lib):
    """get_succ-sized calls (one node, the completion word) and nodes without any successor (every entry of an axis over
    the limit: nA = 0) through the host-pointer entry point."""
    wl = _small_world(engine, 2, 0x03, seed=4242, n_nodes=40)
    wl.nodes[2:4, 5] = 50.0  # far over v_max on both axes: no successor at all
    oenv = oracle_env(wl)
    env = engine_env(engine, wl)
    env.set_lists_route("grid")
    for k in (0, 5, 17, 39):
        one = np.ascontiguousarray(wl.nodes[:, k:k + 1])
        got = env.expand_lists(one)
        assert env.last_grid_kernel() == "lex"
        assert_lists_equal(got, oracle_lib.expand(oenv, one, threads=1), 1, wl.U.shape[0], what="single node %d" % k)
    assert env.expand_lists(np.ascontiguousarray(wl.nodes[:, 5:6]))["count"][0] == 0
    env.close()


@pytest.mark.parametrize("dim,control,n_vals,n_nodes", [(2, 0x03, 17, 200), (2, 0x07, 25, 120), (3, 0x03, 17, 60), (3, 0x01, 20, 40),
                                                         (3, 0x03, 11, 80), (2, 0x03, 32, 64)])
def test_lex_kernel_wide_control_tables(engine, oracle_lib, dim, control, n_vals, n_nodes):
    """Lattices finer than the tests' du = u / 2: 17 .. 32 values per axis (one or two DPP rows per axis, a 3D node's 96
    entries in two rounds of phase T1) and tables of more than 1 024 controls (11^3, 17^3, 20^3) -- only the lexicographic
    kernel takes those; before round 4 they ran the 3.4 x slower workgroup-per-node kernel."""
    wl = _small_world(engine, dim, control, seed=9100 + n_vals + dim, n_nodes=n_nodes)
    vals = np.linspace(-1.0, 1.0, n_vals)
    wl.U = engine.workloads.grid_controls(list(vals), dim)
    wl.nodes[dim:4 * dim, ::9] = 0.0  # nodes at rest: the zero control (odd n_vals) reproduces them
    ref = oracle_lib.expand(oracle_env(wl), wl.nodes, threads=8)
    env = engine_env(engine, wl)
    env.set_lists_route("grid")  # (AUTO sends batches of <= 512 nodes with >= 512 controls to the workgroup-per-node kernel)
    got = env.expand_lists(wl.nodes)
    assert env.last_lists_route() == "grid" and env.last_grid_kernel() == "lex"
    env.set_lists_route("tile")
    try:
        tile = env.expand_lists(wl.nodes)
    except Exception:  # the tiled kernel's own scope may end below this table size
        tile = None
    env.close()
    assert_lists_equal(got, ref, wl.n_nodes, wl.U.shape[0], what="lex %dD ctrl0x%x %d values per axis" % (dim, control, n_vals))
    if tile is not None:
        _same_lists(got, tile, "lex vs tile, %d values per axis" % n_vals)
