"""GPU: expand_pair_kernel.hip -- yaw controls on a potential map over a pre-screened frontier, two nodes per wave.

The kernel serves a launch only when the pre-screen ran (large frontiers); MPLX_GRID_PRESCREEN_MIN=1 makes every
frontier large, so small worlds reach it.  Every case is compared
  * with the reference / oracle (successor set, order, hash, state, iteration counts bit for bit; cost to north_star's
    1e-6, the heading cost uses device trig), and
  * with the general factorised kernel (MPLX_GRID_PAIR=0) on the same inputs, EVERYTHING bit for bit including the
    costs: the two kernels evaluate the same expressions
over Dim 2 / 3, ACCxYAW / JRKxYAW, 2 - 4 yaw rates (the kernel's three instantiations by yaw accumulators), with and without
heading cost, gradient weight, search region, odd survivor counts (the last wave task has one node), several passes
(one row per pass) and the fused heuristic / goal-flag rows.  Full size: tests/test_gpu_fullsize.py (C5, C5 tunnel, 2D)."""
import numpy as np
import pytest

from helpers import assert_lists_equal, engine_env, oracle_env
from test_gpu_parity import _small_world

pytestmark = pytest.mark.gpu
YAW_COST_RTOL = 1e-6


def _world(engine, dim, control, seed, n_nodes, yaw_rates, wyaw, grad, region):
    wl = _small_world(engine, dim, control, seed=seed, n_nodes=n_nodes, potential=True, region=region)
    vals = [-1.0, 0.0, 1.0]
    wl.U = engine.workloads.grid_controls(vals, dim, yaw_rates=yaw_rates)
    wl.params["wyaw"] = wyaw
    wl.params["gradient_weight"] = grad
    wl.params["yaw_max"] = 0.9  # (a wide limit: a good share of the frontier survives the pre-screen)
    return wl


def _lists(engine, wl, monkeypatch, pair, extra_env=None, goal=None):
    monkeypatch.setenv("MPLX_GRID_PRESCREEN_MIN", "1")
    monkeypatch.setenv("MPLX_GRID_PAIR", "1" if pair else "0")
    for k, v in (extra_env or {}).items():
        monkeypatch.setenv(k, v)
    env = engine_env(engine, wl)
    env.set_lists_route("grid")
    fr = env.upload_frontier(wl.nodes)
    lists = env.alloc_lists(wl.n_nodes, want_state=True, want_iters=True)
    hb = fb = None
    if goal is not None:
        ns = lists.n_slots
        hb, fb = engine.env.DeviceArray(env, ns * 8), engine.env.DeviceArray(env, ns)
        env.set_goal(goal, w=10.0, v_max=1.5, tol_pos=0.5)
        lists.heur, lists.flags = hb, fb
    env.expand_lists_resident(fr, lists)
    env.synchronize()
    kernel = env.last_grid_kernel()
    out = lists.download()
    if goal is not None:
        out["heur"] = hb.download(np.float64, (lists.n_slots,))
        out["flags"] = fb.download(np.uint8, (lists.n_slots,))
        lists.heur = lists.flags = None
        hb.free()
        fb.free()
    lists.free()
    fr.free()
    env.close()
    for k in ["MPLX_GRID_PRESCREEN_MIN", "MPLX_GRID_PAIR"] + list((extra_env or {}).keys()):
        monkeypatch.delenv(k, raising=False)
    return out, kernel


def _same_lists(a, b, n_nodes, what):
    """Two engine list sets: identical counts and, over the used prefix of every node's list, identical rows bit for bit."""
    assert np.array_equal(a["count"], b["count"]), what
    stride = int(a["stride"])
    used = (np.arange(stride)[None, :] < a["count"][:, None]).ravel()
    for k in ("action", "hash", "iters", "heur", "flags"):
        if a.get(k) is not None and b.get(k) is not None:
            x, y = np.asarray(a[k])[used], np.asarray(b[k])[used]
            if x.dtype.kind == "f":
                x, y = x.view(np.uint64), y.view(np.uint64)
            assert np.array_equal(x, y), "%s: %s differs" % (what, k)
    assert np.array_equal(a["cost"][used].view(np.uint64), b["cost"][used].view(np.uint64)), "%s: cost differs" % what
    assert np.array_equal(a["state"][:, used].view(np.uint64), b["state"][:, used].view(np.uint64)), "%s: state differs" % what


@pytest.mark.parametrize("dim,control", [(2, 0x13), (3, 0x13), (2, 0x17), (3, 0x17)])
@pytest.mark.parametrize("yaw_rates", [[0.3], [-0.5, 0.5], [-0.5, 0.0, 0.5], [-0.6, -0.2, 0.2, 0.6]])
@pytest.mark.parametrize("variant", ["heading", "plain", "gradient+region"])
def test_pair_kernel_equals_reference_and_general_kernel(engine, oracle_lib, monkeypatch, dim, control, yaw_rates, variant):
    wyaw = 0.0 if variant == "plain" else 1.0
    grad = 0.3 if variant == "gradient+region" else 0.0
    wl = _world(engine, dim, control, seed=7000 + 10 * dim + control + len(yaw_rates), n_nodes=333, yaw_rates=yaw_rates,
                wyaw=wyaw, grad=grad, region=(variant == "gradient+region"))
    nU = wl.U.shape[0]
    ref = oracle_lib.expand(oracle_env(wl), wl.nodes, threads=8)
    got, kernel = _lists(engine, wl, monkeypatch, pair=True)
    assert kernel == "pair", kernel
    what = "dim%d ctrl0x%x %d yaw rates %s" % (dim, control, len(yaw_rates), variant)
    assert_lists_equal(got, ref, wl.n_nodes, nU, cost_rtol=YAW_COST_RTOL, what=what)
    gen, kernel = _lists(engine, wl, monkeypatch, pair=False)
    assert kernel == "grid", kernel
    _same_lists(got, gen, wl.n_nodes, what + " pair vs general")
    st = ref["status"]
    assert np.count_nonzero(st == 1) > 20 and np.count_nonzero(st == 2) > 0


@pytest.mark.parametrize("n_nodes", [1, 2, 3, 64, 65, 257])
def test_pair_kernel_odd_survivor_counts(engine, oracle_lib, monkeypatch, n_nodes):
    """Any number of survivors: the last wave task may carry one node (its second group repeats it and stores nothing)."""
    wl = _world(engine, 3, 0x13, seed=7100 + n_nodes, n_nodes=300, yaw_rates=[-0.5, 0.0, 0.5], wyaw=1.0, grad=0.0, region=False)
    wl.nodes = np.ascontiguousarray(wl.nodes[:, 40:40 + n_nodes])
    ref = oracle_lib.expand(oracle_env(wl), wl.nodes, threads=4)
    got, kernel = _lists(engine, wl, monkeypatch, pair=True)
    assert kernel == "pair"
    assert_lists_equal(got, ref, wl.n_nodes, wl.U.shape[0], cost_rtol=YAW_COST_RTOL, what="%d nodes" % n_nodes)


def test_pair_kernel_several_passes_and_fused_rows(engine, oracle_lib, monkeypatch):
    """One row of cell codes per pass (MPLX_PAIR_RMAX=1: a node with several sample counts takes several passes), and the
    heuristic / goal-flag rows written by the launch (mplx_set_goal): both against the general kernel bit for bit, the
    lists against the reference."""
    wl = _world(engine, 3, 0x13, seed=7200, n_nodes=400, yaw_rates=[-0.5, 0.0, 0.5], wyaw=1.0, grad=0.0, region=False)
    goal = np.ascontiguousarray(wl.nodes[:, 5])
    ref = oracle_lib.expand(oracle_env(wl), wl.nodes, threads=8)
    one, kernel = _lists(engine, wl, monkeypatch, pair=True, extra_env={"MPLX_PAIR_RMAX": "1"}, goal=goal)
    assert kernel == "pair"
    assert_lists_equal(one, ref, wl.n_nodes, wl.U.shape[0], cost_rtol=YAW_COST_RTOL, what="one row per pass")
    gen, kernel = _lists(engine, wl, monkeypatch, pair=False, goal=goal)
    assert kernel == "grid"
    _same_lists(one, gen, wl.n_nodes, "one row per pass + fused rows, pair vs general")
    three, _ = _lists(engine, wl, monkeypatch, pair=True, goal=goal)
    _same_lists(one, three, wl.n_nodes, "one row per pass vs three")


def test_pair_kernel_leaves_the_scope_it_does_not_cover(engine, monkeypatch):
    """Five yaw rates (more accumulators than a lane carries), an occupancy map, or a table that is not lexicographic:
    the general kernel serves the launch."""
    wl = _world(engine, 2, 0x13, seed=7300, n_nodes=100, yaw_rates=[-0.8, -0.4, 0.0, 0.4, 0.8], wyaw=1.0, grad=0.0, region=False)
    _, kernel = _lists(engine, wl, monkeypatch, pair=True)
    assert kernel == "grid"
    wl = _world(engine, 2, 0x13, seed=7301, n_nodes=100, yaw_rates=[-0.5, 0.0, 0.5], wyaw=1.0, grad=0.0, region=False)
    wl.potential = None
    _, kernel = _lists(engine, wl, monkeypatch, pair=True)
    assert kernel == "grid"
    wl = _world(engine, 2, 0x13, seed=7302, n_nodes=100, yaw_rates=[-0.5, 0.0, 0.5], wyaw=1.0, grad=0.0, region=False)
    wl.U = np.ascontiguousarray(wl.U[np.random.default_rng(5).permutation(wl.U.shape[0])])  # (reversed would still be a nested-loop table)
    _, kernel = _lists(engine, wl, monkeypatch, pair=True)
    assert kernel == "grid"
