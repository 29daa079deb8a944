"""GPU: the heading-limit decision of validate_yaw (reference include/mpl_basis/primitive.h:504-525) is pinned to
the HOST libm.

The reference rejects a primitive when d = v_hat . (cos yaw, sin yaw) < cos(yaw_max) with std::cos / std::sin, i.e.
glibc's; the device's cos / sin (OCML) differ from glibc's in the last place on a few per cent of arguments, so a
decision within rounding noise of the threshold could go either way.  The engine flags such nodes and re-expands
them with trig values computed by the host (YawPin, csrc/mplx_internal.h).  Here the frontier is built ON the
threshold -- headings a few 2^-55 rad either side of yaw_max away from the velocity direction, at t = 0 and at
t = T -- and the successor SET must still be the reference build's (oracle/_ref: the reference's own headers compiled
by GCC, so its cos / sin pair is the fused glibc sincos), for the lists kernel, the dense kernel and the
dense -> lists route."""
import numpy as np
import pytest

from helpers import assert_lists_equal, assert_slots_equal, require_reference_build
from oracle import oracle as O

pytestmark = pytest.mark.gpu

YAW_COST_RTOL = 1e-6  # per-sample heading cost: device trig, continuous (north_star: costs within 1e-6)


def threshold_world(engine, n_each=1500, seed=3):
    """2D ACCxYAW on an empty map; nodes whose heading-limit decisions sit within a few ulp of the threshold."""
    m = engine
    rng = np.random.default_rng(seed)
    # a different heading per node: the device's cos / sin differ from glibc's on ~3 % of arguments, so with many
    # distinct arguments some of the decisions below really do depend on whose libm is asked
    yaw_max = 0.5
    yaw = np.round(rng.uniform(-2.6, 2.6, size=2 * n_each), 3)
    U = m.workloads.grid_controls([-1.0, 0.0, 1.0], 2, yaw_rates=[-0.5, 0.0, 0.5])
    delta = rng.integers(-48, 49, size=2 * n_each) * 2.0 ** -55       # a few ulp of an angle near 0.8
    sign = rng.choice([-1.0, 1.0], size=2 * n_each)                   # either side of the heading
    theta = yaw + sign * (yaw_max + delta)
    r = rng.uniform(0.6, 1.4, size=2 * n_each)
    w = np.stack([r * np.cos(theta), r * np.sin(theta)])              # velocity ON the threshold direction
    nodes = np.zeros((10, 2 * n_each))
    nodes[0] = rng.uniform(8.0, 12.0, size=2 * n_each).round(2)
    nodes[1] = rng.uniform(8.0, 12.0, size=2 * n_each).round(2)
    # first half: the node's own velocity is on the threshold (decision at t = 0, the same for every control)
    nodes[2:4, :n_each] = w[:, :n_each]
    # second half: vel(T) = v + u is on the threshold for the control u = -round(heading direction), yaw rate 0, while
    # vel(0) = v = w - u leans towards the heading and passes
    e = np.stack([np.cos(yaw[n_each:]), np.sin(yaw[n_each:])])
    nodes[2:4, n_each:] = w[:, n_each:] + np.round(e)
    nodes[8] = yaw
    nodes[9] = np.arange(2 * n_each) % 7
    grid = np.zeros((200, 200), np.int8)
    params = dict(v_max=3.0, yaw_max=yaw_max, dt=1.0)
    return dict(dim=2, control=m.ACCxYAW, U=U, grid=grid, map_dim=[200, 200], origin=[0.0, 0.0], res=0.1,
                params=params, nodes=nodes)


def make_env(m, wd):
    env = m.EnvMap(wd["dim"], 0)
    env.setMap(wd["origin"], wd["map_dim"], wd["grid"], wd["res"])
    env.set_control(wd["control"])
    env.set_u(wd["U"])
    for k, v in wd["params"].items():
        getattr(env, "set_" + k)(v)
    return env


def oracle_of(wd):
    return O.Env(wd["dim"], wd["control"], wd["U"], wd["grid"], wd["map_dim"], wd["origin"], wd["res"], **wd["params"])


def test_decisions_on_the_threshold_are_the_host_libms(engine, monkeypatch):
    wd = threshold_world(engine)
    n, nU = wd["nodes"].shape[1], wd["U"].shape[0]
    # against the REFERENCE BUILD (its headers as GCC compiles them: cos(w.yaw), sin(w.yaw) of primitive.h:519-520 fused
    # into one glibc sincos()) -- exactly the arithmetic the pinning reproduces; the restatement is not good enough here
    use_ref = require_reference_build()
    ref = O.expand(oracle_of(wd), wd["nodes"], threads=8, ref=use_ref)
    # the premise: the reference itself splits these nodes both ways
    st = ref["status"].reshape(n, nU)
    dead0 = np.all(st[: n // 2] != 1, axis=1)
    assert 0.2 < dead0.mean() < 0.8, dead0.mean()
    env = make_env(engine, wd)
    got = env.expand_lists(wd["nodes"], stride=32)
    assert env.last_lists_route() == "grid"
    assert_lists_equal(got, ref, n, nU, cost_rtol=YAW_COST_RTOL, what="threshold frontier, lists")
    flagged, passes = env.yaw_pin_stats()
    assert flagged >= n // 2 and passes >= 1, (flagged, passes)   # the threshold nodes went through the host libm
    # HBM-resident launch: final after synchronize
    fr = env.upload_frontier(wd["nodes"])
    lists = env.alloc_lists(n, want_state=True, want_iters=True)
    env.expand_lists_resident(fr, lists)
    env.synchronize()
    assert_lists_equal(lists.download(), ref, n, nU, cost_rtol=YAW_COST_RTOL, what="threshold frontier, resident lists")
    # dense kernel (slots) and the dense -> lists route
    dense = env.expand(wd["nodes"])
    assert_slots_equal(dense, ref, cost_rtol=YAW_COST_RTOL, what="threshold frontier, dense slots")
    env.set_lists_route("dense")
    got2 = env.expand_lists(wd["nodes"], stride=32)
    assert env.last_lists_route() == "dense"
    assert_lists_equal(got2, ref, n, nU, cost_rtol=YAW_COST_RTOL, what="threshold frontier, dense -> lists")
    for b in (lists, fr):
        b.free()
    env.close()
    # for the record: the same frontier on raw device trig (pinning off)
    monkeypatch.setenv("MPLX_YAW_PIN", "0")
    raw_env = make_env(engine, wd)
    raw = raw_env.expand(wd["nodes"], want_state=False)
    raw_env.close()
    flips = int(np.count_nonzero(raw["status"] != ref["status"]))
    print("threshold frontier: %d of %d pairs decided differently by raw device trig; 0 with the pinning" % (flips, n * nU))


@pytest.mark.parametrize("margin", ["0.02", "2.0"])
def test_wide_margin_sends_many_nodes_through_the_fix_pass(engine, oracle_lib, monkeypatch, margin):
    """MPLX_YAW_MARGIN widens the detection band: 0.02 flags a share of a C5-like frontier (recorded node list),
    2.0 flags every decision (more than the block records -> the whole launch is re-checked).  Either way the fix
    pass must reproduce the oracle."""
    from helpers import engine_env, oracle_env
    monkeypatch.setenv("MPLX_YAW_MARGIN", margin)
    wl = engine.workloads.make("C5", scale=0.1875, n_nodes=3000)
    nU, N = wl.U.shape[0], wl.n_nodes
    ref = oracle_lib.expand(oracle_env(wl), wl.nodes, threads=8)
    env = engine_env(engine, wl)
    got = env.expand_lists(wl.nodes, stride=96)
    assert env.last_lists_route() == "grid"
    assert_lists_equal(got, ref, N, nU, cost_rtol=YAW_COST_RTOL, what="C5 small, margin %s" % margin)
    flagged, passes = env.yaw_pin_stats()
    assert flagged > 50 and passes >= 1, (flagged, passes)
    if margin == "2.0":
        assert flagged >= N - 1
    dense = env.expand(wl.nodes)
    assert_slots_equal(dense, ref, cost_rtol=YAW_COST_RTOL, what="C5 small dense, margin %s" % margin)
    env.close()


def test_baseline_c5_needs_no_fix_pass(engine):
    """On the lattice of BASELINE's C5 every near-tie is the exact structural one (velocity along x, |yaw| == yaw_max),
    which any libm with an even cosine decides alike: nothing is flagged, the pinning costs nothing there."""
    from helpers import engine_env
    wl = engine.workloads.make("C5", scale=0.25, n_nodes=8192)
    env = engine_env(engine, wl)
    env.expand_lists(wl.nodes, want_state=False)
    assert env.yaw_pin_stats() == (0, 0)
    env.close()


def test_heading_cost_stays_within_a_few_ulp_of_the_reference(engine, oracle_lib):
    """The per-sample heading cost (env_map.h:121-129) normalises the velocity with v_rsq_f64 + two Newton steps instead of
    a square root and two divisions (mplx_device_common.h::heading_unit; its cos / sin are the device's anyway): the
    north-star tolerance for costs is 1e-6 relative -- what the kernels deliver is 1e-13, on the lists and on the dense
    slots alike (which share the function: they agree bit for bit, test_gpu_lists.py)."""
    from helpers import engine_env, oracle_env
    from test_gpu_parity import _small_world
    worst = 0.0
    for dim, control in ((2, 0x13), (3, 0x13), (2, 0x17), (3, 0x11)):
        wl = _small_world(engine, dim, control, seed=6100 + dim + control, n_nodes=400)
        wl.params["wyaw"] = 1.0
        ref = oracle_lib.expand(oracle_env(wl), wl.nodes, threads=8)
        env = engine_env(engine, wl)
        got = env.expand(wl.nodes)
        env.close()
        fin = ref["status"] == 1
        assert np.array_equal(got["status"], ref["status"]) and fin.sum() > 100
        rel = np.abs(got["cost"][fin] - ref["cost"][fin]) / np.abs(ref["cost"][fin])
        worst = max(worst, float(rel.max()))
    print("heading cost: worst relative difference to the reference arithmetic %.3g" % worst)
    assert worst < 1e-13
