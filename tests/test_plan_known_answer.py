"""CPU: BASELINE config C1 -- test_planner_2d on data/corridor.yaml, Control::ACC,
|U| = 9 -- run through the host search of libmplx.so with the CPU ORACLE plugged
in as the successor provider (plumbing, no GPU).

Pins (reference README.md:199-202, the console transcript of test_planner_2d):
    "MPL Planner expanded states: 615"     -> closed set size
    "Total time T: 35.000000"              -> trajectory duration
    "Total J:  J(VEL) = 36.750000, J(ACC) = 1.500000, J(JRK) = 0, J(SNP) = 0"
This is the one result the reference itself publishes for this path; the
oracle (and, when present, the reference's own get_succ from oracle/_ref) must
reproduce it under the restated A*.
"""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "corridor_map.npz")


def corridor():
    z = np.load(GOLD)
    n = int(z["n_cells"])
    occ = np.unpackbits(z["occupied_bits"])[:n].astype(bool)
    cells = np.where(occ, 100, 0).astype(np.int8)  # read_map.hpp:42-45: data > 0 -> 100 else 0
    return dict(cells=cells, dim=[int(x) for x in z["dim"]], origin=[float(x) for x in z["origin"]],
                res=float(z["resolution"]), start=z["start"], goal=z["goal"], raw_counts=z["raw_counts"])


def provider_from_oracle(oenv, ref=False):
    lib = O.load(ref=ref)
    ce = oenv._c()
    single = C.cast(lib.mpl_oracle_get_succ, C.c_void_p)
    batched = C.cast(lib.mpl_oracle_batch, C.c_void_p)
    user = C.cast(C.pointer(ce), C.c_void_p)
    return (single, batched, user), ce  # keep `ce` alive as long as the planner


def run_c1(engine, ref=False, batch=1, t_max=None):
    m = engine
    c = corridor()
    U = m.workloads.grid_controls([-0.5, 0.0, 0.5], 2)  # test_planner_2d.cpp:49-53
    oenv = O.Env(2, O.ACC, U, c["cells"], c["dim"], c["origin"], c["res"], v_max=1.0, a_max=1.0, dt=1.0)
    prov, keep = provider_from_oracle(oenv, ref=ref)
    planner = m.MapPlanner(2, provider=prov)
    mu = m.MapUtil(2)
    mu.setMap(c["origin"], c["dim"], c["cells"], c["res"])
    planner.setMapUtil(mu)
    planner.setVmax(1.0)
    planner.setAmax(1.0)
    planner.setDt(1.0)
    planner.setU(U)
    planner.setBatch(batch)
    if t_max is not None:
        planner.setTmax(t_max)
    start = m.Waypoint(2, m.ACC, pos=c["start"])
    goal = m.Waypoint(2, m.ACC, pos=c["goal"])
    ok = planner.plan(start, goal)
    s = planner.summary()
    traj = planner.getTraj()
    closed = planner.getCloseSet()
    planner.close()
    del keep
    return ok, s, traj, closed


def test_fixture_matches_the_reference_file():
    c = corridor()
    assert c["dim"] == [799, 199] and c["res"] == 0.05 and c["origin"] == [0.0, -5.0]
    assert c["start"].tolist() == [2.5, -3.5] and c["goal"].tolist() == [37.0, 2.5]
    assert c["raw_counts"].tolist() == [123934, 0, 35067]  # raw values {-1, 100}
    assert int((c["cells"] == 100).sum()) == 35067


def test_c1_known_answer_with_oracle(engine):
    ok, s, traj, closed = run_c1(engine)
    assert ok
    assert s["closed"] == 615 and closed.shape == (615, 2)   # README.md:200
    assert traj.getTotalTime() == 35.0                        # README.md:201
    assert traj.J(engine.VEL) == 36.75 and traj.J(engine.ACC) == 1.5  # README.md:202
    assert traj.J(engine.JRK) == 0.0 and traj.J(engine.SNP) == 0.0
    assert s["cost"] == 351.5 and s["segments"] == 35         # g = w*T + J(ACC) = 10*35 + 1.5
    assert s["device_launches"] == s["expansions"] and s["pairs"] == 9 * s["expansions"]


def test_c1_t_max_is_kept_and_ignored_like_the_reference_map_planner_does(engine):
    """PlannerBase::setTmax only reaches env_base::is_goal (env_base.h:24); env_map::is_goal (env_map.h:25-45) overrides it
    without the test, so a horizon shorter than the plan changes nothing in the reference's MapPlanner -- nor here."""
    ok, s, traj, _ = run_c1(engine, batch=16, t_max=3.0)
    assert ok and s["closed"] == 615 and traj.getTotalTime() == 35.0 and s["cost"] == 351.5


def test_c1_batched_expansion_gives_the_same_plan(engine):
    ok1, s1, t1, c1 = run_c1(engine, batch=1)
    ok2, s2, t2, c2 = run_c1(engine, batch=64)
    assert ok1 and ok2
    for k in ("closed", "expansions", "cost", "total_time", "segments", "nodes", "opened"):
        assert s1[k] == s2[k], k
    assert np.array_equal(t1.actions, t2.actions) and np.array_equal(t1.nodes, t2.nodes)
    assert s2["device_launches"] < s1["device_launches"] / 5  # far fewer provider launches


@pytest.mark.skipif(not os.path.exists(os.path.join(O.HERE, "_ref", "libmpl_ref.so")),
                    reason="oracle/_ref not built")
def test_c1_known_answer_with_reference_get_succ(engine):
    """Same search, successors from the reference's own env_map::get_succ."""
    ok, s, traj, _ = run_c1(engine, ref=True)
    assert ok and s["closed"] == 615 and traj.getTotalTime() == 35.0
    assert traj.J(engine.VEL) == 36.75 and traj.J(engine.ACC) == 1.5 and s["cost"] == 351.5


@pytest.mark.skipif(not os.path.exists(O.REF_PLANNER_SO), reason="oracle/_ref/libmpl_ref_planner.so not built")
def test_c1_reference_planner_itself_and_host_search_agree(engine):
    """The reference's unmodified MapPlanner::plan (its own A*, StateSpace and
    get_succ, compiled against the stand-in Eigen/Boost) on config C1: hits the
    README pins, and our restated host search expands exactly as many nodes."""
    c = corridor()
    U = engine.workloads.grid_controls([-0.5, 0.0, 0.5], 2)
    oenv = O.Env(2, O.ACC, U, c["cells"], c["dim"], c["origin"], c["res"], v_max=1.0, a_max=1.0, dt=1.0)
    start = engine.Waypoint(2, engine.ACC, pos=c["start"]).to_row()
    goal = engine.Waypoint(2, engine.ACC, pos=c["goal"]).to_row()
    r = O.ref_plan(oenv, start, goal, use_gpu=False)
    assert r["ok"] and r["closed"] == 615 and r["total_time"] == 35.0
    assert r["J"] == [36.75, 1.5, 0.0, 0.0] and r["cost"] == 351.5
    ok, s, traj, _ = run_c1(engine)
    assert s["expansions"] == r["expansions"] and s["opened"] == r["opened"] and s["segments"] == r["segments"]


@pytest.mark.parametrize("scenario,closed,cost,T,region", [
    ("distance", 2732, 647.0999999999999, 36.0, 17351),
    ("distance_yaw", 25326, 617.7304404847963, 37.0, 20911),
    ("distance_iterative", 3419, 617.45, 37.0, 20911),
    ("yaw", 1342, 352.4275550988982, 35.0, 0),
    ("prior_traj", 628, 353.5, 35.0, 0)])
def test_reference_test_scenarios_on_the_cpu(scenario, closed, cost, T, region):
    """The reference's own MapPlanner through the scenarios of its test programs (oracle/_ref, where built): the
    numbers the GPU drop-in is compared with in tests/test_gpu_plan.py, pinned here so that a change of the
    stand-in headers or of the shim shows up."""
    import motion_primitive_library_amd as m
    if not os.path.exists(O.REF_PLANNER_SO):
        pytest.skip("oracle/_ref/libmpl_ref_planner.so not built")
    c = corridor()
    U = m.workloads.grid_controls([-0.5, 0.0, 0.5], 2)
    oenv = O.Env(2, O.ACC, U, c["cells"], c["dim"], c["origin"], c["res"], v_max=1.0, a_max=1.0, dt=1.0)
    start = m.Waypoint(2, m.ACC, pos=c["start"]).to_row()
    goal = m.Waypoint(2, m.ACC, pos=c["goal"]).to_row()
    r = O.ref_scenario(oenv, start, goal, scenario)
    last = r[0] if scenario == "yaw" else r[1]
    if scenario.startswith("distance"):  # stage 1 is test_planner_2d itself: README.md:199-202
        assert r[0]["closed"] == 615 and r[0]["total_time"] == 35.0 and r[0]["J"][:2] == [36.75, 1.5]
    assert last["ok"] and last["closed"] == closed and last["total_time"] == T
    assert r[1]["region_cells"] == region and abs(last["cost"] - cost) <= 1e-12 * cost


def test_c1_children_that_ride_along_change_nothing_but_the_launch_count(engine, monkeypatch):
    """Speculation on states that are not nodes yet (host_planner.hpp): the children of the first nodes of a launch,
    evaluated on the host, are expanded in the same launch and served later by hash + bitwise equal state.  Same
    expansions, same closed set, same trajectory; fewer provider calls."""
    monkeypatch.setenv("MPLX_PLAN_SPEC", "0")
    ok0, s0, t0, c0 = run_c1(engine, batch=16)
    launches = {0: s0["device_launches"]}
    for parents in (1, 4, 16):
        monkeypatch.setenv("MPLX_PLAN_SPEC", str(parents))
        ok, s, t, c = run_c1(engine, batch=16)
        assert ok and ok0
        for k in ("expansions", "closed", "opened", "nodes", "cost"):
            assert s[k] == s0[k], (parents, k)
        assert np.array_equal(c, c0)
        assert t.getTotalTime() == t0.getTotalTime()
        launches[parents] = s["device_launches"]
    assert launches[0] == 75 and launches[1] < 65 and launches[4] < 55 and launches[16] < 45
    monkeypatch.delenv("MPLX_PLAN_SPEC")
    ok, s, t, c = run_c1(engine, batch=16)      # the default: on for a 9-control table
    assert s["device_launches"] == launches[4] and s["closed"] == 615



def _prior_traj_scenario_on_the_engine_planner(m, make_planner):
    """test_planner_2d_with_prior_traj.cpp:29-102 on the engine's own planner: a VEL plan with unit controls, then a plan
    with position + velocity + acceleration in the state (jerk control) guided by it (PlannerBase::setPriorTrajectory).
    make_planner(control, U) -> a configured MapPlanner (provider of the caller's choice)."""
    c = corridor()
    vals = [-0.5, 0.0, 0.5]
    U = m.workloads.grid_controls(vals, 2)
    first = make_planner(m.VEL, 2.0 * U)
    assert first.plan(m.Waypoint(2, m.VEL, pos=c["start"]), m.Waypoint(2, m.VEL, pos=c["goal"]))
    s1 = first.summary()
    second = make_planner(m.JRK, U)
    second.setEpsilon(1.0)
    second.setW(10)
    second.setTol(0.5)
    second.setPriorTrajectory(first)
    ok = second.plan(m.Waypoint(2, m.JRK, pos=c["start"]), m.Waypoint(2, m.VEL, pos=c["goal"]))  # the goal keeps the VEL flag
    s2, tr = second.summary(), second.getTraj()
    # ... and without the prior trajectory the same planner searches towards the goal itself
    second.setPriorTrajectory(None)
    assert second.plan(m.Waypoint(2, m.JRK, pos=c["start"]), m.Waypoint(2, m.VEL, pos=c["goal"]))
    s3 = second.summary()
    first.close()
    second.close()
    return ok, s1, s2, tr, s3


@pytest.mark.skipif(not os.path.exists(O.REF_PLANNER_SO), reason="oracle/_ref/libmpl_ref_planner.so not built")
def test_prior_trajectory_heuristic_on_the_engine_planner(engine):
    """env_base::get_heur with a prior trajectory (env_base.h:46-52, env_map::set_prior_trajectory env_map.h:189-226) in
    csrc/host_planner.hpp, successors from the CPU oracle: the reference's own MapPlanner on the scenario of its test
    program closes 628 nodes for cost 353.5 (pinned above); so must the engine's search."""
    m = engine
    c = corridor()
    keep = []

    def make(control, U):
        oenv = O.Env(2, control, U, c["cells"], c["dim"], c["origin"], c["res"], v_max=1.0, a_max=1.0, dt=1.0)
        prov, ce = provider_from_oracle(oenv)
        keep.append((oenv, ce))
        pl = m.MapPlanner(2, provider=prov)
        mu = m.MapUtil(2)
        mu.setMap(c["origin"], c["dim"], c["cells"], c["res"])
        pl.setMapUtil(mu)
        pl.setVmax(1.0)
        pl.setAmax(1.0)
        pl.setDt(1.0)
        pl.setU(U)
        return pl

    ok, s1, s2, tr, s3 = _prior_traj_scenario_on_the_engine_planner(m, make)
    U = m.workloads.grid_controls([-0.5, 0.0, 0.5], 2)
    oenv = O.Env(2, O.ACC, U, c["cells"], c["dim"], c["origin"], c["res"], v_max=1.0, a_max=1.0, dt=1.0)
    ref = O.ref_scenario(oenv, m.Waypoint(2, m.ACC, pos=c["start"]).to_row(), m.Waypoint(2, m.ACC, pos=c["goal"]).to_row(), "prior_traj")
    assert ok and s1["closed"] == ref[0]["closed"] and s1["cost"] == ref[0]["cost"]
    for k in ("closed", "expansions", "opened", "cost", "total_time", "segments", "J"):
        assert s2[k] == ref[1][k], (k, s2[k], ref[1][k])
    assert s2["closed"] == 628 and s2["cost"] == 353.5 and tr.getTotalTime() == 35.0
    # without the prior trajectory the search ends in the goal's own region (with one, where the prior ends: env_base.h:295-298):
    # the reference's MapPlanner on that problem (O.ref_plan with the JRK-state start) closes 3 598 nodes for 363.0
    assert s3["cost"] == 363.0 and s3["closed"] == 3598 and s3["total_time"] == 36.0


@pytest.mark.skipif(not os.path.exists(O.REF_PLANNER_SO), reason="oracle/_ref/libmpl_ref_planner.so not built")
@pytest.mark.parametrize("gradient_weight", [0.0, 0.25])
def test_prior_trajectory_with_a_potential_map_on_the_engine_planner(engine, gradient_weight):
    """env_map::set_prior_trajectory / traverse_trajectory WITH potential_map_ (env_map.h:197-216, 241-249) in the
    engine's planner (mplx_planner_set_prior_trajectory_potential, round 6): the scenario of
    test_planner_2d_with_prior_traj.cpp with updatePotentialMap in the second planner before setPriorTrajectory -- the
    prior's remaining cost then carries potential_weight * value + gradient_weight * |vel| of the cells it passes.
    Successors from the CPU oracle; cost, closed set and expansions against the reference's own MapPlanner
    (the guided search re-opens closed nodes: 224 386 expansions for 147 216 closed at gradient weight 0)."""
    m = engine
    c = corridor()
    U = m.workloads.grid_controls([-0.5, 0.0, 0.5], 2)
    oref = O.Env(2, O.ACC, U, c["cells"], c["dim"], c["origin"], c["res"], v_max=1.0, a_max=1.0, dt=1.0, gradient_weight=gradient_weight)
    ref = O.ref_scenario(oref, m.Waypoint(2, m.ACC, pos=c["start"]).to_row(), m.Waypoint(2, m.ACC, pos=c["goal"]).to_row(), "prior_traj_potential")
    keep = []

    def make(control, table, cells, potential=None):
        oenv = O.Env(2, control, table, cells, c["dim"], c["origin"], c["res"], v_max=1.0, a_max=1.0, dt=1.0, potential=potential,
                     potential_weight=0.5, gradient_weight=gradient_weight)
        prov, ce = provider_from_oracle(oenv)
        keep.append((oenv, ce))
        pl = m.MapPlanner(2, provider=prov)
        mu = m.MapUtil(2)
        mu.setMap(c["origin"], c["dim"], cells, c["res"])
        pl.setMapUtil(mu)
        pl.setVmax(1.0)
        pl.setAmax(1.0)
        pl.setDt(1.0)
        pl.setU(table)
        return pl

    first = make(m.VEL, 2.0 * U, c["cells"])
    assert first.plan(m.Waypoint(2, m.VEL, pos=c["start"]), m.Waypoint(2, m.VEL, pos=c["goal"]))
    s1 = first.summary()
    # MapPlanner::updatePotentialMap (reference semantics; it rewrites the MapUtil's map, map_planner.cpp:387)
    pot = O.update_potential_map(c["cells"], c["dim"], c["origin"], c["res"], c["start"], [1.0, 1.0], ref=True)
    assert int(pot.astype(np.int64).sum()) == ref[1]["potential_sum"]
    second = make(m.JRK, U, pot, potential=pot)
    second.setEpsilon(1.0)
    second.setW(10)
    second.setTol(0.5)
    second.setPriorTrajectory(first, potential=pot, potential_weight=0.5, gradient_weight=gradient_weight)
    ok = second.plan(m.Waypoint(2, m.JRK, pos=c["start"]), m.Waypoint(2, m.VEL, pos=c["goal"]))
    s2 = second.summary()
    first.close()
    second.close()
    assert s1["closed"] == ref[0]["closed"] and s1["cost"] == ref[0]["cost"]
    assert ok == ref[1]["ok"]
    for k in ("closed", "expansions", "opened", "cost", "total_time", "segments", "J"):
        assert s2[k] == ref[1][k], (k, s2[k], ref[1][k])
    if gradient_weight == 0.0:
        assert s2["expansions"] > s2["closed"]  # (the prior's heuristic is inconsistent: closed nodes are re-opened)
