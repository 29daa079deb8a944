"""GPU: BASELINE config C1 end to end on the engine -- MapPlanner.plan() with
get_succ served by the MI355X -- against the README pins and the oracle run."""
import os
import time

import numpy as np
import pytest

from oracle import oracle as O
from test_plan_known_answer import corridor, run_c1

pytestmark = pytest.mark.gpu
needs_ref = pytest.mark.skipif(not os.path.exists(O.REF_PLANNER_SO), reason="oracle/_ref/libmpl_ref_planner.so not built")


def _plan_on_gpu(m, batch):
    c = corridor()
    U = m.workloads.grid_controls([-0.5, 0.0, 0.5], 2)
    planner = m.MapPlanner(2, device=0)
    mu = m.MapUtil(2)
    mu.setMap(c["origin"], c["dim"], c["cells"], c["res"])
    planner.setMapUtil(mu)
    planner.setVmax(1.0)
    planner.setAmax(1.0)
    planner.setDt(1.0)
    planner.setU(U)
    planner.setBatch(batch)
    start = m.Waypoint(2, m.ACC, pos=c["start"])
    goal = m.Waypoint(2, m.ACC, pos=c["goal"])
    planner.plan(start, goal)  # warm-up (first launch loads the code object)
    t0 = time.perf_counter()
    ok = planner.plan(start, goal)
    dt = time.perf_counter() - t0
    s, traj, closed = planner.summary(), planner.getTraj(), planner.getCloseSet()
    planner.close()
    return ok, s, traj, closed, dt


@pytest.mark.parametrize("batch", [1, 64, 256])
def test_c1_plan_on_engine_matches_reference_pins(engine, batch):
    ok, s, traj, closed, dt = _plan_on_gpu(engine, batch)
    print("C1 plan() on MI355X, batch=%d: %.2f ms, %d launches, %d pairs" % (batch, dt * 1e3, s["device_launches"], s["pairs"]))
    assert ok and s["closed"] == 615 and closed.shape == (615, 2)
    assert traj.getTotalTime() == 35.0 and traj.J(engine.VEL) == 36.75 and traj.J(engine.ACC) == 1.5
    assert s["cost"] == 351.5
    # identical trajectory to the oracle-driven search
    ok_o, s_o, traj_o, _ = run_c1(engine, batch=1)
    assert np.array_equal(traj.actions, traj_o.actions) and np.array_equal(traj.nodes, traj_o.nodes)
    assert s["expansions"] == s_o["expansions"] and s["nodes"] == s_o["nodes"]


def test_c1_reference_planner_with_dropin_adapter(engine):
    """THE drop-in test: the reference's own MapPlanner / GraphSearch / StateSpace
    code, with only env_map::get_succ replaced by include/mplx_env_map.hpp over
    libmplx.so (oracle/_ref/libmpl_ref_planner.so, prebuilt where the reference
    tree exists).  Must reproduce README.md:199-202."""
    import os
    from oracle import oracle as O
    if not os.path.exists(O.REF_PLANNER_SO):
        pytest.skip("oracle/_ref/libmpl_ref_planner.so not built")
    c = corridor()
    U = engine.workloads.grid_controls([-0.5, 0.0, 0.5], 2)
    oenv = O.Env(2, O.ACC, U, c["cells"], c["dim"], c["origin"], c["res"], v_max=1.0, a_max=1.0, dt=1.0)
    start = engine.Waypoint(2, engine.ACC, pos=c["start"]).to_row()
    goal = engine.Waypoint(2, engine.ACC, pos=c["goal"]).to_row()
    gpu = O.ref_plan(oenv, start, goal, use_gpu=True, reps=2)
    cpu = O.ref_plan(oenv, start, goal, use_gpu=False, reps=2)
    print("reference MapPlanner::plan C1: CPU env_map %.2f ms, GpuMapPlanner adapter %.2f ms" % (cpu["wall_ms"], gpu["wall_ms"]))
    for k in ("ok", "closed", "opened", "expansions", "segments", "cost", "total_time", "J"):
        assert gpu[k] == cpu[k], k
    assert gpu["closed"] == 615 and gpu["total_time"] == 35.0 and gpu["J"][:2] == [36.75, 1.5]
    assert gpu["device_launches"] == 2 * 615  # reps=2, one launch per expansion
    # the adapter's speculative batching: same search, far fewer launches
    for batch in (8, 64):
        b = O.ref_plan(oenv, start, goal, use_gpu=batch, reps=1)
        print("  batch %d: %.2f ms, %d launches" % (batch, b["wall_ms"], b["device_launches"]))
        for k in ("ok", "closed", "opened", "expansions", "segments", "cost", "total_time", "J"):
            assert b[k] == cpu[k], (batch, k)
        assert b["device_launches"] < 615 // 2


@pytest.mark.parametrize("scenario", ["distance", "distance_yaw", "distance_iterative", "yaw", "prior_traj"])
def test_reference_test_scenarios_with_dropin_adapter(engine, scenario):
    """Every planner scenario of the reference's own test programs on corridor.yaml's map, run by the reference's
    MapPlanner on the CPU and by MPL::GpuMapPlanner (get_succ, updatePotentialMap and setSearchRegion on the MI355X):
      distance            test_distance_map_planner_2d.cpp:48-98   plan, then search region + potential map (config 5)
      distance_yaw        ..._with_yaw.cpp:48-104                  the same with ACCxYAW, U x 3 yaw rates, iterativePlan
      distance_iterative  ..._iterative.cpp:48-89                  ACC, iterativePlan
      yaw                 test_planner_2d_with_yaw.cpp:29-67       ACCxYAW from yaw = pi/2, yaw_max 0.7
      prior_traj          test_planner_2d_with_prior_traj.cpp      VEL plan, then a JRK-state plan guided by it
    Both stages must be the same search: set sizes, expansions, trajectory, J, region cells, potential values."""
    import os
    from oracle import oracle as O
    if not os.path.exists(O.REF_PLANNER_SO):
        pytest.skip("oracle/_ref/libmpl_ref_planner.so not built")
    c = corridor()
    U = engine.workloads.grid_controls([-0.5, 0.0, 0.5], 2)
    oenv = O.Env(2, O.ACC, U, c["cells"], c["dim"], c["origin"], c["res"], v_max=1.0, a_max=1.0, dt=1.0)
    start = engine.Waypoint(2, engine.ACC, pos=c["start"]).to_row()
    goal = engine.Waypoint(2, engine.ACC, pos=c["goal"]).to_row()
    cpu = O.ref_scenario(oenv, start, goal, scenario)
    assert cpu[0]["ok"] and (scenario == "yaw" or cpu[1]["ok"])
    yaw_stage = {"distance_yaw": (1,), "yaw": (0,)}.get(scenario, ())
    for batch in (1, 64):
        gpu = O.ref_scenario(oenv, start, goal, scenario, use_gpu=batch)
        last = 0 if scenario == "yaw" else 1
        print("%s batch=%d: last stage CPU %.1f ms (%d expansions), adapter %.1f ms, %d launches" % (
            scenario, batch, cpu[last]["wall_ms"], cpu[last]["expansions"], gpu[last]["wall_ms"],
            gpu[last]["device_launches"]))
        for stage in (0, 1):
            for k in ("ok", "closed", "opened", "expansions", "segments", "total_time", "J"):
                assert gpu[stage][k] == cpu[stage][k], (batch, stage, k, gpu[stage][k], cpu[stage][k])
            # yaw: device cos / sin against glibc's in the heading cost (tests/test_gpu_parity.py::YAW_COST_RTOL)
            tol = 1e-9 if stage in yaw_stage else 0.0
            assert abs(gpu[stage]["cost"] - cpu[stage]["cost"]) <= tol * abs(cpu[stage]["cost"])
            assert abs(gpu[stage]["traj_checksum"] - cpu[stage]["traj_checksum"]) <= tol * abs(cpu[stage]["traj_checksum"])
        assert gpu[1]["region_cells"] == cpu[1]["region_cells"] and gpu[1]["potential_sum"] == cpu[1]["potential_sum"]


@pytest.mark.parametrize("scenario", ["distance", "distance_yaw", "prior_traj"])
def test_dropin_adapter_through_a_base_class_pointer(engine, scenario):
    """INTEGRATION.md's caveat, exercised: an application that only swaps the constructor holds the drop-in as
    MapPlanner<Dim>*.  plan / setSearchRegion / updatePotentialMap / iterativePlan are not virtual in the reference,
    so the BASE versions run (host ray trace, host potential scatter, no device check in plan) and hand their results
    to the device env through what IS virtual or fingerprinted (set_potential_map, is_free, the region fingerprint);
    get_succ is the device's.  Every stage must still be the reference's search, and the device must really have
    served it (launches > 0)."""
    import os
    from oracle import oracle as O
    if not os.path.exists(O.REF_PLANNER_SO):
        pytest.skip("oracle/_ref/libmpl_ref_planner.so not built")
    c = corridor()
    U = engine.workloads.grid_controls([-0.5, 0.0, 0.5], 2)
    oenv = O.Env(2, O.ACC, U, c["cells"], c["dim"], c["origin"], c["res"], v_max=1.0, a_max=1.0, dt=1.0)
    start = engine.Waypoint(2, engine.ACC, pos=c["start"]).to_row()
    goal = engine.Waypoint(2, engine.ACC, pos=c["goal"]).to_row()
    cpu = O.ref_scenario(oenv, start, goal, scenario)
    yaw_stage = {"distance_yaw": (1,)}.get(scenario, ())
    for batch in (1, 64):
        gpu = O.ref_scenario(oenv, start, goal, scenario, use_gpu=batch, via_base=True)
        for stage in (0, 1):
            assert gpu[stage]["device_launches"] > 0
            for k in ("ok", "closed", "opened", "expansions", "segments", "total_time", "J"):
                assert gpu[stage][k] == cpu[stage][k], (batch, stage, k, gpu[stage][k], cpu[stage][k])
            tol = 1e-9 if stage in yaw_stage else 0.0
            assert abs(gpu[stage]["cost"] - cpu[stage]["cost"]) <= tol * abs(cpu[stage]["cost"])
            assert abs(gpu[stage]["traj_checksum"] - cpu[stage]["traj_checksum"]) <= tol * abs(cpu[stage]["traj_checksum"])
        assert gpu[1]["region_cells"] == cpu[1]["region_cells"] and gpu[1]["potential_sum"] == cpu[1]["potential_sum"]


_REF_PINS = {  # the reference's MapPlanner on the CPU (tests/test_plan_known_answer.py pins the same numbers)
    "distance": dict(closed=2732, cost=647.0999999999999, T=36.0, J=[40.08333333333333, 7.0]),
    "distance_iterative": dict(closed=3419, cost=617.45, T=37.0, J=[42.833333333333336, 7.75]),
    "distance_yaw": dict(closed=25326, cost=617.7304404847963, T=37.0, J=[42.833333333333336, 7.75]),
    "yaw": dict(closed=1342, cost=352.4275550988982, T=35.0, J=[36.666666666666664, 2.0]),
}


@pytest.mark.parametrize("scenario", sorted(_REF_PINS))
@pytest.mark.parametrize("batch", [1, 64])
def test_reference_test_scenarios_on_the_engine_planner(engine, scenario, batch):
    """The same scenarios on the engine's own planner (motion_primitive_library_amd.MapPlanner: host A* of
    csrc/host_planner.hpp, get_succ / updatePotentialMap / setSearchRegion on the MI355X, iterativePlan mirrored in
    planner.py) against what the reference's MapPlanner returns on the CPU: same closed set size, cost, duration
    and efforts, whatever the batch size."""
    m = engine
    c = corridor()
    vals = [-0.5, 0.0, 0.5]
    U = m.workloads.grid_controls(vals, 2)
    U_yaw = m.workloads.grid_controls(vals, 2, yaw_rates=[-0.5, 0.0, 0.5])

    def make(table):
        pl = m.MapPlanner(2, device=0)
        mu = m.MapUtil(2)
        mu.setMap(c["origin"], c["dim"], c["cells"].copy(), c["res"])
        pl.setMapUtil(mu)
        pl.setVmax(1.0)
        pl.setAmax(1.0)
        pl.setDt(1.0)
        pl.setU(table)
        pl.setBatch(batch)
        return pl

    t0 = time.perf_counter()
    if scenario == "yaw":
        pl = make(U_yaw)
        pl.setYawmax(0.7)
        ok = pl.plan(m.Waypoint(2, m.ACCxYAW, pos=c["start"], yaw=np.pi / 2), m.Waypoint(2, m.ACCxYAW, pos=c["goal"]))
    else:
        first = make(U)
        assert first.plan(m.Waypoint(2, m.ACC, pos=c["start"]), m.Waypoint(2, m.ACC, pos=c["goal"]))
        assert first.summary()["closed"] == 615
        traj = first.getTraj()
        assert traj.getWaypoints().shape == (36, 10)
        first.close()
        with_yaw = scenario == "distance_yaw"
        pl = make(U_yaw if with_yaw else U)
        pl.setEpsilon(1.0)
        pl.setSearchRadius([0.5, 0.5])
        if scenario == "distance":
            pl.setSearchRegion(traj.getWaypoints()[:, :2])
        pl.setPotentialRadius([1.0, 1.0])
        pl.setPotentialWeight(0.5)
        pl.setGradientWeight(0)
        pl.updatePotentialMap(c["start"])
        start = m.Waypoint(2, m.ACCxYAW if with_yaw else m.ACC, pos=c["start"])
        goal = m.Waypoint(2, m.ACC, pos=c["goal"])
        if with_yaw:
            pl.setYawmax(0.5)
        ok = pl.plan(start, goal) if scenario == "distance" else pl.iterativePlan(start, goal, traj, 10)
    dt = time.perf_counter() - t0
    s, tr = pl.summary(), pl.getTraj()
    pl.close()
    want = _REF_PINS[scenario]
    print("%s on the engine planner, batch=%d: %.1f ms, last plan %d expansions, %d launches" % (
        scenario, batch, dt * 1e3, s["expansions"], s["device_launches"]))
    assert ok and s["closed"] == want["closed"] and tr.getTotalTime() == want["T"]
    rtol = 1e-9 if "yaw" in scenario else 1e-15
    assert abs(s["cost"] - want["cost"]) <= rtol * want["cost"]
    assert tr.J(m.VEL) == want["J"][0] and tr.J(m.ACC) == want["J"][1]


@pytest.mark.parametrize("control,dim", [(0x01, 2), (0x03, 2), (0x07, 2), (0x0F, 2), (0x13, 2), (0x03, 3), (0x07, 3)])
def test_host_evaluated_states_equal_the_devices(engine, monkeypatch, control, dim):
    """The engine's search asks the device for (action, cost, hash) only and evaluates the state of a successor it
    has not seen before on the host (host_planner.hpp::forward_state).  MPLX_PLAN_CHECK_STATES=1 moves the device's
    states as well and counts every disagreement, bit for bit: there must be none, for every control order."""
    monkeypatch.setenv("MPLX_PLAN_CHECK_STATES", "1")
    m = engine
    rng = np.random.default_rng(control * 10 + dim)
    edge = 60 if dim == 2 else 28
    grid = m.workloads.box_map([edge] * dim, 0.1, 0.06, 31 + control, side_m=(0.3, 0.8))
    flat = grid.ravel().copy()
    pl = m.MapPlanner(dim, device=0)
    mu = m.MapUtil(dim)
    mu.setMap([0.0] * dim, [edge] * dim, flat, 0.1)
    pl.setMapUtil(mu)
    pl.setVmax(1.5)
    pl.setAmax(1.5)
    pl.setJmax(3.0)
    pl.setYawmax(0.9)
    pl.setDt(0.5)
    vals = [-1.0, 0.0, 1.0]
    pl.setU(m.workloads.grid_controls(vals, dim, yaw_rates=[-0.4, 0.0, 0.4] if control & 0x10 else None))
    pl.setBatch(32)
    pl.setEpsilon(0.0)  # uniform-cost search: many expansions whatever the start and the goal
    pl.setMaxNum(1500)
    free = np.argwhere(grid.reshape([edge] * dim) == 0)
    a, b = free[0][::-1], free[-1][::-1]  # opposite corners of the map
    start = m.Waypoint(dim, control, pos=(a + 0.5) * 0.1)
    goal = m.Waypoint(dim, control, pos=(b + 0.5) * 0.1)
    pl.plan(start, goal)
    s = pl.summary()
    pl.close()
    assert s["expansions"] > 50 and s["nodes"] > 150, s
    assert s["state_mismatches"] == 0, s


def test_state_mismatch_counter_counts(engine, monkeypatch):
    """Negative self-check of the check above: with one host-evaluated state corrupted on purpose
    (MPLX_PLAN_CHECK_PERTURB = index of the checked state that gets one bit flipped) the counter must report
    exactly one mismatch -- so a zero from the test above means the comparison really ran."""
    monkeypatch.setenv("MPLX_PLAN_CHECK_STATES", "1")
    monkeypatch.setenv("MPLX_PLAN_CHECK_PERTURB", "7")
    m = engine
    edge = 60
    grid = m.workloads.box_map([edge] * 2, 0.1, 0.06, 34, side_m=(0.3, 0.8))
    pl = m.MapPlanner(2, device=0)
    mu = m.MapUtil(2)
    mu.setMap([0.0] * 2, [edge] * 2, grid.ravel().copy(), 0.1)
    pl.setMapUtil(mu)
    pl.setVmax(1.5)
    pl.setDt(0.5)
    pl.setU(m.workloads.grid_controls([-1.0, 0.0, 1.0], 2))
    pl.setBatch(32)
    pl.setEpsilon(0.0)
    pl.setMaxNum(300)
    free = np.argwhere(grid.reshape([edge] * 2) == 0)
    a, b = free[0][::-1], free[-1][::-1]
    pl.plan(m.Waypoint(2, m.ACC, pos=(a + 0.5) * 0.1), m.Waypoint(2, m.ACC, pos=(b + 0.5) * 0.1))
    s = pl.summary()
    pl.close()
    assert s["state_mismatches"] == 1, s


def test_distance_map_planner_3d_with_yaw_three_ways(engine):
    """BASELINE config 5's planner (DistanceMapPlanner: potential map + ACCxYAW) as the reference's
    test_distance_map_planner_2d_with_yaw.cpp:48-104 runs it, on a VOXEL map: plan, then updatePotentialMap + 81 controls +
    iterativePlan in the tunnel (map_planner.cpp:394-430) -- the reference's MapPlanner on the CPU, the same planner with
    the drop-in adapter, and the engine's planner must agree on expansions, closed set, cost and the potential map."""
    from oracle import oracle as O
    if not os.path.exists(O.REF_PLANNER_SO):
        pytest.skip("oracle/_ref/libmpl_ref_planner.so not built")
    m = engine
    W = m.workloads
    edge, res = 60, 0.1
    grid = W.box_map([edge] * 3, res, 0.08, 4242, side_m=(0.5, 2.5))
    flat = grid.ravel()
    vals = [-1.0, 0.0, 1.0]
    U3, U3y = W.grid_controls(vals, 3), W.grid_controls(vals, 3, yaw_rates=[-0.5, 0.0, 0.5])

    def free_near(p):
        cc = np.array([int(x / res) for x in p])
        for r in range(0, 30):
            for d in np.ndindex(2 * r + 1, 2 * r + 1, 2 * r + 1):
                q = cc + np.array(d) - r
                if np.all(q >= 0) and np.all(q < edge) and flat[q[0] + edge * (q[1] + edge * q[2])] == 0:
                    return [(q[i] + 0.5) * res for i in range(3)]

    ps, pg = free_near([1.0, 1.0, 1.0]), free_near([edge * res - 1.0, edge * res - 1.2, edge * res - 1.5])
    oenv = O.Env(3, O.ACC, U3, flat, [edge] * 3, [0.0] * 3, res, v_max=2.0, a_max=2.0, dt=1.0)
    srow, grow = m.Waypoint(3, m.ACC, pos=ps).to_row(), m.Waypoint(3, m.ACC, pos=pg).to_row()
    cpu = O.ref_scenario(oenv, srow, grow, "distance_yaw")
    assert cpu[0]["ok"] and cpu[1]["ok"] and cpu[1]["expansions"] > 100
    for batch in (1, 64):
        gpu = O.ref_scenario(oenv, srow, grow, "distance_yaw", use_gpu=batch)
        for stage in (0, 1):
            for k in ("ok", "closed", "opened", "expansions", "segments", "total_time", "J"):
                assert gpu[stage][k] == cpu[stage][k], (batch, stage, k, gpu[stage][k], cpu[stage][k])
            assert abs(gpu[stage]["cost"] - cpu[stage]["cost"]) <= 1e-9 * abs(cpu[stage]["cost"])
        assert gpu[1]["region_cells"] == cpu[1]["region_cells"] and gpu[1]["potential_sum"] == cpu[1]["potential_sum"]

    def make(table):
        pl = m.MapPlanner(3, device=0)
        mu = m.MapUtil(3)
        mu.setMap([0.0] * 3, [edge] * 3, flat.copy(), res)
        pl.setMapUtil(mu)
        pl.setVmax(2.0)
        pl.setAmax(2.0)
        pl.setDt(1.0)
        pl.setU(table)
        pl.setBatch(64)
        return pl

    first = make(U3)
    assert first.plan(m.Waypoint(3, m.ACC, pos=ps), m.Waypoint(3, m.ACC, pos=pg))
    assert first.summary()["expansions"] == cpu[0]["expansions"]
    traj = first.getTraj()
    first.close()
    pl = make(U3y)
    pl.setEpsilon(1.0)
    pl.setSearchRadius([0.5] * 3)
    pl.setPotentialRadius([1.0] * 3)
    pl.setPotentialWeight(0.5)
    pl.setGradientWeight(0)
    pl.updatePotentialMap(ps)
    pl.setYawmax(0.5)
    assert pl.iterativePlan(m.Waypoint(3, m.ACCxYAW, pos=ps), m.Waypoint(3, m.ACC, pos=pg), traj, 10)
    s = pl.summary()
    pl.close()
    assert s["expansions"] == cpu[1]["expansions"] and s["closed"] == cpu[1]["closed"]
    assert abs(s["cost"] - cpu[1]["cost"]) <= 1e-9 * cpu[1]["cost"]


def test_search_consumes_the_heuristic_the_expansion_launch_computed(engine):
    """SURVEY.md 8f-2 with its consumer: the expansion launches of a plan() write the default heuristic of every
    successor next to its edge (mplx_set_goal + the `heur` row of the lists, fused into the kernels' list stores), and
    the search takes a new node's h from there instead of evaluating the successor's position itself
    (graph_search.h:84-88; MapPlanner.useDeviceHeuristic -- off by default: the row costs more on the PCIe link than
    the few evaluations it saves).  Same search as with the host evaluation, bit for bit: cost, expansions,
    closed and open set, node count, trajectory.  (Lists that outlive their launch are kept without the row: nodes
    created from those get the host evaluation, so not every node's h comes from the device.)"""
    m = engine
    W = m.workloads
    runs = {}
    for dim, edge, vals, ctrl in ((3, 56, np.linspace(-2.0, 2.0, 9), m.ACC), (2, 120, [-1.0, -0.5, 0.0, 0.5, 1.0], m.JRK)):
        grid = W.box_map([edge] * dim, 0.1, 0.07, 77 + dim, side_m=(0.4, 1.2))
        flat = grid.ravel().copy()
        free = np.argwhere(grid.reshape([edge] * dim) == 0)
        a, b = free[3][::-1], free[-3][::-1]
        for host in (False, True):
            pl = m.MapPlanner(dim, device=0)
            pl.useDeviceHeuristic(not host)
            mu = m.MapUtil(dim)
            mu.setMap([0.0] * dim, [edge] * dim, flat, 0.1)
            pl.setMapUtil(mu)
            pl.setVmax(2.0)
            pl.setAmax(2.0)
            pl.setJmax(4.0)
            pl.setDt(1.0)
            pl.setU(W.grid_controls(vals, dim))
            pl.setBatch(64)
            ok = pl.plan(m.Waypoint(dim, ctrl, pos=(a + 0.5) * 0.1), m.Waypoint(dim, ctrl, pos=(b + 0.5) * 0.1))
            s, t, tr = pl.summary(), pl.timing(), pl.getTraj()
            closed = pl.getCloseSet()
            pl.close()
            assert ok and s["expansions"] > 100
            runs[(dim, host)] = (s, t, tr, closed)
        (s0, t0, tr0, c0), (s1, t1, tr1, c1) = runs[(dim, False)], runs[(dim, True)]
        assert t0["heur_from_device"] > 0.2 * s0["nodes"] and t1["heur_from_device"] == 0, (t0, t1)
        for k in ("cost", "expansions", "closed", "opened", "nodes", "segments", "total_time", "J"):
            assert s0[k] == s1[k], (dim, k, s0[k], s1[k])
        assert np.array_equal(tr0.actions, tr1.actions) and np.array_equal(tr0.nodes, tr1.nodes) and np.array_equal(c0, c1)


def test_prior_trajectory_scenario_on_the_engine_planner_with_the_device(engine):
    """test_planner_2d_with_prior_traj.cpp on the engine's planner with get_succ on the MI355X (VEL plan, then the
    JRK-state plan guided by it: PlannerBase::setPriorTrajectory): what the reference's MapPlanner returns on the CPU
    (tests/test_plan_known_answer.py pins 628 closed nodes, cost 353.5, T = 35)."""
    from test_plan_known_answer import _prior_traj_scenario_on_the_engine_planner
    m = engine
    c = corridor()

    def make(control, U):
        pl = m.MapPlanner(2, device=0)
        mu = m.MapUtil(2)
        mu.setMap(c["origin"], c["dim"], c["cells"].copy(), c["res"])
        pl.setMapUtil(mu)
        pl.setVmax(1.0)
        pl.setAmax(1.0)
        pl.setDt(1.0)
        pl.setU(U)
        pl.setBatch(16)
        return pl

    ok, s1, s2, tr, s3 = _prior_traj_scenario_on_the_engine_planner(m, make)
    assert ok and s1["closed"] == 248 and s1["cost"] == 382.0
    assert s2["closed"] == 628 and s2["cost"] == 353.5 and tr.getTotalTime() == 35.0 and s2["J"][2] == 3.5
    assert s3["cost"] == 363.0 and s3["closed"] == 3598


@needs_ref
@pytest.mark.parametrize("gradient_weight", [0.0, 0.25])
def test_prior_trajectory_with_a_potential_map_on_the_device(engine, gradient_weight):
    """The prior-trajectory scenario with a potential map in the second planner (env_map.h:197-216, 241-249), everything on
    the MI355X: updatePotentialMap on the device, setPriorTrajectory reading the potential of the cells the prior passes
    from the device's map (mplx_read_cells), get_succ from the potential-map kernels.  Cost, closed set and expansions
    must be the reference's own MapPlanner's (oracle/_ref)."""
    m = engine
    c = corridor()
    U = m.workloads.grid_controls([-0.5, 0.0, 0.5], 2)
    oref = O.Env(2, O.ACC, U, c["cells"], c["dim"], c["origin"], c["res"], v_max=1.0, a_max=1.0, dt=1.0, gradient_weight=gradient_weight)
    ref = O.ref_scenario(oref, m.Waypoint(2, m.ACC, pos=c["start"]).to_row(), m.Waypoint(2, m.ACC, pos=c["goal"]).to_row(), "prior_traj_potential")

    def make(table):
        pl = m.MapPlanner(2, device=0)
        mu = m.MapUtil(2)
        mu.setMap(c["origin"], c["dim"], c["cells"].copy(), c["res"])
        pl.setMapUtil(mu)
        pl.setVmax(1.0)
        pl.setAmax(1.0)
        pl.setDt(1.0)
        pl.setU(table)
        pl.setBatch(64)
        return pl

    first = make(2.0 * U)
    assert first.plan(m.Waypoint(2, m.VEL, pos=c["start"]), m.Waypoint(2, m.VEL, pos=c["goal"]))
    second = make(U)
    second.setEpsilon(1.0)
    second.setW(10)
    second.setTol(0.5)
    second.setPotentialRadius([1.0, 1.0])
    second.setPotentialWeight(0.5)
    second.setGradientWeight(gradient_weight)
    pot = second.updatePotentialMap(c["start"])
    assert int(np.asarray(pot).astype(np.int64).sum()) == ref[1]["potential_sum"]
    # a few cells of the device's potential map, read back (mplx_read_cells)
    probe = np.array([0, 1234, pot.size // 2, pot.size - 1], dtype=np.int64)
    assert np.array_equal(second.env.read_cells(probe, potential=True), np.asarray(pot).ravel()[probe])
    second.setPriorTrajectory(first)  # (the env holds a potential map: its values come from the device)
    ok = second.plan(m.Waypoint(2, m.JRK, pos=c["start"]), m.Waypoint(2, m.VEL, pos=c["goal"]))
    s2 = second.summary()
    first.close()
    second.close()
    assert ok == ref[1]["ok"]
    for k in ("closed", "expansions", "opened", "cost", "total_time", "segments", "J"):
        assert s2[k] == ref[1][k], (k, s2[k], ref[1][k])
