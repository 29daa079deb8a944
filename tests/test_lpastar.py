"""Incremental re-planning through the reference's own LPA* (PlannerBase::setLPAstar, graph_search.h:194-365) with a
map edit between plans: plan, MapPlanner::getLinkedNodes, block a box of cells on the trajectory +
updateBlockedNodes, plan, clear the box + updateClearedNodes, plan (map_planner.cpp:125-185).

CPU: the scenario on the reference's MapPlanner (oracle/_ref/libmpl_ref_planner.so), pinned.
GPU: the same scenario on MPL::GpuMapPlanner (include/mplx_env_map.hpp), whose getLinkedNodes and updateClearedNodes
run their edge work as ONE mplx_check_edges call each -- every plan and the voxel -> edge table must be identical
(SURVEY.md 8f-4 bound on the reference side)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle as O
from test_plan_known_answer import corridor

needs_ref = pytest.mark.skipif(not os.path.exists(O.REF_PLANNER_SO), reason="oracle/_ref/libmpl_ref_planner.so not built")


def corridor_problem(m):
    c = corridor()
    U = m.workloads.grid_controls([-0.5, 0.0, 0.5], 2)
    oenv = O.Env(2, O.ACC, U, c["cells"], c["dim"], c["origin"], c["res"], v_max=1.0, a_max=1.0, dt=1.0)
    return oenv, m.Waypoint(2, m.ACC, pos=c["start"]).to_row(), m.Waypoint(2, m.ACC, pos=c["goal"]).to_row()


def voxel_problem(m):
    W = m.workloads
    edge = 40
    grid = W.box_map([edge] * 3, 0.1, 0.05, 78, side_m=(0.3, 0.8))
    flat = grid.ravel()
    U = W.grid_controls([-1.0, 0.0, 1.0], 3)
    oenv = O.Env(3, O.ACC, U, flat, [edge] * 3, [0.0] * 3, 0.1, v_max=1.0, a_max=1.0, dt=1.0)

    def free_near(p):
        c = np.array([int(x / 0.1) for x in p])
        for r in range(0, 10):
            for d in np.ndindex(2 * r + 1, 2 * r + 1, 2 * r + 1):
                q = c + np.array(d) - r
                if np.all(q >= 0) and np.all(q < edge) and flat[q[0] + edge * (q[1] + edge * q[2])] == 0:
                    return [(q[i] + 0.5) * 0.1 for i in range(3)]
        raise RuntimeError("no free cell")

    return (oenv, m.Waypoint(3, m.ACC, pos=free_near([0.5, 0.5, 0.5])).to_row(),
            m.Waypoint(3, m.ACC, pos=free_near([3.0, 2.6, 2.2])).to_row())


PINS = {  # (closed, expansions, ok, cost) of the three plans; table (cells, entries, linked_points, edited_cells)
    "corridor": ([(615, 615, True, 351.5), (609, 14, False, float("inf")), (615, 14, True, 351.5)], (30648, 62136, 62136, 81), 4),
    "voxel": ([(88, 88, True, 44.0), (99, 7, False, float("inf")), (88, 6, True, 44.0)], (5247, 12143, 12143, 319), 3),
}


@needs_ref
@pytest.mark.parametrize("which", ["corridor", "voxel"])
def test_reference_lpastar_scenario_on_the_cpu(engine, which):
    oenv, s, g = (corridor_problem if which == "corridor" else voxel_problem)(engine)
    plans_pin, table_pin, box = PINS[which]
    plans, table = O.ref_lpastar(oenv, s, g, use_gpu=False, box_half=box)
    assert [(p["closed"], p["expansions"], p["ok"], p["cost"]) for p in plans] == plans_pin
    assert (table["cells"], table["entries"], table["linked_points"], table["edited_cells"]) == table_pin


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("which", ["corridor", "voxel"])
def test_gpu_adapter_lpastar_is_identical(engine, which):
    oenv, s, g = (corridor_problem if which == "corridor" else voxel_problem)(engine)
    box = PINS[which][2]
    cpu_plans, cpu_table = O.ref_lpastar(oenv, s, g, use_gpu=False, box_half=box)
    gpu_plans, gpu_table = O.ref_lpastar(oenv, s, g, use_gpu=True, box_half=box)
    for a, b in zip(cpu_plans, gpu_plans):
        for k in ("ok", "closed", "opened", "expansions", "segments", "cost", "total_time", "J", "traj_checksum"):
            assert a[k] == b[k], (which, k, a[k], b[k])
    for k in ("cells", "entries", "checksum", "linked_points", "points_checksum", "edited_cells"):
        assert cpu_table[k] == gpu_table[k], (which, k, cpu_table[k], gpu_table[k])
    # a re-plan after an edit of k cells moves k cells to the device (mplx_edit_map: 9 bytes each), not the map
    k = gpu_table["edited_cells"]
    assert gpu_table["replan_upload_bytes"] == [9 * k, 9 * k], (gpu_table["replan_upload_bytes"], k, oenv.map.size)
    assert cpu_table["replan_upload_bytes"] == [0, 0]
    print("%s: getLinkedNodes %d us on the CPU, %d us through the adapter (%d edges); updateClearedNodes %d / %d us" % (
        which, cpu_table["get_linked_nodes_us"], gpu_table["get_linked_nodes_us"], cpu_table["entries"],
        cpu_table["update_cleared_us"], gpu_table["update_cleared_us"]))


@needs_ref
@pytest.mark.gpu
def test_gpu_adapter_reports_a_device_failure_instead_of_no_path(engine):
    """GpuMapPlanner::plan on a device that does not exist: plan() is false AND the failure is queryable
    (deviceOk / deviceError) -- a device error must not look like "no trajectory exists"."""
    oenv, s, g = corridor_problem(engine)
    lib = C.CDLL(O.REF_PLANNER_SO)
    lib.mpl_gpu_plan_on_device.restype = C.c_int
    lib.mpl_gpu_plan_on_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int32),
                                           C.POINTER(C.c_int32), C.c_char_p, C.c_int]
    ce = oenv._c()
    for device, want_ok in ((0, True), (4096, False)):
        ok, dev_ok = C.c_int32(-1), C.c_int32(-1)
        err = C.create_string_buffer(512)
        assert lib.mpl_gpu_plan_on_device(C.byref(ce), s.ctypes.data, g.ctypes.data, device, C.byref(ok), C.byref(dev_ok),
                                          err, 512) == 0
        assert bool(ok.value) == want_ok and bool(dev_ok.value) == want_ok
        assert (err.value == b"") == want_ok, err.value


@pytest.mark.gpu
@needs_ref
def test_device_failure_through_a_base_class_pointer_is_still_latched(engine):
    """The drop-in held as MapPlanner<2>*: PlannerBase::plan (not virtual) runs, so a device failure comes back as
    "no trajectory" -- but the latch is set and an application that asks the derived type learns why
    (hold the planner as GpuMapPlanner, or check deviceOk() after plan())."""
    oenv, s, g = corridor_problem(engine)
    lib = C.CDLL(O.REF_PLANNER_SO)
    lib.mpl_gpu_plan_on_device_via_base.restype = C.c_int
    lib.mpl_gpu_plan_on_device_via_base.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int32),
                                                    C.POINTER(C.c_int32), C.c_char_p, C.c_int]
    ce = oenv._c()
    for device, want_ok in ((0, True), (4096, False)):
        ok, dev_ok = C.c_int32(-1), C.c_int32(-1)
        err = C.create_string_buffer(512)
        assert lib.mpl_gpu_plan_on_device_via_base(C.byref(ce), s.ctypes.data, g.ctypes.data, device, C.byref(ok),
                                                   C.byref(dev_ok), err, 512) == 0
        assert bool(ok.value) == want_ok and bool(dev_ok.value) == want_ok
        assert (err.value == b"") == want_ok, err.value


# ---------------------------------------------------------------- the engine's own LPA* (csrc/host_lpastar.hpp, mplx_planner_*)
def _engine_lpastar_scenario(m, oenv, s_row, g_row, box, provider=None, batch=1):
    """The scenario of oracle/ref_planner_shim.cpp::run_lpastar on the engine's planner: plan, getLinkedNodes, block a box
    of free cells around the middle of the trajectory + updateBlockedNodes, plan, getLinkedNodes, clear + updateClearedNodes,
    plan.  provider: (single, batched, user, edges_fn) of another env implementation (CPU: the oracle); None: the MI355X."""
    D = oenv.dim
    cells = oenv.map  # (the very array the oracle env points at: edits must reach it)
    pl = m.MapPlanner(D, provider=provider[:3] if provider else None)
    mu = m.MapUtil(D)
    mu.setMap(oenv.origin[:D], oenv.map_dim[:D], cells, oenv.res)
    mu.cells = cells  # (no copy: a CPU provider's env reads this array)
    pl.setMapUtil(mu)
    pl.setVmax(oenv.v_max)
    pl.setAmax(oenv.a_max)
    pl.setDt(oenv.dt)
    pl.setU(oenv.U)
    pl.setBatch(batch)
    pl.setLPAstar(True)
    if provider:
        pl.setEdgeProvider(provider[3], provider[2])
    start, goal = m.Waypoint.from_row(D, m.ACC, s_row), m.Waypoint.from_row(D, m.ACC, g_row)
    plans = []

    def record(ok):
        s = pl.summary()
        tr = pl.getTraj()
        plans.append({"ok": ok, "closed": s["closed"], "opened": s["opened"], "expansions": s["expansions"], "cost": s["cost"],
                      "segments": s["segments"], "total_time": s["total_time"], "J": s["J"], "launches": s["device_launches"],
                      "wps": tr.getWaypoints() if ok else None})

    record(pl.plan(start, goal))
    pts, n_cells, n_entries = pl.getLinkedNodes()
    table = {"cells": n_cells, "entries": n_entries, "linked_points": len(pts)}
    # the box, as the shim picks it
    wps = plans[0]["wps"]
    res, org, dims = oenv.res, np.array(oenv.origin[:D]), np.array(oenv.map_dim[:D])
    c_round = lambda x: int(np.sign(x) * np.floor(abs(x) + 0.5))  # C round(): halves away from zero (numpy rounds them to even)
    to_cell = lambda p: np.array([c_round((p[i] - org[i]) / res - 0.5) for i in range(D)])  # MapUtil::floatToInt
    mid, sc, gc = to_cell(wps[len(wps) // 2][:D]), to_cell(s_row[:D]), to_cell(g_row[:D])
    w = 2 * box + 1
    edit = []
    for q in range(w ** D):
        r, pn = q, []
        for i in range(D):
            pn.append(mid[i] + (r % w) - box)
            r //= w
        pn = np.array(pn)
        if np.any(pn < 0) or np.any(pn >= dims):
            continue
        idx = int(pn[0] + dims[0] * (pn[1] + (dims[1] * pn[2] if D == 3 else 0)))
        if not (0 <= cells[idx] < 100):
            continue
        if np.all(np.abs(pn - sc) <= 2) or np.all(np.abs(pn - gc) <= 2):
            continue
        edit.append((pn, idx))
    table["edited_cells"] = len(edit)
    ecells = np.array([e[0] for e in edit], dtype=np.int32)
    eidx = np.array([e[1] for e in edit], dtype=np.int64)
    cells[eidx] = 100
    pl.setMapUtil(mu)
    pl.updateBlockedNodes(ecells, edit_map=False)
    record(pl.plan(start, goal))
    pl.getLinkedNodes(want_points=False)
    cells[eidx] = 0
    pl.setMapUtil(mu)
    pl.updateClearedNodes(ecells, edit_map=False)
    record(pl.plan(start, goal))
    pl.close()
    return plans, table


def _oracle_provider(oenv):
    lib = O.load()
    ce = oenv._c()
    user = C.cast(C.pointer(ce), C.c_void_p)
    return (C.cast(lib.mpl_oracle_get_succ, C.c_void_p), C.cast(lib.mpl_oracle_batch, C.c_void_p), user,
            C.cast(lib.mpl_oracle_check_edges, C.c_void_p)), ce


@needs_ref
@pytest.mark.parametrize("which", ["corridor", "voxel"])
@pytest.mark.parametrize("batch", [1, 16])
def test_engine_lpastar_equals_the_reference_on_the_cpu(engine, which, batch):
    """The engine's LPA* (csrc/host_lpastar.hpp) with the CPU oracle as get_succ and as edge checker: the three plans of the
    scenario and the voxel -> edge table against the reference's own LPA* (oracle/_ref)."""
    oenv, s, g = (corridor_problem if which == "corridor" else voxel_problem)(engine)
    oenv.map = oenv.map.copy()  # (edited in place by the scenario)
    box = PINS[which][2]
    ref_plans, ref_table = O.ref_lpastar(oenv, s, g, use_gpu=False, box_half=box)
    prov, keep = _oracle_provider(oenv)
    plans, table = _engine_lpastar_scenario(engine, oenv, s, g, box, provider=prov, batch=batch)
    del keep
    for k in ("cells", "entries", "linked_points", "edited_cells"):
        assert table[k] == ref_table[k], (k, table[k], ref_table[k])
    for i, (a, b) in enumerate(zip(plans, ref_plans)):
        for k in ("ok", "closed", "opened", "expansions", "cost"):
            assert a[k] == b[k], (which, "plan %d" % i, k, a[k], b[k])
        if a["ok"]:
            for k in ("segments", "total_time", "J"):
                assert a[k] == b[k], (which, i, k)
    assert [(p["closed"], p["expansions"], p["ok"], p["cost"]) for p in plans] == PINS[which][0]
    if batch > 1:
        assert plans[0]["launches"] < plans[0]["expansions"] / 3


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("which", ["corridor", "voxel"])
def test_engine_lpastar_on_the_device_equals_the_reference(engine, which):
    """The same scenario with get_succ and the edge work on the MI355X (mplx_planner_* with an attached context)."""
    oenv, s, g = (corridor_problem if which == "corridor" else voxel_problem)(engine)
    oenv.map = oenv.map.copy()  # (edited in place by the scenario)
    box = PINS[which][2]
    ref_plans, ref_table = O.ref_lpastar(oenv, s, g, use_gpu=False, box_half=box)
    plans, table = _engine_lpastar_scenario(engine, oenv, s, g, box, provider=None, batch=16)
    for k in ("cells", "entries", "linked_points", "edited_cells"):
        assert table[k] == ref_table[k], (k, table[k], ref_table[k])
    for i, (a, b) in enumerate(zip(plans, ref_plans)):
        for k in ("ok", "closed", "opened", "expansions", "cost"):
            assert a[k] == b[k], (which, "plan %d" % i, k, a[k], b[k])


@pytest.mark.parametrize("which", ["corridor", "voxel"])
def test_engine_lpastar_sub_state_space_re_roots_the_tree(engine, which):
    """StateSpace::getSubStateSpace (state_space.h:116-195) through mplx_planner_sub_state_space: after the robot has
    executed k primitives of the plan, the tree is re-rooted at way point k and the next plan from there costs what was
    left of the first one -- found by repairing the kept tree, not by a search from scratch."""
    m = engine
    oenv, s, g = (corridor_problem if which == "corridor" else voxel_problem)(m)
    prov, keep = _oracle_provider(oenv)
    D = oenv.dim
    pl = m.MapPlanner(D, provider=prov[:3])
    mu = m.MapUtil(D)
    mu.setMap(oenv.origin[:D], oenv.map_dim[:D], oenv.map, oenv.res)
    pl.setMapUtil(mu)
    pl.setVmax(oenv.v_max)
    pl.setAmax(oenv.a_max)
    pl.setDt(oenv.dt)
    pl.setU(oenv.U)
    pl.setLPAstar(True)
    pl.setEdgeProvider(prov[3], prov[2])
    assert pl.plan(m.Waypoint.from_row(D, m.ACC, s), m.Waypoint.from_row(D, m.ACC, g))
    first = pl.summary()
    tr = pl.getTraj()
    wps = tr.getWaypoints()
    k = 2
    # cost of the first k primitives: w * dt + J(ACC) each
    U = oenv.U
    spent = sum(oenv.w * oenv.dt + float((U[a] ** 2).sum()) * oenv.dt for a in tr.actions[:k])
    pl.getSubStateSpace(k)
    assert pl.plan(m.Waypoint.from_row(D, m.ACC, wps[k]), m.Waypoint.from_row(D, m.ACC, g))
    second = pl.summary()
    tr2 = pl.getTraj()
    pl.close()
    del keep
    if os.path.exists(O.REF_PLANNER_SO):
        # ... and it is the reference's own re-rooting (getSubStateSpace rebuilds the open list in its hash map's order):
        # both plans against MapPlanner::getSubStateSpace + plan of oracle/_ref
        ref = O.ref_lpastar_substate(oenv, s, g, k)
        for mine, theirs in ((first, ref[0]), (second, ref[1])):
            for key in ("ok", "closed", "opened", "expansions", "cost", "segments", "total_time", "J"):
                assert mine[key] == theirs[key], (which, key, mine[key], theirs[key])
    assert abs(second["cost"] - (first["cost"] - spent)) <= 1e-9 * first["cost"]
    assert second["segments"] == first["segments"] - k and np.array_equal(tr2.actions, tr.actions[k:])
    assert second["expansions"] < first["expansions"] / 4  # repaired, not searched again


@needs_ref
@pytest.mark.parametrize("which,box,k", [("corridor", 1, 3), ("corridor", 3, 3), ("corridor", 5, 6), ("voxel", 2, 2)])
def test_engine_lpastar_map_edit_then_re_rooting(engine, which, box, k):
    """plan, getLinkedNodes, updateBlockedNodes, getSubStateSpace(k), plan -- a map edit COMBINED with re-rooting.
    getSubStateSpace drops the nodes it does not reach from the hash map and clears the queue while a re-opened parent
    still lists them as successors; the reference looks every successor up again (graph_search.h:285-290) and starts a
    fresh State for a dropped one.  (An engine that relaxed the stored node instead erased a heap position of the
    cleared queue: memory corruption, round-5 advisor.)  Both plans against the reference's own LPA*."""
    m = engine
    oenv, s, g = (corridor_problem if which == "corridor" else voxel_problem)(m)
    oenv.map = oenv.map.copy()
    ref, ref_edited = O.ref_lpastar_edit_substate(oenv, s, g, box, k)
    prov, keep = _oracle_provider(oenv)
    D = oenv.dim
    cells = oenv.map
    pl = m.MapPlanner(D, provider=prov[:3])
    mu = m.MapUtil(D)
    mu.setMap(oenv.origin[:D], oenv.map_dim[:D], cells, oenv.res)
    mu.cells = cells
    pl.setMapUtil(mu)
    pl.setVmax(oenv.v_max)
    pl.setAmax(oenv.a_max)
    pl.setDt(oenv.dt)
    pl.setU(oenv.U)
    pl.setLPAstar(True)
    pl.setEdgeProvider(prov[3], prov[2])
    assert pl.plan(m.Waypoint.from_row(D, m.ACC, s), m.Waypoint.from_row(D, m.ACC, g))
    first = pl.summary()
    wps = pl.getTraj().getWaypoints()
    pl.getLinkedNodes(want_points=False)
    res, org, dims = oenv.res, np.array(oenv.origin[:D]), np.array(oenv.map_dim[:D])
    c_round = lambda x: int(np.sign(x) * np.floor(abs(x) + 0.5))
    to_cell = lambda p: np.array([c_round((p[i] - org[i]) / res - 0.5) for i in range(D)])
    mid, sc, gc = to_cell(wps[len(wps) // 2][:D]), to_cell(wps[k][:D]), to_cell(g[:D])
    w = 2 * box + 1
    ecells, eidx = [], []
    for q in range(w ** D):
        r, pn = q, []
        for i in range(D):
            pn.append(mid[i] + (r % w) - box)
            r //= w
        pn = np.array(pn)
        if np.any(pn < 0) or np.any(pn >= dims):
            continue
        idx = int(pn[0] + dims[0] * (pn[1] + (dims[1] * pn[2] if D == 3 else 0)))
        if not (0 <= cells[idx] < 100):
            continue
        if np.all(np.abs(pn - sc) <= 2) or np.all(np.abs(pn - gc) <= 2):
            continue
        ecells.append(pn)
        eidx.append(idx)
    assert len(ecells) == ref_edited
    cells[np.array(eidx, dtype=np.int64)] = 100
    pl.setMapUtil(mu)
    pl.updateBlockedNodes(np.array(ecells, dtype=np.int32), edit_map=False)
    pl.getSubStateSpace(k)
    ok = pl.plan(m.Waypoint.from_row(D, m.ACC, wps[k]), m.Waypoint.from_row(D, m.ACC, g))
    second = pl.summary()
    second["ok"], first["ok"] = ok, True
    pl.close()
    del keep
    for mine, theirs in ((first, ref[0]), (second, ref[1])):
        for key in ("ok", "closed", "opened", "expansions", "cost"):
            assert mine[key] == theirs[key], (which, box, k, key, mine[key], theirs[key])
        if mine["ok"]:
            for key in ("segments", "total_time", "J"):
                assert mine[key] == theirs[key], (which, key, mine[key], theirs[key])
