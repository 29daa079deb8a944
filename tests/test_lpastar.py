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
