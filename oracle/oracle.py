"""ctypes front-end of the CPU parity oracle (oracle/libmpl_oracle.so).

TEST INFRASTRUCTURE, NOT PRODUCT: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module.  The shipped package
(motion_primitive_library_amd) never does.

The same front-end drives oracle/_ref/libmpl_ref.so (the reference's own
headers compiled against stand-in Eigen/Boost headers), which exports the same
C interface (oracle/mpl_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

SKIP_SAME, FINITE, BLOCKED, SKIP_DYN = 0, 1, 2, 3

# Control::Control (reference include/mpl_basis/control.h:10-20)
VEL, ACC, JRK, SNP = 0x01, 0x03, 0x07, 0x0F
VELxYAW, ACCxYAW, JRKxYAW, SNPxYAW = 0x11, 0x13, 0x17, 0x1F


class _Env(C.Structure):
    _fields_ = [
        ("dim", C.c_int32), ("control", C.c_int32),
        ("dt", C.c_double), ("w", C.c_double), ("wyaw", C.c_double),
        ("v_max", C.c_double), ("a_max", C.c_double), ("j_max", C.c_double),
        ("yaw_max", C.c_double),
        ("potential_weight", C.c_double), ("gradient_weight", C.c_double),
        ("map_dim", C.c_int32 * 3), ("origin", C.c_double * 3), ("res", C.c_double),
        ("map", C.c_void_p), ("potential", C.c_void_p), ("region", C.c_void_p),
        ("U", C.c_void_p), ("nU", C.c_int32), ("udim", C.c_int32),
    ]


class _Out(C.Structure):
    _fields_ = [("status", C.c_void_p), ("cost", C.c_void_p), ("hash", C.c_void_p),
                ("state", C.c_void_p), ("iters", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [("pairs", C.c_int64), ("emitted", C.c_int64), ("finite", C.c_int64),
                ("skip_same", C.c_int64), ("skip_dyn", C.c_int64), ("samples", C.c_int64),
                ("sum_finite_cost", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def build(ref=False):
    """(Re)build the oracle with its Makefile; returns the library path."""
    target = ["ref"] if ref else []
    subprocess.run(["make", "-s", "-C", HERE] + target, check=True)
    return os.path.join(HERE, "_ref", "libmpl_ref.so") if ref else os.path.join(HERE, "libmpl_oracle.so")


def _bind(path):
    lib = C.CDLL(path)
    lib.mpl_oracle_expand.restype = C.c_int
    lib.mpl_oracle_expand.argtypes = [C.POINTER(_Env), C.c_void_p, C.c_int64, C.POINTER(_Out),
                                      C.c_int, C.POINTER(Stats)]
    lib.mpl_oracle_time_expand.restype = C.c_double
    lib.mpl_oracle_time_expand.argtypes = [C.POINTER(_Env), C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                           C.POINTER(Stats)]
    lib.mpl_oracle_hash.restype = C.c_uint64
    lib.mpl_oracle_hash.argtypes = [C.c_int32, C.c_int32, C.c_void_p]
    lib.mpl_oracle_heur.restype = C.c_double
    lib.mpl_oracle_heur.argtypes = [C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    lib.mpl_oracle_goal_tol.restype = C.c_int32
    lib.mpl_oracle_goal_tol.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double,
                                        C.c_double]
    lib.mpl_oracle_loop_count.restype = C.c_int32
    lib.mpl_oracle_loop_count.argtypes = [C.c_double, C.c_int32]
    return lib


_LIBS = {}


def load(ref=False):
    key = "ref" if ref else "port"
    if key not in _LIBS:
        path = (os.path.join(HERE, "_ref", "libmpl_ref.so") if ref
                else os.path.join(HERE, "libmpl_oracle.so"))
        if not os.path.exists(path):
            build(ref=ref)
        _LIBS[key] = _bind(path)
    return _LIBS[key]


class Env:
    """Parameters + map of one environment (the oracle-side mirror of the
    reference's env_map<Dim>: env_base.h:368-404, env_map.h:288-296)."""

    def __init__(self, dim, control, U, map_cells, map_dim, origin, res, dt=1.0, w=10.0, wyaw=1.0,
                 v_max=-1.0, a_max=-1.0, j_max=-1.0, yaw_max=-1.0, potential=None, region=None,
                 potential_weight=0.1, gradient_weight=0.0):
        self.dim, self.control = int(dim), int(control)
        self.U = np.ascontiguousarray(U, dtype=np.float64)
        assert self.U.ndim == 2
        self.map = np.ascontiguousarray(map_cells, dtype=np.int8).ravel()
        self.map_dim = [int(x) for x in map_dim] + [1] * (3 - len(map_dim))
        self.origin = [float(x) for x in origin] + [0.0] * (3 - len(origin))
        self.res = float(res)
        ncell = int(np.prod(self.map_dim[: self.dim]))
        assert self.map.size == ncell, (self.map.size, ncell)
        self.potential = None if potential is None else np.ascontiguousarray(potential, dtype=np.int8).ravel()
        self.region = None if region is None else np.ascontiguousarray(region, dtype=np.uint8).ravel()
        self.dt, self.w, self.wyaw = float(dt), float(w), float(wyaw)
        self.v_max, self.a_max, self.j_max, self.yaw_max = map(float, (v_max, a_max, j_max, yaw_max))
        self.potential_weight, self.gradient_weight = float(potential_weight), float(gradient_weight)

    @property
    def n_fields(self):
        return 4 * self.dim + 2

    def _c(self):
        e = _Env()
        e.dim, e.control = self.dim, self.control
        e.dt, e.w, e.wyaw = self.dt, self.w, self.wyaw
        e.v_max, e.a_max, e.j_max, e.yaw_max = self.v_max, self.a_max, self.j_max, self.yaw_max
        e.potential_weight, e.gradient_weight = self.potential_weight, self.gradient_weight
        for i in range(3):
            e.map_dim[i] = self.map_dim[i]
            e.origin[i] = self.origin[i]
        e.res = self.res
        e.map = self.map.ctypes.data
        e.potential = None if self.potential is None else self.potential.ctypes.data
        e.region = None if self.region is None else self.region.ctypes.data
        e.U = self.U.ctypes.data
        e.nU, e.udim = self.U.shape
        return e


def make_nodes(dim, pos, vel=None, acc=None, jrk=None, yaw=None, t=None):
    """Pack per-node arrays into the field-major [4D+2][N] layout."""
    pos = np.atleast_2d(np.asarray(pos, dtype=np.float64))
    n = pos.shape[0]
    out = np.zeros((4 * dim + 2, n), dtype=np.float64)
    for k, a in enumerate((pos, vel, acc, jrk)):
        if a is not None:
            out[k * dim:(k + 1) * dim, :] = np.atleast_2d(np.asarray(a, dtype=np.float64)).T
    if yaw is not None:
        out[4 * dim, :] = np.asarray(yaw, dtype=np.float64)
    if t is not None:
        out[4 * dim + 1, :] = np.asarray(t, dtype=np.float64)
    return out


def expand(env, nodes, threads=1, ref=False, want_state=True):
    """Dense expansion. Returns dict(status, cost, hash, state, iters, stats)."""
    lib = load(ref=ref)
    nodes = np.ascontiguousarray(nodes, dtype=np.float64)
    assert nodes.shape[0] == env.n_fields
    n = nodes.shape[1]
    nslots = n * env.U.shape[0]
    res = {
        "status": np.zeros(nslots, dtype=np.uint8),
        "cost": np.zeros(nslots, dtype=np.float64),
        "hash": np.zeros(nslots, dtype=np.uint64),
        "state": np.zeros((env.n_fields, nslots), dtype=np.float64) if want_state else None,
        "iters": np.zeros(nslots, dtype=np.int32),
    }
    o = _Out()
    o.status, o.cost, o.hash = res["status"].ctypes.data, res["cost"].ctypes.data, res["hash"].ctypes.data
    o.state = res["state"].ctypes.data if want_state else None
    o.iters = res["iters"].ctypes.data
    st = Stats()
    ce = env._c()
    rc = lib.mpl_oracle_expand(C.byref(ce), nodes.ctypes.data, n, C.byref(o), int(threads), C.byref(st))
    if rc != 0:
        raise RuntimeError("mpl_oracle_expand failed: %d" % rc)
    res["stats"] = st.as_dict()
    return res


def time_expand(env, nodes, threads=1, reps=1, ref=False):
    lib = load(ref=ref)
    nodes = np.ascontiguousarray(nodes, dtype=np.float64)
    st = Stats()
    ce = env._c()
    sec = lib.mpl_oracle_time_expand(C.byref(ce), nodes.ctypes.data, nodes.shape[1], int(threads), int(reps),
                                     C.byref(st))
    if sec < 0:
        raise RuntimeError("mpl_oracle_time_expand failed")
    return sec, st.as_dict()


def lattice_hash(dim, control, wp, ref=False):
    wp = np.ascontiguousarray(wp, dtype=np.float64)
    assert wp.size == 4 * dim + 2
    return int(load(ref=ref).mpl_oracle_hash(dim, control, wp.ctypes.data))


def heur(dim, control, w, v_max, wp, goal, ref=False):
    wp = np.ascontiguousarray(wp, dtype=np.float64)
    goal = np.ascontiguousarray(goal, dtype=np.float64)
    return float(load(ref=ref).mpl_oracle_heur(dim, control, w, v_max, wp.ctypes.data, goal.ctypes.data))


def goal_tol(dim, wp, goal, tol_pos, tol_vel=-1.0, tol_acc=-1.0, tol_yaw=-1.0, ref=False):
    """env_map::is_goal without the ray trace; wp / goal: 4D+2 doubles."""
    a = np.ascontiguousarray(wp, dtype=np.float64)
    b = np.ascontiguousarray(goal, dtype=np.float64)
    return bool(load(ref).mpl_oracle_goal_tol(dim, a.ctypes.data, b.ctypes.data, float(tol_pos), float(tol_vel),
                                              float(tol_acc), float(tol_yaw)))


def check_edges(env, parents, actions, cell_cap=0, ref=False):
    """Batched is_free(Primitive) + intrinsic cost (+ linked cells) for edges (parent column, action)."""
    lib = load(ref)
    lib.mpl_oracle_check_edges.restype = C.c_int
    lib.mpl_oracle_check_edges.argtypes = [C.POINTER(_Env), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_int32]
    parents = np.ascontiguousarray(parents, dtype=np.float64)
    actions = np.ascontiguousarray(actions, dtype=np.int32)
    n = actions.size
    out = {"free": np.zeros(n, np.uint8), "cost": np.zeros(n, np.float64)}
    cells = cnt = None
    if cell_cap > 0:
        out["cells"] = np.zeros((n, int(cell_cap)), np.int32)
        out["cell_count"] = np.zeros(n, np.int32)
        cells, cnt = out["cells"].ctypes.data, out["cell_count"].ctypes.data
    ce = env._c()
    rc = lib.mpl_oracle_check_edges(C.byref(ce), parents.ctypes.data, actions.ctypes.data, n, out["free"].ctypes.data,
                                    out["cost"].ctypes.data, cells, cnt, int(cell_cap))
    if rc != 0:
        raise RuntimeError("mpl_oracle_check_edges failed: %d" % rc)
    return out


def loop_count(T, n, ref=False):
    return int(load(ref=ref).mpl_oracle_loop_count(float(T), int(n)))


# ---- the reference's own MapPlanner::plan (oracle/_ref/libmpl_ref_planner.so)
class RefPlanOut(C.Structure):
    _fields_ = [("ok", C.c_int32), ("closed", C.c_int32), ("opened", C.c_int32), ("expansions", C.c_int32),
                ("segments", C.c_int32), ("hm_size", C.c_int32), ("cost", C.c_double), ("total_time", C.c_double),
                ("J", C.c_double * 4), ("wall_ms", C.c_double)]


REF_SO = os.path.join(HERE, "_ref", "libmpl_ref.so")
REF_PLANNER_SO = os.path.join(HERE, "_ref", "libmpl_ref_planner.so")


def ref_plan(env, start_row, goal_row, use_gpu=False, epsilon=1.0, reps=1):
    """Runs the reference's unmodified MapPlanner<Dim>::plan (A*).  use_gpu=True
    swaps in MPL::GpuMapPlanner from include/mplx_env_map.hpp (the drop-in
    adapter over libmplx.so) -- needs a GPU."""
    lib = _LIBS.setdefault("ref_planner", C.CDLL(REF_PLANNER_SO))
    lib.mpl_ref_plan.restype = C.c_int
    lib.mpl_ref_plan.argtypes = [C.POINTER(_Env), C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int,
                                 C.POINTER(RefPlanOut)]
    s = np.ascontiguousarray(start_row, dtype=np.float64)
    g = np.ascontiguousarray(goal_row, dtype=np.float64)
    out = RefPlanOut()
    ce = env._c()
    rc = _LIBS["ref_planner"].mpl_ref_plan(C.byref(ce), s.ctypes.data, g.ctypes.data, int(use_gpu),  # > 1: batch size
                                            float(epsilon), int(reps), C.byref(out))
    if rc != 0:
        raise RuntimeError("mpl_ref_plan failed: %d" % rc)
    return {"ok": bool(out.ok), "closed": out.closed, "opened": out.opened, "expansions": out.expansions,
            "segments": out.segments, "cost": out.cost, "total_time": out.total_time, "J": list(out.J),
            "wall_ms": out.wall_ms, "device_launches": out.hm_size}


SCENARIOS = {"distance": 0, "distance_yaw": 1, "distance_iterative": 2, "yaw": 3, "prior_traj": 4, "prior_traj_potential": 5}


def ref_scenario(env, start_row, goal_row, scenario, use_gpu=False, via_base=False):
    """The scenarios of the reference's own test programs end to end, on the reference's MapPlanner:
    "distance" (test_distance_map_planner_2d.cpp), "distance_yaw" (..._with_yaw.cpp), "distance_iterative"
    (..._iterative.cpp), "yaw" (test_planner_2d_with_yaw.cpp; one stage), "prior_traj"
    (test_planner_2d_with_prior_traj.cpp).  env.U must be a {-u, 0, u}^D table (the tests': {-0.5, 0, 0.5}^2); Dim 3 runs
    the same flows on a voxel map.  use_gpu: MPL::GpuMapPlanner
    instead of MPL::MapPlanner in every stage (> 1: speculative batch size).  Returns the two stages' summaries."""
    lib = _LIBS.setdefault("ref_planner", C.CDLL(REF_PLANNER_SO))
    for fn in (lib.mpl_ref_scenario, lib.mpl_ref_scenario_via_base):
        fn.restype = C.c_int
        fn.argtypes = [C.POINTER(_Env), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(RefPlanOut),
                       C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    s = np.ascontiguousarray(start_row, dtype=np.float64)
    g = np.ascontiguousarray(goal_row, dtype=np.float64)
    out = (RefPlanOut * 2)()
    chk = (C.c_double * 2)()
    region, pot = C.c_int64(0), C.c_int64(0)
    ce = env._c()
    # via_base: the drop-in (MPL::GpuMapPlanner) held through a MapPlanner<Dim> pointer -- the reference's non-virtual
    # plan / setSearchRegion / updatePotentialMap / iterativePlan run, only get_succ is the device's
    fn = lib.mpl_ref_scenario_via_base if via_base else lib.mpl_ref_scenario
    rc = fn(C.byref(ce), s.ctypes.data, g.ctypes.data, int(use_gpu), SCENARIOS[scenario], out, chk,
            C.byref(region), C.byref(pot))
    if rc != 0:
        raise RuntimeError("mpl_ref_scenario failed: %d" % rc)
    res = []
    for i in range(2):
        o = out[i]
        res.append({"ok": bool(o.ok), "closed": o.closed, "opened": o.opened, "expansions": o.expansions,
                    "segments": o.segments, "cost": o.cost, "total_time": o.total_time, "J": list(o.J),
                    "wall_ms": o.wall_ms, "device_launches": o.hm_size, "traj_checksum": chk[i]})
    res[1]["region_cells"], res[1]["potential_sum"] = region.value, pot.value
    if hasattr(lib, "mpl_ref_last_prep_ms"):
        lib.mpl_ref_last_prep_ms.restype = C.c_double
        res[1]["potential_map_ms"] = float(lib.mpl_ref_last_prep_ms())  # wall time of updatePotentialMap (stage 2)
    return res


# ---- map preprocessing: restatement, and the reference's own MapPlanner via the shim
def _prep_lib(ref):
    """ref: False = the restatement, True = the reference's MapPlanner, "gpu" = the reference's
    MapPlanner API on MPL::GpuMapPlanner (the drop-in adapter; needs a GPU)."""
    if not ref:
        lib = load()
        fn_pot, fn_reg = lib.mpl_oracle_update_potential_map, lib.mpl_oracle_search_region
    else:
        if "ref_planner" not in _LIBS:
            ref_plan  # noqa: B018  (same library as ref_plan)
            _LIBS["ref_planner_prep"] = C.CDLL(REF_PLANNER_SO)
        lib = _LIBS.setdefault("ref_planner_prep", C.CDLL(REF_PLANNER_SO))
        if ref == "gpu":
            fn_pot, fn_reg = lib.mpl_gpu_update_potential_map, lib.mpl_gpu_search_region
        else:
            fn_pot, fn_reg = lib.mpl_ref_update_potential_map, lib.mpl_ref_search_region
    fn_pot.restype = C.c_int
    fn_pot.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                       C.c_double, C.c_void_p]
    fn_reg.restype = C.c_int
    fn_reg.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                       C.c_void_p]
    return fn_pot, fn_reg


def update_potential_map(grid, map_dim, origin, res, pos, radius, range_=None, power=1.0, ref=False):
    """MapPlanner::updatePotentialMap: returns the new int8 map (flat, x fastest)."""
    dim = len(map_dim)
    cells = np.ascontiguousarray(grid, dtype=np.int8).ravel()
    md = np.asarray(map_dim, dtype=np.int32)
    org = np.asarray(origin, dtype=np.float64)
    p = np.asarray(pos, dtype=np.float64)
    r = np.asarray(radius, dtype=np.float64)
    g = np.zeros(dim) if range_ is None else np.asarray(range_, dtype=np.float64)
    out = np.empty_like(cells)
    rc = _prep_lib(ref)[0](dim, cells.ctypes.data, md.ctypes.data, org.ctypes.data, float(res), p.ctypes.data,
                           r.ctypes.data, g.ctypes.data, float(power), out.ctypes.data)
    if rc != 0:
        raise RuntimeError("update_potential_map failed: %d" % rc)
    return out


def search_region(map_dim, origin, res, path, search_radius, dense=False, ref=False):
    """MapPlanner::setSearchRegion: returns one byte per cell (flat, x fastest)."""
    dim = len(map_dim)
    md = np.asarray(map_dim, dtype=np.int32)
    org = np.asarray(origin, dtype=np.float64)
    pts = np.ascontiguousarray(path, dtype=np.float64).reshape(-1, dim)
    sr = np.asarray(search_radius, dtype=np.float64)
    out = np.empty(int(np.prod(md)), dtype=np.uint8)
    rc = _prep_lib(ref)[1](dim, md.ctypes.data, org.ctypes.data, float(res), pts.ctypes.data, pts.shape[0],
                           int(bool(dense)), sr.ctypes.data, out.ctypes.data)
    if rc != 0:
        raise RuntimeError("search_region failed: %d" % rc)
    return out


def ref_lpastar_substate(env, start_row, goal_row, time_step):
    """The reference's LPA* with StateSpace::getSubStateSpace: plan, re-root the tree at way point `time_step`, plan
    again from there.  Returns the two plans' summaries."""
    lib = _LIBS.setdefault("ref_planner", C.CDLL(REF_PLANNER_SO))
    lib.mpl_ref_lpastar_substate.restype = C.c_int
    lib.mpl_ref_lpastar_substate.argtypes = [C.POINTER(_Env), C.c_void_p, C.c_void_p, C.c_int, C.POINTER(RefPlanOut),
                                             C.POINTER(C.c_double)]
    s = np.ascontiguousarray(start_row, dtype=np.float64)
    g = np.ascontiguousarray(goal_row, dtype=np.float64)
    out = (RefPlanOut * 2)()
    chk = (C.c_double * 2)()
    ce = env._c()
    rc = lib.mpl_ref_lpastar_substate(C.byref(ce), s.ctypes.data, g.ctypes.data, int(time_step), out, chk)
    if rc != 0:
        raise RuntimeError("mpl_ref_lpastar_substate failed: %d" % rc)
    return [{"ok": bool(o.ok), "closed": o.closed, "opened": o.opened, "expansions": o.expansions, "segments": o.segments,
             "cost": o.cost, "total_time": o.total_time, "J": list(o.J), "traj_checksum": chk[i]} for i, o in enumerate(out)]


def ref_lpastar_edit_substate(env, start_row, goal_row, box_half, time_step):
    """The reference's LPA*: plan, getLinkedNodes, block a box around the middle of the trajectory + updateBlockedNodes,
    getSubStateSpace(time_step), plan from way point `time_step`.  Returns the two plans' summaries and the number of
    edited cells."""
    lib = _LIBS.setdefault("ref_planner", C.CDLL(REF_PLANNER_SO))
    lib.mpl_ref_lpastar_edit_substate.restype = C.c_int
    lib.mpl_ref_lpastar_edit_substate.argtypes = [C.POINTER(_Env), C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                                  C.POINTER(RefPlanOut), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    s = np.ascontiguousarray(start_row, dtype=np.float64)
    g = np.ascontiguousarray(goal_row, dtype=np.float64)
    out = (RefPlanOut * 2)()
    chk = (C.c_double * 2)()
    ed = C.c_int64(0)
    ce = env._c()
    rc = lib.mpl_ref_lpastar_edit_substate(C.byref(ce), s.ctypes.data, g.ctypes.data, int(box_half), int(time_step), out, chk,
                                           C.byref(ed))
    if rc != 0:
        raise RuntimeError("mpl_ref_lpastar_edit_substate failed: %d" % rc)
    return [{"ok": bool(o.ok), "closed": o.closed, "opened": o.opened, "expansions": o.expansions, "segments": o.segments,
             "cost": o.cost, "total_time": o.total_time, "J": list(o.J), "traj_checksum": chk[i]} for i, o in enumerate(out)], ed.value


def ref_lpastar(env, start_row, goal_row, use_gpu=False, box_half=3):
    """The reference's LPA* (PlannerBase::setLPAstar) with a map edit between plans: plan, getLinkedNodes, block a box
    of (2 box_half + 1)^D free cells around the middle of the trajectory + updateBlockedNodes, plan, clear the box +
    updateClearedNodes, plan -- on the reference's MapPlanner (CPU) or on MPL::GpuMapPlanner (use_gpu; the edge work
    of getLinkedNodes / updateClearedNodes batched on the device).  Returns the three plans' summaries and the
    statistics of the voxel -> edge table."""
    lib = _LIBS.setdefault("ref_planner", C.CDLL(REF_PLANNER_SO))
    lib.mpl_ref_lpastar.restype = C.c_int
    lib.mpl_ref_lpastar.argtypes = [C.POINTER(_Env), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(RefPlanOut),
                                    C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    s = np.ascontiguousarray(start_row, dtype=np.float64)
    g = np.ascontiguousarray(goal_row, dtype=np.float64)
    out = (RefPlanOut * 3)()
    chk = (C.c_double * 3)()
    st = (C.c_int64 * 10)()
    ce = env._c()
    rc = lib.mpl_ref_lpastar(C.byref(ce), s.ctypes.data, g.ctypes.data, int(use_gpu), int(box_half), out, chk, st)
    if rc != 0:
        raise RuntimeError("mpl_ref_lpastar failed: %d" % rc)
    plans = []
    for i in range(3):
        o = out[i]
        plans.append({"ok": bool(o.ok), "closed": o.closed, "opened": o.opened, "expansions": o.expansions,
                      "segments": o.segments, "cost": o.cost, "total_time": o.total_time, "J": list(o.J),
                      "wall_ms": o.wall_ms, "traj_checksum": chk[i]})
    lp = np.array([st[4]], dtype=np.int64).view(np.float64)[0]
    table = {"cells": st[0], "entries": st[1], "checksum": st[2], "linked_points": st[3], "points_checksum": float(lp),
             "edited_cells": st[5], "get_linked_nodes_us": st[6], "update_cleared_us": st[7],
             # GpuMapPlanner only: map bytes moved to the device by (updateBlockedNodes + plan 2), (updateClearedNodes + plan 3)
             "replan_upload_bytes": [st[8], st[9]]}
    return plans, table
