"""Host-side mirror of the reference's MapPlanner<Dim> (the caller of the hot
path): same method names and argument meaning as

    PlannerBase  reference include/mpl_planner/common/planner_base.h:170-325
    MapPlanner   reference include/mpl_planner/planner/map_planner.h,
                 src/mpl_planner/map_planner.cpp:14-18 (setMapUtil)

The A* itself is the C++ host search inside libmplx.so (csrc/host_planner.hpp);
successors come from the engine context (get_succ on the MI355X).  Tests may
plug another provider through set_provider() -- that is how the CPU oracle is
run under the very same search to pin config C1.
"""
import ctypes as C

import numpy as np

from . import _abi
from .env import EnvMap, Waypoint


class MapUtil:
    """MapUtil<Dim> as far as the planner needs it (map_util.h:84-90)."""

    def __init__(self, dim):
        self.dim = dim
        self.origin = self.map_dim = self.cells = self.res = None

    def setMap(self, origin, dim, cells, res):
        self.origin = [float(x) for x in origin]
        self.map_dim = [int(x) for x in dim]
        self.cells = np.ascontiguousarray(cells, dtype=np.int8).ravel()
        self.res = float(res)
        assert self.cells.size == int(np.prod(self.map_dim))


class Trajectory:
    """What the reference's tests read off Trajectory<Dim> (trajectory.h)."""

    def __init__(self, total_time, J, nodes, actions, cost, end=None):
        self._T, self._J, self.nodes, self.actions, self.cost, self.end = total_time, J, nodes, actions, cost, end

    def getWaypoints(self):
        """Rows of 4D+2: the start state of every primitive and the state the last one reaches
        (Trajectory::getWaypoints, trajectory.h)."""
        if self.end is None or len(self.nodes) == 0:
            return np.asarray(self.nodes)
        return np.vstack([self.nodes, self.end[None, :]])

    def getTotalTime(self):
        return self._T

    def J(self, control):
        order = {0x01: 0, 0x03: 1, 0x07: 2, 0x0F: 3}[control & 0x0F]
        return self._J[order]


class MapPlanner:
    def __init__(self, dim, device=0, verbose=False, provider=None):
        """provider=None: successors from the HIP engine on `device` (the product
        path; needs a GPU).  provider=(single_fn_ptr, batch_fn_ptr, user_ptr):
        raw C hooks of another env implementation (tests: the CPU oracle)."""
        self.dim = dim
        self._L = _abi.lib()
        p = C.c_void_p()
        rc = self._L.mplx_planner_create(dim, C.byref(p))
        if rc != 0:
            raise _abi.MplxError(rc, "mplx_planner_create failed")
        self._p = p
        self._cfg = _abi.PlannerConfig()
        # nodes per device launch: the result does not depend on it (get_succ is a pure function), the speed does
        self._cfg.control, self._cfg.max_expand, self._cfg.batch = 0x03, -1, (64 if provider is None else 1)
        self._cfg.dt, self._cfg.w, self._cfg.v_max, self._cfg.epsilon = 1.0, 10.0, -1.0, 1.0
        self._cfg.tol_pos, self._cfg.tol_vel, self._cfg.tol_acc, self._cfg.tol_yaw = 0.5, -1.0, -1.0, -1.0
        self.env = None
        self._keep = provider
        self._map_util = None
        self._search_radius = None
        self._potential_radius, self._potential_range, self._pow = None, None, 1.0
        self._traj = None
        if provider is None:
            self.env = EnvMap(dim, device)
            self._check(self._L.mplx_planner_attach_ctx(self._p, self.env._ctx))
        else:
            single, batched, user = provider
            self._check(self._L.mplx_planner_set_provider(self._p, single, batched, user))
        self._summary = None

    def _check(self, rc):
        if rc != 0:
            msg = self._L.mplx_planner_last_error(self._p)
            raise _abi.MplxError(rc, msg.decode() if msg else "?")

    def close(self):
        if self._p:
            self._L.mplx_planner_destroy(self._p)
            self._p = None
        if self.env is not None:
            self.env.close()
            self.env = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- MapPlanner / PlannerBase setters (same names as the reference)
    def setMapUtil(self, map_util):
        self._map_util = map_util
        d = (C.c_int32 * 3)(*(map_util.map_dim + [1] * (3 - len(map_util.map_dim))))
        o = (C.c_double * 3)(*(map_util.origin + [0.0] * (3 - len(map_util.origin))))
        self._check(self._L.mplx_planner_set_map(self._p, map_util.cells.ctypes.data, d, o, map_util.res))
        if self.env is not None:
            self.env.setMap(map_util.origin, map_util.map_dim, map_util.cells, map_util.res)

    def setU(self, U):
        U = np.ascontiguousarray(U, dtype=np.float64)
        self._check(self._L.mplx_planner_set_controls(self._p, U.ctypes.data, U.shape[0], U.shape[1]))
        if self.env is not None:
            self.env.set_u(U)

    def _env(self, name, v):
        if self.env is not None:
            getattr(self.env, name)(v)

    def setVmax(self, v): self._cfg.v_max = float(v); self._env("set_v_max", v)
    def setAmax(self, a): self._env("set_a_max", a)
    def setJmax(self, j): self._env("set_j_max", j)
    def setYawmax(self, y): self._env("set_yaw_max", y)
    def setDt(self, dt): self._cfg.dt = float(dt); self._env("set_dt", dt)
    def setW(self, w): self._cfg.w = float(w); self._env("set_w", w)
    def setWyaw(self, w): self._env("set_wyaw", w)
    def setEpsilon(self, eps): self._cfg.epsilon = float(eps)
    def setMaxNum(self, n): self._cfg.max_expand = int(n)

    def setTmax(self, t):
        """PlannerBase::setTmax (planner_base.h:203-205).  Kept and, as in the reference's MapPlanner, without effect on the
        search: t_max is only read by env_base::is_goal (env_base.h:24), which env_map::is_goal (env_map.h:25-45) overrides
        without it."""
        self._t_max = float(t)

    def setTol(self, tol_pos, tol_vel=-1.0, tol_acc=-1.0):
        self._cfg.tol_pos, self._cfg.tol_vel, self._cfg.tol_acc = float(tol_pos), float(tol_vel), float(tol_acc)

    # ---- MapPlanner's map preprocessing and iterative planning (map_planner.h:25-62, map_planner.cpp:46-95,
    #      246-283, 394-430); the kernels are the engine's (EnvMap.updatePotentialMap / setSearchRegion)
    def setSearchRadius(self, r): self._search_radius = [float(x) for x in r]
    def setPotentialRadius(self, r): self._potential_radius = [float(x) for x in r]
    def setPotentialMapRange(self, r): self._potential_range = [float(x) for x in r]
    def setPotentialWeight(self, w): self._env("set_potential_weight", w)
    def setGradientWeight(self, w): self._env("set_gradient_weight", w)

    def setSearchRegion(self, path, dense=False):
        """Cells within the search radius of `path` ([n][D] positions) become the only traversable ones."""
        if self._search_radius is None:
            raise ValueError("setSearchRadius first")
        return self.env.setSearchRegion(np.asarray(path, dtype=np.float64)[:, :self.dim], self._search_radius, dense)

    def updatePotentialMap(self, pos):
        """Rewrites the MapUtil's map with the potential field around the obstacles (map_planner.cpp:246-283, 387)
        and installs it as the env's potential map."""
        if self._potential_radius is None:
            raise ValueError("setPotentialRadius first")
        new_map = self.env.updatePotentialMap(pos, self._potential_radius, self._potential_range, self._pow)
        mu = self._map_util
        mu.cells = new_map
        d = (C.c_int32 * 3)(*(mu.map_dim + [1] * (3 - len(mu.map_dim))))
        o = (C.c_double * 3)(*(mu.origin + [0.0] * (3 - len(mu.origin))))
        self._check(self._L.mplx_planner_set_map(self._p, mu.cells.ctypes.data, d, o, mu.res))
        return new_map

    def iterativePlan(self, start, goal, raw_traj, max_num):
        """MapPlanner::iterativePlan (map_planner.cpp:394-430): re-plan inside the tunnel around the previous
        trajectory until the cost stops changing."""
        traj = raw_traj
        prev_cost = 0.0
        for _ in range(int(max_num)):
            self.setSearchRegion(traj.getWaypoints()[:, :self.dim], False)
            if not self.plan(start, goal):
                return False
            traj = self.getTraj()
            if prev_cost == self.getTrajCost():
                break
            prev_cost = self.getTrajCost()
        return True

    def setBatch(self, n):
        """Nodes per device launch (1 = the reference's one-node-at-a-time loop; default 64 on the engine)."""
        self._cfg.batch = int(n)

    # ---- PlannerBase::plan
    def plan(self, start, goal):
        self._cfg.control = int(start.control)
        self._cfg.goal_control = int(goal.control) if goal.control else 0  # env_base.h:47 hashes the goal with its own flags
        if self.env is not None:
            self.env.set_control(start.control)
            self.env._flush()
        self._check(self._L.mplx_planner_configure(self._p, C.byref(self._cfg)))
        s = np.ascontiguousarray(start.to_row(), dtype=np.float64)
        g = np.ascontiguousarray(goal.to_row(), dtype=np.float64)
        out = _abi.PlanSummary()
        self._check(self._L.mplx_planner_plan(self._p, s.ctypes.data, g.ctypes.data, C.byref(out)))
        self._summary = out
        return bool(out.ok)

    def summary(self):
        o = self._summary
        return {k: getattr(o, k) for k in ("ok", "expansions", "closed", "opened", "nodes", "device_launches",
                                           "pairs", "cost", "total_time", "segments")} | {
            "J": list(o.J), "state_mismatches": o.state_mismatches}

    # ---- incremental re-planning (PlannerBase::setLPAstar, MapPlanner::getLinkedNodes / updateBlockedNodes /
    #      updateClearedNodes, StateSpace::getSubStateSpace): the state space outlives plan()
    def setLPAstar(self, on=True):
        self._check(self._L.mplx_planner_set_lpastar(self._p, 1 if on else 0))

    def reset(self):
        self._check(self._L.mplx_planner_reset(self._p))

    def getLinkedNodes(self, want_points=True):
        """(Re)builds the voxel -> edge table; returns (points [n][D] or None, number of cells, number of entries)."""
        n, cells, entries = C.c_int64(), C.c_int64(), C.c_int64()
        self._check(self._L.mplx_planner_linked_nodes(self._p, None, 0, C.byref(n), C.byref(cells), C.byref(entries)))
        pts = None
        if want_points:
            pts = np.empty((max(n.value, 1), self.dim), dtype=np.float64)
            self._check(self._L.mplx_planner_linked_nodes(self._p, pts.ctypes.data, n.value, C.byref(n), C.byref(cells), C.byref(entries)))
            pts = pts[:n.value]
        return pts, cells.value, entries.value

    def _edit_map(self, cells, value):
        """The caller's side of a map edit (the reference's test edits its MapUtil and calls setMap): the planner's host
        copy and the device map get the new cells."""
        mu = self._map_util
        cells = np.ascontiguousarray(cells, dtype=np.int32).reshape(-1, self.dim)
        idx = cells[:, 0].astype(np.int64)
        mul = 1
        for i in range(1, self.dim):
            mul *= mu.map_dim[i - 1]
            idx = idx + mul * cells[:, i]
        # O(edited cells) everywhere: the MapUtil's array in place (copied once, only if it cannot be written), the
        # planner's host copy (start / goal tests) through mplx_planner_edit_map, the device through mplx_edit_map
        if not mu.cells.flags.writeable:
            mu.cells = mu.cells.copy()
        mu.cells[idx] = value
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        val = np.full(idx.size, value, dtype=np.int8)
        self._check(self._L.mplx_planner_edit_map(self._p, idx.ctypes.data, val.ctypes.data, idx.size))
        if self.env is not None:
            self.env.editMap(idx, val)
        return cells

    def updateBlockedNodes(self, cells, edit_map=True):
        cells = self._edit_map(cells, 100) if edit_map else np.ascontiguousarray(cells, dtype=np.int32).reshape(-1, self.dim)
        self._check(self._L.mplx_planner_update_blocked_nodes(self._p, cells.ctypes.data, cells.shape[0]))

    def updateClearedNodes(self, cells, edit_map=True):
        cells = self._edit_map(cells, 0) if edit_map else np.ascontiguousarray(cells, dtype=np.int32).reshape(-1, self.dim)
        if self.env is not None:
            self.env._flush()
        self._check(self._L.mplx_planner_update_cleared_nodes(self._p, cells.ctypes.data, cells.shape[0]))

    def getSubStateSpace(self, time_step):
        self._check(self._L.mplx_planner_sub_state_space(self._p, int(time_step)))

    def setEdgeProvider(self, fn_ptr, user_ptr):
        """Tests: another implementation of the batched edge re-validation (the CPU oracle's)."""
        self._check(self._L.mplx_planner_set_edge_provider(self._p, fn_ptr, user_ptr))

    def setPriorTrajectory(self, other, potential=None, potential_weight=None, gradient_weight=None):
        """PlannerBase::setPriorTrajectory: the last trajectory of another planner (still open) guides this one's
        search; None clears it.  Set this planner's map, v_max, w and dt first -- and the potential map, if any
        (env_map::set_prior_trajectory reads potential_map_, env_map.h:197-216): with the engine's own env the values
        of the cells the prior passes through are read from the device's potential map; a planner on another
        provider passes its host copy as `potential` with the two weights."""
        self._check(self._L.mplx_planner_configure(self._p, C.byref(self._cfg)))  # (v_max, w, dt reach the search)
        if other is not None and (potential is not None or (self.env is not None and self.env.has_potential)):
            if potential is not None:
                potential = np.ascontiguousarray(potential, dtype=np.int8).ravel()
                pw, gw = float(potential_weight), float(gradient_weight or 0.0)
                ptr = potential.ctypes.data
            else:
                self.env._flush()
                (pw, gw), ptr = self.env.potential_weights(), None
            self._check(self._L.mplx_planner_set_prior_trajectory_potential(self._p, other._p, ptr, C.c_double(pw), C.c_double(gw)))
            return
        self._check(self._L.mplx_planner_set_prior_trajectory(self._p, other._p if other is not None else None))

    def useDeviceHeuristic(self, on=True):
        """The heuristic of new nodes from the `heur` row the expansion launches write (mplx_set_goal) instead of the
        search's own evaluation: same search, see mplx_planner_use_device_heuristic."""
        self._check(self._L.mplx_planner_use_device_heuristic(self._p, 1 if on else 0))

    def timing(self):
        """Where the wall time of the last plan() went (ms) and what its relaxation loop did (mplx_plan_timing)."""
        t = _abi.PlanTiming()
        self._check(self._L.mplx_planner_timing(self._p, C.byref(t)))
        return {k: getattr(t, k) for k, _ in _abi.PlanTiming._fields_}

    def getCloseSet(self):
        n = C.c_int32()
        self._check(self._L.mplx_planner_closed_set(self._p, None, 0, C.byref(n)))
        pts = np.empty((n.value, self.dim), dtype=np.float64)
        self._check(self._L.mplx_planner_closed_set(self._p, pts.ctypes.data, n.value, C.byref(n)))
        return pts

    def getOpenStates(self):
        """Full states (rows of 4D+2) of the open set of the last plan (PlannerBase::getOpenSet gives positions)."""
        n = C.c_int32()
        self._check(self._L.mplx_planner_open_set(self._p, None, 0, C.byref(n)))
        f = 4 * self.dim + 2
        rows = np.empty((n.value, f), dtype=np.float64)
        self._check(self._L.mplx_planner_open_set(self._p, rows.ctypes.data, n.value, C.byref(n)))
        return rows

    def getTraj(self):
        o = self._summary
        f = 4 * self.dim + 2
        nodes = np.empty((max(o.segments, 1), f), dtype=np.float64)
        acts = np.empty(max(o.segments, 1), dtype=np.int32)
        self._check(self._L.mplx_planner_trajectory(self._p, nodes.ctypes.data, acts.ctypes.data, max(o.segments, 1)))
        end = np.empty(f, dtype=np.float64)
        if o.ok and o.segments > 0:
            self._check(self._L.mplx_planner_trajectory_end(self._p, end.ctypes.data))
        else:
            end = None
        return Trajectory(o.total_time, list(o.J), nodes[:o.segments], acts[:o.segments], o.cost, end)

    def getTrajCost(self):
        return self._summary.cost
