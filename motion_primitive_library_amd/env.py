"""Host-side mirror of the reference's env_map<Dim> for the one path this
package replaces: successor expansion.

Names and argument meaning follow the reference so that code (and tests)
written against MPL read the same:

    reference                                   here
    ----------------------------------------    -------------------------------
    MapUtil<Dim>::setMap       map_util.h:84    EnvMap.setMap(origin, dim, map, res)
    env_base::set_u            env_base.h:234   EnvMap.set_u(U)
    env_base::set_v_max ...    env_base.h:237+  EnvMap.set_v_max(v) ...
    env_base::set_dt/w/wyaw    env_base.h:264+  EnvMap.set_dt / set_w / set_wyaw
    env_map::set_potential_map env_map.h:181    EnvMap.set_potential_map(map)
    env_base::set_search_region env_base.h:301  EnvMap.set_search_region(mask)
    env_map::get_succ          env_map.h:147    EnvMap.get_succ(curr) -> (succ, cost, action)

plus the batched forms that are the point of the engine: EnvMap.expand(nodes)
(host arrays) and EnvMap.expand_resident(frontier, slots) (HBM-resident).

Everything below the method bodies is the C ABI of include/mplx.h; numpy is
used only to own host buffers.  No CPU implementation exists here.
"""
import ctypes as C

import numpy as np

from . import _abi

# Control::Control (reference include/mpl_basis/control.h:10-20)
VEL, ACC, JRK, SNP = 0x01, 0x03, 0x07, 0x0F
VELxYAW, ACCxYAW, JRKxYAW, SNPxYAW = 0x11, 0x13, 0x17, 0x1F

SLOT_SKIP_SAME, SLOT_FINITE, SLOT_BLOCKED, SLOT_SKIP_DYN = 0, 1, 2, 3


class Waypoint:
    """Waypoint<Dim> (reference include/mpl_basis/waypoint.h:23-57)."""

    __slots__ = ("dim", "pos", "vel", "acc", "jrk", "yaw", "t", "control")

    def __init__(self, dim, control=0, pos=None, vel=None, acc=None, jrk=None, yaw=0.0, t=0.0):
        self.dim = dim
        self.control = control
        z = np.zeros(dim)
        self.pos = z.copy() if pos is None else np.asarray(pos, dtype=np.float64).copy()
        self.vel = z.copy() if vel is None else np.asarray(vel, dtype=np.float64).copy()
        self.acc = z.copy() if acc is None else np.asarray(acc, dtype=np.float64).copy()
        self.jrk = z.copy() if jrk is None else np.asarray(jrk, dtype=np.float64).copy()
        self.yaw = float(yaw)
        self.t = float(t)

    def to_row(self):
        return np.concatenate([self.pos, self.vel, self.acc, self.jrk, [self.yaw, self.t]])

    @classmethod
    def from_row(cls, dim, control, row):
        d = dim
        return cls(dim, control, row[0:d], row[d:2 * d], row[2 * d:3 * d], row[3 * d:4 * d], row[4 * d],
                   row[4 * d + 1])

    def __repr__(self):
        return "Waypoint(pos=%s, vel=%s, acc=%s, jrk=%s, yaw=%r, t=%r)" % (
            self.pos.tolist(), self.vel.tolist(), self.acc.tolist(), self.jrk.tolist(), self.yaw, self.t)


class DeviceArray:
    """An HBM allocation made through the C ABI (mplx_device_alloc)."""

    def __init__(self, env, nbytes):
        self._env = env
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        _abi.check(env._ctx, _abi.lib().mplx_device_alloc(env._ctx, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, host):
        host = np.ascontiguousarray(host)
        assert host.nbytes <= self.nbytes
        _abi.check(self._env._ctx, _abi.lib().mplx_memcpy_h2d(self._env._ctx, self.ptr, host.ctypes.data, host.nbytes))

    def download(self, dtype, shape, offset=0):
        """Copy `shape` elements of `dtype` starting `offset` BYTES into the allocation back to the host."""
        out = np.empty(shape, dtype=dtype)
        assert int(offset) + out.nbytes <= self.nbytes
        _abi.check(self._env._ctx, _abi.lib().mplx_memcpy_d2h(self._env._ctx, out.ctypes.data, self.ptr + int(offset),
                                                               out.nbytes))
        return out

    def free(self):
        if self.ptr and self._env._ctx:
            _abi.lib().mplx_device_free(self._env._ctx, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _alloc(env, nbytes, alloc=None):
    """An HBM allocation for the engine: through the C ABI by default, or from `alloc(nbytes)` -- any object with
    .ptr / .nbytes / .download / .free, e.g. shard.TorchArray, so that torch.distributed can move the same memory."""
    return DeviceArray(env, nbytes) if alloc is None else alloc(int(nbytes))


class DeviceArrayView:
    """A raw device pointer the engine did not allocate (e.g. a field of a ready C struct): read-back helper."""

    def __init__(self, env, ptr):
        self._env, self.ptr = env, int(ptr)

    def download_i64(self, index):
        out = np.empty(1, dtype=np.int64)
        _abi.check(self._env._ctx, _abi.lib().mplx_memcpy_d2h(self._env._ctx, out.ctypes.data, self.ptr + 8 * int(index), 8))
        return out[0]


class Slots:
    """HBM-resident dense successor slots for n_nodes x nU pairs."""

    def __init__(self, env, n_nodes, nU, want_state=True, want_iters=False):
        self.n_nodes, self.nU = int(n_nodes), int(nU)
        self.n_slots = self.n_nodes * self.nU
        self.n_fields = env.n_fields
        n = max(self.n_slots, 1)
        self.status = DeviceArray(env, n)
        self.cost = DeviceArray(env, n * 8)
        self.hash = DeviceArray(env, n * 8)
        self.state = DeviceArray(env, n * 8 * self.n_fields) if want_state else None
        self.iters = DeviceArray(env, n * 4) if want_iters else None

    def c_struct(self):
        s = _abi.Succ()
        s.status, s.cost, s.hash = self.status.ptr, self.cost.ptr, self.hash.ptr
        s.state = self.state.ptr if self.state else None
        s.state_stride = self.n_slots
        s.iters = self.iters.ptr if self.iters else None
        return s

    def download(self):
        out = {
            "status": self.status.download(np.uint8, (self.n_slots,)),
            "cost": self.cost.download(np.float64, (self.n_slots,)),
            "hash": self.hash.download(np.uint64, (self.n_slots,)),
        }
        if self.state:
            out["state"] = self.state.download(np.float64, (self.n_fields, self.n_slots))
        if self.iters:
            out["iters"] = self.iters.download(np.int32, (self.n_slots,))
        return out

    def free(self):
        for b in (self.status, self.cost, self.hash, self.state, self.iters):
            if b is not None:
                b.free()


STATE_ROW_PAD = 0  # default padding between the state rows of Lists, in entries (see Lists.state_stride)


class Lists:
    """HBM-resident per-node successor lists (mplx_succ_lists)."""

    def __init__(self, env, n_nodes, nU, want_state=True, want_iters=False, want_hash=True, stride=None, alloc=None,
                 state_pad=None, want_heur=False, want_flags=False):
        self.n_nodes, self.nU = int(n_nodes), int(nU)
        # entries reserved per node: a multiple of 32 keeps every node's rows on 128-byte lines (and lets the
        # kernel complete the last line of each list instead of leaving a partial-line store)
        self.stride = (self.nU + 31) & ~31 if stride is None else int(stride)
        self.n_slots = self.n_nodes * self.stride
        self.n_fields = env.n_fields
        n = max(self.n_slots, 1)
        self.count = _alloc(env, max(self.n_nodes, 1) * 4, alloc)
        self.action = _alloc(env, n * 4, alloc)
        self.cost = _alloc(env, n * 8, alloc)
        self.hash = _alloc(env, n * 8, alloc) if want_hash else None
        # entries between consecutive state rows (mplx_succ_lists::state_stride): n_slots + state_pad
        self.state_stride = n + (STATE_ROW_PAD if state_pad is None else int(state_pad))
        self.state = _alloc(env, self.state_stride * 8 * self.n_fields, alloc) if want_state else None
        self.iters = _alloc(env, n * 4, alloc) if want_iters else None
        # rows the expansion launch fills for the search (mplx_set_goal): default heuristic, goal flags
        self.heur = _alloc(env, n * 8, alloc) if want_heur else None
        self.flags = _alloc(env, n, alloc) if want_flags else None

    def c_struct(self):
        s = _abi.SuccLists()
        s.count, s.action, s.cost = self.count.ptr, self.action.ptr, self.cost.ptr
        s.hash = self.hash.ptr if self.hash else None
        s.state = self.state.ptr if self.state else None
        s.state_stride = self.state_stride
        s.iters = self.iters.ptr if self.iters else None
        s.node_stride = self.stride
        s.heur = self.heur.ptr if self.heur else None
        s.flags = self.flags.ptr if self.flags else None
        return s

    def download(self):
        out = {
            "stride": self.stride,
            "count": self.count.download(np.int32, (self.n_nodes,)),
            "action": self.action.download(np.int32, (self.n_slots,)),
            "cost": self.cost.download(np.float64, (self.n_slots,)),
        }
        if self.hash:
            out["hash"] = self.hash.download(np.uint64, (self.n_slots,))
        if self.state:
            st = self.state.download(np.float64, (self.n_fields, self.state_stride))
            out["state"] = np.ascontiguousarray(st[:, :self.n_slots]) if self.state_stride != self.n_slots else st
        if self.iters:
            out["iters"] = self.iters.download(np.int32, (self.n_slots,))
        if self.heur:
            out["heur"] = self.heur.download(np.float64, (self.n_slots,))
        if self.flags:
            out["flags"] = self.flags.download(np.uint8, (self.n_slots,))
        return out

    def download_nodes(self, lo, hi):
        """The lists of nodes [lo, hi) only, as download() would return them for a frontier of hi - lo nodes
        (lets a host with less memory than the device walk a full-size result chunk by chunk)."""
        lo, hi = int(lo), int(hi)
        n, S = hi - lo, self.stride
        out = {
            "stride": S,
            "count": self.count.download(np.int32, (n,), lo * 4),
            "action": self.action.download(np.int32, (n * S,), lo * S * 4),
            "cost": self.cost.download(np.float64, (n * S,), lo * S * 8),
        }
        if self.hash:
            out["hash"] = self.hash.download(np.uint64, (n * S,), lo * S * 8)
        if self.state:
            st = np.empty((self.n_fields, n * S), np.float64)
            for r in range(self.n_fields):
                st[r] = self.state.download(np.float64, (n * S,), (r * self.state_stride + lo * S) * 8)
            out["state"] = st
        if self.iters:
            out["iters"] = self.iters.download(np.int32, (n * S,), lo * S * 4)
        return out

    def free(self):
        for b in (self.count, self.action, self.cost, self.hash, self.state, self.iters, self.heur, self.flags):
            if b is not None:
                b.free()


class PackedLists:
    """HBM-resident packed successor lists (mplx_packed_lists): node k owns entries [offs[k], offs[k+1]) of every
    row, no padding -- the form the multi-GPU all-gather moves and an on-device consumer reads."""

    def __init__(self, env, n_nodes, capacity, want_state=True, want_hash=True, alloc=None):
        self.n_nodes, self.capacity = int(n_nodes), max(int(capacity), 1)
        self.n_fields = env.n_fields
        c = self.capacity
        self.count = _alloc(env, max(self.n_nodes, 1) * 4, alloc)
        self.offs = _alloc(env, (self.n_nodes + 1) * 8, alloc)
        self.action = _alloc(env, c * 4, alloc)
        self.cost = _alloc(env, c * 8, alloc)
        self.hash = _alloc(env, c * 8, alloc) if want_hash else None
        self.state = _alloc(env, c * 8 * self.n_fields, alloc) if want_state else None

    def c_struct(self):
        s = _abi.PackedLists()
        s.count, s.offs, s.action, s.cost = self.count.ptr, self.offs.ptr, self.action.ptr, self.cost.ptr
        s.hash = self.hash.ptr if self.hash else None
        s.state = self.state.ptr if self.state else None
        s.state_stride = self.capacity
        s.capacity = self.capacity
        return s

    def download(self, n_nodes=None):
        n = self.n_nodes if n_nodes is None else int(n_nodes)
        offs = self.offs.download(np.int64, (n + 1,))
        total = int(offs[n])
        out = {"offs": offs, "total": total, "count": self.count.download(np.int32, (n,)),
               "action": self.action.download(np.int32, (total,)), "cost": self.cost.download(np.float64, (total,))}
        if self.hash:
            out["hash"] = self.hash.download(np.uint64, (total,))
        if self.state:
            out["state"] = self.state.download(np.float64, (self.n_fields, self.capacity))[:, :total]
        return out

    def free(self):
        for b in (self.count, self.offs, self.action, self.cost, self.hash, self.state):
            if b is not None:
                b.free()


def pack_host_lists(lists, n_nodes):
    """Host-side restatement of mplx_pack_lists_device on downloaded lists (tests, CPU stand-ins)."""
    S = int(lists["stride"])
    cnt = np.asarray(lists["count"][:n_nodes], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    src = (np.repeat(np.arange(n_nodes, dtype=np.int64) * S - offs[:-1], cnt) + np.arange(offs[-1], dtype=np.int64))
    out = {"offs": offs, "total": int(offs[-1]), "count": cnt.astype(np.int32)}
    for k in ("action", "cost", "hash"):
        if lists.get(k) is not None:
            out[k] = lists[k][src]
    if lists.get("state") is not None:
        out["state"] = lists["state"][:, src]
    return out


def lists_from_dense(dense, n_nodes, nU):
    """Dense slots -> the per-node list layout of mplx_succ_lists (host-side
    helper for tests and examples).  Unused tail entries are left as zeros."""
    st = dense["status"].reshape(n_nodes, nU)
    emit = (st == 1) | (st == 2)
    count = emit.sum(axis=1).astype(np.int32)
    out = {"count": count, "action": np.zeros(n_nodes * nU, np.int32), "cost": np.zeros(n_nodes * nU, np.float64),
           "hash": np.zeros(n_nodes * nU, np.uint64)}
    if dense.get("state") is not None:
        out["state"] = np.zeros_like(dense["state"])
    if dense.get("iters") is not None:
        out["iters"] = np.zeros(n_nodes * nU, np.int32)
    for k in range(n_nodes):
        ci = np.nonzero(emit[k])[0]
        src = k * nU + ci
        dst = k * nU + np.arange(ci.size)
        out["action"][dst] = ci
        out["cost"][dst] = dense["cost"][src]
        out["hash"][dst] = dense["hash"][src]
        if "state" in out:
            out["state"][:, dst] = dense["state"][:, src]
        if "iters" in out:
            out["iters"][dst] = dense["iters"][src]
    return out


class EnvMap:
    """env_map<Dim> whose get_succ runs on the MI355X (reference env_map.h)."""

    def __init__(self, dim, device=0):
        if dim not in (2, 3):
            raise ValueError("dim must be 2 or 3")
        self.dim = dim
        self._ctx = None
        L = _abi.lib()
        ctx = C.c_void_p()
        rc = L.mplx_create(dim, device, C.byref(ctx))
        if rc != _abi.OK:
            msg = L.mplx_last_error(None)
            raise _abi.MplxError(rc, msg.decode() if msg else "?")
        self._ctx = ctx
        # env_base.h:368-392 / env_map.h:294-296 defaults
        self._p = _abi.Params()
        self._p.control = ACC
        self._p.dt, self._p.w, self._p.wyaw = 1.0, 10.0, 1.0
        self._p.v_max = self._p.a_max = self._p.j_max = self._p.yaw_max = -1.0
        self._p.potential_weight, self._p.gradient_weight = 0.1, 0.0
        self.has_potential = False
        self._dirty = True
        self.nU = 0
        self.map_dim = None

    # ---- lifetime
    def close(self):
        if self._ctx:
            _abi.lib().mplx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def n_fields(self):
        return 4 * self.dim + 2

    def device_info(self):
        buf = C.create_string_buffer(256)
        cus = C.c_int32()
        _abi.check(self._ctx, _abi.lib().mplx_device_info(self._ctx, buf, 256, C.byref(cus)))
        return buf.value.decode(), cus.value

    # ---- map (MapUtil<Dim>::setMap, map_util.h:84-90)
    def setMap(self, origin, dim, cells, res):
        cells = np.ascontiguousarray(cells, dtype=np.int8).ravel()
        d = (C.c_int32 * 3)(*([int(x) for x in dim] + [1] * (3 - len(dim))))
        o = (C.c_double * 3)(*([float(x) for x in origin] + [0.0] * (3 - len(origin))))
        n = int(np.prod([int(x) for x in dim]))
        if cells.size != n:
            raise ValueError("map has %d cells, dim says %d" % (cells.size, n))
        _abi.check(self._ctx, _abi.lib().mplx_set_map(self._ctx, cells.ctypes.data, d, o, float(res)))
        self.map_dim = [int(x) for x in dim]
        self._ncell = n

    def editMap(self, cell_index, values):
        """A few cells of the map already on the device take new values (mplx_edit_map): cell_index = x + dim0 * (y +
        dim1 * z) as MapUtil::getIndex numbers them."""
        idx = np.ascontiguousarray(cell_index, dtype=np.int64).ravel()
        val = np.ascontiguousarray(values, dtype=np.int8).ravel()
        if val.size == 1 and idx.size > 1:
            val = np.full(idx.size, val[0], dtype=np.int8)
        if idx.size != val.size:
            raise ValueError("editMap: %d indices, %d values" % (idx.size, val.size))
        _abi.check(self._ctx, _abi.lib().mplx_edit_map(self._ctx, idx.ctypes.data, val.ctypes.data, idx.size))

    def read_cells(self, cell_index, potential=False):
        """Values of a few cells of the map (or the potential map) the device holds (mplx_read_cells)."""
        idx = np.ascontiguousarray(cell_index, dtype=np.int64).ravel()
        out = np.empty(idx.size, dtype=np.int8)
        _abi.check(self._ctx, _abi.lib().mplx_read_cells(self._ctx, 1 if potential else 0, idx.ctypes.data, idx.size, out.ctypes.data))
        return out

    def potential_weights(self):
        return float(self._p.potential_weight), float(self._p.gradient_weight)

    def map_upload_bytes(self):
        """Host -> device bytes the map calls of this context have moved so far (mplx_map_upload_bytes)."""
        b = C.c_uint64(0)
        _abi.check(self._ctx, _abi.lib().mplx_map_upload_bytes(self._ctx, C.byref(b)))
        return int(b.value)

    # ---- env_base / env_map setters
    def set_control(self, control):
        """The control flag of the search (Waypoint::control of the start node)."""
        self._p.control = int(control)
        self._dirty = True

    def set_u(self, U):
        U = np.ascontiguousarray(U, dtype=np.float64)
        if U.ndim != 2:
            raise ValueError("U must be [nU][udim]")
        _abi.check(self._ctx, _abi.lib().mplx_set_controls(self._ctx, U.ctypes.data, U.shape[0], U.shape[1]))
        self.nU = U.shape[0]

    def _setp(self, name, v):
        setattr(self._p, name, float(v))
        self._dirty = True

    def set_v_max(self, v): self._setp("v_max", v)
    def set_a_max(self, a): self._setp("a_max", a)
    def set_j_max(self, j): self._setp("j_max", j)
    def set_yaw_max(self, y): self._setp("yaw_max", y)
    def set_dt(self, dt): self._setp("dt", dt)
    def set_w(self, w): self._setp("w", w)
    def set_wyaw(self, w): self._setp("wyaw", w)
    def set_potential_weight(self, w): self._setp("potential_weight", w)
    def set_gradient_weight(self, w): self._setp("gradient_weight", w)

    def set_potential_map(self, cells):
        self.has_potential = not (cells is None or len(cells) == 0)
        if not self.has_potential:
            _abi.check(self._ctx, _abi.lib().mplx_set_potential(self._ctx, None))
            return
        cells = np.ascontiguousarray(cells, dtype=np.int8).ravel()
        if cells.size != self._ncell:
            raise ValueError("potential map size mismatch")
        _abi.check(self._ctx, _abi.lib().mplx_set_potential(self._ctx, cells.ctypes.data))

    def set_search_region(self, mask):
        if mask is None or len(mask) == 0:
            _abi.check(self._ctx, _abi.lib().mplx_set_region(self._ctx, None))
            return
        mask = np.ascontiguousarray(np.asarray(mask) != 0, dtype=np.uint8).ravel()
        if mask.size != self._ncell:
            raise ValueError("search region size mismatch")
        _abi.check(self._ctx, _abi.lib().mplx_set_region(self._ctx, mask.ctypes.data))

    # ---- MapPlanner-level map preprocessing on the device (map_planner.cpp:46-95, 246-391)
    def updatePotentialMap(self, pos, radius, range_=None, power=1.0):
        """MapPlanner::updatePotentialMap: stamps the potential field into the device map (which it also
        installs as the potential map, as the reference does) and returns the new int8 map."""
        D = len(self.map_dim)
        p = (C.c_double * 3)(*([float(x) for x in pos] + [0.0] * (3 - D)))
        r = (C.c_double * 3)(*([float(x) for x in radius] + [0.0] * (3 - D)))
        g = None if range_ is None else (C.c_double * 3)(*([float(x) for x in range_] + [0.0] * (3 - D)))
        out = np.empty(self._ncell, dtype=np.int8)
        _abi.check(self._ctx, _abi.lib().mplx_update_potential_map(self._ctx, p, r, g, float(power), out.ctypes.data))
        self.has_potential = True
        return out

    def setSearchRegion(self, path, search_radius, dense=False):
        """MapPlanner::setSearchRegion: installs the tunnel around `path` ([n][D]) and returns it, one byte
        per cell.  `dense` has the reference's (inverted) meaning: False = ray-trace between the points."""
        D = len(self.map_dim)
        pts = np.ascontiguousarray(path, dtype=np.float64).reshape(-1, D)
        sr = (C.c_double * 3)(*([float(x) for x in search_radius] + [0.0] * (3 - D)))
        out = np.empty(self._ncell, dtype=np.uint8)
        _abi.check(self._ctx, _abi.lib().mplx_set_search_region_path(self._ctx, pts.ctypes.data, pts.shape[0],
                                                                      int(bool(dense)), sr, out.ctypes.data))
        return out

    def _flush(self):
        if self._dirty:
            _abi.check(self._ctx, _abi.lib().mplx_set_params(self._ctx, C.byref(self._p)))
            self._dirty = False

    # ---- get_succ, exactly the reference's contract (env_map.h:147-172)
    def get_succ(self, curr):
        """Returns (succ, succ_cost, action_idx): successors in ascending control
        index; blocked ones are included with cost = +inf."""
        self.set_control(curr.control) if curr.control != self._p.control else None
        self._flush()
        F, nU = self.n_fields, self.nU
        node = np.ascontiguousarray(curr.to_row(), dtype=np.float64)
        succ = np.empty((nU, F), dtype=np.float64)
        cost = np.empty(nU, dtype=np.float64)
        act = np.empty(nU, dtype=np.int32)
        n = C.c_int32()
        _abi.check(self._ctx, _abi.lib().mplx_get_succ(self._ctx, node.ctypes.data, succ.ctypes.data,
                                                        cost.ctypes.data, act.ctypes.data, C.byref(n)))
        m = n.value
        return ([Waypoint.from_row(self.dim, curr.control, succ[i]) for i in range(m)],
                cost[:m].tolist(), act[:m].tolist())

    # ---- batched forms
    def expand(self, nodes, want_state=True, want_iters=True):
        """Dense expansion of a host frontier [4D+2][N]; returns host arrays."""
        self._flush()
        nodes = np.ascontiguousarray(nodes, dtype=np.float64)
        if nodes.ndim != 2 or nodes.shape[0] != self.n_fields:
            raise ValueError("nodes must be [%d][N]" % self.n_fields)
        n = nodes.shape[1]
        ns = n * self.nU
        out = {"status": np.empty(ns, np.uint8), "cost": np.empty(ns, np.float64),
               "hash": np.empty(ns, np.uint64)}
        s = _abi.Succ()
        s.status, s.cost, s.hash = out["status"].ctypes.data, out["cost"].ctypes.data, out["hash"].ctypes.data
        if want_state:
            out["state"] = np.empty((self.n_fields, ns), np.float64)
            s.state, s.state_stride = out["state"].ctypes.data, ns
        if want_iters:
            out["iters"] = np.empty(ns, np.int32)
            s.iters = out["iters"].ctypes.data
        _abi.check(self._ctx, _abi.lib().mplx_expand(self._ctx, nodes.ctypes.data, n, n, C.byref(s)))
        return out

    def expand_lists(self, nodes, want_state=True, want_iters=True, stride=None, out=None):
        """Per-node successor lists of a host frontier [4D+2][N] (mplx_expand_lists).
        `stride` = entries reserved per node (default nU; out["stride"] reports it).
        `out` = the dict a previous call with the same shapes returned: its arrays are reused (a C caller
        keeps its buffers too; fresh arrays cost one page fault per 4 KiB written)."""
        self._flush()
        nodes = np.ascontiguousarray(nodes, dtype=np.float64)
        if nodes.ndim != 2 or nodes.shape[0] != self.n_fields:
            raise ValueError("nodes must be [%d][N]" % self.n_fields)
        n = nodes.shape[1]
        stride = self.nU if stride is None else int(stride)
        ns = n * stride
        reuse = out is not None
        if reuse:
            if out["stride"] != stride or out["count"].shape != (n,) or (want_state and "state" not in out) or (
                    want_iters and "iters" not in out):
                raise ValueError("out does not match this call")
        else:
            out = {"stride": stride, "count": np.zeros(n, np.int32), "action": np.zeros(ns, np.int32),
                   "cost": np.zeros(ns, np.float64), "hash": np.zeros(ns, np.uint64)}
        s = _abi.SuccLists()
        s.node_stride = stride
        s.count, s.action = out["count"].ctypes.data, out["action"].ctypes.data
        s.cost, s.hash = out["cost"].ctypes.data, out["hash"].ctypes.data
        if want_state:
            if not reuse:
                out["state"] = np.zeros((self.n_fields, ns), np.float64)
            s.state, s.state_stride = out["state"].ctypes.data, ns
        if want_iters:
            if not reuse:
                out["iters"] = np.zeros(ns, np.int32)
            s.iters = out["iters"].ctypes.data
        _abi.check(self._ctx, _abi.lib().mplx_expand_lists(self._ctx, nodes.ctypes.data, n, n, C.byref(s)))
        return out

    def alloc_lists(self, n_nodes, want_state=True, want_iters=False, want_hash=True, stride=None, alloc=None,
                    state_pad=None, want_heur=False, want_flags=False):
        return Lists(self, n_nodes, self.nU, want_state, want_iters, want_hash, stride, alloc, state_pad, want_heur, want_flags)

    def set_goal(self, goal_row, w=None, v_max=None, tol_pos=0.5, tol_vel=-1.0, tol_acc=-1.0, tol_yaw=-1.0, goal_control=0):
        """env_base::set_goal for the device (mplx_set_goal): the goal the `heur` / `flags` rows of the lists refer to.
        goal_row=None clears it."""
        self._flush()
        if goal_row is None:
            _abi.check(self._ctx, _abi.lib().mplx_set_goal(self._ctx, None))
            return
        goal = np.ascontiguousarray(goal_row, dtype=np.float64)
        g = _abi.GoalSpec()
        g.goal, g.control, g.goal_control = goal.ctypes.data, int(self._p.control), int(goal_control)
        g.w = float(self._p.w if w is None else w)
        g.v_max = float(self._p.v_max if v_max is None else v_max)
        g.tol_pos, g.tol_vel, g.tol_acc, g.tol_yaw = float(tol_pos), float(tol_vel), float(tol_acc), float(tol_yaw)
        _abi.check(self._ctx, _abi.lib().mplx_set_goal(self._ctx, C.byref(g)))

    def alloc_packed(self, n_nodes, capacity=None, want_state=True, want_hash=True, alloc=None):
        """Packed lists for n_nodes nodes; the default capacity n_nodes * nU always suffices."""
        return PackedLists(self, n_nodes, n_nodes * self.nU if capacity is None else capacity, want_state, want_hash, alloc)

    def pack_lists(self, lists, packed, n_nodes=None, want_total=False):
        """mplx_pack_lists_device: asynchronous unless want_total (then returns the number of packed entries)."""
        n = lists.n_nodes if n_nodes is None else int(n_nodes)
        s, p = lists.c_struct(), packed.c_struct()
        total = C.c_int64(-1)
        _abi.check(self._ctx, _abi.lib().mplx_pack_lists_device(self._ctx, C.byref(s), n, C.byref(p),
                                                                 C.byref(total) if want_total else None))
        return total.value if want_total else None

    # ---- RCCL communicator of the context (mplx_comm_*)
    @staticmethod
    def comm_unique_id():
        buf = (C.c_uint8 * _abi.COMM_ID_BYTES)()
        rc = _abi.lib().mplx_comm_unique_id(buf)
        if rc != _abi.OK:
            msg = _abi.lib().mplx_last_error(None)
            raise _abi.MplxError(rc, msg.decode() if msg else "?")
        return bytes(buf)

    def comm_init(self, unique_id, rank, world):
        buf = (C.c_uint8 * _abi.COMM_ID_BYTES).from_buffer_copy(unique_id)
        _abi.check(self._ctx, _abi.lib().mplx_comm_init(self._ctx, buf, int(rank), int(world)))

    def comm_destroy(self):
        _abi.check(self._ctx, _abi.lib().mplx_comm_destroy(self._ctx))

    def comm_broadcast_map(self, root=0):
        _abi.check(self._ctx, _abi.lib().mplx_comm_broadcast_map(self._ctx, int(root)))

    def comm_allgather_lists(self, local, n_local, gathered):
        """mplx_comm_allgather_lists; returns (node_offs, entry_offs) per rank, each [world + 1]."""
        a, b = local.c_struct(), gathered.c_struct()
        no = np.zeros(1025, np.int64)
        eo = np.zeros(1025, np.int64)
        _abi.check(self._ctx, _abi.lib().mplx_comm_allgather_lists(self._ctx, C.byref(a), int(n_local), C.byref(b),
                                                                    no.ctypes.data, eo.ctypes.data))
        return no, eo

    def expand_lists_resident(self, frontier, lists, n_nodes=None):
        """Asynchronous launch on HBM-resident buffers (mplx_expand_lists_device)."""
        self._flush()
        n = frontier.n_nodes if n_nodes is None else int(n_nodes)
        s = lists.c_struct()
        _abi.check(self._ctx, _abi.lib().mplx_expand_lists_device(self._ctx, frontier.ptr, n, frontier.n_nodes,
                                                                  C.byref(s)))

    def post_lists(self, lists, goal_row, w=None, v_max=None, tol_pos=0.5, tol_vel=-1.0, tol_acc=-1.0, tol_yaw=-1.0,
                   n_nodes=None, want_canon=True):
        """Heuristic, goal-tolerance flags and node identity of HBM-resident lists
        (mplx_post_lists_device).  Returns host arrays indexed like the lists."""
        self._flush()
        n = lists.n_nodes if n_nodes is None else int(n_nodes)
        goal = np.ascontiguousarray(goal_row, dtype=np.float64)
        g = _abi.GoalSpec()
        g.goal, g.control = goal.ctypes.data, int(self._p.control)
        g.w = float(self._p.w if w is None else w)
        g.v_max = float(self._p.v_max if v_max is None else v_max)
        g.tol_pos, g.tol_vel, g.tol_acc, g.tol_yaw = float(tol_pos), float(tol_vel), float(tol_acc), float(tol_yaw)
        ns = lists.n_slots
        heur = DeviceArray(self, max(ns, 1) * 8)
        flags = DeviceArray(self, max(ns, 1))
        canon = DeviceArray(self, max(ns, 1) * 4) if want_canon else None
        _abi.check(self._ctx, _abi.lib().mplx_memset(self._ctx, flags.ptr, 0, max(ns, 1)))
        o = _abi.Post()
        o.heur, o.flags, o.canon = heur.ptr, flags.ptr, canon.ptr if canon else None
        s = lists.c_struct()
        _abi.check(self._ctx, _abi.lib().mplx_post_lists_device(self._ctx, C.byref(s), n, C.byref(g), C.byref(o)))
        self.synchronize()
        out = {"heur": heur.download(np.float64, (ns,)), "flags": flags.download(np.uint8, (ns,))}
        if canon:
            out["canon"] = canon.download(np.int32, (ns,))
            canon.free()
        heur.free()
        flags.free()
        return out

    def post_packed(self, packed, n_nodes, goal_row, w=None, v_max=None, tol_pos=0.5, tol_vel=-1.0, tol_acc=-1.0,
                    tol_yaw=-1.0, want_canon=True, alloc=None, download=True):
        """mplx_post_packed_device: heuristic, goal flags and node identity of PACKED lists (e.g. the gathered lists
        of all ranks).  `packed`: an env.PackedLists or a ready _abi.PackedLists struct (device pointers).  Returns
        host arrays (download=True) or the device buffers {"heur", "flags", "canon"} (asynchronous)."""
        self._flush()
        ps = packed.c_struct() if hasattr(packed, "c_struct") else packed
        cap = int(ps.capacity)
        goal = np.ascontiguousarray(goal_row, dtype=np.float64)
        g = _abi.GoalSpec()
        g.goal, g.control = goal.ctypes.data, int(self._p.control)
        g.w = float(self._p.w if w is None else w)
        g.v_max = float(self._p.v_max if v_max is None else v_max)
        g.tol_pos, g.tol_vel, g.tol_acc, g.tol_yaw = float(tol_pos), float(tol_vel), float(tol_acc), float(tol_yaw)
        heur = _alloc(self, cap * 8, alloc)
        flags = _alloc(self, cap, alloc)
        canon = _alloc(self, cap * 4, alloc) if want_canon else None
        _abi.check(self._ctx, _abi.lib().mplx_memset(self._ctx, flags.ptr, 0, cap))
        o = _abi.Post()
        o.heur, o.flags, o.canon = heur.ptr, flags.ptr, canon.ptr if canon else None
        _abi.check(self._ctx, _abi.lib().mplx_post_packed_device(self._ctx, C.byref(ps), int(n_nodes), C.byref(g), C.byref(o)))
        if not download:
            return {"heur": heur, "flags": flags, "canon": canon}
        self.synchronize()
        total = int(DeviceArrayView(self, ps.offs).download_i64(int(n_nodes)))
        out = {"heur": heur.download(np.float64, (total,)), "flags": flags.download(np.uint8, (total,)), "total": total}
        if canon:
            out["canon"] = canon.download(np.int32, (total,))
            canon.free()
        heur.free()
        flags.free()
        return out

    def check_edges(self, parents, actions, cell_cap=0):
        """Batched env_map::is_free(Primitive) + calculate_intrinsic_cost (+ the linked cells of
        MapPlanner::getLinkedNodes when cell_cap > 0) for edges (parent column, action id)."""
        self._flush()
        parents = np.ascontiguousarray(parents, dtype=np.float64)
        actions = np.ascontiguousarray(actions, dtype=np.int32)
        n = actions.size
        if parents.ndim != 2 or parents.shape != (self.n_fields, n):
            raise ValueError("parents must be [%d][%d]" % (self.n_fields, n))
        out = {"free": np.zeros(n, np.uint8), "cost": np.zeros(n, np.float64)}
        o = _abi.EdgesOut()
        o.free_flag, o.cost = out["free"].ctypes.data, out["cost"].ctypes.data
        if cell_cap > 0:
            out["cells"] = np.zeros((n, int(cell_cap)), np.int32)
            out["cell_count"] = np.zeros(n, np.int32)
            o.cells, o.cell_count, o.cell_cap = out["cells"].ctypes.data, out["cell_count"].ctypes.data, int(cell_cap)
        _abi.check(self._ctx, _abi.lib().mplx_check_edges(self._ctx, parents.ctypes.data, actions.ctypes.data, n, n,
                                                           C.byref(o)))
        return out

    def upload_frontier(self, nodes):
        nodes = np.ascontiguousarray(nodes, dtype=np.float64)
        if nodes.ndim != 2 or nodes.shape[0] != self.n_fields:
            raise ValueError("nodes must be [%d][N]" % self.n_fields)
        buf = DeviceArray(self, max(nodes.nbytes, 8))
        buf.upload(nodes)
        buf.n_nodes = nodes.shape[1]
        return buf

    def alloc_slots(self, n_nodes, want_state=True, want_iters=False):
        return Slots(self, n_nodes, self.nU, want_state, want_iters)

    def expand_resident(self, frontier, slots, n_nodes=None, node_offset=0):
        """Asynchronous launch on HBM-resident buffers (mplx_expand_device)."""
        self._flush()
        n = frontier.n_nodes if n_nodes is None else int(n_nodes)
        s = slots.c_struct()
        _abi.check(self._ctx, _abi.lib().mplx_expand_device(
            self._ctx, frontier.ptr + 8 * int(node_offset), n, frontier.n_nodes, C.byref(s)))

    def synchronize(self):
        _abi.check(self._ctx, _abi.lib().mplx_synchronize(self._ctx))

    def timer_begin(self):
        _abi.check(self._ctx, _abi.lib().mplx_timer_begin(self._ctx))

    def timer_end(self):
        ms = C.c_float()
        _abi.check(self._ctx, _abi.lib().mplx_timer_end(self._ctx, C.byref(ms)))
        return ms.value

    def set_lists_route(self, route):
        """Force the kernel behind expand_lists* ("auto", "dense", "tile", "grid"); diagnostic."""
        code = {"auto": _abi.ROUTE_AUTO, "dense": _abi.ROUTE_DENSE, "tile": _abi.ROUTE_TILE,
                "grid": _abi.ROUTE_GRID}[route] if isinstance(route, str) else int(route)
        _abi.check(self._ctx, _abi.lib().mplx_set_lists_route(self._ctx, code))

    def last_lists_route(self):
        code = _abi.lib().mplx_last_lists_route(self._ctx)
        return {_abi.ROUTE_AUTO: "none", _abi.ROUTE_DENSE: "dense", _abi.ROUTE_TILE: "tile",
                _abi.ROUTE_GRID: "grid"}[code]

    def debug_store_model(self, lists, n_nodes=None):
        """DIAGNOSTIC: the list stores of an expansion launch alone, into `lists` (mplx_debug_store_model): every node's first
        count[k] entries of every row, unspecified values.  Overwrites the successor entries; asynchronous."""
        self._flush()
        s = lists.c_struct()
        _abi.check(self._ctx, _abi.lib().mplx_debug_store_model(self._ctx, C.byref(s), lists.n_nodes if n_nodes is None else int(n_nodes)))

    def last_grid_kernel(self):
        """Which kernel of the GRID route the last expand_lists* call ran: "lex" (expand_lex_kernel.hip: lexicographic
        control table, no yaw, occupancy map), "grid" (expand_grid_kernel.hip), "pair" (expand_pair_kernel.hip: yaw controls on
        a potential map over a pre-screened frontier, two nodes per wave), "none" (another route)."""
        return {0: "none", 1: "grid", 2: "lex", 3: "pair"}[_abi.lib().mplx_last_grid_kernel(self._ctx)]

    def last_identity_form(self):
        """Which form of the node-identity pass the last post_lists / post_packed call with canon ran: "table" (in HBM, small
        batches), "claimed" (partition into buckets of fixed capacity), "exact" (partition by histograms),
        "claimed+exact" (a bucket overflowed: the exact form ran after the claimed one).  Same canon[] from all."""
        return {0: "table", 1: "claimed", 2: "exact", 3: "claimed+exact"}[_abi.lib().mplx_last_identity_form(self._ctx)]

    def yaw_pin_stats(self):
        """(nodes re-expanded with the host libm's trig values, fix passes launched) since the context was made."""
        a, b = C.c_int64(), C.c_int64()
        _abi.check(self._ctx, _abi.lib().mplx_yaw_pin_stats(self._ctx, C.byref(a), C.byref(b)))
        return a.value, b.value

    def service(self, mode=-1):
        """The resident kernel behind small synchronous batches (include/mplx.h, mplx_service): mode 1 on, 0 off,
        -1 unchanged; returns dict(requests, launches, failures, resident)."""
        st = (C.c_int64 * 4)()
        _abi.check(self._ctx, _abi.lib().mplx_service(self._ctx, int(mode), st))
        return {"requests": st[0], "launches": st[1], "failures": st[2], "resident": bool(st[3])}

    def selftest_math(self, op, a, b=None):
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = a if b is None else np.ascontiguousarray(b, dtype=np.float64)
        out = np.empty_like(a)
        _abi.check(self._ctx, _abi.lib().mplx_selftest_math(self._ctx, int(op), a.ctypes.data, b.ctypes.data,
                                                             out.ctypes.data, a.size))
        return out
