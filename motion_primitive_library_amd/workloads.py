"""Deterministic synthetic maps, control sets and frontiers for the BASELINE
configurations.  Pure numpy; used by bench.py and the
tests.  Nothing here is on the measured path.

Configurations (BASELINE.json `configs`):
  C2  2D occupancy 1024x1024, Control::ACC, |U| = 25, 4k-node frontier
  C3  3D voxel 256^3, Control::JRK, |U| = 125, 16k-node frontier
  C4  3D voxel 512^3, Control::ACC, |U| = 729, 64k-node frontier   (headline)
  C5  3D voxel 256^3 + potential field, Control::ACCxYAW, |U| = 81, 32k nodes
Random numbers come from numpy's PCG64 (`default_rng(seed)`), map seeds
1002..1005 and frontier seeds 2002..2005.
"""
import itertools

import numpy as np

VEL, ACC, JRK, SNP = 0x01, 0x03, 0x07, 0x0F
VELxYAW, ACCxYAW, JRKxYAW, SNPxYAW = 0x11, 0x13, 0x17, 0x1F


def grid_controls(values, dim, yaw_rates=None):
    """Cartesian control set, first axis slowest -- the loop order of the
    reference's tests (test/test_planner_2d.cpp:52-53)."""
    rows = [list(c) for c in itertools.product(values, repeat=dim)]
    if yaw_rates is not None:
        rows = [r + [y] for r in rows for y in yaw_rates]
    return np.array(rows, dtype=np.float64)


def box_map(dims, res, target_occupancy, seed, side_m=(0.5, 6.0)):
    """Axis-aligned random boxes until the occupied fraction reaches the
    target.  Returns int8 cells {0, 100} in reference order (x fastest), i.e.
    a C-contiguous array indexed [z][y][x] (or [y][x])."""
    rng = np.random.default_rng(seed)
    dims = [int(d) for d in dims]
    shape = tuple(reversed(dims))
    grid = np.zeros(shape, dtype=np.int8)
    total = grid.size
    occupied = 0
    lo, hi = side_m
    while occupied < target_occupancy * total:
        side = np.maximum(1, np.round(rng.uniform(lo, hi, size=len(dims)) / res).astype(np.int64))
        corner = [int(rng.integers(0, max(1, dims[i] - side[i] + 1))) for i in range(len(dims))]
        sl = tuple(slice(corner[i], min(dims[i], corner[i] + int(side[i]))) for i in reversed(range(len(dims))))
        block = grid[sl]
        occupied += int(block.size - np.count_nonzero(block))
        block[...] = 100
    return grid


def lattice_values(limit, q):
    m = int(np.floor(limit / q + 1e-9))
    return np.arange(-m, m + 1, dtype=np.float64) * q


def wrap_angle(a):
    a = np.array(a, dtype=np.float64)
    while np.any(a > np.pi):
        a = np.where(a > np.pi, a - 2.0 * np.pi, a)
    while np.any(a < -np.pi):
        a = np.where(a < -np.pi, a + 2.0 * np.pi, a)
    return a


def random_frontier(grid, origin, res, n_nodes, seed, control, v_lim, v_q, a_lim=None, a_q=None,
                    j_lim=None, j_q=None, dt=1.0):
    """Synthetic frontier: node position = centre of a random
    non-occupied cell + uniform jitter in [-res/2, res/2), rounded to 0.01;
    derivatives drawn from the reachable lattice; t = dt * (k mod 32).
    Returns field-major [4D+2][N] float64."""
    rng = np.random.default_rng(seed)
    dim = grid.ndim
    dims = list(reversed(grid.shape))
    flat = grid.ravel()
    cells = np.empty(0, dtype=np.int64)
    while cells.size < n_nodes:
        cand = rng.integers(0, flat.size, size=2 * n_nodes)
        cand = cand[flat[cand] != 100]
        cells = np.concatenate([cells, cand])
    cells = cells[:n_nodes]
    out = np.zeros((4 * dim + 2, n_nodes), dtype=np.float64)
    rem = cells.copy()
    for i in range(dim):
        ci = rem % dims[i]
        rem //= dims[i]
        jitter = rng.uniform(-0.5, 0.5, size=n_nodes) * res
        out[i] = np.round((ci + 0.5) * res + origin[i] + jitter, 2)
    vv = lattice_values(v_lim, v_q)
    if control & 0x02:
        for i in range(dim):
            out[dim + i] = rng.choice(vv, size=n_nodes)
    if control & 0x04:
        av = lattice_values(a_lim, a_q)
        for i in range(dim):
            out[2 * dim + i] = rng.choice(av, size=n_nodes)
    if control & 0x08:
        jv = lattice_values(j_lim, j_q)
        for i in range(dim):
            out[3 * dim + i] = rng.choice(jv, size=n_nodes)
    if control & 0x10:
        out[4 * dim] = wrap_angle(0.5 * rng.integers(-6, 7, size=n_nodes))
    out[4 * dim + 1] = dt * (np.arange(n_nodes) % 32)
    return out


def potential_field(grid, res, radius_xy, radius_z=None, h_max=100, power=1.0):
    """updatePotentialMap with a global range (reference
    src/mpl_planner/map_planner.cpp:286-391): every cell with value > 0 becomes
    h_max and stamps a cone-shaped int8 stencil around itself with `max`.
    Host-side preprocessing that feeds configuration C5; exact same cell values
    as the reference's double -> int8 truncation."""
    dim = grid.ndim
    occ = grid > 0
    rn = int(np.ceil(radius_xy / res))
    # planar profile a(ox, oy) = 1 - hypot/rn for hypot <= rn, as a max-dilation
    best = np.full(grid.shape, -1.0)
    for ox in range(-rn, rn + 1):
        for oy in range(-rn, rn + 1):
            hyp = float(np.hypot(ox, oy))
            if hyp > rn:
                continue
            a = 1 - hyp / rn
            shifted = _shift(occ, ox, oy, dim)
            best = np.where(shifted & (a > best), a, best)
    out = grid.astype(np.int16).copy()
    out[occ] = h_max
    if dim == 2:
        h = h_max * np.power(np.maximum(best, 0.0), power)
        val = np.where((best >= 0) & (h > 1e-3), np.trunc(h), -128).astype(np.int16)
        out = np.maximum(out, val)
    else:
        hn = int(np.ceil(radius_z / res))
        for oz in range(-hn, hn + 1):
            b = 1 - abs(oz) / hn
            src = _shift_z(best, oz)
            h = h_max * np.power(np.maximum(src, 0.0) * b, power)
            val = np.where((src >= 0) & (h > 1e-3), np.trunc(h), -128).astype(np.int16)
            out = np.maximum(out, val)
    return out.astype(np.int8)


def _shift(a, ox, oy, dim):
    """result[.., y, x] = a[.., y - oy, x - ox] (False outside)."""
    r = np.zeros_like(a)
    ny, nx = a.shape[-2], a.shape[-1]
    ys = slice(max(0, oy), min(ny, ny + oy))
    yd = slice(max(0, -oy), min(ny, ny - oy))
    xs = slice(max(0, ox), min(nx, nx + ox))
    xd = slice(max(0, -ox), min(nx, nx - ox))
    r[..., ys, xs] = a[..., yd, xd]
    return r


def _shift_z(a, oz):
    r = np.full_like(a, -1.0)
    nz = a.shape[0]
    zs = slice(max(0, oz), min(nz, nz + oz))
    zd = slice(max(0, -oz), min(nz, nz - oz))
    r[zs] = a[zd]
    return r


def tunnel_region(dims, origin, res, p0, p1, radius):
    """Byte mask of the cells within `radius` (box dilation, as
    MapPlanner::setSearchRegion map_planner.cpp:46-95 does) of the straight
    segment p0 -> p1 sampled densely."""
    dim = len(dims)
    shape = tuple(reversed(dims))
    mask = np.zeros(shape, dtype=np.uint8)
    p0, p1 = np.asarray(p0, float), np.asarray(p1, float)
    steps = int(np.ceil(np.max(np.abs(p1 - p0)) / res / 0.8)) + 1
    rn = int(np.ceil(radius / res))
    for s in range(steps + 1):
        p = p0 + (p1 - p0) * (s / steps)
        c = np.round((p - np.asarray(origin[:dim])) / res - 0.5).astype(int)
        sl = tuple(slice(max(0, c[i] - rn), min(dims[i], c[i] + rn + 1)) for i in reversed(range(dim)))
        mask[sl] = 1
    return mask


class Workload:
    """One benchmark configuration: map + env parameters + frontier."""

    def __init__(self, name, dim, control, grid, origin, res, U, nodes, params, potential=None, region=None):
        self.name, self.dim, self.control = name, dim, control
        self.grid, self.origin, self.res = grid, list(origin), res
        self.map_dim = list(reversed(grid.shape))
        self.U, self.nodes, self.params = U, nodes, dict(params)
        self.potential, self.region = potential, region

    @property
    def n_nodes(self):
        return self.nodes.shape[1]

    @property
    def n_pairs(self):
        return self.nodes.shape[1] * self.U.shape[0]

    def apply(self, env):
        """Configure an EnvMap-like object (engine or oracle front-end)."""
        env.setMap(self.origin, self.map_dim, self.grid, self.res)
        env.set_control(self.control)
        env.set_u(self.U)
        for k, v in self.params.items():
            getattr(env, "set_" + k)(v)
        env.set_potential_map(self.potential)
        env.set_search_region(self.region)


def device_potential_fn(device=0, stats=None):
    """potential_fn for make("C5"): the potential map is produced by the engine's own reference-semantics
    MapPlanner::updatePotentialMap on the MI355X (mplx_update_potential_map, SURVEY.md 8f-3 -> 8d's C5: "256^3 as C3
    then reference-semantics updatePotentialMap").  stats (a dict) receives the wall time of the call."""
    import time

    def fn(grid, origin, res, radius):
        from .env import EnvMap
        dim = grid.ndim
        env = EnvMap(dim, device)
        env.setMap(origin, list(reversed(grid.shape)), grid, res)
        t0 = time.perf_counter()
        out = env.updatePotentialMap([0.0] * dim, radius, None, 1.0)
        if stats is not None:
            stats["potential_map_ms"] = (time.perf_counter() - t0) * 1e3
        env.close()
        return out.reshape(grid.shape)

    return fn


def make(name, scale=1.0, n_nodes=None, potential_fn=None, shell=False):
    """Build configuration `name` in {"C2","C3","C4","C5"}.  `scale` < 1
    shrinks the map edge (tests); n_nodes overrides the frontier size.
    potential_fn(grid, origin, res, radius) -> int8 map builds C5's potential map (device_potential_fn: on the
    GPU); the default is the numpy restatement potential_field (33 s at 256^3: the checker of the device path).
    shell=True: geometry, controls and parameters only -- an all-free map and an all-zero frontier of the right
    shapes -- for the ranks of a multi-GPU run that receive map and frontier from rank 0."""
    if shell:
        spec = {"C2": (2, 1024, 32, 4096), "C3": (3, 256, 16, 16384), "C4": (3, 512, 16, 65536), "C5": (3, 256, 16, 32768)}[name]
        dim, edge = spec[0], max(spec[2], int(spec[1] * scale))
        real = make(name, scale=16.0 / spec[1] if dim == 3 else 32.0 / spec[1], n_nodes=1,
                    potential_fn=(lambda g, o, r, rad: np.zeros_like(g)) if name == "C5" else None)
        grid = np.zeros((edge,) * dim, dtype=np.int8)
        nodes = np.zeros((4 * dim + 2, n_nodes or spec[3]), dtype=np.float64)
        return Workload(name, dim, real.control, grid, real.origin, real.res, real.U, nodes, real.params)
    if name == "C2":
        edge = max(32, int(1024 * scale))
        grid = box_map([edge, edge], 0.1, 0.20, 1002)
        U = grid_controls([-1, -0.5, 0, 0.5, 1], 2)
        nodes = random_frontier(grid, [0, 0], 0.1, n_nodes or 4096, 2002, ACC, 2.0, 0.5)
        return Workload("C2", 2, ACC, grid, [0, 0], 0.1, U, nodes, {"v_max": 2.0})
    if name == "C3":
        edge = max(16, int(256 * scale))
        grid = box_map([edge] * 3, 0.1, 0.15, 1003)
        U = grid_controls([-2, -1, 0, 1, 2], 3)
        nodes = random_frontier(grid, [0, 0, 0], 0.1, n_nodes or 16384, 2003, JRK, 3.0, 0.5, 2.0, 1.0)
        return Workload("C3", 3, JRK, grid, [0, 0, 0], 0.1, U, nodes, {"v_max": 3.0, "a_max": 2.0})
    if name == "C4":
        edge = max(16, int(512 * scale))
        grid = box_map([edge] * 3, 0.1, 0.15, 1004)
        U = grid_controls(np.arange(-2, 2.01, 0.5), 3)
        nodes = random_frontier(grid, [0, 0, 0], 0.1, n_nodes or 65536, 2004, ACC, 2.0, 0.5)
        return Workload("C4", 3, ACC, grid, [0, 0, 0], 0.1, U, nodes, {"v_max": 2.0})
    if name == "C5":
        edge = max(16, int(256 * scale))
        grid = box_map([edge] * 3, 0.1, 0.15, 1005)
        pot = (potential_field(grid, 0.1, 1.0, 1.0) if potential_fn is None
               else np.ascontiguousarray(potential_fn(grid, [0, 0, 0], 0.1, [1.0, 1.0, 1.0]), dtype=np.int8))
        U = grid_controls([-1, 0, 1], 3, yaw_rates=[-0.5, 0, 0.5])
        nodes = random_frontier(pot, [0, 0, 0], 0.1, n_nodes or 32768, 2005, ACCxYAW, 2.0, 0.5)
        return Workload("C5", 3, ACCxYAW, pot, [0, 0, 0], 0.1, U, nodes,
                        {"v_max": 2.0, "yaw_max": 0.5, "potential_weight": 0.5, "gradient_weight": 0.0},
                        potential=pot)
    # ---- parity-only configurations at BASELINE size (round 6): the controls BASELINE's four do not exercise
    if name == "C3-SNP":
        # C3's map, Control::SNP (state pos / vel / acc / jrk), 5^3 snap controls: primitive.h:117-119, 152-193 with
        # math.h:22-32 (quad) at full size
        edge = max(16, int(256 * scale))
        grid = box_map([edge] * 3, 0.1, 0.15, 1003)
        U = grid_controls([-2, -1, 0, 1, 2], 3)
        nodes = random_frontier(grid, [0, 0, 0], 0.1, n_nodes or 16384, 2013, SNP, 3.0, 0.5, 2.0, 0.5, 2.0, 1.0)
        return Workload("C3-SNP", 3, SNP, grid, [0, 0, 0], 0.1, U, nodes, {"v_max": 3.0, "a_max": 2.0, "j_max": 2.0})
    if name == "C2-VEL":
        # C2's map, Control::VEL (state = position only), 5^2 velocity controls up to 2 m/s
        edge = max(32, int(1024 * scale))
        grid = box_map([edge, edge], 0.1, 0.20, 1002)
        U = grid_controls([-2, -1, 0, 1, 2], 2)
        nodes = random_frontier(grid, [0, 0], 0.1, n_nodes or 4096, 2012, VEL, 2.0, 0.5)
        return Workload("C2-VEL", 2, VEL, grid, [0, 0], 0.1, U, nodes, {"v_max": 2.0})
    if name == "C2-YAWPOT":
        # C2's map with a potential field, Control::ACCxYAW, 3^2 x 3 controls: the 2D DistanceMapPlanner of
        # test/test_distance_map_planner_2d_with_yaw.cpp at C2's size
        edge = max(32, int(1024 * scale))
        grid = box_map([edge, edge], 0.1, 0.20, 1002)
        pot = (potential_field(grid, 0.1, 1.0) if potential_fn is None
               else np.ascontiguousarray(potential_fn(grid, [0, 0], 0.1, [1.0, 1.0]), dtype=np.int8))
        U = grid_controls([-1, 0, 1], 2, yaw_rates=[-0.5, 0, 0.5])
        nodes = random_frontier(pot, [0, 0], 0.1, n_nodes or 4096, 2015, ACCxYAW, 2.0, 0.5)
        return Workload("C2-YAWPOT", 2, ACCxYAW, pot, [0, 0], 0.1, U, nodes,
                        {"v_max": 2.0, "yaw_max": 0.5, "potential_weight": 0.5, "gradient_weight": 0.0}, potential=pot)
    raise ValueError("unknown workload %r" % name)


def wavefront_frontier(wl, n_nodes, device=0):
    """SURVEY.md 8(d)'s second synthetic frontier: the open list of an eps = 0 (uniform-cost) search grown from a
    free start cell until it holds n_nodes states -- realistic spatial locality and lattice velocities, unlike
    the uniformly scattered default.  The search is run by the engine's own host A* (needs a GPU).  Returns
    field-major [4D+2][n_nodes]; deterministic for a given workload."""
    from .planner import MapPlanner, MapUtil
    from .env import Waypoint
    dim = wl.dim
    flat = wl.grid.ravel()
    centre = [d // 2 for d in wl.map_dim]
    start = None
    for r in range(0, max(wl.map_dim)):
        for off in itertools.product(range(-r, r + 1), repeat=dim):
            c = [centre[i] + off[i] for i in range(dim)]
            if all(0 <= c[i] < wl.map_dim[i] for i in range(dim)):
                idx = c[0] + wl.map_dim[0] * (c[1] + (wl.map_dim[1] * c[2] if dim == 3 else 0))
                if flat[idx] == 0:
                    start = [wl.origin[i] + (c[i] + 0.5) * wl.res for i in range(dim)]
                    break
        if start:
            break
    goal = [wl.origin[i] + (wl.map_dim[i] - 0.5) * wl.res for i in range(dim)]  # far corner: never reached in time
    rows = None
    expansions = max(64, n_nodes // 64)
    for _ in range(12):
        pl = MapPlanner(dim, device=device)
        mu = MapUtil(dim)
        mu.setMap(wl.origin, wl.map_dim, flat, wl.res)
        pl.setMapUtil(mu)
        for name, setter in (("v_max", pl.setVmax), ("a_max", pl.setAmax), ("j_max", pl.setJmax)):
            if name in wl.params:
                setter(wl.params[name])
        pl.setDt(wl.params.get("dt", 1.0))
        pl.setU(wl.U)
        pl.setEpsilon(0.0)
        pl.setMaxNum(expansions)
        pl.setBatch(256)
        pl.plan(Waypoint(dim, wl.control, pos=start), Waypoint(dim, wl.control, pos=goal))
        rows = pl.getOpenStates()
        pl.close()
        if rows.shape[0] >= n_nodes:
            break
        expansions *= 2
    if rows is None or rows.shape[0] < n_nodes:
        raise RuntimeError("wavefront frontier: open list has only %d states" % (0 if rows is None else rows.shape[0]))
    return np.ascontiguousarray(rows[:n_nodes].T)
