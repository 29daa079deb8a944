"""Random interleavings around the resident kernel: small batches (served by it), other calls that end it (parameters, map
edits, device-side launches, large batches, route changes, switching it off and on), idle pauses longer than its
watchdog, two contexts at once -- every small batch compared with the oracle.
    python profiles/micro/service_lifecycle_fuzz.py [seconds, default 30] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MPLX_SERVICE_IDLE_US", "400")
import motion_primitive_library_amd as m  # noqa: E402
from helpers import assert_lists_equal, engine_env, oracle_env  # noqa: E402
from oracle import oracle as O  # noqa: E402
from test_gpu_parity import _small_world  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)


class World:
    def __init__(self, dim, s):
        self.wl = _small_world(m, dim, 0x03, seed=s, n_nodes=300)
        self.wl.U = m.workloads.grid_controls([-1.0, 0.0, 1.0] if dim == 2 else [-1.0, -0.5, 0.0, 0.5, 1.0], dim)
        self.env = engine_env(m, self.wl)
        self.v_max = self.wl.params.get("v_max", 1.5)
        self.blocked = False
        self.ref = None


def oracle_with_map(w):
    """the oracle on the map and the velocity limit the engine has right now"""
    import copy
    wl = copy.copy(w.wl)
    wl.grid = np.full_like(w.wl.grid, 100) if w.blocked else w.wl.grid
    wl.params = dict(w.wl.params, v_max=w.v_max)
    return O.expand(oracle_env(wl), wl.nodes, threads=8)


worlds = [World(2, 9301), World(3, 9302)]
for w in worlds:
    w.ref = oracle_with_map(w)
ops = {"small": 0, "params": 0, "map": 0, "device": 0, "large": 0, "route": 0, "toggle": 0, "pause": 0}
t0 = time.time()
while time.time() - t0 < seconds:
    w = worlds[int(rng.integers(0, 2))]
    env, wl = w.env, w.wl
    nU = wl.U.shape[0]
    r = rng.random()
    if r < 0.80:
        n = int(rng.integers(1, 65))
        ids = rng.integers(0, wl.n_nodes, size=n)
        got = env.expand_lists(np.ascontiguousarray(wl.nodes[:, ids]))
        slots = (ids[:, None] * nU + np.arange(nU)[None, :]).ravel()
        sub = {k: (w.ref[k][:, slots] if k == "state" else w.ref[k][slots]) for k in ("status", "cost", "hash", "state", "iters")}
        assert_lists_equal(got, sub, n, nU, what="small batch after %s" % ops)
        ops["small"] += 1
    elif r < 0.84:
        w.v_max = float(rng.choice([0.6, 1.0, 1.5]))
        env.set_v_max(w.v_max)
        w.ref = oracle_with_map(w)
        ops["params"] += 1
    elif r < 0.87:
        w.blocked = not w.blocked
        env.setMap(wl.origin, wl.map_dim, np.full_like(wl.grid, 100) if w.blocked else wl.grid, wl.res)
        w.ref = oracle_with_map(w)
        ops["map"] += 1
    elif r < 0.90:
        fr = env.upload_frontier(wl.nodes)
        lists = env.alloc_lists(wl.n_nodes)
        env.expand_lists_resident(fr, lists)
        env.synchronize()
        lists.free()
        fr.free()
        ops["device"] += 1
    elif r < 0.93:
        got = env.expand_lists(wl.nodes)  # 300 nodes: above the service's limit
        assert_lists_equal(got, w.ref, wl.n_nodes, nU, what="large batch")
        ops["large"] += 1
    elif r < 0.95:
        env.set_lists_route(str(rng.choice(["auto", "tile", "grid", "dense"])))
        ops["route"] += 1
    elif r < 0.97:
        env.service(int(rng.integers(0, 2)))
        ops["toggle"] += 1
    else:
        time.sleep(float(rng.choice([0.0002, 0.001, 0.003])))
        ops["pause"] += 1
print("ok:", ops, [w.env.service() for w in worlds])
for w in worlds:
    w.env.close()
