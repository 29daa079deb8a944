"""How often would raw device trig decide validate_yaw (primitive.h:504-525) differently from the host libm, and
what does the pinning (YawPin, csrc/mplx_internal.h) cost?  (VERDICT r1 "What's weak" #2.)

    python profiles/yaw_flip.py > profiles/r02_yaw_flip.json

  1. last-place differences of OCML cos / sin against glibc on 1 M random arguments;
  2. 1 M random (velocity direction, yaw) decisions d < cos(yaw_max): device trig vs glibc;
  3. full BASELINE C5 (2.65 M pairs): nodes flagged by the pinning, pairs that differ from the oracle with the pinning
     off / on (oracle = glibc on this box's host);
  4. an adversarial frontier built ON the threshold (tests/test_gpu_yaw_pin.py): flips with the pinning off / on."""
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import motion_primitive_library_amd as m  # noqa: E402
from oracle import oracle as O  # noqa: E402

out = {}
rng = np.random.default_rng(17)
env = m.EnvMap(2, 0)
x = rng.uniform(-math.pi, math.pi, size=1_000_000)
dc, ds = env.selftest_math(2, x), env.selftest_math(3, x)
hc = np.array([math.cos(v) for v in x])
hs = np.array([math.sin(v) for v in x])
out["trig_1M_random_args"] = {"cos_differs": int(np.count_nonzero(dc != hc)), "sin_differs": int(np.count_nonzero(ds != hs)),
                              "max_ulp": int(max(np.abs(dc.view(np.int64) - hc.view(np.int64)).max(),
                                                 np.abs(ds.view(np.int64) - hs.view(np.int64)).max()))}
# random decisions: unit velocity direction phi, heading x, yaw_max 0.5
phi = rng.uniform(-math.pi, math.pi, size=x.size)
vx, vy = np.cos(phi) * 1.3, np.sin(phi) * 1.3
sn = np.sqrt(vx * vx + vy * vy)
cl_h = math.cos(0.5)
cl_d = float(env.selftest_math(2, np.array([0.5]))[0])
d_h = vx / sn * hc + vy / sn * hs
d_d = vx / sn * dc + vy / sn * ds
out["decisions_1M_random"] = {"flips": int(np.count_nonzero((d_h < cl_h) != (d_d < cl_d))),
                              "within_2^-46_of_threshold": int(np.count_nonzero(np.abs(d_d - cl_d) <= 2.0 ** -46)),
                              "cos_yaw_max_device_equals_host": bool(cl_h == cl_d)}
env.close()


def run(wl_like, pin):
    os.environ["MPLX_YAW_PIN"] = "1" if pin else "0"
    if hasattr(wl_like, "apply"):
        e = m.EnvMap(wl_like.dim, 0)
        wl_like.apply(e)
        nodes = wl_like.nodes
    else:
        from test_gpu_yaw_pin import make_env
        e = make_env(m, wl_like)
        nodes = wl_like["nodes"]
    fr = e.upload_frontier(nodes)
    slots = e.alloc_slots(nodes.shape[1], want_state=False, want_iters=False)
    lists = e.alloc_lists(nodes.shape[1], want_state=True)
    e.expand_resident(fr, slots)
    e.synchronize()
    res = slots.download()
    t = []
    for _ in range(5):
        e.synchronize()
        t0 = time.perf_counter()
        e.expand_lists_resident(fr, lists)
        e.synchronize()
        t.append((time.perf_counter() - t0) * 1e3)
    stats = e.yaw_pin_stats()
    e.close()
    return res, stats, min(t)


stats = {}
wl = m.workloads.make("C5", potential_fn=m.workloads.device_potential_fn(0, stats))
oenv = O.Env(wl.dim, wl.control, wl.U, wl.grid, wl.map_dim, wl.origin, wl.res, potential=wl.potential, **wl.params)
ref = O.expand(oenv, wl.nodes, threads=os.cpu_count(), want_state=False)
off, _, t_off = run(wl, False)
on, st_on, t_on = run(wl, True)
out["C5_full"] = {"pairs": int(ref["status"].size),
                  "pairs_differing_from_oracle_pin_off": int(np.count_nonzero(off["status"] != ref["status"])),
                  "pairs_differing_from_oracle_pin_on": int(np.count_nonzero(on["status"] != ref["status"])),
                  "nodes_flagged": st_on[0], "fix_passes": st_on[1],
                  "launch_plus_sync_ms_pin_off": t_off, "launch_plus_sync_ms_pin_on": t_on}

from test_gpu_yaw_pin import oracle_of, threshold_world  # noqa: E402
wd = threshold_world(m, n_each=20000, seed=9)
ref = O.expand(oracle_of(wd), wd["nodes"], threads=os.cpu_count(), want_state=False)
off, _, _ = run(wd, False)
on, st_on, _ = run(wd, True)
out["threshold_frontier"] = {"pairs": int(ref["status"].size), "nodes": int(wd["nodes"].shape[1]),
                             "pairs_differing_from_oracle_pin_off": int(np.count_nonzero(off["status"] != ref["status"])),
                             "nodes_differing_pin_off": int(np.count_nonzero(np.any(
                                 (off["status"] != ref["status"]).reshape(wd["nodes"].shape[1], -1), axis=1))),
                             "pairs_differing_from_oracle_pin_on": int(np.count_nonzero(on["status"] != ref["status"])),
                             "nodes_flagged": st_on[0], "fix_passes": st_on[1]}
print(json.dumps(out))
