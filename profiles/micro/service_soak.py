"""Long soak of the resident kernel: requests back to back for `seconds`, every one compared with the precomputed lists of
a launch of the same batch (100 distinct batches of 1 .. 64 nodes, 2D 9 controls and 3D 125 controls).
    python profiles/micro/service_soak.py [seconds per table, default 20]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import motion_primitive_library_amd as m  # noqa: E402
from helpers import engine_env  # noqa: E402
from test_gpu_parity import _small_world  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
for dim, vals in ((2, [-1.0, 0.0, 1.0]), (3, [-1.0, -0.5, 0.0, 0.5, 1.0])):
    wl = _small_world(m, dim, 0x03, seed=9100 + dim, n_nodes=600)
    wl.U = m.workloads.grid_controls(vals, dim)
    rng = np.random.default_rng(3)
    batches = [np.ascontiguousarray(wl.nodes[:, rng.integers(0, wl.n_nodes, size=int(rng.integers(1, 65)))]) for _ in range(100)]
    ref = engine_env(m, wl)
    ref.service(0)
    want = [ref.expand_lists(b, want_iters=False) for b in batches]
    ref.close()
    env = engine_env(m, wl)
    outs = [None] * 100
    n = bad = 0
    t0 = time.time()
    while time.time() - t0 < seconds:
        for k in rng.permutation(100):
            outs[k] = env.expand_lists(batches[k], want_iters=False, out=outs[k])
            a, b = outs[k], want[k]
            S = a["stride"]
            used = (np.arange(S)[None, :] < b["count"][:, None]).ravel()
            ok = (np.array_equal(a["count"], b["count"]) and np.array_equal(a["action"][used], b["action"][used])
                  and np.array_equal(a["hash"][used], b["hash"][used])
                  and np.array_equal(a["cost"][used].view(np.uint64), b["cost"][used].view(np.uint64))
                  and np.array_equal(a["state"][:, used].view(np.uint64), b["state"][:, used].view(np.uint64)))
            bad += 0 if ok else 1
            n += 1
    st = env.service()
    print("dim %d, %d controls: %d requests in %.0f s, %d wrong; service %s" % (dim, wl.U.shape[0], n, seconds, bad, st), flush=True)
    env.close()
