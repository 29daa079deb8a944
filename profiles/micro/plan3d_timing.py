"""Where does the 3D plan() (120^3 voxels, |U| = 729, 64 nodes per launch) spend its time, with the search's batches as
launches of their own (MPLX_SERVICE=0) and through the resident kernel?   python profiles/micro/plan3d_timing.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    os.environ["MPLX_PLAN_TIMING"] = "1"
    import numpy as np
    import bench
    import motion_primitive_library_amd as m
    W = m.workloads
    edge, res = int(sys.argv[2]), 0.1
    grid = W.box_map([edge] * 3, res, 0.08, 4242, side_m=(0.5, 2.5))
    flat = grid.ravel()
    U3 = W.grid_controls(np.linspace(-2.0, 2.0, 9), 3)

    def free_near(p):
        cc = np.array([int(x / res) for x in p])
        for r in range(0, 30):
            for d in np.ndindex(2 * r + 1, 2 * r + 1, 2 * r + 1):
                q = cc + np.array(d) - r
                if np.all(q >= 0) and np.all(q < edge) and flat[q[0] + edge * (q[1] + edge * q[2])] == 0:
                    return [(q[i] + 0.5) * res for i in range(3)]
        raise RuntimeError("no free cell")

    s3 = m.Waypoint(3, m.ACC, pos=free_near([1.0, 1.0, 1.0]))
    g3 = m.Waypoint(3, m.ACC, pos=free_near([edge * res - 1.0, edge * res - 1.2, edge * res - 1.5]))
    for batch in (16, 64, 256):
        r = bench.engine_plan(m, 3, [0.0] * 3, [edge] * 3, flat, res, U3, s3, g3, 2.0, 2.0, batch, reps=3)
        print("batch %3d: %.2f ms, %d launches, %d expansions" % (batch, r["wall_ms"], r["launches"], r["expansions"]), flush=True)
else:
    for svc in ("0", "1"):
        env = dict(os.environ, MPLX_SERVICE=svc)
        print("MPLX_SERVICE=%s" % svc, flush=True)
        out = subprocess.run([sys.executable, __file__, "child", "120"], env=env, capture_output=True, text=True)
        lines = (out.stdout + out.stderr).splitlines()
        keep = [l for l in lines if l.startswith("batch")]
        prov = [l for l in lines if "host_planner" in l]
        for k, l in enumerate(keep):
            print("  " + l + "   | last run: " + (prov[4 * k + 3] if len(prov) > 4 * k + 3 else "?"))
