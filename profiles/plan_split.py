"""Where the engine planner's wall time goes on the 3D problems of the bench line (120^3: 3.3 k expansions, 160^3:
64 k expansions; ACC, |U| = 729): wall time, launches and the split of mplx_planner_timing, for a few launch sizes.

    python profiles/plan_split.py [--edges 120,160] [--batches 64,256]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import motion_primitive_library_amd as m  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--edges", default="120,160")
ap.add_argument("--batches", default="64,256")
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
W = m.workloads
out = {}
for edge in [int(x) for x in a.edges.split(",")]:
    res = 0.1
    flat = W.box_map([edge] * 3, res, 0.08, 4242, side_m=(0.5, 2.5)).ravel()
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
    for b in [int(x) for x in a.batches.split(",")]:
        r = bench.engine_plan(m, 3, [0.0] * 3, [edge] * 3, flat, res, U3, s3, g3, 2.0, 2.0, b, reps=a.reps)
        out["%d^3 batch %d" % (edge, b)] = r
        print("%d^3 batch %d: %.1f ms, %d launches, %s" % (edge, b, r["wall_ms"], r["launches"], r["timing_split"]), file=sys.stderr)
print(json.dumps(out))
