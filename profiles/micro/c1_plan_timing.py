"""Where does C1's plan() wall time go?  The engine's host search (MPLX_PLAN_TIMING) against the reference planner with
the drop-in adapter (MPLX_ADAPTER_TIMING), same 75 launches of <= 64 nodes each.  Run on the GPU box."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["MPLX_PLAN_TIMING"] = "1"
os.environ["MPLX_ADAPTER_TIMING"] = "1"
import bench  # noqa: E402
import motion_primitive_library_amd as m  # noqa: E402
from oracle import oracle as O  # noqa: E402

c = bench.corridor_fixture()
U = m.workloads.grid_controls([-0.5, 0.0, 0.5], 2)
start, goal = m.Waypoint(2, m.ACC, pos=c["start"]), m.Waypoint(2, m.ACC, pos=c["goal"])
for batch in (64, 16, 256):
    r = bench.engine_plan(m, 2, c["origin"], c["dim"], c["cells"], c["res"], U, start, goal, 1.0, 1.0, batch, reps=5)
    print("engine batch %3d: %.3f ms, %d launches, %d expansions" % (batch, r["wall_ms"], r["launches"], r["expansions"]), flush=True)
oenv = O.Env(2, O.ACC, U, c["cells"], c["dim"], c["origin"], c["res"], v_max=1.0, a_max=1.0, dt=1.0)
for batch in (64, 16, 256):
    ad = min((O.ref_plan(oenv, start.to_row(), goal.to_row(), use_gpu=batch) for _ in range(5)), key=lambda r: r["wall_ms"])
    print("adapter batch %3d: %.3f ms, %d launches" % (batch, ad["wall_ms"], ad["device_launches"]), flush=True)
cpu = min((O.ref_plan(oenv, start.to_row(), goal.to_row(), use_gpu=False) for _ in range(5)), key=lambda r: r["wall_ms"])
print("reference CPU: %.3f ms" % cpu["wall_ms"])
