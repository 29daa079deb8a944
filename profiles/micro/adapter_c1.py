import json, os, sys
sys.path.insert(0, os.getcwd())
import motion_primitive_library_amd as m
import bench
from oracle import oracle as O
c = bench.corridor_fixture()
U = m.workloads.grid_controls([-0.5, 0.0, 0.5], 2)
start, goal = m.Waypoint(2, m.ACC, pos=c["start"]), m.Waypoint(2, m.ACC, pos=c["goal"])
oenv = O.Env(2, O.ACC, U, c["cells"], c["dim"], c["origin"], c["res"], v_max=1.0, a_max=1.0, dt=1.0)
ad = sorted(O.ref_plan(oenv, start.to_row(), goal.to_row(), use_gpu=64)["wall_ms"] for _ in range(7))
print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("MPLX_")}, "adapter_ms": [round(x, 3) for x in ad]}))
