O=gpurun_out/r04_b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_plan.py -x -q -m gpu -k "distance_map_planner_3d" > $O/plan3d.log 2>&1; tail -n 5 $O/plan3d.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_yaw_pin.py -x -q -m gpu -k "C5 or yaw" > $O/c5.log 2>&1; tail -n 3 $O/c5.log
timeout 600 python - > $O/plan_leg.json 2> $O/plan_leg.err <<'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
os.environ["MPLX_BENCH_SKIP_PLAN_160"] = "1"
import motion_primitive_library_amd as m
import bench
out = bench.extra_plan(m)
print(json.dumps({k: out[k] for k in ("distance_map_3D", "3D", "C1")}))
PY
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_b/plan_leg.json").read().strip().splitlines()[-1])
for k, v in d.items():
    print(k, json.dumps(v)[:1500])
PY
tail -n 3 $O/plan_leg.err
timeout 300 python profiles/micro/env_ab.py C5 --allocs 2 MPLX_GRID_LEX=1 >> $O/c5_time.txt 2>> $O/c5_time.err; cat $O/c5_time.txt
