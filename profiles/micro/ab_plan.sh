#!/bin/bash
# A/B of two builds of libmplx.so (profiles/micro/ab_old/libmplx.so against the in-tree one) on the engine's 3D plan()
L=motion_primitive_library_amd/csrc/libmplx.so
cp $L /tmp/new.so
for rep in 1 2 3; do
  for which in old new; do
    if [ $which = old ]; then cp profiles/micro/ab_old/libmplx.so $L; else cp /tmp/new.so $L; fi
    python - <<PY
import sys, time, numpy as np
sys.path.insert(0, ".")
import bench, motion_primitive_library_amd as m
r = bench.extra_plan(m)
print("$which", "3D engine %.2f ms adapter %.1f ms | C1 engine %.2f ms adapter %.2f ms" % (r["3D"]["engine_host_search"]["wall_ms"], r["3D"]["reference_planner_gpu_adapter"]["wall_ms"], r["C1"]["engine_host_search"]["wall_ms"], r["C1"]["reference_planner_gpu_adapter"]["wall_ms"]))
PY
  done
done
cp /tmp/new.so $L
