# host search on the GPU box's host: per-pass share and wall time of the 3D problems
mkdir -p gpurun_out/r05b
MPLX_PLAN_PASS_TIMING=1 MPLX_PLAN_TIMING=1 python profiles/plan_split.py --edges 120,160 --batches 64,256 --reps 3 2>&1 | grep "host_planner\]\|\^3" | cut -c1-330 > gpurun_out/r05b/knobs2.log
cat gpurun_out/r05b/knobs2.log
