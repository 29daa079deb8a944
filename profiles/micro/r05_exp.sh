mkdir -p gpurun_out/r05b
{ for i in 1 2; do MPLX_PLAN_PASS_TIMING=1 MPLX_PLAN_TIMING=1 python profiles/plan_split.py --edges 120,160 --batches 64,256 --reps 3 2>&1 | grep "host_planner\]\|\^3" | cut -c1-200 | grep -v "batch 64: \|batch 256: " | awk 'NR%6==5||NR%6==0'; done; } > gpurun_out/r05b/split3.log
cat gpurun_out/r05b/split3.log
