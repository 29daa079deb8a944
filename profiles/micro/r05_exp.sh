MPLX_PLAN_PASS_TIMING=1 MPLX_PLAN_TIMING=1 python profiles/plan_split.py --edges 120,160 --batches 256 --reps 4 2>&1 | grep "passes\|\^3" | cut -c1-300
