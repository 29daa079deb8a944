#!/bin/bash
# where the engine planner's wall time goes on the 120^3 problem of profiles/plan_bench.py (MPLX_PLAN_TIMING)
MPLX_PLAN_TIMING=1 python profiles/plan_bench.py --edge 120 --skip-adapter 2>&1 | grep -E "host_planner|engine batch" | tail -20
