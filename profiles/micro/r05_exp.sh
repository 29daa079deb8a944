mkdir -p gpurun_out/r05g
timeout 600 python -m pytest tests/test_gpu_post.py tests/test_lpastar.py -x -q -m gpu 2>&1 | grep -a "passed\|failed\|rc=\|Error\|assert" | tail -4
MPLX_PLAN_PASS_TIMING=1 MPLX_PLAN_TIMING=1 python profiles/plan_split.py --edges 120,160 --batches 64,256 --reps 4 2>&1 | grep "\^3\|passes" | cut -c1-330 > gpurun_out/r05g/split.log
cat gpurun_out/r05g/split.log
