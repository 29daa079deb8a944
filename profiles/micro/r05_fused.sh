mkdir -p gpurun_out/r05d
timeout 900 python -m pytest tests/test_gpu_post.py tests/test_gpu_plan.py tests/test_gpu_service.py tests/test_gpu_lists.py -x -q -m gpu > gpurun_out/r05d/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r05d/tests.log
grep -av "^t: \|^total cost\|^start\|^goal" gpurun_out/r05d/tests.log | tail -30
for h in 0 1; do
  if [ $h = 1 ]; then export MPLX_PLAN_HOST_HEUR=1; fi
  MPLX_PLAN_TIMING=1 python profiles/plan_split.py --edges 120,160 --batches 64,256 --reps 3 2>&1 | grep "\^3" | cut -c1-420
done > gpurun_out/r05d/heur_ab.log
cat gpurun_out/r05d/heur_ab.log
