set -x
mkdir -p gpurun_out/r05a
lscpu | grep -E "Model name|^CPU\(s\)|L3|L2" > gpurun_out/r05a/lscpu.txt
timeout 900 python -m pytest tests/test_gpu_plan.py tests/test_lpastar.py tests/test_gpu_service.py tests/test_plan_reopen.py -x -q -m gpu > gpurun_out/r05a/plan_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r05a/plan_tests.log
tail -3 gpurun_out/r05a/plan_tests.log
MPLX_PLAN_TIMING=1 timeout 600 python profiles/plan_split.py --batches 64,256,1024 > gpurun_out/r05a/plan_split.json 2> gpurun_out/r05a/plan_split.log
tail -12 gpurun_out/r05a/plan_split.log
