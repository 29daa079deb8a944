mkdir -p gpurun_out/r05f
timeout 600 python -m pytest tests/test_gpu_post.py tests/test_lpastar.py tests/test_gpu_plan.py -x -q -m gpu > gpurun_out/r05f/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r05f/tests.log
grep -av "^t: \|^total cost\|^start\|^goal" gpurun_out/r05f/tests.log | grep -a "passed\|failed\|rc=\|Error\|assert" | tail -6
python bench.py > gpurun_out/r05f/bench.json 2> gpurun_out/r05f/bench.err; echo "bench rc=$?"
tail -c 300 gpurun_out/r05f/bench.json
