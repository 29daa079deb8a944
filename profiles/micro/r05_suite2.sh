mkdir -p gpurun_out/r05e
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r05e/gputests.log 2>&1; echo "rc=$?" >> gpurun_out/r05e/gputests.log
grep -av "^t: \|^total cost\|^start\|^goal" gpurun_out/r05e/gputests.log | grep -a "passed\|failed\|rc=\|Error\|assert" | tail -12
python bench.py > gpurun_out/r05e/bench.json 2> gpurun_out/r05e/bench.err; echo "bench rc=$?"
tail -c 600 gpurun_out/r05e/bench.json
