# round 5: GPU test suite, per-configuration counters, the bench line
mkdir -p gpurun_out/r05c
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/r05c/gputests.log 2>&1; echo "rc=$?" >> gpurun_out/r05c/gputests.log
grep -v "^t: \|^total cost\|^start\|^goal" gpurun_out/r05c/gputests.log | tail -5
bash profiles/run_round5_counters.sh r05ctr > gpurun_out/r05c/counters.log 2>&1
tail -5 gpurun_out/r05c/counters.log
python bench.py > gpurun_out/r05c/bench.json 2> gpurun_out/r05c/bench.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/r05c/bench.json
