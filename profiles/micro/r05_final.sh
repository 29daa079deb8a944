# closing runs of round 5: the GPU suite, the smoke entry, the bench line alone and under rocprofv3
mkdir -p gpurun_out/r05z
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r05z/gputests.log 2>&1; echo "rc=$?" >> gpurun_out/r05z/gputests.log
grep -av "^t: \|^total cost\|^start\|^goal" gpurun_out/r05z/gputests.log | grep -a "passed\|failed\|rc=\|Error\|assert" | tail -6
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05z/smoke.log 2>&1; tail -1 gpurun_out/r05z/smoke.log
python bench.py > gpurun_out/r05z/bench.json 2> gpurun_out/r05z/bench.err; echo "bench rc=$?"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/r05z/kt -o kt -- python bench.py --no-extras --no-cpu-baseline > gpurun_out/r05z/bench_under_rocprof.json 2> gpurun_out/r05z/kt.err
for f in $(find gpurun_out/r05z/kt -name '*kernel_stats.csv'); do cp "$f" gpurun_out/r05z/c4_kernel_stats_final.csv; done
head -3 gpurun_out/r05z/c4_kernel_stats_final.csv | cut -c1-200
tail -c 200 gpurun_out/r05z/bench_under_rocprof.json
