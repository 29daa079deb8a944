mkdir -p gpurun_out/r05i
timeout 120 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | grep -a "passed\|failed" | tail -1
timeout 1500 python profiles/micro/fuzz_parity.py 30000 1200 > gpurun_out/r05i/fuzz.log 2>&1; echo "rc=$?"
tail -2 gpurun_out/r05i/fuzz.log
grep -c " ok:" gpurun_out/r05i/fuzz.log
