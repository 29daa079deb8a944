mkdir -p gpurun_out/r05i
timeout 900 python profiles/micro/identity_fuzz.py 1000 150 > gpurun_out/r05i/identity_fuzz.log 2>&1; echo "rc=$?"
tail -2 gpurun_out/r05i/identity_fuzz.log | cut -c1-300
