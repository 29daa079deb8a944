mkdir -p gpurun_out/r04_final
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r04_final/gpu_tests.log 2>&1; tail -3 gpurun_out/r04_final/gpu_tests.log
timeout 200 python __graft_entry__.py --smoke > gpurun_out/r04_final/smoke.log 2>&1; tail -2 gpurun_out/r04_final/smoke.log
bash profiles/micro/identity_kernels.sh > gpurun_out/r04_final/identity_per_kernel.txt 2>&1; cat gpurun_out/r04_final/identity_per_kernel.txt
cp gpurun_out/id_prof/id_kernel_stats.csv gpurun_out/r04_final/identity_kernel_stats.csv
