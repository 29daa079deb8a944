#!/bin/bash
# per-kernel times of the node-identity pass (C4 lists, random + wavefront frontier), both partition forms:
#   bash profiles/micro/identity_kernels.sh   -> one line per (frontier, form), gpurun_out/id_prof/*_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/id_prof; mkdir -p gpurun_out/id_prof
ID_BENCH_SKIP_TABLE=1 timeout 300 rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/id_prof -o id -- python profiles/micro/identity_bench.py > gpurun_out/id_prof/bench.log 2>&1
python profiles/micro/identity_trace_split.py gpurun_out/id_prof/id_kernel_trace.csv
grep -c '"canon_equal_to_table_route": true' gpurun_out/id_prof/bench.log
rm -f gpurun_out/id_prof/id_kernel_trace.csv
