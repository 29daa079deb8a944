#!/bin/bash
# per-kernel times of the node-identity pass (C4 lists, random + wavefront frontier), both partition forms:
#   bash profiles/micro/identity_kernels.sh   -> gpurun_out/id_prof/*_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/id_prof
ID_BENCH_SKIP_TABLE=1 rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/id_prof -o id -- python profiles/micro/identity_bench.py > gpurun_out/id_prof/bench.log 2>&1
find gpurun_out/id_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'grep -E "Name|id_|post_" {} | cut -c1-200'
find gpurun_out/id_prof -name "*.db" -delete; find gpurun_out/id_prof -name "*kernel_trace.csv" -size +20M -delete
