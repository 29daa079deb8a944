#!/bin/bash
# C4: resident waves per CU against rows per pass (LDS per workgroup decides how many workgroups fit); run through gpurun
B="python bench.py --no-extras --no-cpu-baseline --steps 40 --warmup 5 --workload C4"
for cfg in "0 0" "3 16" "3 20" "2 20" "2 24" "3 24"; do
  set -- $cfg
  r=$(MPLX_GRID_RMAX=$1 MPLX_GRID_WAVES_PER_CU=$2 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms parity=%s' % (d['roofline']['kernel_ms'], d['parity_sample_ok']))")
  echo "C4 rmax=$1 waves_per_cu=$2 : $r"
done
