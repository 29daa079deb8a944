#!/bin/bash
# Occupancy / LDS sweep of the factorised kernel on the small configurations (run through gpurun):
#   profiles/micro/occupancy_sweep.sh > gpurun_out/occ_sweep.txt
# kernel_ms of `bench.py --workload W` for caps on resident waves per CU and rows per pass.
B="python bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 5"
for W in C5 C3 C2; do
  for WPC in 0 16 20 24 28 32; do
    for RMAX in 0 2 1; do
      r=$(MPLX_GRID_WAVES_PER_CU=$WPC MPLX_GRID_RMAX=$RMAX $B --workload $W 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms  parity=%s' % (d['roofline']['kernel_ms'], d['parity_sample_ok']))")
      echo "$W waves_per_cu=$WPC rmax=$RMAX : $r"
    done
  done
done
