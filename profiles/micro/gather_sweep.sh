#!/bin/bash
# gather mode / free-box query / occupancy on the small configurations (run through gpurun)
B="python bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 5"
for W in C2 C3 C4; do
  for G in 0 1; do for S in 0 1; do for WPC in 16 24 32; do
    r=$(MPLX_GRID_GATHER=$G MPLX_GRID_SAT=$S MPLX_GRID_WAVES_PER_CU=$WPC $B --workload $W 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms parity=%s' % (d['roofline']['kernel_ms'], d['parity_sample_ok']))")
    echo "$W gather=$G sat=$S waves_per_cu=$WPC : $r"
  done; done; done
done
