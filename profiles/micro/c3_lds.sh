#!/bin/bash
# C3 / C2: smaller staged box (LDS) against more resident waves; run through gpurun
B="python bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 5"
for W in C3 C2; do for BC in 0 600 300 150; do for WPC in 16 24 32; do
  r=$(MPLX_GRID_BOXCAP=$BC MPLX_GRID_WAVES_PER_CU=$WPC $B --workload $W 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms parity=%s' % (d['roofline']['kernel_ms'], d['parity_sample_ok']))")
  echo "$W boxcap=$BC waves_per_cu=$WPC : $r"
done; done; done
