#!/bin/bash
# Rows-per-pass sweep of the factorised kernel (run through gpurun): profiles/micro/rmax_sweep.sh
B="python bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 5"
for W in C2 C3 C5 C4; do
  for RMAX in 0 6 8 12 16 24; do
    r=$(MPLX_GRID_RMAX=$RMAX $B --workload $W 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms  parity=%s' % (d['roofline']['kernel_ms'], d['parity_sample_ok']))")
    echo "$W rmax=$RMAX : $r"
  done
done
