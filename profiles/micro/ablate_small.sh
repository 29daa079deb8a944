#!/bin/bash
# Where does a node's time go on the small configurations?  Timing ablations of the factorised kernel
# (MPLX_TILE_DBG bits: 1 no sampling [rows, box staging, sample loops], 2 no action/hash/state stores, 4 no cost stores,
# 8 no staging loads; results are wrong by design, only kernel_ms is read).  Run through gpurun.
B="python bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 5"
for W in C2 C3 C5 C4; do
  for DBG in 0 1 3 7 8 6; do
    r=$(MPLX_TILE_DBG=$DBG $B --workload $W 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms' % (d['roofline']['kernel_ms']))")
    echo "$W dbg=$DBG : $r"
  done
  r=$(MPLX_GRID_NOSAT=1 $B --workload $W 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms' % (d['roofline']['kernel_ms']))")
  echo "$W nosat : $r"
  r=$(MPLX_GRID_NOLEX=1 $B --workload $W 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms' % (d['roofline']['kernel_ms']))")
  echo "$W nolex : $r"
done
