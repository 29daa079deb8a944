#!/bin/bash
# after the rows were packed: rows budget (MPLX_GRID_RMAX x (n_max + 1) slots per entry) against resident waves
B="python bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 5"
run() { r=$(MPLX_GRID_RMAX=$2 MPLX_GRID_WAVES_PER_CU=$3 $B --workload $1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms parity=%s' % (d['roofline']['kernel_ms'], d['parity_sample_ok']))"); echo "$1 rmax=$2 waves_per_cu=$3 : $r"; }
run C4 0 0; run C4 3 16; run C4 3 20; run C4 2 20; run C4 2 24
run C3 0 0; run C3 3 16; run C3 3 20; run C3 5 16; run C3 6 16; run C3 2 24
run C5 0 0; run C5 3 16; run C5 2 20; run C5 2 16; run C5 6 12
run C2 0 0; run C2 2 16; run C2 3 16
