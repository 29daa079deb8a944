#!/bin/bash
# C5: rows per pass (MPLX_GRID_RMAX) against resident workgroups (45 KB of LDS per workgroup = 3 resident, 35 KB = 4); run through gpurun
run() { env "$@" MPLX_GRID_VERBOSE=1 python bench.py --workload C5 --no-extras --no-cpu-baseline --steps 50 --warmup 5 2>/tmp/e.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', 'kernel_ms=%.4f parity=%s' % (d['roofline']['kernel_ms'], d['parity_sample_ok']), end=' | ')"; grep "mplx: grid" /tmp/e.txt | head -1; }
for rep in 1 2; do
run A=1
run MPLX_GRID_RMAX=3
run MPLX_GRID_RMAX=2
run MPLX_GRID_RMAX=3 MPLX_GRID_WAVES_PER_CU=12
done
