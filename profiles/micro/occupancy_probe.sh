#!/bin/bash
# C4: rows per pass / staged-box size pairs that change the LDS per workgroup (round 1); run through gpurun
for cfg in "4 576" "3 484" "2 500" "2 400"; do set -- $cfg
MPLX_GRID_RMAX=$1 MPLX_GRID_BOXCAP=$2 timeout 120 python bench.py --no-cpu-baseline --steps 10 --warmup 2 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('rmax=$1 boxcap=$2', round(j['roofline']['kernel_ms'],4), j.get('parity_sample_ok'))"; done
