#!/bin/bash
# C4 kernel time against the number of persistent workgroups (MPLX_GRID_BLOCKS); run through gpurun
for b in 1024 1023 1021 1000 960 896 768 767 640 1016; do
MPLX_GRID_BLOCKS=$b timeout 120 python bench.py --no-cpu-baseline --steps 10 --warmup 2 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('blocks=$b', round(j['roofline']['kernel_ms'],4))"; done
