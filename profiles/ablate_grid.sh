#!/bin/bash
# kernel-time ablations of expand_grid_kernel (MPLX_TILE_DBG bits: 1 no rows/box/sampling, 2 no list writes,
# 4 no cost writes, 8 no staging loads, 16 no row build); results are NOT valid outputs, timing only
for d in ${@:-0 1 2 3 10 18 26}; do MPLX_TILE_DBG=$d timeout 120 python bench.py --no-cpu-baseline --steps 10 --warmup 2 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('dbg=$d', round(j['roofline']['kernel_ms'],4), j['config']['kernel'][:20])"; done
