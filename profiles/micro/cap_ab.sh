#!/bin/bash
# resident-wave cap 16 against 20 with the in-tree library, alternating, on one box
run() { env "$@" MPLX_GRID_VERBOSE=1 python bench.py --workload $W --no-extras --no-cpu-baseline --steps 50 --warmup 5 2>/tmp/e.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$W $*', 'kernel_ms=%.4f parity=%s' % (d['roofline']['kernel_ms'], d['parity_sample_ok']), end=' | ')"; grep "mplx: grid" /tmp/e.txt | head -1; }
for rep in 1 2 3; do
  for W in C4 C3 C2 C5; do
    run MPLX_GRID_WAVES_PER_CU=16
    run MPLX_GRID_WAVES_PER_CU=20
  done
done
