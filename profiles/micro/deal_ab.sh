#!/bin/bash
# node chunks dealt round-robin over the 64 counters (default) against one contiguous block per counter
# (MPLX_GRID_BLOCKED=1), alternating processes on one box
run() { env "$@" python bench.py --workload $W --no-extras --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$W $*', 'kernel_ms=%.4f parity=%s' % (d['roofline']['kernel_ms'], d['parity_sample_ok']))"; }
for rep in 1 2 3; do
  for W in C4 C3 C5; do
    run MPLX_GRID_BLOCKED=1
    run MPLX_GRID_BLOCKED=0
  done
done
W=C4; for f in wavefront; do
  for b in 1 0; do env MPLX_GRID_BLOCKED=$b python bench.py --frontier wavefront --no-extras --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C4 wavefront blocked=$b', 'kernel_ms=%.4f parity=%s' % (d['roofline']['kernel_ms'], d['parity_sample_ok']))"; done
done
