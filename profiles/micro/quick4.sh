#!/bin/bash
# kernel_ms of the four configurations with the in-tree library (run through gpurun); extra env passes through
for W in C2 C3 C5 C4; do
  python bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 5 --workload $W 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W %.4f ms parity=%s' % (d['roofline']['kernel_ms'], d['parity_sample_ok']))"
done
