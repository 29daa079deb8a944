#!/bin/bash
# the driver's command line with and without the clock spin-up, alternating on one box
for rep in 1 2 3; do
  for sp in 0 200; do
    python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --spinup-ms $sp 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('spinup $sp: ms_per_step %.4f kernel_ms %.4f frac %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
  done
done
