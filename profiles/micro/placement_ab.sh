#!/bin/bash
# the driver's command line, with and without the placement probe, alternating processes on one box
for rep in 1 2 3 4; do
  for pt in 1 4; do
    python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --placement-trials $pt 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('trials $pt: ms_per_step %.4f kernel_ms %.4f frac %.3f parity %s probes %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['parity_sample_ok'], d['config']['output_placement']))"
  done
done
