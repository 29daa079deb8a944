#!/bin/bash
# times C4 with every library under profiles/micro/st_variants/ (alternating, two rounds) on one box
CS=motion_primitive_library_amd/csrc
cp $CS/libmplx.so /tmp/keep.so
for rep in 1 2; do
  for d in profiles/micro/st_variants/*/; do
    name=$(basename $d)
    cp $d/libmplx.so $CS/libmplx.so
    python bench.py --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['roofline']['kernel_ms'])"
  done
done
cp /tmp/keep.so $CS/libmplx.so
