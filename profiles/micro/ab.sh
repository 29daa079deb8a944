#!/bin/bash
# A/B of two builds of libmplx.so on the same box: profiles/micro/ab_old/libmplx.so (built from an earlier commit)
# against the in-tree one.  Prints kernel_ms of each workload, alternating the libraries.
L=motion_primitive_library_amd/csrc/libmplx.so
cp $L /tmp/new.so
for rep in 1 2; do
  for which in old new; do
    if [ $which = old ]; then cp profiles/micro/ab_old/libmplx.so $L; else cp /tmp/new.so $L; fi
    for w in C4 C3 C5 C2; do
      python bench.py --workload $w --no-extras --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$which $w', d['roofline']['kernel_ms'], d['ms_per_step'])"
    done
  done
done
cp /tmp/new.so $L
