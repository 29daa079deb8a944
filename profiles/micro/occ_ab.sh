#!/bin/bash
# Same-box comparison of register-occupancy variants of the factorised kernel (built with
#   python -m motion_primitive_library_amd.build --define MPLX_OCC_K3=5 ... --out profiles/micro/ab_var/libmplx_X.so)
# against the in-tree library and profiles/micro/ab_old/libmplx.so, at several resident-wave caps.
run() {  # lib workload cap
  local lib=$1 w=$2 cap=$3
  MPLX_LIB=$lib MPLX_GRID_WAVES_PER_CU=$cap MPLX_GRID_VERBOSE=1 python bench.py --workload $w --no-extras --no-cpu-baseline --steps 50 --warmup 5 2>/tmp/occ_err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$(basename $lib) $w cap=$cap kernel_ms=%.4f parity=%s' % (d['roofline']['kernel_ms'], d['parity_sample_ok']), end=' | ')"
  grep "mplx: grid" /tmp/occ_err.txt | head -1
}
NEW=motion_primitive_library_amd/csrc/libmplx.so
OLD=profiles/micro/ab_old/libmplx.so
A=profiles/micro/ab_var/libmplx_A.so
B=profiles/micro/ab_var/libmplx_B.so
for rep in 1 2; do
  for w in C5 C3 C4 C2; do run $OLD $w 16; run $NEW $w 16; done
  run $A C3 20; run $A C3 16; run $A C4 16; run $A C4 20; run $A C2 16; run $A C2 24
  run $B C5 16; run $B C5 20; run $B C4 16; run $B C4 20; run $NEW C5 12
done
