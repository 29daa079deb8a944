#!/bin/bash
# Cache-policy bits of the successor-list stores of expand_grid_kernel (experiment):
#   profiles/micro/store_policy.sh build   -- here: one libmplx.so per policy under profiles/micro/st_variants/
#   profiles/micro/store_policy.sh run     -- on the GPU box: C4 kernel time with each of them, alternating
set -e
cd "$(dirname "$0")/../.."
CS=motion_primitive_library_amd/csrc
SRC="expand_kernel.hip expand_tile_kernel.hip expand_grid_kernel.hip map_prep_kernel.hip map_prep_api.cpp post_kernel.hip post_api.cpp edge_kernel.hip edge_api.cpp mplx_api.cpp planner_capi.cpp pack_kernel.hip lists_copy_api.cpp"
declare -A POL=( [builtin_nt]="" [asm_nt]="nt" [asm_plain]="" [asm_sc1]="sc1" [asm_sc0sc1]="sc0 sc1" [asm_sc0sc1nt]="sc0 sc1 nt" [asm_sc1nt]="sc1 nt" )
if [ "$1" = build ]; then
  for name in "${!POL[@]}"; do
    mkdir -p profiles/micro/st_variants/$name
    DEF="-DMPLX_ST_BUILTIN"
    if [ $name != builtin_nt ]; then DEF="-DMPLX_ST_ASM=\"${POL[$name]}\""; fi
    (cd $CS && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared "$DEF" -o ../../profiles/micro/st_variants/$name/libmplx.so $SRC) &
  done
  wait
  ls -la profiles/micro/st_variants/*/libmplx.so
else
  cp $CS/libmplx.so /tmp/keep.so
  for rep in 1 2; do
    for name in builtin_nt asm_nt asm_plain asm_sc1 asm_sc0sc1 asm_sc0sc1nt asm_sc1nt; do
      cp profiles/micro/st_variants/$name/libmplx.so $CS/libmplx.so
      python bench.py --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['roofline']['kernel_ms'])"
    done
  done
  cp /tmp/keep.so $CS/libmplx.so
fi
