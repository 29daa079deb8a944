#!/bin/bash
# builds libmplx.so of commit $1 (default HEAD) into profiles/micro/ab_old/ for profiles/micro/ab.sh
set -e
REV=${1:-HEAD}
cd "$(git rev-parse --show-toplevel)"
rm -rf gpurun_out/wt && mkdir -p gpurun_out
git worktree add -f gpurun_out/wt $REV >/dev/null 2>&1
(cd gpurun_out/wt && python -m motion_primitive_library_amd.build >/dev/null)
mkdir -p profiles/micro/ab_old
cp gpurun_out/wt/motion_primitive_library_amd/csrc/libmplx.so profiles/micro/ab_old/libmplx.so
git worktree remove --force gpurun_out/wt
echo built $REV
