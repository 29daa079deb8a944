#!/bin/bash
# Collects the evidence profiles/README.md cites, on the GPU box (run through gpurun):
#   profiles/run_profile.sh <tag> [bench args...]
# 1. bench line (JSON)  2. rocprofv3 --kernel-trace --stats  3/4. PMC passes (own runs).
# Everything lands under gpurun_out/<tag>/ ; the summaries worth judging are copied to profiles/ by hand.
set -u
TAG=${1:-run}; shift || true
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 3 "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -c 3000 "$OUT/bench.json"
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/kt" -o kt -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > "$OUT/kt.log" 2>&1
rocprofv3 --pmc FETCH_SIZE -f csv -d "$OUT/pmc_fetch" -o fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -f csv -d "$OUT/pmc_write" -o write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$OUT/pmc_write.log" 2>&1
find "$OUT" -name '*.csv' | head -20
