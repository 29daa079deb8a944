#!/bin/bash
# Round 6: a kernel trace of BOTH output-placement modes of the C4 launch from the same binary, each from ONE process with the
# bench JSON written under rocprofv3 beside its kernel statistics (the round-5 review's item 4c).  Fresh processes, with
# 0 .. 5 allocate-touch-free cycles before the timed allocation (--prealloc-cycles), until a fast (< 0.505 ms) and a slow
# (> 0.525 ms) launch have both been seen.  Through gpurun:  bash profiles/run_round6_modes.sh [tag]
set -u
TAG=${1:-r06modes}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
fast=0; slow=0
for cyc in 0 1 2 3 4 5 0 2; do
  D="$OUT/try_${cyc}_$RANDOM"; mkdir -p "$D"
  MPLX_BENCH_DETAIL="$D/bench_detail.json" rocprofv3 --kernel-trace --stats -f csv -d "$D/kt" -o kt -- \
      python bench.py --no-extras --no-cpu-baseline --prealloc-cycles $cyc > "$D/bench_line.json" 2> "$D/err.txt"
  ms=$(python -c "import json,sys; print(json.loads([l for l in open('$D/bench_line.json') if l.startswith('{')][-1])['roofline']['kernel_ms'])" 2>/dev/null || echo 0)
  for f in $(find "$D/kt" -name '*kernel_stats.csv'); do cp "$f" "$D/kernel_stats.csv"; done
  mode=$(python -c "ms=float('$ms'); print('fast' if 0 < ms < 0.505 else ('slow' if ms > 0.525 else 'between'))")
  echo "prealloc-cycles $cyc: kernel_ms $ms -> $mode"
  grep -a "expand_lex_kernel" "$D/kernel_stats.csv" | head -1 | cut -c1-160
  if [ "$mode" = fast ] && [ $fast = 0 ]; then fast=1; cp "$D/bench_line.json" "$OUT/bench_c4_fast_under_rocprof.json"; cp "$D/kernel_stats.csv" "$OUT/c4_kernel_stats_fast.csv"; fi
  if [ "$mode" = slow ] && [ $slow = 0 ]; then slow=1; cp "$D/bench_line.json" "$OUT/bench_c4_slow_under_rocprof.json"; cp "$D/kernel_stats.csv" "$OUT/c4_kernel_stats_slow.csv"; fi
  if [ "$mode" = between ]; then cp "$D/bench_line.json" "$OUT/bench_c4_between_under_rocprof.json"; cp "$D/kernel_stats.csv" "$OUT/c4_kernel_stats_between.csv"; fi
  rm -rf "$D/kt"
  if [ $fast = 1 ] && [ $slow = 1 ]; then break; fi
done
echo "fast seen: $fast, slow seen: $slow"
