#!/bin/bash
# rocprofv3 kernel stats of any command, on the GPU box:  profiles/micro/kstats.sh <tag> <command...>
# prints the top kernels (calls, average ns, share); the CSV stays under gpurun_out/<tag>/
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/kt" -o kt -- "$@" > "$OUT/kt.log" 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/kt/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:int(20)]:
        print("%-70s calls %6s avg %10.1f us  %5s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
