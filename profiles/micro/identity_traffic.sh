#!/bin/bash
# HBM traffic of the identity pass per kernel (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, each its own pass), random frontier
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-id_traffic}; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c -f csv -d $OUT/$c -o p -- python profiles/micro/identity_bench.py > $OUT/$c.log 2>&1
done
python - $OUT <<'PY'
import csv, glob, collections, re, sys
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(sys.argv[1] + "/" + c + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(id_\w+?)(<\d>)?\(", r["Kernel_Name"])
            if m:
                tot[m.group(0).rstrip("(")][c].append(float(r["Counter_Value"]) * 1024)
sf = sw = 0
for k, v in tot.items():
    n = len(v["FETCH_SIZE"]) // 2  # first half of the dispatches: the random frontier
    f = sum(v["FETCH_SIZE"][2:n]) / max(1, n - 2); w = sum(v["WRITE_SIZE"][2:n]) / max(1, n - 2)
    mult = 2 if "scan" in k else 1
    sf += f * mult; sw += w * mult
    print("%-24s fetch %8.1f MB  write %8.1f MB" % (k, f / 1e6, w / 1e6))
print("%-24s fetch %8.1f MB  write %8.1f MB  (random frontier, 20.4 M successors)" % ("sum", sf / 1e6, sw / 1e6))
PY
