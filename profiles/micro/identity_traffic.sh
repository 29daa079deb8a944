#!/bin/bash
# HBM-side traffic of the identity pass per kernel (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, each its own pass), C4's
# random frontier, both partition forms.  KiB as reported x 1024 (FETCH_SIZE counts 128-byte requests at 64 B for wide
# coalesced loads on gfx950 -- MI355X_MICROARCH.md -- these kernels load 4 - 8 bytes per lane: uncalibrated, read as ratios).
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-id_traffic}; rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  ID_BENCH_SKIP_TABLE=1 ID_BENCH_RANDOM_ONLY=1 timeout -s KILL 300 rocprofv3 --pmc $c -f csv -d $OUT/$c -o p -- python profiles/micro/identity_bench.py > $OUT/$c.log 2>&1
done
python - $OUT <<'PY'
import csv, glob, collections, re, sys
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(sys.argv[1] + "/" + c + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(id_\w+?_kernel(<\d>)?|post_lists_kernel)", r["Kernel_Name"])
            if m:
                tot[m.group(1)][c].append(float(r["Counter_Value"]) * 1024)
claimed = ("id_part1_kernel", "id_seg_kernel", "id_part2_kernel", "id_tables_claimed_kernel")
sums = {"claimed": [0.0, 0.0], "exact": [0.0, 0.0]}
for k, v in sorted(tot.items()):
    f = sum(v["FETCH_SIZE"][2:]) / max(1, len(v["FETCH_SIZE"]) - 2); w = sum(v["WRITE_SIZE"][2:]) / max(1, len(v["WRITE_SIZE"]) - 2)
    if k != "post_lists_kernel":
        s = sums["claimed" if k in claimed else "exact"]; mult = 2 if "scan" in k else 1
        s[0] += f * mult; s[1] += w * mult
    print("%-28s fetch %8.1f MB  write %8.1f MB  (%d dispatches)" % (k, f / 1e6, w / 1e6, len(v["FETCH_SIZE"])))
for form, (f, w) in sums.items():
    print("sum %-8s fetch %8.1f MB  write %8.1f MB  (identity kernels, 20.4 M successors in 48.2 M list slots)" % (form, f / 1e6, w / 1e6))
PY
find $OUT -name "*.csv" -size +5M -delete
