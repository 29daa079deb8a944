#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, own passes) of the expansion kernel on the small configurations against B_alg
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-small_traffic}; mkdir -p $OUT
for W in C2 C3 C5; do for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $c -f csv -d $OUT/${W}_$c -o p -- python bench.py --no-extras --no-cpu-baseline --workload $W --steps 3 --warmup 1 --placement-trials 1 --spinup-ms 0 > $OUT/${W}_$c.json 2> $OUT/${W}_$c.err
done; done
python - $OUT <<'PY'
import csv, glob, json, sys
for W in ("C2", "C3", "C5"):
    v = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = []
        for f in glob.glob("%s/%s_%s/**/*counter_collection.csv" % (sys.argv[1], W, c), recursive=True):
            vals += [float(r["Counter_Value"]) * 1024 for r in csv.DictReader(open(f)) if "expand_grid" in r["Kernel_Name"]]
        v[c] = sum(vals[-3:]) / max(1, len(vals[-3:]))
    d = json.loads([l for l in open("%s/%s_FETCH_SIZE.json" % (sys.argv[1], W)) if l.startswith("{")][-1])
    b = d["roofline"]["algorithmic_bytes_per_launch"]
    print("%s: fetch %.1f MB, write %.1f MB, B_alg %.1f MB -> traffic / B_alg = %.2f" % (W, v["FETCH_SIZE"] / 1e6, v["WRITE_SIZE"] / 1e6, b / 1e6, (v["FETCH_SIZE"] + v["WRITE_SIZE"]) / b))
PY
