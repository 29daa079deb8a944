#!/bin/bash
# dynamic VALU / SALU / LDS instructions per node of expand_grid_kernel under the timing ablations (MPLX_TILE_DBG):
#   0 full | 32 rows + staging built, no sample loop | 1 no sampling at all | 7 no sampling, no list stores
# -> sample loops = d0 - d32, rows + pass set-up + staging = d32 - d1, list stores = d1 - d7, the rest = d7
OUT=$PWD/gpurun_out/valu_phase; mkdir -p $OUT; export TMPDIR=/tmp
for W in ${@:-C3 C2 C5 C4}; do for DBG in 0 32 1 7; do
  MPLX_TILE_DBG=$DBG timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM -f csv -d $OUT/${W}_$DBG -o p -- python bench.py --no-extras --no-cpu-baseline --workload $W --steps 3 --warmup 1 --placement-trials 1 --spinup-ms 0 > $OUT/${W}_$DBG.log 2>&1
  python - $OUT/${W}_$DBG $W $DBG <<'PY'
import csv, collections, glob, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "expand_grid" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
n = {"C2": 4096, "C3": 16384, "C5": 32768, "C4": 65536}[sys.argv[2]]
print("%s dbg=%-2s per node: %s" % (sys.argv[2], sys.argv[3], {k.replace("SQ_INSTS_", ""): round(sum(v[-3:]) / len(v[-3:]) / n, 1) for k, v in sorted(agg.items())}))
PY
done; done
