#!/bin/bash
# dynamic instructions per node under the ablations: what part of a node costs what (run through gpurun)
OUT=$PWD/gpurun_out/valu_split; mkdir -p $OUT; export TMPDIR=/tmp
for W in ${1:-C4 C2}; do for DBG in 0 1 3 7; do for EXTRA in "" "MPLX_GRID_NOSAT=1"; do
  env MPLX_TILE_DBG=$DBG $EXTRA rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM -f csv -d $OUT/${W}_${DBG}_$EXTRA -o p -- python bench.py --no-extras --no-cpu-baseline --workload $W --steps 3 --warmup 1 > $OUT/${W}_$DBG.log 2>&1
  python - <<PY
import csv,collections,glob
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/${W}_${DBG}_$EXTRA/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'expand_grid' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
n={"C2":4096,"C3":16384,"C5":32768,"C4":65536}["$W"]
print("$W dbg=$DBG $EXTRA per node:", {k.replace('SQ_INSTS_',''): round(sum(v)/len(v)/n,1) for k,v in sorted(agg.items())})
PY
done; done; done
