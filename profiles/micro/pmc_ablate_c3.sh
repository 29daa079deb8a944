#!/bin/bash
# Dynamic instruction counts of C3 / C2 under the timing ablations (which phase executes how many instructions);
# run through gpurun.  MPLX_TILE_DBG: 1 no sampling, 3 + no action/hash/state stores, 7 + no cost stores.
OUT=$PWD/gpurun_out/pmc_ablate; mkdir -p $OUT; export TMPDIR=/tmp
for W in C3 C2; do for DBG in 0 1 7; do
  MPLX_TILE_DBG=$DBG rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -f csv -d $OUT/${W}_$DBG -o p -- python bench.py --no-extras --no-cpu-baseline --workload $W --steps 3 --warmup 1 > $OUT/${W}_$DBG.log 2>&1
  python - <<PY
import csv,collections,glob
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/${W}_$DBG/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'expand_grid' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
print("$W dbg=$DBG", {k: round(sum(v)/len(v)) for k,v in sorted(agg.items())})
PY
done; done
