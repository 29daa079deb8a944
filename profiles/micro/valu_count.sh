#!/bin/bash
# dynamic instructions per node of the factorised kernel on the four configurations (run through gpurun)
OUT=$PWD/gpurun_out/valu_count; mkdir -p $OUT; export TMPDIR=/tmp
for W in C2 C3 C5 C4; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -f csv -d $OUT/$W -o p -- python bench.py --no-extras --no-cpu-baseline --workload $W --steps 3 --warmup 1 > $OUT/$W.log 2>&1
  python - <<PY
import csv,collections,glob
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/$W/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'expand_grid' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
n={"C2":4096,"C3":16384,"C5":32768,"C4":65536}["$W"]
print("$W per node:", {k: round(sum(v)/len(v)/n,1) for k,v in sorted(agg.items()) if 'INSTS' in k})
PY
done
