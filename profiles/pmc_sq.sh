#!/bin/bash
# SQ counter passes for the bench kernel: profiles/pmc_sq.sh <tag>
TAG=${1:-sq}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES -f csv -d $OUT/p1 -o p1 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -f csv -d $OUT/p2 -o p2 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/p2.log 2>&1
python - <<PY
import csv,collections,glob
for f in glob.glob("$OUT/p*/*counter_collection.csv"):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'grid' in r['Kernel_Name'] or 'tile' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()): print("%-24s %.4g (n=%d)"%(k,sum(v)/len(v),len(v)))
PY
tail -3 $OUT/p1.log | cut -c1-300
