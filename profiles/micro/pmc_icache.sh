#!/bin/bash
# instruction-fetch side of the factorised kernel (run through gpurun)
OUT=$PWD/gpurun_out/pmc_icache; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQC_DCACHE[A-Z_]*\|SQ_INSTS_BRANCH\|SQ_WAIT_IFETCH[A-Z_]*\|SQ_IFETCH_LEVEL" | sort -u | tr '\n' ' '; echo
for W in C3 C4; do
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES -f csv -d $OUT/$W -o p -- python bench.py --no-extras --no-cpu-baseline --workload $W --steps 3 --warmup 1 > $OUT/$W.log 2>&1
  tail -n 3 $OUT/$W.log | cut -c1-200
  python - <<PY
import csv,collections,glob
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/$W/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'expand_grid' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
print("$W", {k: round(sum(v)/len(v)) for k,v in sorted(agg.items())})
PY
done
