#!/bin/bash
# SQ counters of the node-identity kernels (C4 lists; the bench's random + wavefront legs, both forms), two passes.
OUT=$PWD/gpurun_out/id_pmc; mkdir -p $OUT; export TMPDIR=/tmp
ID_BENCH_SKIP_TABLE=1 timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES -f csv -d $OUT/p1 -o p -- python profiles/micro/identity_bench.py > $OUT/p1.log 2>&1
ID_BENCH_SKIP_TABLE=1 timeout -s KILL 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM -f csv -d $OUT/p2 -o p -- python profiles/micro/identity_bench.py > $OUT/p2.log 2>&1
python - $OUT <<'PY'
import csv, collections, glob, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(id_\w+?_kernel(<\d>)?)", r["Kernel_Name"])
        if m:
            agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(agg.items()):
    print(k, "  ".join("%s %.3g" % (n.replace("SQ_", ""), sum(v) / len(v)) for n, v in sorted(c.items())))
PY
find $OUT -name "*.csv" -size +5M -delete
