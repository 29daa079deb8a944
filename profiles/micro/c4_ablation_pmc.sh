#!/bin/bash
# SQ counters of every ablation variant of profiles/micro/c4_ablation.py (one rocprofv3 --pmc pass per variant)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-c4_abl}; mkdir -p $OUT
python profiles/micro/c4_ablation.py $OUT/timing.json > $OUT/timing.log 2>&1; cat $OUT/timing.log | tail -40
for v in full no_state no_stores no_sampling no_freebox compute_only; do
  timeout -s KILL 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES -f csv -d $OUT/$v -o p -- python profiles/micro/c4_ablation.py --only $v > $OUT/$v.log 2>&1
  python - $OUT/$v $v <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "expand_grid" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("%-13s " % sys.argv[2] + "  ".join("%s %.4g" % (k.replace("SQ_", ""), sum(v[-20:]) / len(v[-20:])) for k, v in sorted(agg.items())))
PY
done
