#!/bin/bash
# TCC -> memory counters of the C4 kernel per allocation (slow against fast placement): c4_stride.py cycles allocations
# (50 dispatches each), rocprofv3 records the counters per dispatch; the summary groups them by allocation.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${PMC_TAG:-pmc_place}; rm -rf $OUT; mkdir -p $OUT
# at most two TCC counters per pass: more "exceeds the capabilities of the hardware", rocprofv3 aborts and then hangs in its
# signal handler (cost 15 GPU-minutes once) -- hence the hard timeout
CNT="${1:-TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_WRITE_GMI_32B_sum}"
timeout -s KILL 240 rocprofv3 --pmc $CNT -f csv -d $OUT -o p -- python profiles/micro/c4_stride.py 8 736 > $OUT/run.log 2>&1
grep node_stride $OUT/run.log
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "expand_grid" in r["Kernel_Name"]]
by = collections.defaultdict(dict)
for r in rows:
    by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(by)
names = sorted({n for d in by.values() for n in d})
print("dispatches", len(ids))
for g in range(0, len(ids), 50):
    grp = ids[g:g + 50][30:]   # the timed 20
    if not grp: break
    print("alloc %2d: " % (g // 50) + "  ".join("%s %.4g" % (n.replace("TCC_", "").replace("_sum", ""), sum(by[i].get(n, 0) for i in grp) / len(grp)) for n in names))
PY
