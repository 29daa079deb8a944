#!/bin/bash
# Round-4 evidence, on the GPU box (run through gpurun):  profiles/run_round4.sh [tag]
#  per configuration (C4 headline, C2, C3, C5): rocprofv3 --kernel-trace --stats, two SQ counter passes;
#  C4 also FETCH_SIZE / WRITE_SIZE (each its own pass).  Summaries land in gpurun_out/<tag>/ and are copied to
#  profiles/ by hand.
set -u
TAG=${1:-r4prof}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
B="python bench.py --no-extras --no-cpu-baseline"
for W in C4 C2 C3 C5; do
  rocprofv3 --kernel-trace --stats -f csv -d "$OUT/kt_$W" -o kt -- $B --workload $W --steps 20 --warmup 5 > "$OUT/kt_$W.log" 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES -f csv -d "$OUT/p1_$W" -o p1 -- $B --workload $W --steps 3 --warmup 1 > "$OUT/p1_$W.log" 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -f csv -d "$OUT/p2_$W" -o p2 -- $B --workload $W --steps 3 --warmup 1 > "$OUT/p2_$W.log" 2>&1
done
rocprofv3 --pmc FETCH_SIZE -f csv -d "$OUT/fetch_C4" -o fetch -- $B --workload C4 --steps 3 --warmup 1 > "$OUT/fetch_C4.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -f csv -d "$OUT/write_C4" -o write -- $B --workload C4 --steps 3 --warmup 1 > "$OUT/write_C4.log" 2>&1
python - <<PY > "$OUT/summary.txt"
import csv, collections, glob, os
out = "$OUT"
for W in ("C4", "C2", "C3", "C5"):
    print("==", W)
    for f in glob.glob(out + "/kt_%s/**/*kernel_stats.csv" % W, recursive=True):
        for r in csv.DictReader(open(f)):
            if any(k in r["Name"] for k in ("expand", "compact")):
                print("  kernel_stats", r["Name"][:60], "calls", r["Calls"], "avg_ns", r["AverageNs"], "min_ns", r["MinNs"], "max_ns", r["MaxNs"])
    for p in ("p1", "p2"):
        for f in glob.glob(out + "/%s_%s/**/*counter_collection.csv" % (p, W), recursive=True):
            agg = collections.defaultdict(list)
            meta = {}
            for r in csv.DictReader(open(f)):
                if any(k in r["Kernel_Name"] for k in ("expand_lex", "expand_grid_kernel", "expand_tile")):
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    meta = {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size")}
            for k, v in sorted(agg.items()):
                print("  %-24s %.5g (n=%d)" % (k, sum(v) / len(v), len(v)))
            if meta:
                print("  ", meta)
# the timed dispatches alone: bench.py launches spin-up / placement-probe / warm-up steps before them, the stats average
# covers all of those; the last `steps` dispatches of the kernel are the timed region
for W in ("C4", "C2", "C3", "C5"):
    for f in glob.glob(out + "/kt_%s/**/*kernel_trace.csv" % W, recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in ("expand_lex", "expand_grid_kernel", "expand_tile"))]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
        if len(d) >= 20:
            print("%s kernel trace: %d dispatches, all avg %.1f ns; last 20 (the timed steps) avg %.1f ns, min %d, max %d" % (
                W, len(d), sum(d) / len(d), sum(d[-20:]) / 20, min(d[-20:]), max(d[-20:])))
for name in ("fetch", "write"):
    for f in glob.glob(out + "/%s_C4/**/*counter_collection.csv" % name, recursive=True):
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "expand_lex" in r["Kernel_Name"] or "expand_grid_kernel" in r["Kernel_Name"]]
        if vals:
            print("C4 %s_SIZE per launch (KiB as reported): %.6g (n=%d)" % (name.upper(), sum(vals) / len(vals), len(vals)))
PY
cat "$OUT/summary.txt"
mkdir -p "$OUT/csv"
for W in C4 C2 C3 C5; do
  for f in $(find "$OUT/kt_$W" -name '*kernel_stats.csv'); do cp "$f" "$OUT/csv/${W}_kernel_stats.csv"; done
done
for f in $(find "$OUT/fetch_C4" -name '*counter_collection.csv'); do cp "$f" "$OUT/csv/C4_pmc_fetch.csv"; done
for f in $(find "$OUT/write_C4" -name '*counter_collection.csv'); do cp "$f" "$OUT/csv/C4_pmc_write.csv"; done
