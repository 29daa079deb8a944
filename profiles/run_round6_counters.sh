#!/bin/bash
# Round-6 counters (the round-5 script, re-run on the round-6 binary) per configuration, on the GPU box (run through gpurun):  profiles/run_round6_counters.sh [tag]
#   for C2, C3, C4, C5: rocprofv3 --kernel-trace --stats of the bench command (the JSON written under rocprof is kept
#   beside the CSV), then --pmc SQ_INSTS_VALU (+ SALU / LDS / VMEM), --pmc FETCH_SIZE, --pmc WRITE_SIZE, each its own
#   pass (the TCC counters do not fit one pass; gpurun refuses --pmc together with the trace domains).
# Writes gpurun_out/<tag>/r06_counters.json (copied to profiles/ by hand) + the CSVs.
set -u
TAG=${1:-r06ctr}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT/csv"; export TMPDIR=/tmp
B="python bench.py --no-extras --no-cpu-baseline"
for W in ${WL:-C4 C2 C3 C5}; do
  rocprofv3 --kernel-trace --stats -f csv -d "$OUT/kt_$W" -o kt -- $B --workload $W --steps 20 --warmup 5 > "$OUT/bench_${W}_under_rocprof.json" 2> "$OUT/kt_$W.err"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES -f csv -d "$OUT/sq_$W" -o sq -- $B --workload $W --steps 3 --warmup 1 > "$OUT/sq_$W.log" 2>&1
  rocprofv3 --pmc FETCH_SIZE -f csv -d "$OUT/fetch_$W" -o fetch -- $B --workload $W --steps 3 --warmup 1 > "$OUT/fetch_$W.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE -f csv -d "$OUT/write_$W" -o write -- $B --workload $W --steps 3 --warmup 1 > "$OUT/write_$W.log" 2>&1
  for f in $(find "$OUT/kt_$W" -name '*kernel_stats.csv'); do cp "$f" "$OUT/csv/${W}_kernel_stats.csv"; done
  for n in fetch write sq; do for f in $(find "$OUT/${n}_$W" -name '*counter_collection.csv'); do cp "$f" "$OUT/csv/${W}_pmc_$n.csv"; done; done
done
python - "$OUT" <<'PY'
import csv, collections, glob, json, sys
out = sys.argv[1]
KERNELS = ("expand_lex_kernel", "expand_grid_kernel", "expand_tile_kernel", "expand_pair_kernel")
res = {}
for W in ("C2", "C3", "C4", "C5"):
    rec = {}
    agg, name = collections.defaultdict(list), None
    for tag in ("sq", "fetch", "write"):
        for f in glob.glob("%s/csv/%s_pmc_%s.csv" % (out, W, tag)):
            for r in csv.DictReader(open(f)):
                if any(k in r["Kernel_Name"] for k in KERNELS) and "prescreen" not in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    name = r["Kernel_Name"].replace("void mplx::", "").replace("(anonymous namespace)::", "").split("(")[0]
    if not agg:
        continue
    def last(k, n=3):  # the timed steps are the last dispatches of the kernel
        v = agg.get(k, [])
        return sum(v[-n:]) / max(1, len(v[-n:])) if v else None
    rec["kernel"] = name
    rec["valu_insts_per_launch"] = last("SQ_INSTS_VALU")
    for k in ("SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_WAVES"):
        rec[k.lower() + "_per_launch"] = last(k)
    fe, wr = last("FETCH_SIZE"), last("WRITE_SIZE")
    # rocprofv3 reports both in KiB (rounds 1 - 4 calibrated the write figure against the kernel's own stores: C4's
    # 2.75 GB of list entries come out as 2.747e9 B)
    rec["fetch_bytes"] = fe * 1024 if fe is not None else None
    rec["write_bytes"] = wr * 1024 if wr is not None else None
    rec["traffic_bytes"] = (fe + wr) * 1024 if fe is not None and wr is not None else None
    rec["dispatches_seen"] = {k: len(v) for k, v in agg.items()}
    for f in glob.glob("%s/csv/%s_kernel_stats.csv" % (out, W)):
        for r in csv.DictReader(open(f)):
            if any(k in r["Name"] for k in KERNELS) and "prescreen" not in r["Name"]:
                rec["kernel_stats"] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])}
    try:
        line = [l for l in open("%s/bench_%s_under_rocprof.json" % (out, W)) if l.startswith("{")][-1]
        b = json.loads(line)
        rec["bench_under_rocprof"] = {"ms_per_step": b["ms_per_step"], "kernel_ms": b["roofline"]["kernel_ms"],
                                      "algorithmic_bytes_per_launch": b["roofline"]["algorithmic_bytes_per_launch"]}
        if rec.get("traffic_bytes"):
            rec["traffic_over_algorithmic"] = rec["traffic_bytes"] / b["roofline"]["algorithmic_bytes_per_launch"]
    except Exception as e:
        rec["bench_under_rocprof"] = {"error": str(e)}
    rec["source"] = "profiles/run_round6_counters.sh: rocprofv3 --pmc SQ_INSTS_VALU.. / FETCH_SIZE / WRITE_SIZE, each its own pass; averages of the last 3 dispatches (the timed steps); KiB as reported x 1024"
    res[W] = rec
import os
prev = {}
if os.path.exists("profiles/r06_counters.json"):  # (a partial run -- WL="C5" -- keeps the other configurations' entries)
    prev = json.load(open("profiles/r06_counters.json"))
prev.update(res)
res = prev
json.dump(res, open(out + "/r06_counters.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
