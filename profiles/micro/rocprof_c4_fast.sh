#!/bin/bash
# rocprofv3 --kernel-trace --stats of the driver's bench command, up to 3 times (the output placement differs per
# process): per attempt the HIP-event kernel_ms of the JSON line against the trace's last 20 dispatches.
export TMPDIR=/tmp
for i in 1 2 3; do
  OUT=$PWD/gpurun_out/kt_fast_$i; rm -rf $OUT; mkdir -p $OUT
  timeout -s KILL 150 rocprofv3 --kernel-trace --stats -f csv -d $OUT -o kt -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/run.log 2>&1
  grep '^{"metric"' $OUT/run.log > $OUT/bench.json
  python - <<PY
import csv, glob, json
d = json.loads(open("$OUT/bench.json").read())
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "expand_grid" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
print("attempt $i: HIP events %.4f ms, frac %.3f, probes %s | trace: %d dispatches, last 20 avg %.4f ms (min %.4f max %.4f), all avg %.4f ms" % (
    d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["config"]["output_placement"]["probe_ms"], len(t),
    sum(t[-20:]) / 20e6, min(t[-20:]) / 1e6, max(t[-20:]) / 1e6, sum(t) / len(t) / 1e6))
PY
done
