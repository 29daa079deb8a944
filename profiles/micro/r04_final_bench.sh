#!/bin/bash
# the round's closing evidence on one box: the default bench line, and rocprofv3 kernel stats of the same command for
# C4 (headline) and C3 (changed last: four samples per step)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04_close; mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
for W in C4 C3; do
  rocprofv3 --kernel-trace --stats -f csv -d $OUT/kt_$W -o kt -- python bench.py --no-extras --no-cpu-baseline --workload $W --steps 20 --warmup 5 > $OUT/kt_$W.log 2>&1
  cp $(find $OUT/kt_$W -name "*kernel_stats.csv" | head -1) $OUT/${W}_kernel_stats.csv
  rm -rf $OUT/kt_$W
done
python - <<PY
import json
j = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print({k: j.get(k) for k in ("value", "ms_per_step", "post_identity_ms", "wavefront_ratio_to_random")})
print(j["roofline"]["frac"], j["roofline"]["kernel_ms"], j["roofline"].get("store_only_ms"))
print({k: v.get("kernel_ms") for k, v in j.get("other_configs", {}).items() if isinstance(v, dict)})
PY
grep -h "expand" $OUT/C4_kernel_stats.csv $OUT/C3_kernel_stats.csv | cut -c1-160
