#!/bin/bash
# Dynamic instruction counts of one configuration's kernel under MPLX_TILE_DBG ablation words (and any other environment):
#   bash profiles/micro/pmc_variants.sh C5 "MPLX_TILE_DBG=384" "MPLX_TILE_DBG=385" ...        (through gpurun)
# One rocprofv3 --pmc pass per variant; prints per kernel the average SQ_INSTS_VALU / SALU / LDS / VMEM per launch.
export TMPDIR=/tmp
W=$1; shift
i=0
for v in "$@"; do
  i=$((i+1)); OUT=$PWD/gpurun_out/pv_$W_$i; rm -rf $OUT; mkdir -p $OUT
  env $(echo $v | tr ',' ' ') rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES -f csv -d $OUT -o p -- python profiles/micro/one_config.py $W 3 > $OUT/log 2>&1
  python - "$OUT" "$v" <<'PY'
import csv, collections, glob, sys
out, v = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void mplx::", "").replace("(anonymous namespace)::", "").split("(")[0]
        if any(s in k for s in ("expand_", "prescreen")):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print("%-34s %-44s" % (v, k), " ".join("%s=%.5g" % (c[3:], sum(x) / len(x)) for c, x in sorted(d.items())))
PY
done
