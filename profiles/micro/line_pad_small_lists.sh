export TMPDIR=/tmp
python profiles/micro/c5_sweep.py C5 - MPLX_NO_LINE_PAD=1 2>/dev/null | tail -1
python profiles/micro/c5_sweep.py C3 - MPLX_NO_LINE_PAD=1 2>/dev/null | tail -1
python profiles/micro/c5_sweep.py C2 - MPLX_NO_LINE_PAD=1 2>/dev/null | tail -1
for v in "" "MPLX_NO_LINE_PAD=1"; do
for C in FETCH_SIZE WRITE_SIZE; do
  OUT=/tmp/pmc_$C; rm -rf $OUT
  env $v rocprofv3 --pmc $C -f csv -d $OUT -o p -- python profiles/micro/one_config.py C5 3 > /dev/null 2>&1
  python - "$OUT" "$C" "$v" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void mplx::", "").replace("(anonymous namespace)::", "").split("(")[0]
        if "expand_" in k or "prescreen" in k: agg[k].append(float(r["Counter_Value"]))
for k, v in agg.items(): print(sys.argv[3] or "default", sys.argv[2], k, "KiB per launch: %.1f" % (sum(v[-3:]) / len(v[-3:])))
PY
done; done
