export TMPDIR=/tmp; OUT=$PWD/gpurun_out/r06v; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -f csv -d $OUT/kt -o kt -- python profiles/micro/one_config.py C5 200 > $OUT/log 2>&1
for f in $(find $OUT/kt -name '*kernel_stats.csv'); do cp $f $OUT/c5_kernel_stats.csv; done; rm -rf $OUT/kt
head -5 $OUT/c5_kernel_stats.csv | cut -c1-200
bash profiles/micro/pmc_variants.sh C5 MPLX_GRID_PAIR=1 MPLX_GRID_PAIR=0 2>&1 | tee $OUT/pmc_c5.txt
for n in fetch write; do
  C=$( [ $n = fetch ] && echo FETCH_SIZE || echo WRITE_SIZE )
  rocprofv3 --pmc $C -f csv -d $OUT/$n -o p -- python profiles/micro/one_config.py C5 3 > $OUT/$n.log 2>&1
  python - "$OUT/$n" $C <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void mplx::", "").replace("(anonymous namespace)::", "").split("(")[0]
        if "expand_" in k or "prescreen" in k: agg[k].append(float(r["Counter_Value"]))
for k, v in agg.items(): print(sys.argv[2], k, "KiB per launch: %.1f" % (sum(v[-3:]) / len(v[-3:])))
PY
done 2>&1 | tee -a $OUT/pmc_c5.txt
