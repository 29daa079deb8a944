OUT=$PWD/gpurun_out/valu_c5; mkdir -p $OUT; export TMPDIR=/tmp
for Y in 1 0; do
  MPLX_GRID_LEX_YAW=$Y timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -f csv -d $OUT/y$Y -o p -- python bench.py --no-extras --no-cpu-baseline --workload C5 --steps 3 --warmup 1 --spinup-ms 0 > $OUT/y$Y.log 2>&1
  python - $OUT/y$Y $Y <<'PY'
import csv, collections, glob, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "expand_grid_kernel" in r["Kernel_Name"] or "expand_lex_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("lex_yaw=%s" % sys.argv[2], {k.replace("SQ_", ""): "%.4g" % (sum(v[-3:]) / len(v[-3:])) for k, v in sorted(agg.items())})
PY
done
