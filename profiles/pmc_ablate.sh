export TMPDIR=/tmp
for d in ${@:-0 2}; do
OUT=$PWD/gpurun_out/pa$d; mkdir -p $OUT
MPLX_TILE_DBG=$d rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU -f csv -d $OUT -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/log 2>&1
python - <<PY
import csv,collections,glob
for f in glob.glob("$OUT/*counter_collection.csv"):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'grid' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print("dbg=$d", " ".join("%s=%.4g"%(k[3:],sum(v)/len(v)/65536) for k,v in sorted(agg.items())))
PY
done
