#!/bin/bash
# round 4: dynamic instructions per node of the two factorised kernels (MPLX_GRID_LEX=1: expand_lex_kernel, 0: expand_grid_kernel)
# on the BASELINE configurations; MPLX_TILE_DBG ablations as in valu_phase_split.sh.  Run through gpurun.
OUT=$PWD/gpurun_out/valu_r04; mkdir -p $OUT; export TMPDIR=/tmp
for W in ${WL:-C2 C3 C4}; do for LEX in ${LEXS:-1 0}; do for DBG in ${DBGS:-0}; do
  MPLX_GRID_LEX=$LEX MPLX_TILE_DBG=$DBG timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM -f csv -d $OUT/${W}_${LEX}_$DBG -o p -- python bench.py --no-extras --no-cpu-baseline --workload $W --steps 3 --warmup 1 --placement-trials 1 --spinup-ms 0 > $OUT/${W}_${LEX}_$DBG.log 2>&1
  python - $OUT/${W}_${LEX}_$DBG $W $LEX $DBG <<'PY'
import csv, collections, glob, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "expand_grid_kernel" in r["Kernel_Name"] or "expand_lex_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
n = {"C2": 4096, "C3": 16384, "C5": 32768, "C4": 65536}[sys.argv[2]]
print("%s lex=%s dbg=%-2s per node: %s   per launch VALU %.4g" % (sys.argv[2], sys.argv[3], sys.argv[4], {k.replace("SQ_INSTS_", ""): round(sum(v[-3:]) / len(v[-3:]) / n, 1) for k, v in sorted(agg.items())}, sum(agg["SQ_INSTS_VALU"][-3:]) / max(1, len(agg["SQ_INSTS_VALU"][-3:]))))
PY
done; done; done
