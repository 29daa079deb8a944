mkdir -p gpurun_out/r06i
for ub in 8 4 16 24; do
  if [ $ub = 8 ]; then lib=""; else lib="MPLX_LIB=$PWD/profiles/micro/libmplx_ub$ub.so"; fi
  echo "UB=$ub"; env $lib python profiles/micro/c5_sweep.py C5 - 2>/dev/null
  env $lib python profiles/micro/c5_sweep.py C3 - MPLX_GRID_LEX=0 2>/dev/null
done | tee gpurun_out/r06i/ub_sweep.txt
