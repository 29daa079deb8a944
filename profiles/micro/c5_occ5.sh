# C5 at 5 waves per SIMD (amdgpu_waves_per_eu(5): 96 VGPRs, 84 B of scratch per lane) against the default 4: one node per
# wave for the 5.2 k live nodes instead of a second round for a quarter of the waves.  Through gpurun.
mkdir -p gpurun_out/r06i
L="MPLX_LIB=$PWD/profiles/micro/libmplx_occ5.so"
( echo default; python profiles/micro/c5_sweep.py C5 - 2>/dev/null
  echo occ5; MPLX_GRID_VERBOSE=1 env $L python profiles/micro/c5_sweep.py C5 MPLX_GRID_WAVES_PER_CU=20,MPLX_GRID_RMAX=2 MPLX_GRID_WAVES_PER_CU=20,MPLX_GRID_RMAX=1 MPLX_GRID_WAVES_PER_CU=20,MPLX_GRID_RMAX=3 MPLX_GRID_WAVES_PER_CU=16 2>&1 | grep -v "^$" | sort | uniq ) | tee gpurun_out/r06i/c5_occ5.txt
