O=gpurun_out/r04_e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_lex.py -x -q -m gpu > $O/test_lex.log 2>&1; tail -n 3 $O/test_lex.log
for w in C2 C3 C4; do MPLX_GRID_VERBOSE=1 timeout 300 python profiles/micro/env_ab.py $w --allocs 1 MPLX_GRID_WAVES_PER_CU=16 MPLX_GRID_LEX=1 MPLX_GRID_LEX=0 >> $O/ab.txt 2>> $O/ab.err; done
MPLX_GRID_VERBOSE=1 timeout 300 python profiles/micro/env_ab.py C4 --edges --allocs 1 MPLX_GRID_WAVES_PER_CU=16 MPLX_GRID_LEX=1 MPLX_GRID_LEX=0 >> $O/ab.txt 2>> $O/ab.err
cat $O/ab.txt; grep "mplx:" $O/ab.err | sort | uniq
