O=gpurun_out/r04_lex2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "grid or wavefront" > $O/fullsize.log 2>&1
tail -n 5 $O/fullsize.log
for w in C2 C3 C4; do timeout 300 python profiles/micro/env_ab.py $w MPLX_GRID_LEX=0 MPLX_GRID_LEX=1 >> $O/ab.txt 2>> $O/ab.err; done
timeout 300 python profiles/micro/env_ab.py C4 --edges MPLX_GRID_LEX=0 MPLX_GRID_LEX=1 >> $O/ab.txt 2>> $O/ab.err
timeout 300 python profiles/micro/env_ab.py C4 --wavefront --allocs 2 MPLX_GRID_LEX=0 MPLX_GRID_LEX=1 >> $O/ab.txt 2>> $O/ab.err
cat $O/ab.txt; tail -n 3 $O/ab.err
