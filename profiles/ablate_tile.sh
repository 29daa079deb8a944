mkdir -p gpurun_out/abl
for d in 0 8 1 2 4 16 3 7; do
  MPLX_TILE_DBG=$d python bench.py --no-cpu-baseline --steps 10 --warmup 2 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('dbg=$d kernel_ms=%.4f'%j['roofline']['kernel_ms'])
" | tee -a gpurun_out/abl/abl.txt
done
