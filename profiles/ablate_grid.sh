for d in 0 1 2 3; do MPLX_TILE_DBG=$d timeout 120 python bench.py --no-cpu-baseline --steps 10 --warmup 2 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('dbg=$d', j['roofline']['kernel_ms'], j['config']['kernel'][:20], j.get('parity_sample_ok'))"; done
