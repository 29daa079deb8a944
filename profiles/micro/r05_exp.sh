mkdir -p gpurun_out/r05h
timeout 300 python -m pytest tests/test_lpastar.py -x -q -m gpu 2>&1 | grep -a "passed\|failed" | tail -2
MPLX_BENCH_SKIP_PLAN_160=1 python bench.py > gpurun_out/r05h/bench2.json 2> gpurun_out/r05h/bench2.err; echo "bench rc=$?"
python - <<'PY'
import json
b=json.loads([l for l in open('gpurun_out/r05h/bench2.json') if l.startswith('{')][-1])
r=b['plan']['replan_3D']
print({k:r['engine_lpastar'][k] for k in r['engine_lpastar'] if k.endswith('_ms')})
print({k:r['reference_cpu'][k] for k in r['reference_cpu'] if k.endswith('_ms')})
print(r['agree'], r['speedup_replan_engine_vs_reference_cpu'])
PY
