mkdir -p gpurun_out/r05y
python bench.py > gpurun_out/r05y/bench.json 2> gpurun_out/r05y/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
b=json.loads([l for l in open('gpurun_out/r05y/bench.json') if l.startswith('{')][-1])
print(b['metric'], b['value'], b['ms_per_step'], b['roofline']['frac'], b['roofline']['issue_frac'], b['roofline']['placement_modes'] is not None, b['parity_sample_ok'])
print({k:(v.get('bound'), v.get('issue_frac') and round(v['issue_frac'],3), 'cpu_baseline' in v) for k,v in b['other_configs'].items() if k in ('C2','C3','C5')})
p=b['plan']; print({k:round(p[k]['engine_host_search']['wall_ms'],2) for k in ('C1','3D','3D_160','distance_map_3D')}, p['replan_3D'].get('speedup_replan_engine_vs_reference_cpu'), p['replan_3D'].get('agree'))
print([k for k in b if isinstance(b[k],dict) and 'error' in b[k]])
PY
