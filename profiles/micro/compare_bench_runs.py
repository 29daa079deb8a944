"""CPU baselines of two bench runs of one box side by side: python profiles/micro/compare_bench_runs.py DIR with DIR/line_{1,2}.json
(the stdout of `python bench.py`) and DIR/detail_{1,2}.json (MPLX_BENCH_DETAIL).  Prints, per configuration, the two all-core medians,
their difference, the repetition lengths and spreads, and the one-thread rates."""
import json, sys
d0=sys.argv[1]
r={}
for i in (1,2):
    raw=open(d0+"/line_%d.json"%i).read().strip().splitlines()
    print("stdout lines:", len(raw), "bytes:", len(raw[-1]))
    d=json.load(open(d0+"/detail_%d.json"%i))
    r[i]={"C4":d["cpu_baseline"]}
    for k,v in (d.get("other_configs") or {}).items():
        if isinstance(v,dict) and isinstance(v.get("cpu_baseline"),dict): r[i][k]=v["cpu_baseline"]
    l=json.loads(raw[-1]); print(i, l["value"], l["ms_per_step"], l["roofline"]["frac"])
for k in r[1]:
    a,b=r[1][k],r[2][k]
    pa,pb=a["protocol"]["all_cores"],b["protocol"]["all_cores"]
    print(k, "%.3g %.3g diff %.1f%%"%(a["value"],b["value"],abs(a["value"]-b["value"])/min(a["value"],b["value"])*100), "rep_s %.2f %.2f spread %.2f %.2f"%(pa["rep_seconds"],pb["rep_seconds"],pa["spread"],pb["spread"]), "1thr %.3g %.3g"%(a["protocol"]["one_thread"]["value"],b["protocol"]["one_thread"]["value"]))
