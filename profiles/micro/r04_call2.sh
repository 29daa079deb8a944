#!/bin/bash
# round 4, second GPU call: ablation in whatever mode the first allocation of a fresh process gets; the whole bench
# (does the wavefront leg reproduce 0.9 ms?); the wavefront frontier as the headline frontier
O=gpurun_out/r04_call2; mkdir -p $O
timeout 300 python profiles/micro/c4_ablation.py $O/ablation_first_alloc.json > $O/ablation.log 2>&1
timeout 600 python bench.py > $O/bench_head.json 2> $O/bench_head.err
timeout 300 python bench.py --frontier wavefront --no-extras --no-cpu-baseline > $O/bench_wavefront.json 2> $O/bench_wavefront.err
timeout 300 python bench.py --frontier wavefront --no-extras --no-cpu-baseline --placement-trials 1 > $O/bench_wavefront_p1.json 2> $O/bench_wavefront_p1.err
for i in 1 2; do timeout 200 python profiles/micro/placement_table.py 8 >> $O/placement.txt 2>> $O/placement.err; done
tail -c 600 $O/ablation.log; python - <<'PY'
import json
for f in ("bench_head", "bench_wavefront", "bench_wavefront_p1"):
    try:
        p = json.loads(open("gpurun_out/r04_call2/%s.json" % f).read().strip().splitlines()[-1])
        print(f, p["ms_per_step"], p["config"].get("output_placement"), {k: (p[k].get("kernel_ms") if isinstance(p.get(k), dict) else None) for k in ("wavefront", "edges_only")})
    except Exception as e:
        print(f, "failed", e)
PY
cat $O/placement.txt
