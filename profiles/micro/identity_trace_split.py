"""Per-call breakdown of the node-identity kernels from a rocprofv3 kernel trace of profiles/micro/identity_bench.py:
    python profiles/micro/identity_trace_split.py gpurun_out/id_prof/id_kernel_trace.csv
One line per (frontier, form): mean microseconds per kernel over the timed calls."""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
groups, cur = [], []
for r in rows:
    m = re.search(r"(id_\w+?_kernel(<\d>)?|post_lists_kernel)", r["Kernel_Name"])
    if not m:
        continue
    cur.append((m.group(1).replace("_kernel", "").replace("id_", ""), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    if m.group(1) == "post_lists_kernel":
        groups.append(cur)
        cur = []
agg = collections.OrderedDict()
for g in groups:
    key = tuple(s for s, _ in g)
    agg.setdefault(key, []).append([d for _, d in g])
# consecutive runs of the same launch sequence = one (frontier, form) leg
legs, last = [], None
for g in groups:
    key = tuple(s for s, _ in g)
    if key != last:
        legs.append((key, []))
        last = key
    legs[-1][1].append([d for _, d in g])
for key, runs in legs:
    if len(key) == 1:
        continue
    n = len(runs)
    mean = [sum(r[i] for r in runs) / n for i in range(len(key))]
    print("%2d calls: " % n + "  ".join("%s %.1f" % (k, v) for k, v in zip(key, mean)) + "  | identity sum %.1f us" % sum(mean[:-1]))
