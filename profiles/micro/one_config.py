"""A few launches of one BASELINE configuration's resident-lists kernel and nothing else (for rocprofv3 --pmc passes and
ablations: far lighter than bench.py):  python profiles/micro/one_config.py C5 [launches]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import motion_primitive_library_amd as m  # noqa: E402

name = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
wl = m.workloads.make(name, potential_fn=m.workloads.device_potential_fn(0) if name == "C5" else None)
env = m.EnvMap(wl.dim, 0)
wl.apply(env)
fr = env.upload_frontier(wl.nodes)
lists = env.alloc_lists(wl.n_nodes, want_state=True)
for _ in range(n):
    env.expand_lists_resident(fr, lists)
env.synchronize()
print(name, env.last_lists_route(), env.last_grid_kernel())
env.close()
