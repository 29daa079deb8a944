"""Partition-camping probe.  C4's 14 state rows lie n_slots * 8 B = 368 MiB = 23 * 2^24 B apart, so the 14 stores of one
successor differ only in address bits >= 24: if the HBM channel hash depends mostly on lower bits they all queue on one
channel.  Here the rows are skewed by a few hundred bytes to kilobytes (Lists(state_pad=...)) and each variant is
allocated / timed / freed several times in one process (placement differs per allocation).  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import motion_primitive_library_amd as m

wl = m.workloads.make("C4")
env = m.EnvMap(wl.dim, 0)
wl.apply(env)
fr = env.upload_frontier(wl.nodes)
N = wl.nodes.shape[1]

def timeit(lists, k=20):
    for _ in range(30):  # clocks
        env.expand_lists_resident(fr, lists)
    env.synchronize()
    env.timer_begin()
    for _ in range(k):
        env.expand_lists_resident(fr, lists)
    return env.timer_end() / k

pads = [0, 32 * 1031, 32 * 4099]
res = {p: [] for p in pads}
for rep in range(24):
    for p in pads:
        lists = env.alloc_lists(N, want_state=True, want_iters=False, state_pad=p)
        res[p].append(round(timeit(lists), 4))
        lists.free()
for p in pads:
    print("state_pad %6d entries (%7d B): %s  min %.4f" % (p, p * 8, res[p], min(res[p])))
