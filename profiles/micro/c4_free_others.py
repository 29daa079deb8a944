"""Does the time of a C4 launch into ONE allocation of the output lists depend on what ELSE is allocated?  (The round-2
driver run: four probes 0.566 / 0.565 / 0.565 / 0.564 ms, then the kept allocation timed at 0.517 ms once the other
three were freed.)  One process: allocate four pools and time each; free all but one and time it again; allocate three
again and time it again; then ballast of 25 / 100 GB.  Run on the GPU box."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import motion_primitive_library_amd as m  # noqa: E402
from motion_primitive_library_amd.env import DeviceArray  # noqa: E402

wl = m.workloads.make("C4")
env = m.EnvMap(wl.dim, 0)
wl.apply(env)
fr = env.upload_frontier(wl.nodes)
N = wl.n_nodes


def ms(lists, k=20):
    for _ in range(40):
        env.expand_lists_resident(fr, lists)
    env.synchronize()
    env.timer_begin()
    for _ in range(k):
        env.expand_lists_resident(fr, lists)
    return round(env.timer_end() / k, 4)


for trial in range(3):
    pools = [env.alloc_lists(N, want_state=True, want_iters=False) for _ in range(4)]
    t4 = [ms(p) for p in pools]
    keep = pools[trial % 4]
    for p in pools:
        if p is not keep:
            p.free()
    alone = [ms(keep) for _ in range(2)]
    again = [env.alloc_lists(N, want_state=True, want_iters=False) for _ in range(3)]
    crowded = ms(keep)
    for p in again:
        p.free()
    alone2 = ms(keep)
    ballast = DeviceArray(env, 25 << 30)
    b25 = ms(keep)
    b2 = DeviceArray(env, 75 << 30)
    b100 = ms(keep)
    ballast.free()
    b2.free()
    after = ms(keep)
    print("trial %d: four pools held %s | pool %d alone %s | + three new pools %.4f | alone %.4f | + 25 GB ballast %.4f | "
          "+ 100 GB %.4f | alone %.4f" % (trial, t4, trial % 4, alone, crowded, alone2, b25, b100, after), flush=True)
    keep.free()
