#!/usr/bin/env python3
"""Round-4 bisect of the wavefront-frontier regression (C4: 0.63 ms at b0be7b5 -> 0.91-0.97 ms from 57dcdcb on).

    python profiles/micro/wavefront_bisect.py <tree> <tag> [other.npy ...]

<tree> holds a `motion_primitive_library_amd/` package with its own built csrc/libmplx.so (variants/<sha>/ =
`git archive <sha> motion_primitive_library_amd`, or "." for HEAD).  In ONE process and ONE allocation of the lists:
the random C4 frontier, the wavefront frontier this tree's host search generates (order hash + set hash, saved as
gpurun_out/wf_<tag>.npy), and every other frontier file given (the frontiers OTHER trees generated, so that kernel
and frontier are separated), plus the own frontier shuffled and sorted by cell.  Prints one JSON line.
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

tree, tag = os.path.abspath(sys.argv[1]), sys.argv[2]
others = sys.argv[3:]
sys.path.insert(0, tree)
import motion_primitive_library_amd as m  # noqa: E402

assert os.path.abspath(m.__file__).startswith(tree), m.__file__
OUT = os.path.join(os.getcwd(), "gpurun_out")
os.makedirs(OUT, exist_ok=True)


def time_lists(env, fr, lists, steps=20, warmup=5):
    for _ in range(warmup):
        env.expand_lists_resident(fr, lists)
    env.synchronize()
    env.timer_begin()
    for _ in range(steps):
        env.expand_lists_resident(fr, lists)
    return env.timer_end() / steps


wl = m.workloads.make("C4")
res = {"tag": tag, "env": {k: v for k, v in os.environ.items() if k.startswith("MPLX_")}}
t0 = time.time()
wf = m.workloads.wavefront_frontier(wl, wl.n_nodes, 0)
res["wavefront_gen_s"] = round(time.time() - t0, 2)
np.save(os.path.join(OUT, "wf_%s.npy" % tag), wf)
res["order_sha"] = hashlib.sha1(np.ascontiguousarray(wf).tobytes()).hexdigest()[:16]
cols = np.ascontiguousarray(wf.T)
srt = cols[np.lexsort(cols.T[::-1])]
res["set_sha"] = hashlib.sha1(srt.tobytes()).hexdigest()[:16]

env = m.EnvMap(wl.dim, 0)
wl.apply(env)
lists = env.alloc_lists(wl.n_nodes, want_state=True, want_iters=False)


def run(nodes, reps=3):
    fr = env.upload_frontier(np.ascontiguousarray(nodes))
    ms = [time_lists(env, fr, lists) for _ in range(reps)]
    fr.free()
    return [round(x, 4) for x in ms]


# spin-up: the clock ramp
fr0 = env.upload_frontier(wl.nodes)
for _ in range(200):
    env.expand_lists_resident(fr0, lists)
env.synchronize()
fr0.free()
res["random"] = run(wl.nodes)
res["wavefront_own"] = run(wf)
rng = np.random.default_rng(7)
perm = rng.permutation(wf.shape[1])
res["wavefront_shuffled"] = run(wf[:, perm])
# sorted by cell (z, y, x of the position): neighbours in the array are neighbours in the map
key = np.lexsort((wf[0], wf[1], wf[2]))
res["wavefront_sorted_by_cell"] = run(wf[:, key])
res["wavefront_reversed"] = run(wf[:, ::-1])
for p in others:
    if os.path.exists(p):
        o = np.load(p)
        res["other:" + os.path.basename(p)] = {"ms": run(o), "same_order": bool(np.array_equal(o, wf)),
                                              "same_set": bool(np.array_equal(np.sort(o, axis=1), np.sort(wf, axis=1)))}
res["random_again"] = run(wl.nodes)
lists.free()
env.close()
print(json.dumps(res))
