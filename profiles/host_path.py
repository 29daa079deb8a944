"""PCIe-inclusive rate of the host-pointer entry point: mplx_expand_lists on BASELINE config C4
with the frontier and the successor lists in (pageable) host memory, against the HBM-resident launch bench.py times.
    python profiles/host_path.py [--workload C4] [--reps 3]
Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import motion_primitive_library_amd as m  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C4")
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    wl = m.workloads.make(a.workload)
    env = m.EnvMap(wl.dim, 0)
    wl.apply(env)
    nU = wl.U.shape[0]
    out = env.expand_lists(wl.nodes)  # warm-up: code object, device scratch, page faults of the output arrays
    emitted = int(out["count"].sum(dtype=np.int64))
    times = []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        out = env.expand_lists(wl.nodes, out=out)  # the caller's buffers are reused, as a C caller would
        times.append(time.perf_counter() - t0)
    # resident launch for comparison
    fr = env.upload_frontier(wl.nodes)
    lists = env.alloc_lists(wl.n_nodes, want_state=True, want_iters=True)
    env.expand_lists_resident(fr, lists)
    env.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        env.expand_lists_resident(fr, lists)
    env.synchronize()
    t_res = (time.perf_counter() - t0) / 10
    stride = int(out["stride"])
    F = 4 * wl.dim + 2
    bytes_out = wl.n_nodes * 4 + emitted * (4 + 8 + 8 + 4 + F * 8)  # only the used prefixes cross the link
    best = min(times)
    print(json.dumps({
        "workload": a.workload, "pairs": wl.n_pairs, "emitted": emitted,
        "host_pointer_call_ms": [round(t * 1e3, 2) for t in times],
        "host_pointer_pairs_per_s": wl.n_pairs / best,
        "bytes_copied_back": bytes_out, "copy_back_GBps": bytes_out / best / 1e9,
        "resident_launch_ms": round(t_res * 1e3, 4), "resident_pairs_per_s": wl.n_pairs / t_res,
    }))
    env.close()


if __name__ == "__main__":
    main()
