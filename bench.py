#!/usr/bin/env python3
"""bench.py -- throughput of the successor-expansion hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C4]

A "step" is one pass of the hot path (one mplx_expand_device launch) over one
synthetic frontier batch that is already resident in HBM.  The default workload
is BASELINE.json configs[3], the one the metric is quoted on:
    C4 = 3D VoxelMapUtil 512^3, Control::ACC, |U| = 729, 64k-node frontier.
N > 1 (launched by torch.distributed.run, one rank per GPU): the frontier is
sharded by node, every rank expands its own 64k-node shard against its own
replica of the map -- no data-path collective -- so scaling is
"weak" and value = all ranks' pairs / max-over-ranks time.

Rank 0 prints ONE JSON line (see the task contract) including
  roofline     : algorithmic bytes per launch / HIP-event kernel time vs 8 TB/s
  cpu_baseline : the reference's own headers compiled into oracle/_ref ("reference";
                 the CPU oracle, "port", when that build is absent) timed on this
                 box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

WORKLOAD_DESC = {
    "C2": "C2: 2D OccMapUtil 1024x1024, Control::ACC |U|=25, 4k-node synthetic frontier",
    "C3": "C3: 3D VoxelMapUtil 256^3, Control::JRK |U|=125, 16k-node synthetic frontier",
    "C4": "C4: 3D VoxelMapUtil 512^3, Control::ACC |U|=729 (9^3), 64k-node synthetic frontier",
    "C5": "C5: 3D 256^3 potential map, Control::ACCxYAW |U|=81, 32k-node synthetic frontier",
}


def algorithmic_bytes(wl, n_emit, n_samples):
    """B_alg = N*S_wp + |U|*udim*8 + samples*(1 + r/8) + N_emit*(S_wp + 8 + 4)."""
    s_wp = (4 * wl.dim + 2) * 8
    r = 1 if wl.region is not None else 0
    return (wl.n_nodes * s_wp + wl.U.size * 8 + n_samples * (1 + r / 8.0) + n_emit * (s_wp + 8 + 4))


def measured_traffic(workload, kernel):
    """HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, each its own
    run; profiles/README.md): bench.py cannot collect counters itself, so it reports the figure measured for
    this workload + kernel, or None when no such profile is committed."""
    path = os.path.join(ROOT, "profiles", "r01_%s_traffic.json" % workload.lower())
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None
    return rec["traffic_bytes"] if rec.get("kernel", "").split("<")[0] in kernel else None


def cpu_baseline(wl, target_seconds=12.0):
    """Times the CPU oracle (reference-structured port) on a bounded sample of
    the workload's frontier, on all host cores and on one core."""
    from oracle import oracle as O

    oenv = O.Env(wl.dim, wl.control, wl.U, wl.grid, wl.map_dim, wl.origin, wl.res,
                 potential=wl.potential, region=wl.region, **wl.params)
    cores = os.cpu_count() or 1
    nU = wl.U.shape[0]
    probe = min(wl.n_nodes, max(cores * 8, 64))
    sec, _ = O.time_expand(oenv, wl.nodes[:, :probe], threads=cores, reps=1)
    rate = probe * nU / max(sec, 1e-9)
    n = int(min(wl.n_nodes, max(probe, rate * target_seconds * 0.6 / nU)))
    sec_all, st = O.time_expand(oenv, wl.nodes[:, :n], threads=cores, reps=2)
    n1 = int(max(8, min(n, n // cores)))
    sec_1, _ = O.time_expand(oenv, wl.nodes[:, :n1], threads=1, reps=1)
    port = {"value": n * nU / sec_all, "cores": cores, "value_1thread": n1 * nU / sec_1}
    sample = "first %d of %d frontier nodes x %d controls (%d pairs, %d map samples) of %s" % (
        n, wl.n_nodes, nU, n * nU, st["samples"], wl.name)
    if os.path.exists(O.REF_SO):
        # the reference's own headers (env_map.h / primitive.h / map_util.h) compiled where they lay into
        # oracle/_ref/libmpl_ref.so (prebuilt; Eigen / Boost replaced by the stand-in headers of oracle/stub_include,
        # neither is installed): the reference's CPU path itself, one env_map per thread over one shared MapUtil
        sec_r, _ = O.time_expand(oenv, wl.nodes[:, :n], threads=cores, reps=2, ref=True)
        sec_r1, _ = O.time_expand(oenv, wl.nodes[:, :n1], threads=1, reps=1, ref=True)
        return {
            "value": n * nU / sec_r, "unit": "pairs/s", "cores": cores, "kind": "reference",
            "sample": sample + " in %.2f s on %d threads (reference headers built against stand-in Eigen/Boost); "
                               "1 thread: %.4g pairs/s on %d nodes" % (sec_r, cores, n1 * nU / sec_r1, n1),
            "value_1thread": n1 * nU / sec_r1,
            "port": port,
        }, oenv, n
    return {
        "value": port["value"], "unit": "pairs/s", "cores": cores, "kind": "port",
        "sample": sample + " in %.2f s on %d threads; 1 thread: %.4g pairs/s on %d nodes" % (
            sec_all, cores, port["value_1thread"], n1),
        "value_1thread": port["value_1thread"],
    }, oenv, n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C4", choices=sorted(WORKLOAD_DESC))
    ap.add_argument("--scale", type=float, default=1.0, help="map edge scale (debug only; invalidates the metric)")
    ap.add_argument("--nodes", type=int, default=None, help="frontier size override (debug only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--frontier", default="random", choices=["random", "wavefront"],
                    help="random: SURVEY 8(d)'s uniformly scattered frontier (the headline); wavefront: the open list of "
                         "an eps = 0 search from the map centre (realistic locality; reported in profiles/README.md)")
    ap.add_argument("--output", default="lists", choices=["lists", "dense", "dense-compact"],
                    help="lists: per-node successor lists, the reference's output shape (count, action, cost, hash, "
                         "full Waypoint) -- default; dense: one 129-B slot per pair; dense-compact: status+cost+hash only")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py: --gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`"
                     % (args.gpus, args.gpus))
        args.gpus = world

    import torch
    import torch.distributed as dist

    import motion_primitive_library_amd as m

    if not torch.cuda.is_available():
        sys.exit("bench.py: no GPU visible; the engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    # MPLX_BENCH_FORCE_DIST=1 runs the N > 1 code path (RCCL init, barrier, all-reduce of the timing) with whatever
    # world size the launcher gave, including 1 -- the only way to exercise it on a one-GPU box
    force_dist = os.environ.get("MPLX_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    # ---- build the workload; rank r gets shard r of the global frontier
    t_gen = time.time()
    wl = m.workloads.make(args.workload, scale=args.scale, n_nodes=args.nodes)
    if rank > 0:
        seed = {"C2": 2002, "C3": 2003, "C4": 2004, "C5": 2005}[args.workload] + 100 * rank
        kw = {"C2": (2.0, 0.5), "C3": (3.0, 0.5, 2.0, 1.0), "C4": (2.0, 0.5), "C5": (2.0, 0.5)}[args.workload]
        wl.nodes = m.workloads.random_frontier(wl.grid, wl.origin, wl.res, wl.n_nodes, seed, wl.control, *kw)
    if args.frontier == "wavefront":
        wl.nodes = m.workloads.wavefront_frontier(wl, wl.n_nodes, local_rank)
    t_gen = time.time() - t_gen

    env = m.EnvMap(wl.dim, local_rank)
    wl.apply(env)
    frontier = env.upload_frontier(wl.nodes)
    if args.output == "lists":
        slots = env.alloc_lists(wl.n_nodes, want_state=True, want_iters=False)
        launch = lambda: env.expand_lists_resident(frontier, slots)
    else:
        slots = env.alloc_slots(wl.n_nodes, want_state=(args.output == "dense"), want_iters=False)
        launch = lambda: env.expand_resident(frontier, slots)
    dev_name, cus = env.device_info()

    # ---- one untimed verification launch: counts for the algorithmic bytes
    vs = env.alloc_slots(wl.n_nodes, want_state=False, want_iters=True)
    env.expand_resident(frontier, vs)
    env.synchronize()
    v = vs.download()
    vs.free()
    n_emit = int(np.count_nonzero((v["status"] == 1) | (v["status"] == 2)))
    n_finite = int(np.count_nonzero(v["status"] == 1))
    n_samples = int(v["iters"].sum(dtype=np.int64))
    b_alg = algorithmic_bytes(wl, n_emit, n_samples)

    def barrier():
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        launch()
    barrier()
    t0 = time.perf_counter()
    env.timer_begin()
    for _ in range(args.steps):
        launch()
    kernel_ms_total = env.timer_end()  # HIP events on the engine's own stream
    barrier()
    elapsed = time.perf_counter() - t0

    if world > 1 or force_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        cnt = torch.tensor([wl.n_pairs], dtype=torch.float64, device="cuda")
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        total_pairs = float(cnt.item())
    else:
        total_pairs = float(wl.n_pairs)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        out_kernel = {"grid": "expand_grid_kernel", "tile": "expand_tile_kernel", "dense": "expand_kernel",
                      "none": "expand_kernel"}[env.last_lists_route()]
        kernel_ms = kernel_ms_total / args.steps
        achieved = b_alg / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "node-expansions/s (frontier x |U| pair evaluations per second)",
            "value": total_pairs * args.steps / elapsed,
            "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": WORKLOAD_DESC[args.workload] + ("" if args.scale == 1.0 and args.nodes is None else
                                                           " [DEBUG scale=%g nodes=%s]" % (args.scale, args.nodes)),
                "frontier": args.frontier, "frontier_nodes_per_gpu": wl.n_nodes, "controls": int(wl.U.shape[0]), "dim": wl.dim,
                "pairs_per_step_per_gpu": wl.n_pairs, "map_cells": int(wl.grid.size),
                "output": {"lists": "per-node successor lists: count + action + cost + hash + full Waypoint (4D+2 doubles), emitted successors only",
                           "dense": "dense slots: status + cost + hash + full Waypoint for every pair",
                           "dense-compact": "dense slots: status + cost + hash"}[args.output],
                "sharding": "frontier nodes block-partitioned over ranks, map replicated, no collective",
                "device": dev_name, "compute_units": cus,
                "kernel": {"grid": "expand_grid_kernel (per-axis factorised tables in LDS)",
                           "tile": "expand_tile_kernel", "dense": "expand_kernel + compact_lists_kernel",
                           "none": "expand_kernel"}[env.last_lists_route()],
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": measured_traffic(args.workload, out_kernel) if args.output == "lists" else None,
                "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": b_alg,
                "emitted": n_emit, "finite": n_finite, "map_samples": n_samples,
            },
        }
        if not args.no_cpu_baseline and world >= 1:
            cb, oenv, n_chk = cpu_baseline(wl)
            out["cpu_baseline"] = cb
            out["speedup_vs_cpu_all_cores"] = out["value"] / cb["value"]
            # cheap consistency check of the measured run against the oracle on a slice
            from oracle import oracle as O
            n_chk = min(n_chk, 512)
            ref = O.expand(oenv, wl.nodes[:, :n_chk], threads=os.cpu_count() or 1, want_state=False)
            k = n_chk * wl.U.shape[0]
            out["parity_sample_ok"] = bool(np.array_equal(v["status"][:k], ref["status"]) and
                                           np.array_equal(v["hash"][:k], ref["hash"]) and
                                           np.array_equal(v["iters"][:k], ref["iters"]))
        print(json.dumps(out), flush=True)

    slots.free()
    frontier.free()
    env.close()
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
