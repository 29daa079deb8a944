#!/usr/bin/env python3
"""bench.py -- throughput of the successor-expansion hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C4]

A "step" is one pass of the hot path (one mplx_expand_lists_device launch) over
one synthetic frontier batch that is already resident in HBM.  The default
workload is BASELINE.json configs[3], the one the metric is quoted on:
    C4 = 3D VoxelMapUtil 512^3, Control::ACC, |U| = 729, 64k-node frontier.

N > 1 (launched by torch.distributed.run, one rank per GPU): THE frontier of the
workload is block-partitioned by node over the ranks (shard.partition; rank r
expands nodes [r*N/G, (r+1)*N/G) against its own replica of the map) -- no
data-path collective, so `value` = the workload's pairs / the
slowest rank's time per step and scaling is "strong" (BASELINE configs[3]:
"64k-node frontier, 1 -> 8 x MI355X shard").  The same run also reports
  weak       every rank a full-size frontier of its own (per-GPU work fixed)
  allgather  the optional exchange for a consumer that needs the whole successor
             set on every GPU: lists packed on the device, all-gathered by RCCL
             over the engine's own buffers (shard.all_gather_packed)
MPLX_BENCH_FORCE_DIST=1 runs that code path with a world of one.

Rank 0 prints ONE JSON line (see the task contract) including
  roofline     : algorithmic bytes per launch / HIP-event kernel time vs 8 TB/s
  cpu_baseline : the reference's own headers compiled into oracle/_ref ("reference";
                 the CPU oracle, "port", when that build is absent) timed on this
                 box's host cores on a bounded sample of the same workload
and, at N = 1, the rest of BASELINE's metric (--no-extras skips them):
  e2e            the same batch through host pointers (H2D + kernel + D2H)
  wavefront      the frontier with realistic locality (open list of a search)
  other_configs  C2, C3, C5 (C5's potential map made on the device)
  plan           plan() wall time, reference CPU planner vs the drop-in adapter
                 vs the engine's host search, on C1 and on a 3D |U| = 729 problem
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
FRONTIER_SEED = {"C2": 2002, "C3": 2003, "C4": 2004, "C5": 2005}
FRONTIER_KW = {"C2": (2.0, 0.5), "C3": (3.0, 0.5, 2.0, 1.0), "C4": (2.0, 0.5), "C5": (2.0, 0.5)}
KERNEL_NAME = {"grid": "expand_grid_kernel", "tile": "expand_tile_kernel", "dense": "expand_kernel", "none": "expand_kernel"}


def kernel_name(env, route):
    """The kernel the last lists launch ran: the GRID route has two (mplx_last_grid_kernel)."""
    if route == "grid" and env.last_grid_kernel() == "lex":
        return "expand_lex_kernel"
    if route == "grid" and env.last_grid_kernel() == "pair":
        return "expand_pair_kernel"
    return KERNEL_NAME[route]


def algorithmic_bytes(wl, n_nodes, n_emit, n_samples):
    """B_alg = N*S_wp + |U|*udim*8 + samples*(1 + r/8) + N_emit*(S_wp + 8 + 4)."""
    s_wp = (4 * wl.dim + 2) * 8
    r = 1 if wl.region is not None else 0
    return n_nodes * s_wp + wl.U.size * 8 + n_samples * (1 + r / 8.0) + n_emit * (s_wp + 8 + 4)


def measured_traffic(workload, kernel):
    """HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, each its own
    run; profiles/README.md): bench.py cannot collect counters itself, so it reports the figure measured for
    this workload + kernel, or None when no such profile is committed."""
    rec = committed_counters(workload, kernel)  # round 5: every configuration in one file
    if rec and rec.get("traffic_bytes"):
        return rec["traffic_bytes"]
    for rnd in ("r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", "%s_%s_traffic.json" % (rnd, workload.lower()))
        try:
            rec = json.load(open(path))
        except (OSError, ValueError):
            continue
        if rec.get("kernel", "").split("<")[0] in kernel:
            return rec["traffic_bytes"]
    return None


def committed_counters(workload, kernel):
    """Per-launch counters of this workload's kernel from the committed round-5 PMC passes (profiles/r05_counters.json,
    written by profiles/run_round5_counters.sh on the GPU box: SQ_INSTS_VALU, FETCH_SIZE and WRITE_SIZE, each its own
    rocprofv3 pass): bench.py cannot collect counters itself.  None when nothing is committed for this kernel."""
    for fname in ("r06_counters.json", "r05_counters.json"):  # (the kernels of round 6 are round 5's: either file describes them)
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", fname))).get(workload)
        except (OSError, ValueError):
            continue
        if rec and rec.get("kernel", "").split("<")[0] in kernel:
            rec = dict(rec)
            rec["file"] = "profiles/" + fname
            return rec
    return None


N_SIMD, SHADER_CLOCK_HZ = 1024, 2.4e9  # 256 CUs x 4 SIMDs; /opt/skills/guides/MI355X_MICROARCH.md


def bound_of(rec, kernel_ms, frac_hbm):
    """Which ceiling a kernel of the path runs against.  VALU issue: a wave64 VALU instruction occupies its SIMD for 4
    cycles, so `SQ_INSTS_VALU x 4 / (1024 SIMDs x 2.4 GHz)` is the time the launch's vector instructions alone need with
    every SIMD busy every cycle; issue_frac = that / the measured kernel time.  >= 0.4: the instruction stream is the
    bound ("valu-issue"); HBM fraction >= 0.4: "hbm"; neither: "latency" (dependent chains / occupancy: more of either
    resource would not be used)."""
    out = {"bound": "hbm" if frac_hbm >= 0.4 else "latency", "issue_frac": None}
    if rec and rec.get("valu_insts_per_launch"):
        issue_ms = rec["valu_insts_per_launch"] * 4.0 / (N_SIMD * SHADER_CLOCK_HZ) * 1e3
        out["issue_frac"] = issue_ms / kernel_ms
        out["valu_issue_ms"] = issue_ms
        out["valu_insts_per_launch"] = rec["valu_insts_per_launch"]
        if frac_hbm < 0.4 and out["issue_frac"] >= 0.4:
            out["bound"] = "valu-issue"
    if rec and rec.get("traffic_bytes"):
        out["traffic"] = rec["traffic_bytes"]
    out["counters_source"] = ("%s (rocprofv3 --pmc passes of this workload + kernel on an MI355X, committed; "
                              "not collected in this run)" % rec.get("file", "profiles/")) if rec else None
    return out


def count_work(env, frontier, n_nodes):
    """One untimed launch of the dense kernel with diagnostic outputs: emitted / finite successors and map samples
    of this frontier (the terms of the algorithmic bytes)."""
    vs = env.alloc_slots(n_nodes, want_state=False, want_iters=True)
    env.expand_resident(frontier, vs)
    env.synchronize()
    v = vs.download()
    vs.free()
    st = v["status"]
    return (int(np.count_nonzero((st == 1) | (st == 2))), int(np.count_nonzero(st == 1)),
            int(v["iters"].sum(dtype=np.int64)))


def time_lists(env, frontier, lists, steps, warmup):
    """Kernel time per launch (ms), HIP events on the engine's own stream."""
    for _ in range(warmup):
        env.expand_lists_resident(frontier, lists)
    env.synchronize()
    env.timer_begin()
    for _ in range(steps):
        env.expand_lists_resident(frontier, lists)
    return env.timer_end() / steps


def spin_up(env, frontier, lists, max_groups=30):
    """Clocks up from idle before a leg is timed (--spinup-ms for the headline): groups of 10 launches until
    three consecutive groups agree within 1.5 %, at most 10 * max_groups launches.  A leg that starts after seconds of host
    work otherwise measures the power-state transient: round 3's wavefront leg read 0.75 - 0.97 ms for a 0.61 ms kernel."""
    n = 0
    for _ in range(max_groups // 3):
        g = [time_lists(env, frontier, lists, 10, 0) for _ in range(3)]
        n += 30
        if max(g) <= 1.015 * min(g):
            break
    return n


def run_config(m, wl, steps, warmup, device=0, route=None):
    """Resident-lists kernel rate of one workload (used for the configurations next to the headline).  route: force
    "tile" / "dense" (the general kernels behind the factorised one) instead of the automatic choice."""
    env = m.EnvMap(wl.dim, device)
    wl.apply(env)
    if route:
        env.set_lists_route(route)
    fr = env.upload_frontier(wl.nodes)
    lists = env.alloc_lists(wl.n_nodes, want_state=True, want_iters=False)
    n_emit, n_fin, n_samples = count_work(env, fr, wl.n_nodes)
    spin_up(env, fr, lists, 9 if route else 30)
    ms = time_lists(env, fr, lists, steps, warmup)
    route = env.last_lists_route()
    kname = kernel_name(env, route)
    lists.free()
    fr.free()
    env.close()
    b_alg = algorithmic_bytes(wl, wl.n_nodes, n_emit, n_samples)
    return {"kernel_ms": ms, "pairs_per_s": wl.n_pairs / (ms * 1e-3), "algorithmic_bytes_per_launch": b_alg,
            "achieved_GBps": b_alg / (ms * 1e-3) / 1e9, "frac": b_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "pairs": wl.n_pairs, "emitted": n_emit, "finite": n_fin, "map_samples": n_samples, "kernel": kname}


def cpu_baseline(wl, rep_seconds=1.5, rep_seconds_1thread=0.4, reps=7):
    """The reference's CPU path (oracle/_ref, "reference"; the restatement, "port", when that build is absent) timed on
    this box's host cores on a bounded sample of the workload, to SURVEY 8(d)'s protocol: one warm-up repetition, then
    `reps` >= 5 timed repetitions (7), the MEDIAN reported (min / max beside it); every repetition is one call that runs at
    least ~0.3 s -- a frontier too small for that is walked several times inside the repetition (the nodes tiled), so
    thread start-up is not what is measured.  All host threads, and one thread."""
    from oracle import oracle as O

    oenv = O.Env(wl.dim, wl.control, wl.U, wl.grid, wl.map_dim, wl.origin, wl.res,
                 potential=wl.potential, region=wl.region, **wl.params)
    cores = os.cpu_count() or 1
    nU = wl.U.shape[0]
    use_ref = os.path.exists(O.REF_SO)

    def protocol(threads, want_seconds):
        # size of one repetition: whole passes over the frontier when it is small, a prefix when it is large.  Sized from a
        # probe, then corrected from the repetition it produced (a probe of a few nodes per thread measures thread start-up,
        # not throughput: C2's first sizing gave repetitions of 0.15 s) -- these sizing runs are the warm-up
        def sized(want):
            if want >= wl.n_nodes:
                loops = int(min(4096, max(1, round(want / wl.n_nodes))))
                return (np.ascontiguousarray(np.tile(wl.nodes, (1, loops))) if loops > 1 else wl.nodes), loops, wl.n_nodes
            return np.ascontiguousarray(wl.nodes[:, :want]), 1, want

        probe = min(wl.n_nodes, max(threads * 8, 64))
        sec, _ = O.time_expand(oenv, wl.nodes[:, :probe], threads=threads, reps=1, ref=use_ref)
        want = max(probe, int(probe / max(sec, 1e-9) * want_seconds))
        cap = 64 * 1024 * 1024 // max(1, wl.nodes.shape[0])  # (nodes of a repetition: 512 MiB of frontier at most)
        for attempt in range(3):
            for _ in range(4):
                nodes, loops, n_once = sized(min(want, cap))
                sec, _ = O.time_expand(oenv, nodes, threads=threads, reps=1, ref=use_ref)  # sizing + warm-up repetition
                if 0.8 * want_seconds <= sec <= 1.6 * want_seconds or (sec < want_seconds and nodes.shape[1] >= cap):
                    break
                want = max(threads, int(nodes.shape[1] * want_seconds / max(sec, 1e-9) * 1.05))  # (either direction)
            n = nodes.shape[1]
            secs, st = [], None
            for _ in range(reps):
                sec, st = O.time_expand(oenv, nodes, threads=threads, reps=1, ref=use_ref)
                secs.append(sec)
            med_s = sorted(secs)[len(secs) // 2]
            if med_s >= 0.6 * want_seconds or n >= cap:  # (else the sizing run was an outlier: once more with its own figure)
                break
            want = int(n * want_seconds / max(med_s, 1e-9) * 1.05)
        rates = sorted(n * nU / t for t in secs)
        med = rates[len(rates) // 2]
        return {"value": med, "min": rates[0], "max": rates[-1], "spread": (rates[-1] - rates[0]) / med, "reps": reps,
                "rep_seconds": sorted(secs)[len(secs) // 2], "nodes_per_rep": n, "frontier_passes_per_rep": loops,
                "frontier_nodes_per_pass": n_once, "map_samples_per_rep": st["samples"]}

    allc = protocol(cores, rep_seconds)
    one = protocol(1, rep_seconds_1thread)
    kind = "reference" if use_ref else "port"
    sample = "%s: %d x first %d of %d nodes x %d controls per rep; 1 warm-up + %d reps of %.2f s, median; %d threads" % (
        wl.name, allc["frontier_passes_per_rep"], allc["frontier_nodes_per_pass"], wl.n_nodes, nU, reps, allc["rep_seconds"], cores)
    return {"value": allc["value"], "unit": "pairs/s", "cores": cores, "kind": kind, "sample": sample,
            "value_1thread": one["value"], "protocol": {"all_cores": allc, "one_thread": one},
            "what": ("oracle/_ref: the reference's own headers built against the stand-in Eigen / Boost of oracle/stub_include"
                     if use_ref else "oracle/: the restatement of the reference's algorithm")}, oenv, allc["nodes_per_rep"]


# ------------------------------------------------------------------ extras (N = 1)
def post_identity(m, env, wl, lists, reps=5):
    """mplx_post_lists_device on resident lists: heuristic + goal flags alone, and with the node identity (canon[] = first
    successor of the batch with the same lattice hash) -- the difference is the identity pass (identity_kernel.hip)."""
    import ctypes as C
    from motion_primitive_library_amd import _abi
    L = _abi.lib()
    ns = lists.n_slots
    heur, flags, canon = (m.env.DeviceArray(env, ns * 8), m.env.DeviceArray(env, ns), m.env.DeviceArray(env, ns * 4))
    goal = np.ascontiguousarray(wl.nodes[:, 0], dtype=np.float64)
    g = _abi.GoalSpec()
    g.goal, g.control, g.w, g.v_max = goal.ctypes.data, wl.control, 10.0, 2.0
    g.tol_pos, g.tol_vel, g.tol_acc, g.tol_yaw = 0.5, -1.0, -1.0, -1.0
    s = lists.c_struct()

    def run(want_canon):
        o = _abi.Post()
        o.heur, o.flags, o.canon = heur.ptr, flags.ptr, canon.ptr if want_canon else None
        for _ in range(2):
            _abi.check(env._ctx, L.mplx_post_lists_device(env._ctx, C.byref(s), wl.n_nodes, C.byref(g), C.byref(o)))
        env.synchronize()
        loops = []
        for _ in range(3):  # median of three timed loops (one loop of a closing run of round 4 came out 30 % high)
            env.timer_begin()
            for _ in range(reps):
                _abi.check(env._ctx, L.mplx_post_lists_device(env._ctx, C.byref(s), wl.n_nodes, C.byref(g), C.byref(o)))
            loops.append(env.timer_end() / reps)
        return sorted(loops)[1]

    base = run(False)
    full = run(True)
    form = env.last_identity_form()
    c = canon.download(np.int32, (ns,))
    cnt = lists.count.download(np.int32, (wl.n_nodes,))
    valid = (np.arange(lists.stride)[None, :] < cnt[:, None]).ravel()
    firsts = int(np.count_nonzero(c[valid] == np.nonzero(valid)[0]))
    for b in (heur, flags, canon):
        b.free()
    n = int(cnt.sum(dtype=np.int64))
    return {"successors": n, "first_occurrences": firsts, "heuristic_flags_ms": base, "with_identity_ms": full,
            "identity_ms": full - base, "identity_form": form, "G_successors_per_s": n / max(full - base, 1e-9) / 1e6}


def extra_e2e(m, wl, reps=10, want_state=True):
    """The same batch through host pointers (mplx_expand_lists): H2D of the frontier, kernel, D2H of the used list
    prefixes into the caller's pageable arrays -- SURVEY 8(d) "end-to-end incl. H2D + D2H".  want_state=False: the
    edges-only output (action + cost + hash, 20 B per successor), what the engine's own search asks for."""
    env = m.EnvMap(wl.dim, 0)
    wl.apply(env)
    out = env.expand_lists(wl.nodes, want_state=want_state, want_iters=False)  # warm-up: code object, device scratch, page faults
    emitted = int(out["count"].sum(dtype=np.int64))
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = env.expand_lists(wl.nodes, want_state=want_state, want_iters=False, out=out)  # buffers reused, as a C caller would
        times.append(time.perf_counter() - t0)
    env.close()
    F = 4 * wl.dim + 2
    bytes_out = wl.n_nodes * 4 + emitted * (4 + 8 + 8 + (F * 8 if want_state else 0))
    best = sorted(times)[len(times) // 2]  # >= 10 repetitions, median
    return {"e2e_ms_per_step": best * 1e3, "e2e_pairs_per_s": wl.n_pairs / best, "bytes_copied_back": bytes_out,
            "copy_back_GBps": bytes_out / best / 1e9, "calls_ms": [round(t * 1e3, 2) for t in times],
            "e2e_ms_min": min(times) * 1e3, "e2e_ms_max": max(times) * 1e3, "reps": len(times), "statistic": "median",
            "what": "mplx_expand_lists on host pointers: frontier H2D, kernel, only the used list prefixes D2H into pageable arrays"}


def corridor_fixture():
    z = np.load(os.path.join(ROOT, "tests", "golden", "corridor_map.npz"))
    n = int(z["n_cells"])
    occ = np.unpackbits(z["occupied_bits"])[:n].astype(bool)
    return dict(cells=np.where(occ, 100, 0).astype(np.int8), dim=[int(x) for x in z["dim"]],
                origin=[float(x) for x in z["origin"]], res=float(z["resolution"]), start=z["start"], goal=z["goal"])


def engine_plan(m, dim, origin, md, cells, res, U, start, goal, v_max, a_max, batch, reps=3):
    pl = m.MapPlanner(dim, device=0)
    mu = m.MapUtil(dim)
    mu.setMap(origin, md, cells, res)
    pl.setMapUtil(mu)
    pl.setVmax(v_max)
    pl.setAmax(a_max)
    pl.setDt(1.0)
    pl.setU(U)
    pl.setBatch(batch)
    pl.plan(start, goal)  # warm-up
    best, split = 1e30, None
    for _ in range(reps):
        t0 = time.perf_counter()
        ok = pl.plan(start, goal)
        ms = (time.perf_counter() - t0) * 1e3
        if ms < best:
            best, split = ms, pl.timing()  # where that plan's time went (mplx_planner_timing, host_planner.hpp)
    s = pl.summary()
    pl.close()
    sp = {k: round(split[k], 3) for k in ("provider_ms", "fill_ms", "pick_ms", "relax_ms", "recover_ms")}
    sp.update({k: int(split[k]) for k in ("relaxed", "improved", "pushes", "materialised", "heur_from_device")})
    sp["what"] = ("provider = launches + completion + transfers (the device's share); relax = relaxation passes, node table, "
                  "heap and goal tests on one host thread; fill = lists moved out of the landing buffer; pick = choice of the "
                  "next launch's nodes")
    return {"wall_ms": best, "ok": bool(ok), "cost": s["cost"], "expansions": s["expansions"], "closed": s["closed"],
            "nodes": s["nodes"], "launches": s["device_launches"], "timing_split": sp}


def extra_plan(m):
    """plan() wall time (BASELINE metric, second half): the reference's own MapPlanner on the host CPU
    (oracle/_ref/libmpl_ref_planner.so, 1 thread -- the reference's search is sequential), the same planner with
    only get_succ swapped for the MI355X (include/mplx_env_map.hpp, speculative batches), and the engine's host A*
    with batched expansion.  All three must agree on the trajectory cost."""
    from oracle import oracle as O
    out = {}
    W = m.workloads
    have_ref = os.path.exists(O.REF_PLANNER_SO)
    # ---- C1: test_planner_2d on corridor.yaml
    c = corridor_fixture()
    U = W.grid_controls([-0.5, 0.0, 0.5], 2)
    start, goal = m.Waypoint(2, m.ACC, pos=c["start"]), m.Waypoint(2, m.ACC, pos=c["goal"])
    c1 = {"problem": "C1: test_planner_2d on data/corridor.yaml, ACC, |U| = 9"}
    c1["engine_host_search"] = engine_plan(m, 2, c["origin"], c["dim"], c["cells"], c["res"], U, start, goal, 1.0, 1.0, 16)
    c1["engine_host_search"]["batch"] = 16  # a 5 535-pair problem: a small speculative batch (64: ~30 % slower)
    if have_ref:
        oenv = O.Env(2, O.ACC, U, c["cells"], c["dim"], c["origin"], c["res"], v_max=1.0, a_max=1.0, dt=1.0)
        cpu = min((O.ref_plan(oenv, start.to_row(), goal.to_row(), use_gpu=False) for _ in range(3)), key=lambda r: r["wall_ms"])
        ad = min((O.ref_plan(oenv, start.to_row(), goal.to_row(), use_gpu=64) for _ in range(3)), key=lambda r: r["wall_ms"])
        c1["reference_cpu"] = {k: cpu[k] for k in ("wall_ms", "ok", "cost", "expansions")}
        c1["reference_planner_gpu_adapter"] = {"wall_ms": ad["wall_ms"], "ok": ad["ok"], "cost": ad["cost"],
                                               "expansions": ad["expansions"], "launches": ad["device_launches"]}
        c1["agree"] = bool(cpu["cost"] == ad["cost"] == c1["engine_host_search"]["cost"] and cpu["closed"] == 615)
    out["C1"] = c1
    # ---- 3D, |U| = 729: large enough for batching to matter; 120^3 (3.3 k expansions) and 160^3 (64 k expansions)
    def problem_3d(edge, with_adapter, engine_batch, reps):
        res = 0.1
        grid = W.box_map([edge] * 3, res, 0.08, 4242, side_m=(0.5, 2.5))
        flat = grid.ravel()
        U3 = W.grid_controls(np.linspace(-2.0, 2.0, 9), 3)

        def free_near(p):
            cc = np.array([int(x / res) for x in p])
            for r in range(0, 30):
                for d in np.ndindex(2 * r + 1, 2 * r + 1, 2 * r + 1):
                    q = cc + np.array(d) - r
                    if np.all(q >= 0) and np.all(q < edge) and flat[q[0] + edge * (q[1] + edge * q[2])] == 0:
                        return [(q[i] + 0.5) * res for i in range(3)]
            raise RuntimeError("no free cell")

        s3 = m.Waypoint(3, m.ACC, pos=free_near([1.0, 1.0, 1.0]))
        g3 = m.Waypoint(3, m.ACC, pos=free_near([edge * res - 1.0, edge * res - 1.2, edge * res - 1.5]))
        p3 = {"problem": "3D %d^3 voxels, ACC, |U| = 729, v_max 2" % edge}
        p3["engine_host_search"] = engine_plan(m, 3, [0.0] * 3, [edge] * 3, flat, res, U3, s3, g3, 2.0, 2.0, engine_batch, reps=reps)
        p3["engine_host_search"]["batch"] = engine_batch
        if have_ref:
            oenv = O.Env(3, O.ACC, U3, flat, [edge] * 3, [0.0] * 3, res, v_max=2.0, a_max=2.0, dt=1.0)
            cpu = O.ref_plan(oenv, s3.to_row(), g3.to_row(), use_gpu=False)
            p3["reference_cpu"] = {k: cpu[k] for k in ("wall_ms", "ok", "cost", "expansions")}
            agree = cpu["cost"] == p3["engine_host_search"]["cost"] and cpu["expansions"] == p3["engine_host_search"]["expansions"]
            if with_adapter:
                ad = O.ref_plan(oenv, s3.to_row(), g3.to_row(), use_gpu=64)
                p3["reference_planner_gpu_adapter"] = {"wall_ms": ad["wall_ms"], "ok": ad["ok"], "cost": ad["cost"],
                                                       "expansions": ad["expansions"], "launches": ad["device_launches"]}
                agree = agree and cpu["cost"] == ad["cost"]
                p3["speedup_adapter_vs_reference_cpu"] = cpu["wall_ms"] / ad["wall_ms"]
            p3["agree"] = bool(agree)
            p3["speedup_engine_vs_reference_cpu"] = cpu["wall_ms"] / p3["engine_host_search"]["wall_ms"]
        return p3

    def distance_3d(edge, batch):
        """BASELINE config 5's planner as the reference's test_distance_map_planner_2d_with_yaw.cpp:48-104 runs it, on a voxel
        map: plan (ACC, 27 controls); then a DistanceMapPlanner -- updatePotentialMap around the start, ACCxYAW with 3 yaw
        rates (81 controls), yaw_max 0.5, iterativePlan inside the tunnel around the first trajectory (map_planner.cpp:394-430)."""
        res = 0.1
        grid = W.box_map([edge] * 3, res, 0.08, 4242, side_m=(0.5, 2.5))
        flat = grid.ravel()
        vals = [-1.0, 0.0, 1.0]
        U3, U3y = W.grid_controls(vals, 3), W.grid_controls(vals, 3, yaw_rates=[-0.5, 0.0, 0.5])

        def free_near(p):
            cc = np.array([int(x / res) for x in p])
            for r in range(0, 30):
                for d in np.ndindex(2 * r + 1, 2 * r + 1, 2 * r + 1):
                    q = cc + np.array(d) - r
                    if np.all(q >= 0) and np.all(q < edge) and flat[q[0] + edge * (q[1] + edge * q[2])] == 0:
                        return [(q[i] + 0.5) * res for i in range(3)]
            raise RuntimeError("no free cell")

        ps, pg = free_near([1.0, 1.0, 1.0]), free_near([edge * res - 1.0, edge * res - 1.2, edge * res - 1.5])
        d = {"problem": "DistanceMapPlanner, 3D %d^3 voxels: plan (ACC, |U| = 27), then updatePotentialMap + ACCxYAW |U| = 81 + "
                        "iterativePlan in the tunnel around the first trajectory (test_distance_map_planner_2d_with_yaw.cpp:48-104)" % edge}

        def make(table):
            pl = m.MapPlanner(3, device=0)
            mu = m.MapUtil(3)
            mu.setMap([0.0] * 3, [edge] * 3, flat.copy(), res)
            pl.setMapUtil(mu)
            pl.setVmax(2.0)
            pl.setAmax(2.0)
            pl.setDt(1.0)
            pl.setU(table)
            pl.setBatch(batch)
            return pl

        best = None
        for _ in range(2):
            first = make(U3)
            t0 = time.perf_counter()
            ok1 = first.plan(m.Waypoint(3, m.ACC, pos=ps), m.Waypoint(3, m.ACC, pos=pg))
            t1 = time.perf_counter()
            traj = first.getTraj()
            s1 = first.summary()
            first.close()
            pl = make(U3y)
            pl.setEpsilon(1.0)
            pl.setSearchRadius([0.5] * 3)
            pl.setPotentialRadius([1.0] * 3)
            pl.setPotentialWeight(0.5)
            pl.setGradientWeight(0)
            t2 = time.perf_counter()
            pl.updatePotentialMap(ps)
            t3 = time.perf_counter()
            pl.setYawmax(0.5)
            ok2 = pl.iterativePlan(m.Waypoint(3, m.ACCxYAW, pos=ps), m.Waypoint(3, m.ACC, pos=pg), traj, 10)
            t4 = time.perf_counter()
            s2 = pl.summary()
            pl.close()
            rec = {"first_plan_ms": (t1 - t0) * 1e3, "potential_map_ms": (t3 - t2) * 1e3, "wall_ms": (t4 - t3) * 1e3,
                   "ok": bool(ok1 and ok2), "cost": s2["cost"], "expansions": s2["expansions"], "closed": s2["closed"],
                   "launches": s2["device_launches"], "first_plan_expansions": s1["expansions"], "batch": batch}
            if best is None or rec["wall_ms"] < best["wall_ms"]:
                best = rec
        d["engine_host_search"] = best
        if have_ref:
            oenv = O.Env(3, O.ACC, U3, flat, [edge] * 3, [0.0] * 3, res, v_max=2.0, a_max=2.0, dt=1.0)
            srow, grow = m.Waypoint(3, m.ACC, pos=ps).to_row(), m.Waypoint(3, m.ACC, pos=pg).to_row()
            keys = ("wall_ms", "ok", "cost", "expansions", "closed", "potential_map_ms")
            cpu = O.ref_scenario(oenv, srow, grow, "distance_yaw")
            ad = min((O.ref_scenario(oenv, srow, grow, "distance_yaw", use_gpu=batch) for _ in range(2)), key=lambda r: r[1]["wall_ms"])
            d["reference_cpu"] = dict({k: cpu[1][k] for k in keys}, first_plan_ms=cpu[0]["wall_ms"])
            d["reference_planner_gpu_adapter"] = dict({k: ad[1][k] for k in keys}, first_plan_ms=ad[0]["wall_ms"],
                                                      launches=ad[1]["device_launches"])
            close = lambda a, b: abs(a - b) <= 1e-9 * abs(b)  # the per-sample heading cost uses cos / sin (glibc vs OCML)
            d["agree"] = bool(cpu[1]["expansions"] == ad[1]["expansions"] == best["expansions"] and
                              cpu[1]["closed"] == ad[1]["closed"] == best["closed"] and
                              close(ad[1]["cost"], cpu[1]["cost"]) and close(best["cost"], cpu[1]["cost"]) and
                              ad[1]["potential_sum"] == cpu[1]["potential_sum"])
            d["speedup_engine_vs_reference_cpu"] = cpu[1]["wall_ms"] / best["wall_ms"]
            d["speedup_adapter_vs_reference_cpu"] = cpu[1]["wall_ms"] / ad[1]["wall_ms"]
            d["speedup_whole_stage_engine_vs_reference_cpu"] = ((cpu[1]["wall_ms"] + cpu[1]["potential_map_ms"]) /
                                                                (best["wall_ms"] + best["potential_map_ms"]))
        return d

    def replan_3d(edge):
        """Incremental re-planning (LPA*, PlannerBase::setLPAstar): plan, MapPlanner::getLinkedNodes, a box of cells on the
        trajectory becomes occupied + updateBlockedNodes, plan, the box is cleared + updateClearedNodes, plan
        (map_planner.cpp:125-185; the scenario of tests/test_lpastar.py on a larger voxel map) -- the reference's LPA* on
        one host core, the same with the drop-in adapter, and the engine's own LPA* (csrc/host_lpastar.hpp)."""
        res = 0.1
        flat = W.box_map([edge] * 3, res, 0.05, 78, side_m=(0.3, 0.8)).ravel().copy()
        U3 = W.grid_controls(np.linspace(-2.0, 2.0, 5), 3)

        def free_near(p):
            cc = np.array([int(x / res) for x in p])
            for r in range(0, 12):
                for d in np.ndindex(2 * r + 1, 2 * r + 1, 2 * r + 1):
                    q = cc + np.array(d) - r
                    if np.all(q >= 0) and np.all(q < edge) and flat[q[0] + edge * (q[1] + edge * q[2])] == 0:
                        return [(q[i] + 0.5) * res for i in range(3)]
            raise RuntimeError("no free cell")

        s3 = m.Waypoint(3, m.ACC, pos=free_near([0.5, 0.5, 0.5]))
        g3 = m.Waypoint(3, m.ACC, pos=free_near([edge * res - 1.0, edge * res - 1.4, edge * res - 1.8]))
        box = 3
        rp = {"problem": "3D %d^3 voxels, ACC, |U| = 125, v_max 2: LPA* with a %d^3-cell box blocked on the trajectory, then cleared" % (edge, 2 * box + 1)}
        # ---- the engine's own LPA*
        pl = m.MapPlanner(3, device=0)
        mu = m.MapUtil(3)
        mu.setMap([0.0] * 3, [edge] * 3, flat.copy(), res)
        pl.setMapUtil(mu)
        pl.setVmax(2.0)
        pl.setAmax(2.0)
        pl.setDt(1.0)
        pl.setU(U3)
        pl.setBatch(16)
        pl.setLPAstar(True)
        plans, t_ms = [], {}

        def timed(name, fn):
            t0 = time.perf_counter()
            r = fn()
            t_ms[name] = (time.perf_counter() - t0) * 1e3
            return r

        def rec(ok):
            s_ = pl.summary()
            plans.append({"ok": bool(ok), "cost": s_["cost"], "expansions": s_["expansions"], "closed": s_["closed"]})

        rec(timed("plan1_ms", lambda: pl.plan(s3, g3)))
        wps = pl.getTraj().getWaypoints()
        _, n_cells, n_entries = timed("linked_nodes_ms", lambda: pl.getLinkedNodes(want_points=False))
        c_round = lambda x: int(np.sign(x) * np.floor(abs(x) + 0.5))
        to_cell = lambda p: np.array([c_round(p[i] / res - 0.5) for i in range(3)])
        mid, sc, gc = to_cell(wps[len(wps) // 2][:3]), to_cell(s3.pos), to_cell(g3.pos)
        edit = []
        w_ = 2 * box + 1
        for q in range(w_ ** 3):
            r_, pn = q, []
            for i in range(3):
                pn.append(mid[i] + (r_ % w_) - box)
                r_ //= w_
            pn = np.array(pn)
            if np.any(pn < 0) or np.any(pn >= edge) or flat[pn[0] + edge * (pn[1] + edge * pn[2])] != 0:
                continue
            if np.all(np.abs(pn - sc) <= 2) or np.all(np.abs(pn - gc) <= 2):
                continue
            edit.append(pn)
        edit = np.array(edit, dtype=np.int32)
        timed("update_blocked_ms", lambda: pl.updateBlockedNodes(edit))
        rec(timed("plan2_ms", lambda: pl.plan(s3, g3)))
        pl.getLinkedNodes(want_points=False)
        timed("update_cleared_ms", lambda: pl.updateClearedNodes(edit))
        rec(timed("plan3_ms", lambda: pl.plan(s3, g3)))
        pl.close()
        rp["engine_lpastar"] = dict({k: round(v, 3) for k, v in t_ms.items()}, plans=plans, table_cells=n_cells, table_entries=n_entries,
                                    edited_cells=int(edit.shape[0]))
        if have_ref:
            oenv = O.Env(3, O.ACC, U3, flat, [edge] * 3, [0.0] * 3, res, v_max=2.0, a_max=2.0, dt=1.0)
            for label, gpu in (("reference_cpu", False), ("reference_planner_gpu_adapter", True)):
                rp_, tb = O.ref_lpastar(oenv, s3.to_row(), g3.to_row(), use_gpu=gpu, box_half=box)
                rp[label] = {"plan1_ms": rp_[0]["wall_ms"], "plan2_ms": rp_[1]["wall_ms"], "plan3_ms": rp_[2]["wall_ms"],
                             "linked_nodes_ms": tb["get_linked_nodes_us"] / 1e3, "update_cleared_ms": tb["update_cleared_us"] / 1e3,
                             "plans": [{"ok": p_["ok"], "cost": p_["cost"], "expansions": p_["expansions"], "closed": p_["closed"]} for p_ in rp_],
                             "table_cells": tb["cells"], "table_entries": tb["entries"], "edited_cells": tb["edited_cells"],
                             # map bytes moved to the device by (updateBlockedNodes + plan 2), (updateClearedNodes + plan 3):
                             # the adapter patches the edited cells (round 6; before: the whole map, twice)
                             "replan_upload_bytes": tb.get("replan_upload_bytes")}
            ref = rp["reference_cpu"]
            rp["agree"] = bool(all(a == b for a, b in zip(plans, ref["plans"])) and n_cells == ref["table_cells"] and n_entries == ref["table_entries"])
            tot = lambda d: d["plan2_ms"] + d["plan3_ms"] + d["linked_nodes_ms"] + d["update_cleared_ms"]
            rp["speedup_replan_engine_vs_reference_cpu"] = tot(ref) / tot(rp["engine_lpastar"])
            rp["what"] = "speedup_replan_*: (second + third plan + getLinkedNodes + updateClearedNodes) of the reference on one host core over the engine's"
        return rp

    try:
        out["replan_3D"] = replan_3d(100)
    except Exception as e:  # noqa: BLE001
        out["replan_3D"] = {"error": "%s: %s" % (type(e).__name__, e)}
    out["3D"] = problem_3d(120, True, 64, 2)
    out["distance_map_3D"] = distance_3d(120, 64)
    # the larger problem: the reference's search alone takes ~20 s of one host core here, so one run each and no adapter leg
    # (its 2.8 x is the 120^3 figure: most of the adapter's time is the reference's own StateSpace, not get_succ)
    if os.environ.get("MPLX_BENCH_SKIP_PLAN_160") != "1":
        out["3D_160"] = problem_3d(160, False, 256, 1)
    out["cpu_threads_used"] = 1
    return out


def extras(m, args, wl, out):
    """Everything BASELINE's metric names beyond the headline kernel rate; each leg is independent and a failure is
    recorded instead of losing the line."""
    def leg(name, fn):
        t0 = time.time()
        try:
            out[name] = fn()
        except Exception as e:  # noqa: BLE001 -- reported, never fatal for the headline
            out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        if isinstance(out[name], dict):
            out[name]["leg_seconds"] = round(time.time() - t0, 2)

    leg("e2e", lambda: extra_e2e(m, wl))

    def edges_only():
        """The same launch without the 14 state rows: count + action + cost + hash per successor -- what the engine's
        own host search asks for (it evaluates the states of NEW nodes itself, bit-identically, host_planner.hpp)."""
        env = m.EnvMap(wl.dim, 0)
        wl.apply(env)
        fr = env.upload_frontier(wl.nodes)
        lists = env.alloc_lists(wl.n_nodes, want_state=False, want_iters=False)
        spin_up(env, fr, lists)
        ms = time_lists(env, fr, lists, args.steps, args.warmup)
        lists.free()
        fr.free()
        env.close()
        e2e = extra_e2e(m, wl, want_state=False)
        return {"kernel_ms": ms, "value": wl.n_pairs / (ms * 1e-3), "unit": "pairs/s",
                "e2e_ms_per_step": e2e["e2e_ms_per_step"], "e2e_pairs_per_s": e2e["e2e_pairs_per_s"],
                "e2e_bytes_copied_back": e2e["bytes_copied_back"], "e2e_calls_ms": e2e["calls_ms"],
                "what": "resident lists without the Waypoint rows (action + cost + hash: 20 B per successor instead of 132); "
                        "e2e_*: the same through mplx_expand_lists on host pointers"}
    leg("edges_only", edges_only)

    def wavefront():
        """The frontier a search produces (graph_search.h:63-75) against the synthetic one, on ONE allocation of the
        lists in ONE context, alternating, after a clock spin-up: the ratio does not depend on where the lists landed
        (round 3's leg measured each frontier in its own allocation and straight after seconds of host work)."""
        nodes_w = m.workloads.wavefront_frontier(wl, wl.n_nodes, 0)
        env = m.EnvMap(wl.dim, 0)
        wl.apply(env)
        fr_w, fr_r = env.upload_frontier(nodes_w), env.upload_frontier(wl.nodes)
        lists = env.alloc_lists(wl.n_nodes, want_state=True, want_iters=False)
        n_emit, n_fin, n_samples = count_work(env, fr_w, wl.n_nodes)
        cold = time_lists(env, fr_w, lists, args.steps, args.warmup)  # what round 3 reported: no spin-up
        spin_up(env, fr_w, lists)
        rounds = [(time_lists(env, fr_r, lists, args.steps, 2), time_lists(env, fr_w, lists, args.steps, 2)) for _ in range(3)]
        ms_r, ms_w = sorted(r[0] for r in rounds)[1], sorted(r[1] for r in rounds)[1]
        kname = kernel_name(env, env.last_lists_route())
        # what the search does with the successors next: heuristic + goal flags + node identity of the
        # whole batch on the device, on both frontiers' lists (same allocation)
        post = {}
        try:
            for label, fr_x, emitted in (("random", fr_r, None), ("wavefront", fr_w, n_emit)):
                env.expand_lists_resident(fr_x, lists)
                env.synchronize()
                post[label] = post_identity(m, env, wl, lists)
        except Exception as e:  # noqa: BLE001 -- never lose the leg to the optional measurement
            post = {"error": "%s: %s" % (type(e).__name__, e)}
        # ---- the same work FUSED into the expansion launch (ABI v8: mplx_set_goal + the heur / flags rows of the lists):
        # lists-only against lists + heur + flags on the same allocation, and the identity pass on lists whose flags
        # row the launch wrote (no second pass over hash and position rows)
        fused = {}
        try:
            import ctypes as C
            from motion_primitive_library_amd import _abi
            ns = lists.n_slots
            hb, fb, cb = m.env.DeviceArray(env, ns * 8), m.env.DeviceArray(env, ns), m.env.DeviceArray(env, ns * 4)
            goal = np.ascontiguousarray(wl.nodes[:, 0], dtype=np.float64)
            env.set_goal(goal, w=10.0, v_max=2.0, tol_pos=0.5)
            for label, fr_x in (("random", fr_r), ("wavefront", fr_w)):
                lists.heur = lists.flags = None
                plain = sorted(time_lists(env, fr_x, lists, args.steps, 2) for _ in range(3))[1]
                lists.heur, lists.flags = hb, fb
                both = sorted(time_lists(env, fr_x, lists, args.steps, 2) for _ in range(3))[1]
                g = _abi.GoalSpec()
                g.goal, g.control, g.w, g.v_max = goal.ctypes.data, wl.control, 10.0, 2.0
                g.tol_pos, g.tol_vel, g.tol_acc, g.tol_yaw = 0.5, -1.0, -1.0, -1.0
                o = _abi.Post()
                o.heur, o.flags, o.canon = None, fb.ptr, cb.ptr
                st = lists.c_struct()
                st.state = None  # the identity pass alone: canon + bit 2 of the flags row the launch wrote
                loops = []
                env.expand_lists_resident(fr_x, lists)
                env.synchronize()
                for _ in range(2):
                    _abi.check(env._ctx, _abi.lib().mplx_post_lists_device(env._ctx, C.byref(st), wl.n_nodes, C.byref(g), C.byref(o)))
                env.synchronize()
                for _ in range(3):  # (bit 2 is OR-ed into the row: idempotent, so the calls can simply be repeated)
                    env.timer_begin()
                    for _ in range(5):
                        _abi.check(env._ctx, _abi.lib().mplx_post_lists_device(env._ctx, C.byref(st), wl.n_nodes, C.byref(g), C.byref(o)))
                    loops.append(env.timer_end() / 5)
                fused[label] = {"lists_only_ms": plain, "lists_heur_flags_ms": both, "ratio": both / plain,
                                "identity_only_ms": sorted(loops)[1], "identity_form": env.last_identity_form()}
            lists.heur = lists.flags = None
            env.set_goal(None)
            for b_ in (hb, fb, cb):
                b_.free()
            fused["what"] = ("lists_heur_flags_ms: the expansion launch writing heuristic (8 B) and goal flags (1 B) per successor "
                             "itself, against lists_only_ms on the same allocation; identity_only_ms: mplx_post_lists_device for "
                             "canon + the first-occurrence bit on those lists (no post_lists_kernel pass over hash and state rows). "
                             "Compare lists_heur_flags_ms + identity_only_ms with kernel_ms + post.*.with_identity_ms")
        except Exception as e:  # noqa: BLE001
            fused = {"error": "%s: %s" % (type(e).__name__, e)}
        lists.free()
        fr_w.free()
        fr_r.free()
        env.close()
        b_alg = algorithmic_bytes(wl, wl.n_nodes, n_emit, n_samples)
        return {"kernel_ms": ms_w, "post": post, "post_fused": fused, "value": wl.n_pairs / (ms_w * 1e-3), "random_frontier_same_allocation_ms": ms_r,
                "ratio_to_random": ms_w / ms_r, "kernel_ms_cold": cold, "rounds_ms": [[round(a, 4), round(b, 4)] for a, b in rounds],
                "algorithmic_bytes_per_launch": b_alg, "achieved_GBps": b_alg / (ms_w * 1e-3) / 1e9,
                "frac": b_alg / (ms_w * 1e-3) / 1e9 / HBM_PEAK_GBS, "pairs": wl.n_pairs, "emitted": n_emit, "finite": n_fin,
                "map_samples": n_samples, "kernel": kname,
                "what": "same workload, frontier = the first %d open-list nodes of an eps = 0 search from the map centre; "
                        "median of 3 alternating rounds against the random frontier on the same allocation of the lists "
                        "(kernel_ms_cold: %d launches straight after set-up, no clock spin-up)" % (wl.n_nodes, args.steps)}
    leg("wavefront", wavefront)

    def others():
        res = {}
        for name in ("C2", "C3", "C5"):
            if name == args.workload:
                continue
            stats = {}
            w = m.workloads.make(name, potential_fn=m.workloads.device_potential_fn(0, stats) if name == "C5" else None)
            r = run_config(m, w, args.steps, args.warmup)
            r.update(stats)
            r["workload"] = WORKLOAD_DESC[name]
            r.update(bound_of(committed_counters(name, r["kernel"]), r["kernel_ms"], r["frac"]))
            if not args.no_cpu_baseline:
                # per configuration the reference's CPU path beside the kernel (1 warm-up + 5 repetitions of
                # >= 0.3 s each, median; the small frontiers are walked several times per repetition)
                try:
                    cb, _, _ = cpu_baseline(w, rep_seconds=0.8, rep_seconds_1thread=0.3)
                    r["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "sample", "value_1thread", "protocol")}
                    r["speedup_vs_cpu_all_cores"] = r["pairs_per_s"] / cb["value"]
                    r["speedup_vs_cpu_1thread"] = r["pairs_per_s"] / cb["value_1thread"]
                except Exception as e:  # noqa: BLE001
                    r["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
            res[name] = r
        # a lattice twice as fine as C4's on every axis (du = u / 8: 17 values per axis, 4 913 controls): the lexicographic
        # kernel's since round 4 (before: the general routes below); 8 192 nodes keep the lists at 5.3 GB
        try:
            import copy
            w17 = copy.copy(wl)
            w17.U = m.workloads.grid_controls(np.linspace(-2.0, 2.0, 17), 3)
            w17.nodes = np.ascontiguousarray(wl.nodes[:, :8192])
            for tag, route in (("C4_17cubed", None), ("C4_17cubed_dense_route", "dense")):
                r = run_config(m, w17, max(3, args.steps // 4), 1, route=route)
                r["workload"] = "C4's map, 17^3 = 4 913 ACC controls, 8 192-node frontier" + (" through the dense route" if route else "")
                res[tag] = r
        except Exception as e:  # noqa: BLE001
            res["C4_17cubed"] = {"error": "%s: %s" % (type(e).__name__, e)}
        # the general kernels on the headline workload: what a control table outside the factorised kernel's scope
        # (more than 16 distinct values per axis, gradient_weight != 0, SNP with yaw ...) costs at this size
        for route in ("tile", "dense"):
            r = run_config(m, wl, max(3, args.steps // 4), 1, route=route)
            r["workload"] = WORKLOAD_DESC[args.workload] + " through the %s route" % route
            res[args.workload + "_" + route + "_route"] = r
        return res
    leg("other_configs", others)

    def strong_bound():
        """The curve an 8-GPU node can at best produce for THE frontier (strong scaling, no data-path collective): the
        kernel time of one rank's shard -- the first N/G nodes of the frontier, as shard.partition deals them -- on this
        one GPU, each after its own clock spin-up.  Aggregate bound = all pairs / the shard's time (ranks run
        concurrently on their own GPUs; launch gaps and the barrier come on top)."""
        from motion_primitive_library_amd import shard
        env = m.EnvMap(wl.dim, 0)
        wl.apply(env)
        res = []
        for g in (1, 2, 4, 8):
            lo, hi = shard.partition(wl.n_nodes, g, 0)
            fr = env.upload_frontier(np.ascontiguousarray(wl.nodes[:, lo:hi]))
            lists = env.alloc_lists(hi - lo, want_state=True, want_iters=False)
            spin_up(env, fr, lists)
            ms = sorted(time_lists(env, fr, lists, args.steps, 2) for _ in range(3))[1]
            res.append({"gpus": g, "nodes_per_gpu": hi - lo, "shard_kernel_ms": ms,
                        "aggregate_pairs_per_s_bound": wl.n_pairs / (ms * 1e-3),
                        "efficiency_bound": None})
            lists.free()
            fr.free()
        env.close()
        for r in res:
            r["efficiency_bound"] = res[0]["shard_kernel_ms"] / (r["gpus"] * r["shard_kernel_ms"])
        return {"curve": res, "kernel": "expand_lex_kernel", "what": "compute bound of the strong-scaling run measured on ONE GPU (rank 0's "
                "shard per world size); not a multi-GPU measurement"}
    leg("strong_scaling_compute_bound", strong_bound)
    leg("plan", lambda: extra_plan(m))



def _r(x, sig=5):
    """Numbers on the printed line: `sig` significant digits (the detail file keeps full precision)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    try:
        return float("%.*g" % (sig, float(x)))
    except (TypeError, ValueError):
        return x


def compact_line(out, detail_path):
    """The ONE JSON line of the contract, from the long form: the contract's keys, `roofline`, `cpu_baseline`, and of every
    extra leg its numbers only -- no prose, no per-call lists.  Everything else is in the detail file."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data")
    line = {k: (_r(out[k], 7) if k in ("value", "ms_per_step") else out[k]) for k in keep if k in out}
    cfg = out.get("config", {})
    line["config"] = {k: cfg[k] for k in ("workload", "frontier", "frontier_nodes", "frontier_nodes_per_gpu", "controls", "dim",
                                          "pairs_per_step", "map_cells", "device", "compute_units") if k in cfg}
    line["config"]["kernel"] = str(cfg.get("kernel", "")).split(" ")[0]
    rf = out.get("roofline", {})
    line["roofline"] = {k: _r(rf.get(k), 6) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms",
                                                      "algorithmic_bytes_per_launch", "issue_frac", "store_only_ms",
                                                      "kernel_over_store_only", "emitted", "map_samples")}
    for k in ("ms_per_step_events", "value_events", "parity_sample_ok", "speedup_vs_cpu_all_cores",
              "speedup_vs_cpu_all_cores_e2e_host_pointers", "speedup_vs_cpu_all_cores_e2e_edges_only", "map_broadcast_ms"):
        if k in out:
            line[k] = _r(out[k])
    if "rank_kernel_ms" in out and out.get("n_gpus", 1) > 1:
        line["rank_kernel_ms"] = out["rank_kernel_ms"]

    def cpu(cb):
        if not isinstance(cb, dict) or "value" not in cb:
            return cb
        pr = (cb.get("protocol") or {}).get("all_cores", {})
        return {"value": _r(cb["value"]), "unit": cb.get("unit", "pairs/s"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                "sample": cb.get("sample"), "value_1thread": _r(cb.get("value_1thread")), "reps": pr.get("reps"),
                "statistic": "median", "spread": _r(pr.get("spread"), 3), "rep_seconds": _r(pr.get("rep_seconds"), 3)}

    if "cpu_baseline" in out:
        line["cpu_baseline"] = cpu(out["cpu_baseline"])
    if isinstance(out.get("weak"), dict):
        line["weak"] = {k: _r(out["weak"].get(k)) for k in ("value", "ms_per_step", "frontier_nodes_per_gpu")}
    if isinstance(out.get("allgather"), dict):
        ag = out["allgather"]
        line["allgather"] = {k: ({kk: _r(vv) for kk, vv in v.items() if kk != "what"} if isinstance(v, dict) else v)
                             for k, v in ag.items() if k != "what"}

    def pick(d, keys):
        return {k: _r(d.get(k)) for k in keys if isinstance(d, dict) and k in d} if isinstance(d, dict) and "error" not in d else d

    if "e2e" in out:
        line["e2e"] = pick(out["e2e"], ("e2e_ms_per_step", "e2e_ms_min", "e2e_ms_max", "reps", "statistic", "e2e_pairs_per_s",
                                        "bytes_copied_back", "copy_back_GBps"))
        if isinstance(line["e2e"], dict) and "cpu_baseline" in out and "e2e_pairs_per_s" in line["e2e"]:
            # the drop-in's own output (get_succ returns Waypoints) over the PCIe link: the bound of the >= 100 x target
            # through host pointers is the link, not the kernel
            line["e2e"]["bound"] = "pcie"
    if "edges_only" in out:
        line["edges_only"] = pick(out["edges_only"], ("kernel_ms", "value", "e2e_ms_per_step", "e2e_pairs_per_s", "e2e_bytes_copied_back"))
    if "wavefront" in out:
        w = out["wavefront"]
        line["wavefront"] = pick(w, ("kernel_ms", "value", "ratio_to_random", "frac", "emitted", "map_samples"))
        if isinstance(w, dict) and isinstance(w.get("post_fused"), dict):
            pf = w["post_fused"]
            line["wavefront"]["post_fused"] = {k: pick(pf.get(k), ("lists_only_ms", "lists_heur_flags_ms", "ratio", "identity_only_ms"))
                                               for k in ("random", "wavefront") if k in pf}
        if isinstance(w, dict) and isinstance(w.get("post"), dict):
            line["wavefront"]["post"] = {k: pick(w["post"].get(k), ("heuristic_flags_ms", "identity_ms", "successors", "first_occurrences"))
                                         for k in ("random", "wavefront") if k in w["post"]}
    if isinstance(out.get("other_configs"), dict):
        oc = {}
        for name, r in out["other_configs"].items():
            if name == "leg_seconds":
                continue
            if not isinstance(r, dict) or "kernel_ms" not in r:
                oc[name] = r
                continue
            if "_route" in name:  # the general kernels on the headline workload: two numbers each
                oc[name] = pick(r, ("kernel", "kernel_ms", "pairs_per_s"))
                continue
            e = pick(r, ("kernel", "kernel_ms", "pairs_per_s", "frac", "achieved_GBps", "algorithmic_bytes_per_launch", "bound",
                         "issue_frac", "traffic", "speedup_vs_cpu_all_cores", "speedup_vs_cpu_1thread"))
            if "cpu_baseline" in r:
                e["cpu_baseline"] = cpu(r["cpu_baseline"])
                if isinstance(e["cpu_baseline"], dict):
                    e["cpu_baseline"].pop("sample", None)  # (its text is in the detail file; the protocol is the headline's)
            oc[name] = e
        line["other_configs"] = oc
    if isinstance(out.get("strong_scaling_compute_bound"), dict) and "curve" in out["strong_scaling_compute_bound"]:
        line["strong_scaling_compute_bound"] = [{"gpus": c["gpus"], "shard_kernel_ms": _r(c["shard_kernel_ms"]),
                                                 "efficiency_bound": _r(c["efficiency_bound"], 3)}
                                                for c in out["strong_scaling_compute_bound"]["curve"]]
    elif "strong_scaling_compute_bound" in out:
        line["strong_scaling_compute_bound"] = out["strong_scaling_compute_bound"]
    if isinstance(out.get("plan"), dict):
        pl = {}
        for name, r in out["plan"].items():
            if not isinstance(r, dict):
                continue
            if "error" in r:
                pl[name] = r
                continue
            e = {}
            for who in ("engine_host_search", "reference_cpu", "reference_planner_gpu_adapter", "engine_lpastar"):
                d = r.get(who)
                if not isinstance(d, dict):
                    continue
                e[who] = pick(d, ("wall_ms", "cost", "expansions", "launches", "potential_map_ms", "first_plan_ms", "plan1_ms",
                                  "plan2_ms", "plan3_ms", "linked_nodes_ms", "update_blocked_ms", "update_cleared_ms"))
                if isinstance(d.get("replan_upload_bytes"), list):
                    e[who]["replan_upload_bytes"] = d["replan_upload_bytes"]
                ts = d.get("timing_split")
                if isinstance(ts, dict):
                    e[who]["timing_split"] = {k: _r(ts.get(k), 4) for k in ("provider_ms", "fill_ms", "pick_ms", "relax_ms", "recover_ms")}
            for k, v in r.items():
                if k == "agree" or k.startswith("speedup_"):
                    e[k] = _r(v, 4)
            pl[name] = e
        line["plan"] = pl
    line["detail"] = detail_path
    # a hard ceiling, whatever a leg grows into: optional subtrees go (they stay in the detail file) until the line fits
    for path in (("wavefront", "post"), ("wavefront", "post_fused"), ("allgather",), ("plan", "replan_3D"), ("edges_only",)):
        if len(json.dumps(line, separators=(",", ":"))) <= 7800:
            break
        d = line
        for k in path[:-1]:
            d = d.get(k, {}) if isinstance(d, dict) else {}
        if isinstance(d, dict):
            d.pop(path[-1], None)
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C4", choices=sorted(WORKLOAD_DESC))
    ap.add_argument("--scale", type=float, default=1.0, help="map edge scale (debug only; invalidates the metric)")
    ap.add_argument("--nodes", type=int, default=None, help="frontier size override (debug only)")
    ap.add_argument("--spinup-ms", type=float, default=100.0,
                    help="untimed launches of the same step for at least this long, until their time is steady, BEFORE "
                         "the W warm-up steps: the workload is generated on the host for seconds while the GPU idles at "
                         "its lowest clocks, and the first ~25 ms of launches after that run up to 25 %% slower "
                         "(profiles/micro/c4_ramp.py: 0.60, 0.54, 0.50, 0.48 ms ... steady 0.476); 0 = off")
    ap.add_argument("--placement-trials", type=int, default=1,
                    help="DIAGNOSTIC (default 1 = off since round 4).  The C4 kernel's time is the WRITE bandwidth of the "
                         "memory its 5.4 GB of state rows landed in: per allocation 0.49 ... 0.565 ms, and pure stores into "
                         "the same allocation show the same modes (5.9 against 5.15 TB/s, profiles/r04_store_layouts_vs_kernel.txt); "
                         "no layout, skew, stride or allocation flag moves it.  `value` is what a process that allocates its "
                         "output once measures.  > 1: up to this many allocations (held while probing), a 20-launch probe of "
                         "each, the fastest kept -- listed in config.output_placement; never at N > 1")
    ap.add_argument("--prealloc-cycles", type=int, default=0,
                    help="DIAGNOSTIC (default 0).  Allocate, touch and free the output lists this many times before the "
                         "allocation that is timed: a fresh process's first allocation and one that reuses freed memory "
                         "land in different placement modes about half the time (profiles/README.md) -- "
                         "profiles/run_round6_modes.sh uses it to get a trace of BOTH modes from the same binary")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip e2e / wavefront / other_configs / plan (N = 1 only)")
    ap.add_argument("--frontier", default="random", choices=["random", "wavefront"],
                    help="random: SURVEY 8(d)'s uniformly scattered frontier (the headline); wavefront: the open list of "
                         "an eps = 0 search from the map centre (realistic locality; also reported as an extra)")
    args = ap.parse_args()

    verbose = os.environ.get("MPLX_BENCH_VERBOSE") == "1"

    def note(msg):
        if verbose:
            print("[bench %.1fs] %s" % (time.time() - t_start, msg), file=sys.stderr, flush=True)

    t_start = time.time()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # Plain `python bench.py --gpus N`: become the launcher -- one rank per GPU under torch.distributed.run on
        # this node, rendezvous on 127.0.0.1 (the container hostname may not resolve), same arguments.  exec, so the
        # exit status and rank 0's one JSON line are the launcher's.  (Under an external launcher WORLD_SIZE is set
        # and this is skipped.)
        import socket
        s_ = socket.socket()
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
        s_.close()
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it here
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        note("self-launch: %s" % " ".join(cmd))
        if os.environ.get("MPLX_BENCH_BACKEND", "nccl") == "nccl":
            import torch
            if torch.cuda.device_count() < args.gpus:
                sys.exit("bench.py: --gpus %d needs %d GPUs, %d visible (self-launch prepared: %s)"
                         % (args.gpus, args.gpus, torch.cuda.device_count(), " ".join(cmd[1:8])))
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    if world != args.gpus:
        args.gpus = world

    # stdout carries the ONE JSON line and nothing else: libraries under the legs print through C stdio (RCCL's version
    # banner, the reference build's "[PlannerBase] use Lifelong Planning A*" of planner_base.h:173), so descriptor 1 is
    # pointed at stderr for the duration of the run and the line goes to the saved descriptor at the end.
    sys.stdout.flush()
    line_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    import motion_primitive_library_amd as m
    from motion_primitive_library_amd import shard

    if not torch.cuda.is_available():
        sys.exit("bench.py: no GPU visible; the engine has no CPU fallback")
    if world > torch.cuda.device_count() and os.environ.get("MPLX_BENCH_BACKEND", "nccl") == "nccl":
        sys.exit("bench.py: --gpus %d needs %d GPUs, %d visible (one rank per GPU; MPLX_BENCH_BACKEND=gloo rehearses the "
                 "flow with several ranks on one GPU)" % (world, world, torch.cuda.device_count()))
    # MPLX_BENCH_BACKEND=gloo (diagnostic): several ranks may then share one GPU -- RCCL refuses two ranks on one device,
    # gloo moves CUDA tensors through the host -- so the whole N > 1 flow can be rehearsed on a one-GPU box
    backend = os.environ.get("MPLX_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    # MPLX_BENCH_FORCE_DIST=1 runs the N > 1 code path (RCCL init, barrier, all-reduce of the timing, the list
    # gather) with whatever world size the launcher gave, including 1 -- the only way to exercise it on a one-GPU box
    force_dist = os.environ.get("MPLX_BENCH_FORCE_DIST") == "1"
    distributed = world > 1 or force_dist
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # ---- the workload; rank r owns block r of ITS frontier
    # N > 1: rank 0 makes the map and THE frontier once; the other ranks get the frontier from it and the map by
    # mplx_comm_broadcast_map (RCCL over xGMI, device to device) instead of G host generations + G uploads.
    # MPLX_BENCH_LOCAL_MAPS=1: every rank generates its own copy (the workload is deterministic).
    stats = {}
    share = distributed and world > 1 and os.environ.get("MPLX_BENCH_LOCAL_MAPS") != "1"
    pot_fn = m.workloads.device_potential_fn(local_rank, stats) if args.workload == "C5" else None
    wl = m.workloads.make(args.workload, scale=args.scale, n_nodes=args.nodes, potential_fn=pot_fn,
                          shell=share and rank != 0)
    if args.frontier == "wavefront" and not (share and rank != 0):
        wl.nodes = m.workloads.wavefront_frontier(wl, wl.n_nodes, local_rank)

    def from_rank0(arr):
        """A numpy array of rank 0 on every rank (same shape and dtype everywhere)."""
        t = torch.from_numpy(np.ascontiguousarray(arr)).cuda()
        dist.broadcast(t, 0)
        return t.cpu().numpy()

    if share:
        wl.nodes = from_rank0(wl.nodes)
    N, nU = wl.n_nodes, wl.U.shape[0]
    lo, hi = shard.partition(N, world, rank)
    n_loc = hi - lo
    my_nodes = np.ascontiguousarray(wl.nodes[:, lo:hi])

    env = m.EnvMap(wl.dim, local_rank)
    wl.apply(env)  # (ranks other than 0 of a shared run: the geometry with an empty map)
    map_source = "generated and uploaded by every rank" if distributed and world > 1 else "generated and uploaded once"
    if share:
        t_map = time.perf_counter()
        def all_ranks_ok(ok):
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return int(t.item()) == 1

        via_abi = False
        if backend == "nccl":
            # The C ABI's own communicator (csrc/comm_api.cpp).  Every step is agreed on by all ranks before the next one, and
            # any failure sends ALL ranks to the torch.distributed copy below: the map broadcast is set-up, not the metric.
            comm_err = None
            try:
                raw = bytearray(m.EnvMap.comm_unique_id()) if rank == 0 else bytearray(128)
            except Exception as e:  # noqa: BLE001
                raw, comm_err = bytearray(128), "comm_unique_id: %s" % e
            uid = torch.frombuffer(raw, dtype=torch.uint8).cuda()
            dist.broadcast(uid, 0)
            ok = comm_err is None and any(uid.cpu().numpy().tobytes())
            if all_ranks_ok(ok):
                try:
                    env.comm_init(bytes(uid.cpu().numpy().tobytes()), rank, world)
                except Exception as e:  # noqa: BLE001
                    ok, comm_err = False, "comm_init: %s" % e
                if all_ranks_ok(ok):
                    try:
                        env.comm_broadcast_map(0)  # map (+ potential map / search region when rank 0 has them)
                    except Exception as e:  # noqa: BLE001
                        ok, comm_err = False, "comm_broadcast_map: %s" % e
                    via_abi = all_ranks_ok(ok)
            if via_abi:
                map_source = "rank 0's map replicated by mplx_comm_broadcast_map (ncclBroadcast over xGMI, device to device)"
            elif comm_err:
                stats["comm_error"] = comm_err
        if not via_abi:  # rehearsal backends (several ranks on one GPU: RCCL refuses that), or the fallback of the above
            has_pot = int(from_rank0(np.array([wl.potential is not None], np.int64))[0])
            wl.grid = from_rank0(wl.grid)
            env.setMap(wl.origin, wl.map_dim, wl.grid, wl.res)
            if has_pot:
                wl.potential = from_rank0(wl.potential if rank == 0 else np.zeros_like(wl.grid))
                env.set_potential_map(wl.potential)
            map_source = "rank 0's map broadcast through torch.distributed (%s%s)" % (backend, " rehearsal" if backend != "nccl" else
                                                                                       "; the C ABI's communicator failed on a rank")
        stats["map_broadcast_ms"] = (time.perf_counter() - t_map) * 1e3
    alloc = shard.torch_alloc("cuda:%d" % local_rank) if distributed else None  # RCCL moves these very buffers
    frontier = env.upload_frontier(my_nodes)
    for _ in range(max(0, args.prealloc_cycles)):  # (diagnostic, see --prealloc-cycles)
        tmp = env.alloc_lists(n_loc, want_state=True, want_iters=False, alloc=alloc)
        env.expand_lists_resident(frontier, tmp)
        env.synchronize()
        tmp.free()
    slots = env.alloc_lists(n_loc, want_state=True, want_iters=False, alloc=alloc)
    launch = lambda: env.expand_lists_resident(frontier, slots)
    dev_name, cus = env.device_info()

    # ---- one untimed verification launch: counts for the algorithmic bytes of THIS rank's launch
    n_emit, n_finite, n_samples = count_work(env, frontier, n_loc)
    b_alg = algorithmic_bytes(wl, n_loc, n_emit, n_samples)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps):
        """W warm-ups were done by the caller; K steps bracketed by barrier + synchronize on both sides.
        Returns (wall seconds incl. both barriers, max over ranks; this rank's HIP-event ms per step; the slowest
        rank's HIP-event ms per step).  At N > 1 a rank's K steps are a fraction of a millisecond each, so the two
        barriers are a visible share of the wall figure: `value` uses the slowest rank's HIP-event time (the K steps
        themselves, on the engine's stream) and the wall figure is reported beside it."""
        barrier()
        t0 = time.perf_counter()
        env.timer_begin()
        for _ in range(steps):
            fn()
        kernel_ms_total = env.timer_end()  # HIP events on the engine's own stream (synchronises it)
        barrier()
        el = time.perf_counter() - t0
        slowest = kernel_ms_total
        per_rank = [kernel_ms_total / steps]
        if distributed:
            t = torch.tensor([el, kernel_ms_total], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el, slowest = float(t[0].item()), float(t[1].item())
            # every rank's own HIP-event time per step: the placement mode of each rank's output allocation is visible
            # (the step is the slowest rank's)
            mine = torch.tensor([kernel_ms_total / steps], dtype=torch.float64, device="cuda")
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            per_rank = [float(x.item()) for x in allr]
        timed.per_rank_ms = per_rank
        return el, kernel_ms_total / steps, slowest / steps

    note("workload ready, %d of %d nodes on this rank" % (n_loc, N))
    spin = {"ms": 0.0, "groups": 0}
    if args.spinup_ms > 0:
        # Clocks up from idle (see --spinup-ms): groups of 10 launches until five consecutive groups agree within 1.5 %
        # (and at least --spinup-ms have passed), at most 3 s.  Not counted as warm-up or timed steps.
        t_spin = time.perf_counter()
        recent = []
        while True:
            env.timer_begin()
            for _ in range(10):
                launch()
            recent = (recent + [env.timer_end()])[-5:]
            spin["groups"] += 1
            el_ms = (time.perf_counter() - t_spin) * 1e3
            steady = len(recent) == 5 and max(recent) <= 1.015 * min(recent)
            if (steady and el_ms >= args.spinup_ms) or el_ms >= 3000.0:
                break
        spin["ms"] = (time.perf_counter() - t_spin) * 1e3
    placement = {"probe_ms": [], "chosen": 0}
    if args.placement_trials > 1 and world == 1:  # see --placement-trials (diagnostic; a rank of an N > 1 run never probes)
        def probe():
            env.timer_begin()
            for _ in range(20):
                launch()
            return env.timer_end() / 20

        tried = [slots]
        placement["probe_ms"].append(probe())
        # (the modes are a ladder -- 0.49, 0.51, 0.54, 0.565 ms on C4, profiles/r03_c4_placement_counters.txt -- so the search
        # only stops early for a probe that is a clear 10 % below another one)
        while len(tried) < args.placement_trials and not min(placement["probe_ms"]) <= 0.905 * max(placement["probe_ms"]):
            try:
                slots = env.alloc_lists(n_loc, want_state=True, want_iters=False, alloc=alloc)  # (the earlier ones stay allocated)
            except Exception as e:  # noqa: BLE001 -- out of memory for another trial: keep what there is
                note("placement trial %d not allocated: %s" % (len(tried), e))
                break
            tried.append(slots)
            for _ in range(5):
                launch()
            placement["probe_ms"].append(probe())
        placement["chosen"] = int(np.argmin(placement["probe_ms"]))
        slots = tried[placement["chosen"]]
        for i, l in enumerate(tried):
            if i != placement["chosen"]:
                l.free()
    for _ in range(args.warmup):
        launch()
    elapsed, kernel_ms, kernel_ms_slowest = timed(launch, args.steps)
    rank_kernel_ms = list(timed.per_rank_ms)
    note("timed region done: %.3f ms per step" % (elapsed / args.steps * 1e3))
    route = env.last_lists_route()

    # ---- parity of the TIMED output: the lists the timed launches wrote, against the oracle
    timed_head = slots.download_nodes(0, min(n_loc, 512)) if rank == 0 else None
    # ---- the list stores of that launch on their own, into the SAME allocation (mplx_debug_store_model overwrites the
    # entries: the head has just been taken): what the launch costs when its arithmetic is free
    store_only_ms = None
    if rank == 0 and not distributed:  # (at N > 1 the gather leg still reads these lists)
        try:
            for _ in range(5):
                env.debug_store_model(slots, n_loc)
            env.synchronize()
            env.timer_begin()
            for _ in range(args.steps):
                env.debug_store_model(slots, n_loc)
            store_only_ms = env.timer_end() / args.steps
        except Exception as e:  # noqa: BLE001 -- diagnostic only
            note("store model not timed: %s" % e)

    weak = gather = None
    if distributed:
        # ---- weak scaling: every rank a full-size frontier of its own
        def weak_frontier(r):
            return wl.nodes if r == 0 else m.workloads.random_frontier(
                wl.grid, wl.origin, wl.res, N, FRONTIER_SEED[args.workload] + 100 * r, wl.control, *FRONTIER_KW[args.workload])

        if share and backend == "nccl":  # only rank 0 holds the map on the host: it draws every rank's frontier
            nodes_w = wl.nodes
            for r in range(1, world):
                got = from_rank0(weak_frontier(r) if rank == 0 else wl.nodes)
                if r == rank:
                    nodes_w = got
        else:
            nodes_w = weak_frontier(rank)
        fw = env.upload_frontier(nodes_w)
        sw = env.alloc_lists(N, want_state=True, want_iters=False)
        launch_w = lambda: env.expand_lists_resident(fw, sw)
        for _ in range(args.warmup):
            launch_w()
        el_w, k_w, k_w_slowest = timed(launch_w, args.steps)
        weak = {"value": float(N) * nU * world / (k_w_slowest * 1e-3), "unit": "pairs/s", "ms_per_step": k_w_slowest,
                "ms_per_step_wall": el_w / args.steps * 1e3, "value_wall": float(N) * nU * world * args.steps / el_w,
                "kernel_ms_rank0": k_w, "frontier_nodes_per_gpu": N, "scaling": "weak"}
        sw.free()
        fw.free()
        note("weak leg done")
        # ---- the optional all-gather of the successor lists (packed on the device, moved by RCCL)
        try:
            packed = env.alloc_packed(n_loc, capacity=(n_loc + 1) * nU, want_state=True, alloc=alloc)

            def pack_and_gather(want_state):
                env.pack_lists(slots, packed)
                env.synchronize()  # engine stream -> torch's
                cnt, offs, rows = shard.packed_views(packed, n_loc)
                if not want_state:
                    rows = {k: v for k, v in rows.items() if k != "state"}
                res = shard.all_gather_packed(cnt, offs, rows, n_loc)
                torch.cuda.synchronize()
                return res

            gather = {}
            for label, ws in (("edges", False), ("full", True)):
                pack_and_gather(ws)  # warm-up
                barrier()
                t0 = time.perf_counter()
                reps = 3
                for _ in range(reps):
                    res = pack_and_gather(ws)
                barrier()
                dt_ = (time.perf_counter() - t0) / reps
                entries = int(res[4][-1])
                bpe = 20 + (8 * (4 * wl.dim + 2) if ws else 0)
                gather[label] = {"ms": dt_ * 1e3, "entries": entries, "bytes_per_entry": bpe,
                                 "received_GBps_per_rank": entries * bpe * (world - 1) / max(world, 1) / dt_ / 1e9}
            # ---- the consumer of the gather: the on-device merge of ALL ranks' successors (mplx_post_packed_device on
            # the gathered lists: heuristic, goal flags, first occurrence of every lattice state across the whole frontier)
            cnt_all, offs_all, rows_all = res[0], res[1], res[2]
            ps = shard.packed_struct(cnt_all, offs_all, rows_all)
            goal_row = np.ascontiguousarray(wl.nodes[:, 0])
            n_all = int(res[3][-1])
            bufs = env.post_packed(ps, n_all, goal_row, alloc=alloc, download=False)  # warm-up
            env.synchronize()
            for b_ in bufs.values():
                if b_ is not None:
                    b_.free()
            barrier()
            t0 = time.perf_counter()
            bufs = env.post_packed(ps, n_all, goal_row, alloc=alloc, download=False)
            env.synchronize()
            merge_ms = (time.perf_counter() - t0) * 1e3
            unique = int((bufs["flags"].view(torch.uint8, int(res[4][-1])) & 4).ne(0).sum().item())
            # cross-check of the merge on the device: the number of distinct lattice hashes among the gathered successors
            distinct = int(torch.unique(rows_all["hash"][: int(res[4][-1])]).numel())
            gather["merge"] = {"ms": merge_ms, "entries": int(res[4][-1]), "first_occurrences": unique,
                               "distinct_hashes_torch_unique": distinct, "ok": bool(unique == distinct),
                               "what": "mplx_post_packed_device on the gathered lists of all ranks, on every rank: "
                                       "heuristic + goal flags + node identity (first occurrence of each lattice state)"}
            for b_ in bufs.values():
                if b_ is not None:
                    b_.free()
            gather["what"] = ("pack on the device + torch.distributed all_gather_into_tensor (RCCL) of the packed rows of every "
                              "rank, padded to the largest rank, compacted on the device; edges = action + cost + hash, "
                              "full = + the 4D+2 state rows")
            packed.free()
        except Exception as e:  # noqa: BLE001 -- never lose the headline to the optional exchange
            gather = {"error": "%s: %s" % (type(e).__name__, e)}
        note("gather leg done: %s" % (gather,))
        # global work of the strong-scaling step, for the record
        tot = torch.tensor([n_emit, n_samples], dtype=torch.float64, device="cuda")
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        n_emit_all, n_samples_all = int(tot[0].item()), int(tot[1].item())
    else:
        n_emit_all, n_samples_all = n_emit, n_samples

    if rank == 0:
        ms_wall = elapsed / args.steps * 1e3
        # N = 1: the contract's wall clock around the K steps.  N > 1: the slowest rank's K steps on its stream (HIP
        # events, max over ranks) -- see timed(); the barrier-inclusive wall figure is kept beside it.
        # `value` is the wall clock for every N (round-3 advice: the HIP-event figure of N > 1 was not comparable with the
        # N = 1 definition or with earlier rounds); the slowest rank's HIP-event time is reported beside it.
        ms_per_step = ms_wall
        out_kernel = kernel_name(env, route)
        achieved = b_alg / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "node-expansions/s (frontier x |U| pair evaluations per second)",
            "value": float(N) * nU / (ms_per_step * 1e-3),
            "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "ms_per_step_wall": ms_wall, "value_wall": float(N) * nU * args.steps / elapsed,
            "ms_per_step_events": kernel_ms_slowest, "value_events": float(N) * nU / (kernel_ms_slowest * 1e-3),
            "rank_kernel_ms": [round(x, 5) for x in rank_kernel_ms],
            "timing": ("K steps between barrier + synchronize on both sides, host clock, max over ranks = `value` for every "
                       "N; `value_events` / `ms_per_step_events`: the slowest rank's HIP-event time of its K steps on the "
                       "engine's stream (no barrier, no launch gaps)"),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": WORKLOAD_DESC[args.workload] + ("" if args.scale == 1.0 and args.nodes is None else
                                                           " [DEBUG scale=%g nodes=%s]" % (args.scale, args.nodes)),
                "frontier": args.frontier, "frontier_nodes": N, "frontier_nodes_per_gpu": n_loc,
                "controls": int(nU), "dim": wl.dim, "pairs_per_step": N * nU, "map_cells": int(wl.grid.size),
                "output": "per-node successor lists: count + action + cost + hash + full Waypoint (4D+2 doubles), emitted successors only",
                "sharding": "the frontier block-partitioned by node over the ranks (strong scaling), map replicated per "
                            "rank, no data-path collective; the optional list all-gather is timed separately",
                "map": map_source,
                "device": dev_name, "compute_units": cus,
                "clock_spinup_ms": round(spin["ms"], 1),  # untimed launches before the W warm-up steps (see --spinup-ms)
                "output_placement": {"probe_ms": [round(x, 4) for x in placement["probe_ms"]], "chosen": placement["chosen"],
                                     "what": "allocations of the output lists probed before the warm-up, fastest kept "
                                             "(see --placement-trials)"},
                "kernel": "expand_lex_kernel (per-axis factorised tables in LDS, lexicographic control table)" if out_kernel == "expand_lex_kernel" else {"grid": "expand_grid_kernel (per-axis factorised tables in LDS)", "tile": "expand_tile_kernel",
                           "dense": "expand_kernel + compact_lists_kernel", "none": "expand_kernel"}[route],
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": measured_traffic(args.workload, out_kernel) if world == 1 and args.frontier == "random" else None,
                "traffic_source": "committed rocprofv3 --pmc passes of this workload + kernel (profiles/), not collected in this run",
                # the launch's time is the write bandwidth of the pages its 2.7 GB of list entries landed in, and that has
                # two modes per allocation (profiles/README.md, rounds 2 - 4): this run's, and the committed figures of both
                "placement_modes": ({"this_run_ms": kernel_ms, "this_run_frac": achieved / HBM_PEAK_GBS,
                                     "fast_mode": {"ms": 0.480, "frac": 0.75}, "slow_mode": {"ms": 0.555, "frac": 0.65},
                                     "source": "profiles/r04_placement_per_box.txt (6 boxes, first allocations: 4 slow, 2 fast); round 6, this "
                                               "binary, one process each under rocprofv3: profiles/r06_bench_c4_fast_under_rocprof.json + "
                                               "r06_c4_kernel_stats_fast.csv (0.4824 ms; trace 484.5 us), r06_bench_c4_slow_under_rocprof.json + "
                                               "r06_c4_kernel_stats_slow.csv (0.5567 ms; trace 557.3 us)"}
                                    if args.workload == "C4" and world == 1 and args.scale == 1.0 and args.nodes is None else None),
                "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": b_alg,
                # (the committed instruction count is that of the whole frontier in one launch: N = 1 only)
                "issue_frac": (bound_of(committed_counters(args.workload, out_kernel), kernel_ms, achieved / HBM_PEAK_GBS).get("issue_frac")
                               if world == 1 and args.scale == 1.0 and args.nodes is None and args.frontier == "random" else None),
                # what a process that allocates its output lists ONCE and never probes measures: the first probe (after
                # the clock spin-up, before any other allocation existed)
                "ms_per_step_first_allocation": placement["probe_ms"][0] if placement["probe_ms"] else kernel_ms,
                "frac_first_allocation": (b_alg / ((placement["probe_ms"][0] if placement["probe_ms"] else kernel_ms) * 1e-3)
                                          / 1e9 / HBM_PEAK_GBS),
                # the same lists written by mplx_debug_store_model alone (every row, count[k] entries per node, same order and
                # store policy): the launch cannot be shorter than this in this allocation
                "store_only_ms": store_only_ms,
                "kernel_over_store_only": (kernel_ms / store_only_ms) if store_only_ms else None,
                "launch": "rank 0's launch: %d nodes" % n_loc,
                "emitted": n_emit, "finite": n_finite, "map_samples": n_samples,
                "emitted_all_ranks": n_emit_all, "map_samples_all_ranks": n_samples_all,
            },
        }
        out.update({k: v for k, v in stats.items()})
        if weak is not None:
            out["weak"] = weak
        if gather is not None:
            out["allgather"] = gather
        if not args.no_cpu_baseline and world == 1:
            cb, oenv, n_chk = cpu_baseline(wl)
            out["cpu_baseline"] = cb
            out["speedup_vs_cpu_all_cores"] = out["value"] / cb["value"]
        # the lists the TIMED launches wrote (first <= 512 nodes of rank 0) against the oracle: same successors, same
        # order, bit-identical hash and state, cost exact (1e-6 for yaw controls: trig in the per-sample cost)
        try:
            from oracle import compare as OC
            from oracle import oracle as O
            oenv = O.Env(wl.dim, wl.control, wl.U, wl.grid, wl.map_dim, wl.origin, wl.res,
                         potential=wl.potential, region=wl.region, **wl.params)
            n_chk = min(n_loc, 512)
            use_ref = os.path.exists(O.REF_SO)
            ref = O.expand(oenv, my_nodes[:, :n_chk], threads=os.cpu_count() or 1, ref=use_ref)
            problems = OC.lists_mismatches(timed_head, ref, n_chk, nU, cost_rtol=1e-6 if wl.control & 0x10 else 0.0)
            out["parity_sample_ok"] = not problems
            out["parity_sample"] = {"what": "lists written by the timed %s launches, first %d nodes, vs %s" % (
                out_kernel, n_chk, "oracle/_ref (reference headers)" if use_ref else "the oracle"), "problems": problems}
        except Exception as e:  # noqa: BLE001
            out["parity_sample_ok"] = False
            out["parity_sample"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not distributed and not args.no_extras and args.scale == 1.0 and args.nodes is None:
            slots.free()
            frontier.free()
            env.close()
            env = None
            extras(m, args, wl, out)
            if "cpu_baseline" in out and isinstance(out.get("e2e"), dict) and "e2e_pairs_per_s" in out["e2e"]:
                # BASELINE's ">= 100 x the CPU" holds for lists that stay in HBM (`value`) and for an on-device consumer;
                # through host pointers the PCIe link is the bound and the ratio is this one
                out["speedup_vs_cpu_all_cores_e2e_host_pointers"] = out["e2e"]["e2e_pairs_per_s"] / out["cpu_baseline"]["value"]
                if isinstance(out.get("edges_only"), dict) and "e2e_pairs_per_s" in out["edges_only"]:
                    out["speedup_vs_cpu_all_cores_e2e_edges_only"] = out["edges_only"]["e2e_pairs_per_s"] / out["cpu_baseline"]["value"]
                if isinstance(out.get("wavefront"), dict) and "ratio_to_random" in out["wavefront"]:
                    out["wavefront_ratio_to_random"] = out["wavefront"]["ratio_to_random"]
                    pst = out["wavefront"].get("post") or {}
                    if isinstance(pst.get("random"), dict) and isinstance(pst.get("wavefront"), dict):
                        out["post_identity_ms"] = {"random": pst["random"]["identity_ms"], "wavefront": pst["wavefront"]["identity_ms"],
                                                   "form": pst["random"]["identity_form"]}
                out["speedup_note"] = ("speedup_vs_cpu_all_cores: HBM-resident lists (the metric's configuration); "
                                       "..._e2e_host_pointers: the same batch through mplx_expand_lists on pageable host "
                                       "arrays, bound by the PCIe link (2.8 GB of list entries per step)")

    if env is not None:
        slots.free()
        frontier.free()
        env.close()
    # Every rank flushes what C stdio still buffers (to stderr, see the top of main), then rank 0 prints the line.
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    if distributed:
        dist.barrier()
    if rank == 0:
        # The long form (every leg with its prose, per-call times, the plans' timing splits) goes to a file; the ONE line
        # on stdout is numbers only and stays under 8 KB, so that a record which keeps the tail of stdout keeps all of it.
        detail_path = os.environ.get("MPLX_BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json"))
        try:
            with open(detail_path, "w") as f:
                json.dump(out, f, indent=1)
        except OSError as e:
            detail_path = "not written: %s" % e
        line = compact_line(out, detail_path)
        os.write(line_fd, (json.dumps(line, separators=(",", ":")) + "\n").encode())
    os.close(line_fd)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
