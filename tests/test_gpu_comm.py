"""GPU: packed lists (mplx_pack_lists_device) and the multi-GPU exchange on the one GPU a test box has
(world size 1: RCCL initialises, the collectives run, the data must come out unchanged) -- both the C-ABI route
(mplx_comm_*: RCCL loaded by libmplx.so itself) and the torch.distributed route over the engine's own buffers
(shard.all_gather_packed on TorchArray-backed lists, backend nccl = RCCL).  The N > 1 logic of the torch route is
covered on the CPU by tests/test_shard_gloo.py (gloo, world size 2)."""
import os
import socket

import numpy as np
import pytest
import torch  # noqa: F401 -- first: PyTorch-ROCm brings its own librccl.so, and a process must use ONE RCCL (the
#               library's mplx_comm_* then binds the copy that is already loaded)

from helpers import engine_env

pytestmark = pytest.mark.gpu


def _expand(engine, env, wl, alloc=None, want_state=True):
    fr = env.upload_frontier(wl.nodes)
    lists = env.alloc_lists(wl.n_nodes, want_state=want_state, want_iters=False, alloc=alloc)
    env.expand_lists_resident(fr, lists)
    env.synchronize()
    return fr, lists


def _assert_packed_equal(got, want, state=True):
    assert got["total"] == want["total"] and np.array_equal(got["offs"], want["offs"])
    assert np.array_equal(got["count"], want["count"])
    assert np.array_equal(got["action"], want["action"]) and np.array_equal(got["hash"], want["hash"])
    assert np.array_equal(got["cost"].view(np.int64), want["cost"].view(np.int64))
    if state:
        assert np.array_equal(got["state"].view(np.int64), want["state"].view(np.int64))


@pytest.mark.parametrize("name,n_nodes", [("C2", 3000), ("C4", 700), ("C5", 900)])
def test_pack_lists_on_the_device(engine, name, n_nodes):
    wl = engine.workloads.make(name, scale=0.25, n_nodes=n_nodes)
    env = engine_env(engine, wl)
    fr, lists = _expand(engine, env, wl)
    host = lists.download()
    want = engine.pack_host_lists(host, wl.n_nodes)
    packed = env.alloc_packed(wl.n_nodes)
    total = env.pack_lists(lists, packed, want_total=True)
    assert total == want["total"] and total > 0
    _assert_packed_equal(packed.download(), want)
    # an exactly sized capacity works, one entry less is refused
    tight = env.alloc_packed(wl.n_nodes, capacity=total)
    env.pack_lists(lists, tight)
    env.synchronize()
    _assert_packed_equal(tight.download(), want)
    small = env.alloc_packed(wl.n_nodes, capacity=total - 1)
    with pytest.raises(engine._abi.MplxError):
        env.pack_lists(lists, small)
    # rows the caller does not ask for are skipped
    lean = env.alloc_packed(wl.n_nodes, want_state=False)
    env.pack_lists(lists, lean)
    env.synchronize()
    _assert_packed_equal(lean.download(), want, state=False)
    for b in (packed, tight, small, lean, lists, fr):
        b.free()
    env.close()


def test_c_abi_comm_world_of_one(engine):
    """mplx_comm_*: RCCL is loaded by the library, a communicator of one rank is made on the context's GPU, the map
    broadcast and the all-gather of the packed lists run and reproduce the local data."""
    wl = engine.workloads.make("C4", scale=0.25, n_nodes=600)
    env = engine_env(engine, wl)
    uid = engine.EnvMap.comm_unique_id()
    assert len(uid) == 128
    env.comm_init(uid, 0, 1)
    with pytest.raises(engine._abi.MplxError):
        env.comm_init(uid, 0, 1)  # one communicator per context
    env.comm_broadcast_map(0)
    fr, lists = _expand(engine, env, wl)
    want = engine.pack_host_lists(lists.download(), wl.n_nodes)
    packed = env.alloc_packed(wl.n_nodes)
    env.pack_lists(lists, packed)
    gathered = env.alloc_packed(wl.n_nodes, capacity=want["total"])
    noff, eoff = env.comm_allgather_lists(packed, wl.n_nodes, gathered)
    assert noff[:2].tolist() == [0, wl.n_nodes] and eoff[:2].tolist() == [0, want["total"]]
    _assert_packed_equal(gathered.download(), want)
    # the expansion after a map broadcast still gives the same lists
    fr2, lists2 = _expand(engine, env, wl)
    _assert_packed_equal(engine.pack_host_lists(lists2.download(), wl.n_nodes), want)
    env.comm_destroy()
    env.comm_destroy()  # idempotent
    for b in (packed, gathered, lists, lists2, fr, fr2):
        b.free()
    env.close()


def test_torch_distributed_gather_of_engine_buffers_world_of_one(engine):
    """The bench's N > 1 data path on one GPU: lists and packed lists live in torch-owned HBM (TorchArray), the
    kernels write them through data_ptr, shard.all_gather_packed moves them with RCCL (backend nccl)."""
    import torch
    import torch.distributed as dist
    from motion_primitive_library_amd import shard

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        wl = engine.workloads.make("C4", scale=0.25, n_nodes=500)
        env = engine_env(engine, wl)
        alloc = shard.torch_alloc("cuda:0")
        fr, lists = _expand(engine, env, wl, alloc=alloc)
        want = engine.pack_host_lists(lists.download(), wl.n_nodes)
        packed = env.alloc_packed(wl.n_nodes, capacity=(wl.n_nodes + 1) * env.nU, alloc=alloc)
        env.pack_lists(lists, packed)
        env.synchronize()  # the engine's stream -> torch's
        cnt, offs, rows = shard.packed_views(packed, wl.n_nodes)
        cnt_all, offs_all, rows_all, noff, eoff = shard.all_gather_packed(cnt, offs, rows, wl.n_nodes)
        torch.cuda.synchronize()
        got = {"total": int(eoff[-1]), "offs": offs_all.cpu().numpy(), "count": cnt_all.cpu().numpy(),
               "action": rows_all["action"].cpu().numpy(), "hash": rows_all["hash"].cpu().numpy().view(np.uint64),
               "cost": rows_all["cost"].cpu().numpy(), "state": rows_all["state"].cpu().numpy()}
        _assert_packed_equal(got, want)
        env.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("launcher", ["self", "external"])
def test_bench_two_ranks_share_the_gpu_over_gloo(tmp_path, launcher):
    """The whole `bench.py --gpus 2` flow (one process per rank under torch.distributed.run) rehearsed on the one GPU of
    the box: MPLX_BENCH_BACKEND=gloo lets both ranks use device 0 (RCCL refuses two ranks on one device).  THE frontier
    is partitioned, both legs run, the packed lists of both ranks are gathered on both, the timed lists pass the
    parity check.  Timings of this rehearsal mean nothing; the logic is what N = 2 ... 8 on a real node runs."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    detail = str(tmp_path / "bench_detail.json")
    env = dict(os.environ, MPLX_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MPLX_BENCH_DETAIL=detail)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    tail = [os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
            "--workload", "C4", "--scale", "0.25", "--nodes", "6001"]
    launchers = {"external": [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                              "--master-addr", "127.0.0.1", "--master-port", str(port)],
                 "self": [sys.executable]}  # plain `python bench.py --gpus 2`: bench.py becomes the launcher itself
    proc = subprocess.run(launchers[launcher] + tail, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    line = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1]
    short = json.loads(line)
    # the printed line: numbers only, under 8 KB, every rank's kernel time on it; the long form is in the detail file
    assert len(line) < 8000 and short["detail"] == detail and short["n_gpus"] == 2 and short["parity_sample_ok"] is True
    assert len(short["rank_kernel_ms"]) == 2 and min(short["rank_kernel_ms"]) > 0
    assert short["weak"]["frontier_nodes_per_gpu"] == 6001 and short["allgather"]["merge"]["ok"] is True
    d = json.load(open(detail))
    assert d["value"] == pytest.approx(short["value"], rel=1e-5)
    assert "rank 0's map broadcast" in d["config"]["map"] and d["ms_per_step_wall"] >= d["ms_per_step"] > 0
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["parity_sample_ok"] is True
    assert d["config"]["frontier_nodes"] == 6001 and d["config"]["frontier_nodes_per_gpu"] == 3001  # rank 0's block
    assert d["weak"]["frontier_nodes_per_gpu"] == 6001
    g = d["allgather"]
    assert "error" not in g, g
    assert g["edges"]["entries"] == g["full"]["entries"] == d["roofline"]["emitted_all_ranks"] > 0
    # the consumer of the gather: first occurrences reported by the merge = distinct hashes counted by torch.unique
    assert g["merge"]["ok"] is True and g["merge"]["first_occurrences"] == g["merge"]["distinct_hashes_torch_unique"] > 0


def test_c_abi_allgather_refuses_a_short_gathered_side_collectively(engine):
    """The capacity check is part of the collective verdict (mplx_comm_schedule on the gathered meta): with a world of
    one it is simply the error; with more ranks every rank returns it instead of one leaving the others in the group."""
    wl = engine.workloads.make("C4", scale=0.25, n_nodes=200)
    env = engine_env(engine, wl)
    env.comm_init(engine.EnvMap.comm_unique_id(), 0, 1)
    fr, lists = _expand(engine, env, wl)
    want = engine.pack_host_lists(lists.download(), wl.n_nodes)
    packed = env.alloc_packed(wl.n_nodes)
    env.pack_lists(lists, packed)
    short = env.alloc_packed(wl.n_nodes, capacity=want["total"] - 1)
    with pytest.raises(engine._abi.MplxError) as e:
        env.comm_allgather_lists(packed, wl.n_nodes, short)
    assert e.value.code == engine._abi.ERR_ARG and "capacity" in str(e.value)
    lean_local = env.alloc_packed(wl.n_nodes, want_state=False)
    env.pack_lists(lists, lean_local)
    full = env.alloc_packed(wl.n_nodes, capacity=want["total"])
    with pytest.raises(engine._abi.MplxError):  # a gathered row the local side lacks
        env.comm_allgather_lists(lean_local, wl.n_nodes, full)
    noff, eoff = env.comm_allgather_lists(packed, wl.n_nodes, full)  # the communicator is still usable
    assert eoff[1] == want["total"]
    _assert_packed_equal(full.download(), want)
    env.comm_destroy()
    for b in (packed, short, lean_local, full, lists, fr):
        b.free()
    env.close()


def _two_gpu_rank(rank, world, id_q, out_q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import torch  # noqa: F401 -- one RCCL per process: torch's
    import motion_primitive_library_amd as m
    from motion_primitive_library_amd.shard import partition

    wl = m.workloads.make("C4", scale=0.25, n_nodes=1001)  # not divisible by the world size
    lo, hi = partition(wl.n_nodes, world, rank)
    env = m.EnvMap(wl.dim, rank)
    wl.apply(env)
    if rank == 0:
        uid = m.EnvMap.comm_unique_id()
        for _ in range(world - 1):
            id_q.put(uid)
    else:
        uid = id_q.get(timeout=120)
    env.comm_init(uid, rank, world)
    env.comm_broadcast_map(0)
    fr = env.upload_frontier(np.ascontiguousarray(wl.nodes[:, lo:hi]))
    lists = env.alloc_lists(hi - lo, want_state=True, want_iters=False)
    env.expand_lists_resident(fr, lists)
    packed = env.alloc_packed(hi - lo)
    env.pack_lists(lists, packed)
    gathered = env.alloc_packed(wl.n_nodes, capacity=wl.n_nodes * env.nU)
    noff, eoff = env.comm_allgather_lists(packed, hi - lo, gathered)
    got = gathered.download(wl.n_nodes)
    env.comm_destroy()
    env.close()
    out_q.put((rank, noff[:world + 1].tolist(), eoff[:world + 1].tolist(), got))


def test_c_abi_all_pairs_exchange_on_two_gpus(engine):
    """The G > 1 branch of mplx_comm_allgather_lists (all-pairs ncclSend / ncclRecv in one group) on real devices:
    one process per GPU, every rank must end up with the lists of the whole frontier.  Skipped on one-GPU boxes, where
    tests/test_comm_schedule.py executes the same schedule against a host-memory transport."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (%d visible); the schedule itself is executed by tests/test_comm_schedule.py" % torch.cuda.device_count())
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    id_q, out_q = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=_two_gpu_rank, args=(r, world, id_q, out_q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [out_q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    wl = engine.workloads.make("C4", scale=0.25, n_nodes=1001)
    env = engine_env(engine, wl)
    fr, lists = _expand(engine, env, wl)
    want = engine.pack_host_lists(lists.download(), wl.n_nodes)
    env.close()
    for rank, noff, eoff, got in res:
        assert noff == [0, 501, 1001] and eoff[-1] == want["total"]
        _assert_packed_equal(got, want)


# ---------------------------------------------------------------------------------------------------------------------
# The G > 1 branch of the C-ABI exchange, executed for real on a ONE-GPU box: several processes share device 0 and
# libmplx.so binds tests/fake_rccl (the NCCL C API over a shared mapping; RCCL itself refuses two ranks on one device).
FAKE_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_rccl")
FAKE_SO = os.path.join(FAKE_DIR, "libfake_rccl.so")


def _build_fake_rccl():
    import shutil
    import subprocess
    src = os.path.join(FAKE_DIR, "fake_rccl.cpp")
    if os.path.exists(FAKE_SO) and os.path.getmtime(FAKE_SO) >= os.path.getmtime(src):
        return FAKE_SO
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    subprocess.run([hipcc, "-O2", "-fPIC", "-shared", "-o", FAKE_SO, src], check=True, cwd=FAKE_DIR)
    return FAKE_SO


def _fake_rank(rank, world, n_nodes, short_rank, id_q, out_q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MPLX_RCCL_LIB"] = FAKE_SO
    import motion_primitive_library_amd as m
    from motion_primitive_library_amd.shard import partition
    try:
        real = m.workloads.make("C4", scale=0.25, n_nodes=n_nodes)
        wl = real if rank == 0 else m.workloads.make("C4", scale=0.25, n_nodes=n_nodes, shell=True)  # others: an empty map
        lo, hi = partition(n_nodes, world, rank)
        env = m.EnvMap(wl.dim, 0)  # every rank on device 0
        wl.apply(env)
        if rank == 0:
            uid = m.EnvMap.comm_unique_id()
            for _ in range(world - 1):
                id_q.put(uid)
        else:
            uid = id_q.get(timeout=120)
        env.comm_init(uid, rank, world)
        env.comm_broadcast_map(0)  # ranks > 0 receive rank 0's map
        fr = env.upload_frontier(np.ascontiguousarray(real.nodes[:, lo:hi]))
        lists = env.alloc_lists(hi - lo, want_state=True, want_iters=False)
        env.expand_lists_resident(fr, lists)
        packed = env.alloc_packed(hi - lo)
        env.pack_lists(lists, packed)
        cap = n_nodes * env.nU
        err = None
        if short_rank is not None:  # one rank's gathered side is too small: EVERY rank must get the error, none may hang
            try:
                env.comm_allgather_lists(packed, hi - lo, env.alloc_packed(n_nodes, capacity=64 if rank == short_rank else cap))
            except m._abi.MplxError as e:
                err = (e.code, str(e))
        gathered = env.alloc_packed(n_nodes, capacity=cap)
        noff, eoff = env.comm_allgather_lists(packed, hi - lo, gathered)  # and the communicator still works afterwards
        got = gathered.download(n_nodes)
        env.comm_destroy()
        env.close()
        out_q.put((rank, noff[:world + 1].tolist(), eoff[:world + 1].tolist(), got, err))
    except Exception as e:  # noqa: BLE001 -- reported to the parent instead of a silent hang
        out_q.put((rank, None, None, None, ("exception", repr(e))))


@pytest.mark.parametrize("world,n_nodes,short_rank", [(2, 131, None), (3, 181, None), (4, 40, None), (3, 100, 1)])
def test_c_abi_all_pairs_exchange_between_processes_on_one_gpu(engine, world, n_nodes, short_rank):
    """mplx_comm_init / _broadcast_map / _allgather_lists with G = 2, 3, 4 ranks: one process per rank, all on device 0,
    the transport a shared-memory stand-in with the NCCL API (MPLX_RCCL_LIB -> tests/fake_rccl).  Every rank must end up
    with the packed lists of the WHOLE frontier, identical to a single-process expansion; ranks other than 0 start with an
    empty map and get rank 0's through the broadcast; a rank with a short gathered side makes all ranks fail together."""
    import torch.multiprocessing as mp
    _build_fake_rccl()
    ctx = mp.get_context("spawn")
    id_q, out_q = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=_fake_rank, args=(r, world, n_nodes, short_rank, id_q, out_q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [out_q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    wl = engine.workloads.make("C4", scale=0.25, n_nodes=n_nodes)
    env = engine_env(engine, wl)
    fr, lists = _expand(engine, env, wl)
    want = engine.pack_host_lists(lists.download(), wl.n_nodes)
    env.close()
    from motion_primitive_library_amd.shard import partition
    for rank, noff, eoff, got, err in sorted(res, key=lambda r: r[0]):
        assert got is not None, err
        assert noff == [0] + [partition(n_nodes, world, r)[1] for r in range(world)] and eoff[-1] == want["total"]
        _assert_packed_equal(got, want)
        if short_rank is not None:
            assert err is not None and err[0] == engine._abi.ERR_ARG and "capacity" in err[1] and "rank %d" % short_rank in err[1], err
