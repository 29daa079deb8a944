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


def test_bench_two_ranks_share_the_gpu_over_gloo(tmp_path):
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
    env = dict(os.environ, MPLX_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--workload", "C4", "--scale", "0.25", "--nodes", "6001"]
    proc = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    line = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["parity_sample_ok"] is True
    assert d["config"]["frontier_nodes"] == 6001 and d["config"]["frontier_nodes_per_gpu"] == 3001  # rank 0's block
    assert d["weak"]["frontier_nodes_per_gpu"] == 6001
    g = d["allgather"]
    assert "error" not in g, g
    assert g["edges"]["entries"] == g["full"]["entries"] == d["roofline"]["emitted_all_ranks"] > 0
