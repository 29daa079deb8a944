"""CPU, world_size = 2 over gloo: the multi-GPU sharding logic (partition +
optional all-gather of compact successor records).  The per-rank expansion is
done by the CPU oracle here (test stand-in for the engine: no GPU in this
container); what is under test is that sharding + gather reproduces the
single-process result exactly."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_covers_frontier_exactly():
    from motion_primitive_library_amd.shard import partition
    for n in (0, 1, 7, 64, 65536, 65537):
        for world in (1, 2, 3, 8):
            spans = [partition(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        partition(10, 2, 2)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from helpers import oracle_env
    from oracle import oracle as O
    import motion_primitive_library_amd.workloads as W
    from motion_primitive_library_amd.shard import all_gather_records, compact_records, partition

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    wl = W.make("C4", scale=0.125, n_nodes=37)  # deliberately not divisible by the world size
    lo, hi = partition(wl.n_nodes, world, rank)
    shard = np.ascontiguousarray(wl.nodes[:, lo:hi])
    r = O.expand(oracle_env(wl), shard, want_state=False)
    gs, c, h = compact_records(r["status"], r["cost"], r["hash"], lo, wl.U.shape[0])
    gs, c, h = all_gather_records(gs, c, h)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, gs, c, h))


def test_two_rank_shard_and_gather_matches_single_process():
    import torch.multiprocessing as mp
    from helpers import oracle_env
    from oracle import oracle as O
    import motion_primitive_library_amd.workloads as W
    from motion_primitive_library_amd.shard import compact_records

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    wl = W.make("C4", scale=0.125, n_nodes=37)
    full = O.expand(oracle_env(wl), wl.nodes, want_state=False)
    gs, c, h = compact_records(full["status"], full["cost"], full["hash"], 0, wl.U.shape[0])
    assert gs.size > 1000
    for rank, g2, c2, h2 in got:  # every rank holds the complete, identically ordered set
        assert np.array_equal(g2, gs) and np.array_equal(h2, h.view(np.uint64))
        assert np.array_equal(c2.view(np.int64), c.view(np.int64))
