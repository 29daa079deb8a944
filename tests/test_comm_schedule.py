"""CPU: the all-pairs exchange of mplx_comm_allgather_lists (csrc/comm_api.cpp) for G > 1, executed against a
host-memory fake transport ("multi-rank logic via a host-memory fake communicator in CI").

mplx_comm_schedule (include/mplx.h) is the pure function the C entry point executes with ncclSend / ncclRecv: peer
order, row list, byte offsets, sizes.  Here every rank's ops are executed on numpy arrays with in-memory mailboxes
that pair the k-th send of a -> b with the k-th receive of b from a -- RCCL's matching rule inside one group -- and the
result on EVERY rank must be pack_host_lists of the whole frontier.  The per-rank lists come from the CPU oracle
(test stand-in for the kernel: no GPU here); what is under test is the exchange."""
import collections

import numpy as np
import pytest

from helpers import oracle_env


def _rank_rows(p, cap_extra=3):
    """A rank's packed rows as the C side sees them: one byte buffer per row id."""
    F = p["state"].shape[0]
    rows = {0: p["count"].astype(np.int32).tobytes(), 1: p["action"].astype(np.int32).tobytes(),
            2: p["cost"].astype(np.float64).tobytes(), 3: p["hash"].astype(np.uint64).tobytes()}
    for f in range(F):
        rows[4 + f] = np.ascontiguousarray(p["state"][f]).tobytes()
    return rows


def _run_fake_exchange(engine, packs, n_fields, mask=15, capacity=None):
    """Executes every rank's schedule; returns per rank {row id: bytearray} of the gathered side + the offsets."""
    A = engine._abi
    G = len(packs)
    total_e = sum(int(p["total"]) for p in packs)
    total_n = sum(len(p["count"]) for p in packs)
    meta = np.zeros((G, A.COMM_META), np.int64)
    for r, p in enumerate(packs):
        meta[r, :5] = (len(p["count"]), int(p["total"]), mask, capacity if capacity is not None else total_e + r, 0)
    scheds = [engine.shard.comm_schedule(G, r, meta, n_fields) for r in range(G)]
    for ops, noff, eoff in scheds:  # every rank computes the same offsets
        assert noff.tolist() == scheds[0][1].tolist() and eoff.tolist() == scheds[0][2].tolist()
        assert noff[-1] == total_n and eoff[-1] == total_e
    src = [_rank_rows(p) for p in packs]
    dst = []
    for r in range(G):
        d = {0: bytearray(4 * total_n)}
        for row in range(1, 4 + n_fields):
            d[row] = bytearray((4 if row == 1 else 8) * total_e)
        dst.append(d)
    box = collections.defaultdict(collections.deque)  # (from, to) -> FIFO of (row, payload)
    for r, (ops, _, _) in enumerate(scheds):
        seen_net = False
        for o in ops:
            assert o["bytes"] > 0 and o["bytes"] % o["elem"] == 0
            if o["kind"] == A.COMM_COPY:
                assert not seen_net and o["peer"] == r  # local copies first
                dst[r][o["row"]][o["dst_off"]:o["dst_off"] + o["bytes"]] = src[r][o["row"]][o["src_off"]:o["src_off"] + o["bytes"]]
            elif o["kind"] == A.COMM_SEND:
                seen_net = True
                assert o["peer"] != r
                payload = src[r][o["row"]][o["src_off"]:o["src_off"] + o["bytes"]]
                assert len(payload) == o["bytes"]  # never reads past the local row
                box[(r, o["peer"])].append((o["row"], payload))
            else:
                seen_net = True
    for r, (ops, _, _) in enumerate(scheds):
        for o in ops:
            if o["kind"] != A.COMM_RECV:
                continue
            assert box[(o["peer"], r)], "rank %d waits for a message rank %d never sends" % (r, o["peer"])
            row, payload = box[(o["peer"], r)].popleft()
            assert row == o["row"] and len(payload) == o["bytes"], "send / receive of a pair do not match"
            assert o["dst_off"] + o["bytes"] <= len(dst[r][row])
            dst[r][row][o["dst_off"]:o["dst_off"] + o["bytes"]] = payload
    assert all(not q for q in box.values()), "messages nobody receives"
    return dst, scheds


def _peer_pairs_per_step(engine, G):
    """In step d every rank sends to a different peer and receives from a different one (all links busy)."""
    A = engine._abi
    meta = np.zeros((G, A.COMM_META), np.int64)
    meta[:, 0], meta[:, 1], meta[:, 2], meta[:, 3] = 2, 5, 1, 5 * G
    firsts = []
    for r in range(G):
        ops, _, _ = engine.shard.comm_schedule(G, r, meta, 0)
        sends = [o["peer"] for o in ops if o["kind"] == A.COMM_SEND and o["row"] == A.ROW_COUNT]
        recvs = [o["peer"] for o in ops if o["kind"] == A.COMM_RECV and o["row"] == A.ROW_COUNT]
        assert sends == [(r + d) % G for d in range(1, G)] and recvs == [(r - d) % G for d in range(1, G)]
        firsts.append(sends)
    for d in range(G - 1):
        assert sorted(f[d] for f in firsts) == list(range(G))


@pytest.mark.parametrize("world,cuts", [(2, [0, 20, 37]), (3, [0, 5, 5, 37]), (8, [0, 1, 9, 9, 14, 22, 30, 36, 37]),
                                         (8, [0, 0, 0, 37, 37, 37, 37, 37, 37])])
def test_all_pairs_schedule_reproduces_the_whole_frontier_on_every_rank(engine, world, cuts):
    from oracle import oracle as O
    wl = engine.workloads.make("C4", scale=0.125, n_nodes=37)
    nU = wl.U.shape[0]
    F = 4 * wl.dim + 2

    def packed(nodes):
        r = O.expand(oracle_env(wl), nodes, want_state=True)
        lists = engine.lists_from_dense(r, nodes.shape[1], nU)
        lists["stride"] = nU
        return engine.pack_host_lists(lists, nodes.shape[1])

    want = packed(wl.nodes)
    assert want["total"] > 1000
    packs = []
    for r in range(world):
        lo, hi = cuts[r], cuts[r + 1]
        if hi > lo:
            packs.append(packed(np.ascontiguousarray(wl.nodes[:, lo:hi])))
        else:  # a rank with no nodes at all (a frontier smaller than the world)
            packs.append({"count": np.zeros(0, np.int32), "offs": np.zeros(1, np.int64), "total": 0, "action": np.zeros(0, np.int32),
                          "cost": np.zeros(0), "hash": np.zeros(0, np.uint64), "state": np.zeros((F, 0))})
    dst, scheds = _run_fake_exchange(engine, packs, F)
    for r in range(world):
        assert scheds[r][1].tolist() == cuts
        assert np.array_equal(np.frombuffer(dst[r][0], np.int32), want["count"])
        assert np.array_equal(np.frombuffer(dst[r][1], np.int32), want["action"])
        assert np.array_equal(np.frombuffer(dst[r][2], np.int64), want["cost"].view(np.int64))
        assert np.array_equal(np.frombuffer(dst[r][3], np.uint64), want["hash"])
        for f in range(F):
            assert np.array_equal(np.frombuffer(dst[r][4 + f], np.int64), np.ascontiguousarray(want["state"][f]).view(np.int64))


def test_edges_only_gather_and_nodes_without_successors(engine):
    """Row mask without the state; a rank whose nodes emitted nothing sends its counts and no entry row."""
    A = engine._abi
    rng = np.random.default_rng(5)
    packs = []
    for counts in ([3, 0, 2], [0, 0], [4]):
        c = np.array(counts, np.int32)
        t = int(c.sum())
        packs.append({"count": c, "total": t, "action": rng.integers(0, 9, t).astype(np.int32), "cost": rng.random(t),
                      "hash": rng.integers(0, 2**63, t).astype(np.uint64), "state": np.zeros((0, t))})
    dst, scheds = _run_fake_exchange(engine, packs, 10, mask=A.ROWBIT_ACTION | A.ROWBIT_COST | A.ROWBIT_HASH)
    for r in range(3):
        ops = scheds[r][0]
        assert all(o["row"] < A.ROW_STATE0 for o in ops)
        assert not any(o["kind"] == A.COMM_RECV and o["peer"] == 1 and o["row"] != A.ROW_COUNT for o in ops)
        assert np.frombuffer(dst[r][0], np.int32).tolist() == [3, 0, 2, 0, 0, 4]
        assert np.array_equal(np.frombuffer(dst[r][1], np.int32), np.concatenate([p["action"] for p in packs]))
        assert np.array_equal(np.frombuffer(dst[r][3], np.uint64), np.concatenate([p["hash"] for p in packs]))


@pytest.mark.parametrize("world", [2, 3, 5, 8])
def test_every_step_uses_a_different_peer_pair(engine, world):
    _peer_pairs_per_step(engine, world)


def test_the_verdict_is_collective(engine):
    """Any inconsistency makes EVERY rank fail with the same code (no rank enters the group alone)."""
    A = engine._abi
    good = np.zeros((3, A.COMM_META), np.int64)
    good[:, 0], good[:, 1], good[:, 2], good[:, 3] = [4, 0, 2], [10, 0, 7], 15, 17
    for r in range(3):
        engine.shard.comm_schedule(3, r, good, 14)

    def codes(meta):
        out = []
        for r in range(3):
            with pytest.raises(A.MplxError) as e:
                engine.shard.comm_schedule(3, r, meta, 14)
            out.append(e.value.code)
        assert len(set(out)) == 1
        return out[0]

    m = good.copy(); m[1, 3] = 16            # rank 1's gathered side is one entry short
    assert codes(m) == A.ERR_ARG
    m = good.copy(); m[2, 2] = 7             # rank 2 gathers without the state, the others with it
    assert codes(m) == A.ERR_ARG
    m = good.copy(); m[0, 4] = A.ERR_ARG     # rank 0 failed its own argument checks
    assert codes(m) == A.ERR_STATE
    m = good.copy(); m[1, 1] = -1
    assert codes(m) == A.ERR_ARG
    L = A.lib()
    assert L.mplx_comm_schedule(0, 0, good.ctypes.data, 14, None, 0, None, None) == A.ERR_ARG
    assert L.mplx_comm_schedule(3, 3, good.ctypes.data, 14, None, 0, None, None) == A.ERR_ARG
    # a too small op array: the count is still returned and nothing is written past the capacity
    ops = (A.CommOp * 2)()
    n = L.mplx_comm_schedule(3, 0, good.ctypes.data, 14, ops, 1, None, None)
    assert n > 2 and ops[1].bytes == 0
