"""GPU: BASELINE.json's full-size configurations.  The oracle cannot finish these
in seconds, so they are checked through size-independent properties:
  * the three independently written kernels (dense lane-per-pair, factorised
    grid lists) must agree on every one of the 47.8 M pairs of C4 / 2 M of C3 /
    102 k of C2 -- successor set, order, lattice hash, cost bit for bit;
  * a slice of the frontier is checked against the oracle directly;
  * expansion is a pure per-node function: permuting the frontier permutes the
    lists, and expanding a node twice gives the same list (idempotence);
  * conservation: emitted = finite + blocked, counts sum to the emitted total."""
import numpy as np
import pytest

from helpers import assert_lists_equal, engine_env, oracle_env

pytestmark = pytest.mark.gpu


def _lists_resident(env, nodes, want_state=False):
    fr = env.upload_frontier(nodes)
    lists = env.alloc_lists(nodes.shape[1], want_state=want_state, want_iters=True)
    env.expand_lists_resident(fr, lists)
    env.synchronize()
    out = lists.download()
    lists.free()
    fr.free()
    return out


@pytest.mark.parametrize("name", ["C2", "C3", "C4", "C5", "C5-tunnel"])
def test_full_size_kernels_agree_and_slice_matches_oracle(engine, oracle_lib, name):
    wl = engine.workloads.make(name.split("-")[0])  # BASELINE.json size: full map, full frontier
    if name == "C5-tunnel":
        # SURVEY 8(d)'s second C5 variant: a search region of radius 0.5 m around a straight start-goal path
        edge = wl.map_dim[0]
        wl.region = engine.workloads.tunnel_region(wl.map_dim, wl.origin, wl.res, [2.0] * 3, [edge * wl.res - 2.0] * 3, 0.5)
        # half of the frontier inside the tunnel, or nothing would be traversable
        t = np.linspace(0.0, 1.0, wl.n_nodes // 2)
        rng = np.random.default_rng(5)
        for i in range(3):
            wl.nodes[i, ::2] = np.round(2.0 + t * (edge * wl.res - 4.0) + rng.uniform(-0.3, 0.3, size=t.size), 2)
    nU, N = wl.U.shape[0], wl.n_nodes
    env = engine_env(engine, wl)
    L = _lists_resident(env, wl.nodes)
    assert env.last_lists_route() == "grid"
    # dense kernel over the same frontier, compact outputs only (status, cost, hash, iters)
    fr = env.upload_frontier(wl.nodes)
    slots = env.alloc_slots(N, want_state=False, want_iters=True)
    env.expand_resident(fr, slots)
    env.synchronize()
    Dn = slots.download()
    slots.free()
    fr.free()
    Dn["state"] = None
    L["state"] = None
    assert_lists_equal(L, Dn, N, nU, what="%s full size, grid lists vs dense kernel" % name)
    st = Dn["status"]
    n_emit = int(np.count_nonzero((st == 1) | (st == 2)))
    assert int(L["count"].sum(dtype=np.int64)) == n_emit
    assert n_emit == int(np.count_nonzero(st == 1)) + int(np.count_nonzero(st == 2))
    print("%s full size: %d pairs, %d emitted, %d finite" % (name, st.size, n_emit, int(np.count_nonzero(st == 1))))
    assert np.count_nonzero(st == 1) > 0 and np.count_nonzero(st == 2) > 0 and np.count_nonzero(st == 3) > 0
    # a slice against the oracle itself (state included)
    n_chk = 96
    sub = np.ascontiguousarray(wl.nodes[:, :n_chk])
    ref = oracle_lib.expand(oracle_env(wl), sub, threads=16)
    got = _lists_resident(env, sub, want_state=True)
    # xYAW: glibc's cos / sin against the device's (tests/test_gpu_parity.py::YAW_COST_RTOL)
    assert_lists_equal(got, ref, n_chk, nU, cost_rtol=1e-6 if wl.control & 0x10 else 0.0, what="%s oracle slice" % name)
    env.close()


def test_full_size_c4_permutation_and_idempotence(engine):
    wl = engine.workloads.make("C4")
    nU, N = wl.U.shape[0], wl.n_nodes
    env = engine_env(engine, wl)
    A = _lists_resident(env, wl.nodes)
    rng = np.random.default_rng(11)
    perm = rng.permutation(N)
    perm[:1000] = perm[1000:2000]  # duplicates: the same node expanded twice in one batch
    B = _lists_resident(env, np.ascontiguousarray(wl.nodes[:, perm]))
    env.close()
    S = A["stride"]
    assert np.array_equal(B["count"], A["count"][perm])
    for key in ("action", "hash", "cost", "iters"):
        a = A[key].reshape(N, S)[perm]
        b = B[key].reshape(N, S)
        mask = np.arange(S)[None, :] < B["count"][:, None]
        assert np.array_equal(a[mask].view(np.uint64 if a.dtype.itemsize == 8 else a.dtype),
                              b[mask].view(np.uint64 if b.dtype.itemsize == 8 else b.dtype)), key
