"""GPU: the service -- small synchronous batches of mplx_expand_lists / mplx_get_succ through a kernel that stays
resident between the calls and takes its requests from a mailbox in pinned host memory (include/mplx.h mplx_service,
expand_tile_kernel.hip "SERVICE MODE").  The lists must be the bytes an ordinary launch produces (and the oracle's),
whatever ends, restarts or bypasses the resident kernel in between."""
import time

import numpy as np
import pytest

from helpers import assert_lists_equal, engine_env, oracle_env
from test_gpu_parity import YAW_COST_RTOL, _small_world

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _patient_kernel(monkeypatch):
    """The tests below count launches of the resident kernel: it must not leave on its own (2 ms without a request by
    default) while Python allocates the next result arrays.  The test of that exit sets its own limit."""
    monkeypatch.setenv("MPLX_SERVICE_IDLE_US", "500000")

KEYS = ("count", "action", "cost", "hash", "state", "iters")


def _same(a, b, n_nodes):
    assert np.array_equal(a["count"], b["count"])
    S = a["stride"]
    used = (np.arange(S)[None, :] < a["count"][:, None]).ravel()
    assert used.sum() > 0
    for k in ("action", "hash", "iters"):
        assert np.array_equal(a[k][used], b[k][used]), k
    assert np.array_equal(a["cost"][used].view(np.uint64), b["cost"][used].view(np.uint64))
    assert np.array_equal(a["state"][:, used].view(np.uint64), b["state"][:, used].view(np.uint64))


def _reference_lists(engine, wl, batches):
    """every batch as a launch of its own (service off)"""
    env = engine_env(engine, wl)
    env.service(0)
    out = [env.expand_lists(b) for b in batches]
    assert env.service()["requests"] == 0
    env.close()
    return out


@pytest.mark.parametrize("dim,control,nU_vals", [
    (2, 0x03, None),                       # the reference's 2D test problem: several nodes per workgroup
    (3, 0x03, [-1.0, -0.5, 0.0, 0.5, 1.0]),  # 125 controls
    (3, 0x07, None),                       # jerk control
    (2, 0x01, None),                       # velocity control
])
def test_batches_through_the_resident_kernel_equal_launches(engine, dim, control, nU_vals):
    wl = _small_world(engine, dim, control, seed=7100 + 16 * dim + control, n_nodes=400)
    if nU_vals is not None:
        wl.U = engine.workloads.grid_controls(nU_vals, dim)
    rng = np.random.default_rng(5)
    sizes = [16, 16, 1, 7, 64, 33, 16, 2, 64, 5]
    batches = [np.ascontiguousarray(wl.nodes[:, rng.integers(0, wl.n_nodes, size=n)]) for n in sizes]
    want = _reference_lists(engine, wl, batches)
    env = engine_env(engine, wl)
    got = [env.expand_lists(b) for b in batches]
    st = env.service()
    assert st["failures"] == 0 and st["launches"] == 1 and st["resident"]
    assert st["requests"] == len(batches) - 1  # the first call of a row is a launch of its own
    for a, b, n in zip(got, want, sizes):
        _same(a, b, n)
    assert env.last_lists_route() == "tile"
    env.close()


def test_one_workgroup_per_node_and_many_workgroups(engine):
    """|U| = 729: a node is a workgroup (the ONE form, node state through scalar loads), 64 resident workgroups that
    the coordinator waits for.  (Larger batches of this table are launches of their own: the service stops at 64
    workgroups, and 64 nodes x 729 list entries x 14 state rows is what the 8 MB landing block holds.)"""
    wl = engine.workloads.make("C4", scale=0.125, n_nodes=600)
    rng = np.random.default_rng(11)
    sizes = [64, 64, 1, 63, 2, 17, 64, 3, 100, 5, 5]
    batches = [np.ascontiguousarray(wl.nodes[:, rng.integers(0, wl.n_nodes, size=n)]) for n in sizes]
    want = _reference_lists(engine, wl, batches)
    env = engine_env(engine, wl)
    got = [env.expand_lists(b, want_iters=True) for b in batches]
    st = env.service()
    # 100 nodes are 100 workgroups, more than the handshake is worth: that batch is an ordinary launch, which ends the
    # resident kernel and the row; the second small batch after it starts the next one
    assert st["failures"] == 0 and st["requests"] == 8 and st["launches"] == 2
    for a, b, n in zip(got, want, sizes):
        _same(a, b, n)
    env.close()


def test_capacity_grows_with_the_batches(engine):
    wl = _small_world(engine, 2, 0x03, seed=7200, n_nodes=400)
    sizes = [8, 8, 100, 100, 200, 256, 8, 64]
    batches = [np.ascontiguousarray(wl.nodes[:, :n]) for n in sizes]
    want = _reference_lists(engine, wl, batches)
    env = engine_env(engine, wl)
    got = [env.expand_lists(b) for b in batches]
    st = env.service()
    assert st["failures"] == 0 and st["requests"] == 7 and st["launches"] == 3  # capacity 64, 128, 256
    for a, b, n in zip(got, want, sizes):
        _same(a, b, n)
    env.close()


def test_same_slot_different_nodes(engine):
    """The landing block is reused request after request: a node read through a stale cache line (vector L1 / L2 or
    the scalar cache the ONE form reads its node through) would return the previous request's lists."""
    for make in (lambda: _small_world(engine, 2, 0x03, seed=7300, n_nodes=300),
                 lambda: engine.workloads.make("C4", scale=0.125, n_nodes=300)):
        wl = make()
        env = engine_env(engine, wl)
        ref = engine_env(engine, wl)
        ref.service(0)
        for k in range(40):
            b = np.ascontiguousarray(wl.nodes[:, [k, (7 * k) % 300, 299 - k]])
            _same(env.expand_lists(b), ref.expand_lists(b), 3)
        assert env.service()["requests"] == 39 and ref.service()["requests"] == 0
        env.close()
        ref.close()


def test_get_succ_in_a_row_matches_the_oracle(engine, oracle_lib):
    """the adapter's pattern: one node per call (mplx_get_succ), against the CPU restatement"""
    wl = _small_world(engine, 2, 0x03, seed=7400, n_nodes=60)
    env = engine_env(engine, wl)
    oenv = oracle_env(wl)
    ref = oracle_lib.expand(oenv, wl.nodes)
    nU = wl.U.shape[0]
    n_succ = 0
    for k in range(60):
        wp = engine.Waypoint.from_row(2, 0x03, wl.nodes[:, k])
        succ, cost, act = env.get_succ(wp)
        st = ref["status"][k * nU:(k + 1) * nU]
        want_act = [i for i in range(nU) if st[i] in (1, 2)]
        assert act == want_act
        for j, i in enumerate(want_act):
            assert np.array_equal(succ[j].to_row().view(np.uint64), ref["state"][:, k * nU + i].view(np.uint64))
            c = ref["cost"][k * nU + i]
            assert cost[j] == c or (np.isinf(cost[j]) and np.isinf(c))
        n_succ += len(act)
    assert n_succ > 100
    st = env.service()
    assert st["requests"] == 59 and st["failures"] == 0
    env.close()


def test_other_calls_end_the_resident_kernel_and_see_its_results(engine):
    """a map edit between two batches (the LPA* pattern), a device-side call, parameters: each ends the resident
    kernel; the next batches are served by a new one and reflect the change"""
    wl = _small_world(engine, 2, 0x03, seed=7500, n_nodes=200)
    env = engine_env(engine, wl)
    b = np.ascontiguousarray(wl.nodes[:, :24])
    first = env.expand_lists(b)
    env.expand_lists(b)
    assert env.service()["resident"]
    # 1. map edit: block everything -> every successor that moves is blocked (cost inf), then restore
    blocked = np.full_like(wl.grid, 100)
    env.setMap(wl.origin, wl.map_dim, blocked, wl.res)
    assert not env.service()["resident"]
    x = env.expand_lists(b)
    y = env.expand_lists(b)  # resident again
    assert env.service()["resident"] and env.service()["launches"] == 2
    _same(x, y, 24)
    used = (np.arange(x["stride"])[None, :] < x["count"][:, None]).ravel()
    assert np.isinf(y["cost"][used]).sum() > 0.5 * used.sum()
    env.setMap(wl.origin, wl.map_dim, wl.grid, wl.res)
    env.expand_lists(b)
    _same(env.expand_lists(b), first, 24)
    # 2. a device-side call on the context's own stream in between
    fr = env.upload_frontier(wl.nodes)
    lists = env.alloc_lists(wl.n_nodes)
    env.expand_lists_resident(fr, lists)
    env.synchronize()
    assert not env.service()["resident"]
    env.expand_lists(b)
    _same(env.expand_lists(b), first, 24)
    # 3. parameters: a tighter velocity limit removes successors
    env.set_v_max(0.6)
    z0 = env.expand_lists(b)
    z1 = env.expand_lists(b)
    _same(z0, z1, 24)
    assert z1["count"].sum() < first["count"].sum()
    assert env.service()["failures"] == 0
    lists.free()
    fr.free()
    env.close()


def test_idle_kernel_leaves_and_the_next_request_brings_it_back(engine, monkeypatch):
    monkeypatch.setenv("MPLX_SERVICE_IDLE_US", "300")
    wl = _small_world(engine, 2, 0x03, seed=7600, n_nodes=100)
    env = engine_env(engine, wl)
    ref = engine_env(engine, wl)
    ref.service(0)
    b = np.ascontiguousarray(wl.nodes[:, :16])
    want = ref.expand_lists(b)
    env.expand_lists(b)
    env.expand_lists(b)
    for _ in range(3):
        time.sleep(0.02)  # 20 ms >> 300 us: the resident kernel has left on its own
        _same(env.expand_lists(b), want, 16)
        _same(env.expand_lists(b), want, 16)
    st = env.service()
    # (a slow moment of the interpreter between two calls is one more exit and one more launch)
    assert 4 <= st["launches"] <= 7 and st["failures"] == 0 and st["requests"] == 7
    env.close()
    ref.close()


def test_configurations_outside_the_tiled_kernel_and_large_batches_bypass_it(engine):
    # yaw controls: not the tiled kernel's -> never resident
    wl = _small_world(engine, 2, 0x13, seed=7700, n_nodes=60)
    env = engine_env(engine, wl)
    b = np.ascontiguousarray(wl.nodes[:, :8])
    for _ in range(4):
        env.expand_lists(b)
    assert env.service() == {"requests": 0, "launches": 0, "failures": 0, "resident": False}
    env.close()
    # a forced route other than the tiled kernel, and batches above the limit
    wl = _small_world(engine, 2, 0x03, seed=7701, n_nodes=700)
    env = engine_env(engine, wl)
    env.set_lists_route("grid")
    for _ in range(3):
        env.expand_lists(np.ascontiguousarray(wl.nodes[:, :4]))
    assert env.service()["launches"] == 0
    env.set_lists_route("auto")
    big = np.ascontiguousarray(wl.nodes[:, :600])
    small = np.ascontiguousarray(wl.nodes[:, :16])
    r0 = env.expand_lists(small)
    r1 = env.expand_lists(small)
    assert env.service()["resident"]
    env.expand_lists(big)                       # an ordinary launch; it ends the resident kernel first
    assert not env.service()["resident"]
    _same(r0, r1, 16)
    # switched off by the caller
    env.service(0)
    for _ in range(3):
        env.expand_lists(small)
    assert env.service()["launches"] == 1
    env.close()


def test_two_contexts_resident_at_once(engine):
    """two searches in one process (two contexts): both kernels resident, requests interleaved"""
    wa = _small_world(engine, 2, 0x03, seed=7800, n_nodes=100)
    wb = _small_world(engine, 3, 0x03, seed=7801, n_nodes=100)
    ea, eb = engine_env(engine, wa), engine_env(engine, wb)
    ra, rb = engine_env(engine, wa), engine_env(engine, wb)
    ra.service(0)
    rb.service(0)
    for k in range(12):
        a = np.ascontiguousarray(wa.nodes[:, k:k + 9])
        b = np.ascontiguousarray(wb.nodes[:, k:k + 5])
        _same(ea.expand_lists(a), ra.expand_lists(a), 9)
        _same(eb.expand_lists(b), rb.expand_lists(b), 5)
    assert ea.service()["requests"] == 11 and eb.service()["requests"] == 11
    assert ea.service()["resident"] and eb.service()["resident"]
    for e in (ea, eb, ra, rb):
        e.close()


@pytest.mark.parametrize("which", ["2d_9_controls", "3d_125_controls"])
def test_soak_every_request_is_complete_when_done_is_seen(engine, which):
    """Thousands of requests back to back, each compared with a launch of the same batch: a list entry that reached
    the landing block after `done` did (a store overtaking the release) would show up as a stale or torn entry."""
    if which == "2d_9_controls":
        wl = _small_world(engine, 2, 0x03, seed=7900, n_nodes=500)
        wl.U = engine.workloads.grid_controls([-1.0, 0.0, 1.0], 2)
        rounds = 2500
    else:
        wl = _small_world(engine, 3, 0x03, seed=7901, n_nodes=500)
        wl.U = engine.workloads.grid_controls([-1.0, -0.5, 0.0, 0.5, 1.0], 3)
        rounds = 1200
    env = engine_env(engine, wl)
    ref = engine_env(engine, wl)
    ref.service(0)
    rng = np.random.default_rng(17)
    out_a = out_b = None
    n_prev = -1
    for _ in range(rounds):
        n = int(rng.integers(1, 65))
        b = np.ascontiguousarray(wl.nodes[:, rng.integers(0, wl.n_nodes, size=n)])
        if n != n_prev:
            out_a = out_b = None
            n_prev = n
        out_a = env.expand_lists(b, out=out_a)
        out_b = ref.expand_lists(b, out=out_b)
        _same(out_a, out_b, n)
    st = env.service()
    assert st["failures"] == 0 and st["requests"] == rounds - 1
    env.close()
    ref.close()


@pytest.mark.parametrize("control,potential", [(0x13, False), (0x03, True), (0x03, False)])
def test_launches_that_end_with_the_completion_word_against_the_oracle(engine, oracle_lib, control, potential):
    """Small batches that are launches of their own (yaw controls and potential maps never go to the resident kernel;
    the plain case with the service switched off): the host reads the landing block as soon as the kernel's last wave
    has written the completion word, without synchronising the stream.  Hundreds of launches against the CPU
    restatement; a list entry that arrived after the word would differ."""
    wl = _small_world(engine, 2, control, seed=8000 + control + (100 if potential else 0), n_nodes=300, potential=potential)
    ref = oracle_lib.expand(oracle_env(wl), wl.nodes, threads=8)
    nU = wl.U.shape[0]
    env = engine_env(engine, wl)
    env.service(0)
    rng = np.random.default_rng(23)
    rtol = YAW_COST_RTOL if control & 0x10 else 0.0
    for k in range(400):
        n = int(rng.integers(1, 49))
        ids = rng.integers(0, wl.n_nodes, size=n)
        got = env.expand_lists(np.ascontiguousarray(wl.nodes[:, ids]))
        slots = (ids[:, None] * nU + np.arange(nU)[None, :]).ravel()
        sub = {key: (ref[key][:, slots] if key == "state" else ref[key][slots]) for key in ("status", "cost", "hash", "state", "iters")}
        assert_lists_equal(got, sub, n, nU, cost_rtol=rtol, what="launch %d" % k)
    assert env.service()["requests"] == 0
    env.close()

