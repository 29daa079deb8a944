"""GPU: successor post-processing (post_kernel.hip through
mplx_post_lists_device) against the oracle: the default heuristic
(env_base.h:46-64), the goal tolerances of env_map::is_goal (env_map.h:25-37)
and the node identity the search derives from the lattice hash."""
import numpy as np
import pytest

from helpers import engine_env
from test_gpu_parity import _small_world

pytestmark = pytest.mark.gpu


def _emitted_indices(lists):
    S = lists["stride"]
    idx = np.concatenate([k * S + np.arange(c) for k, c in enumerate(lists["count"])]) if lists["count"].sum() else \
        np.zeros(0, np.int64)
    return idx.astype(np.int64)


@pytest.mark.parametrize("dim,control", [(2, 0x03), (3, 0x03), (3, 0x07), (2, 0x13)])
def test_heuristic_goal_flags_and_identity(engine, oracle_lib, dim, control):
    O = oracle_lib
    wl = _small_world(engine, dim, control, seed=4000 + 10 * dim + control, n_nodes=150)
    # duplicates across the batch: repeat some nodes (same successors), and nodes that reach common states
    wl.nodes[:, 100:150] = wl.nodes[:, 0:50]
    env = engine_env(engine, wl)
    fr = env.upload_frontier(wl.nodes)
    lists = env.alloc_lists(wl.n_nodes, want_state=True)
    env.expand_lists_resident(fr, lists)
    env.synchronize()
    L = lists.download()
    idx = _emitted_indices(L)
    assert idx.size > 500
    # goal: the state of one emitted successor, so that "same lattice state" and the tolerances fire
    goal = L["state"][:, idx[idx.size // 3]].copy()
    w, v_max = 10.0, 1.5
    for tols in ((0.5, -1.0, -1.0, -1.0), (1.0, 0.5, -1.0, -1.0), (1.0, 1.0, 1.5, 0.6), (0.0, 0.0, 0.0, 0.0)):
        got = env.post_lists(lists, goal, w=w, v_max=v_max, tol_pos=tols[0], tol_vel=tols[1], tol_acc=tols[2],
                             tol_yaw=tols[3])
        gh = O.lattice_hash(dim, control, goal)
        n_goal = n_same = 0
        for g in idx[::7]:  # every 7th successor through the scalar oracle entry points
            wp = L["state"][:, g]
            assert got["heur"][g] == O.heur(dim, control, w, v_max, wp, goal), "heur @%d" % g
            want_goal = O.goal_tol(dim, wp, goal, *tols)
            assert bool(got["flags"][g] & 1) == want_goal, "goal flag @%d" % g
            assert bool(got["flags"][g] & 2) == (int(L["hash"][g]) == gh)
            n_goal += want_goal
            n_same += int(L["hash"][g]) == gh
        # vectorised check of every successor
        d = np.abs(L["state"][:dim, idx] - goal[:dim, None]).max(axis=0)
        same = L["hash"][idx] == np.uint64(gh)
        assert np.array_equal(got["heur"][idx], np.where(same, 0.0, w * d / v_max))
        assert same.any() and (got["flags"][idx] & 1).any()
        # node identity: canon = smallest list index with the same hash; bit 2 marks it
        h = L["hash"][idx]
        order = np.lexsort((idx, h))
        hs, ids = h[order], idx[order]
        first = np.concatenate([[True], hs[1:] != hs[:-1]])
        canon_sorted = np.maximum.accumulate(np.where(first, np.arange(ids.size), 0))
        want_canon = np.empty(idx.size, np.int64)
        want_canon[order] = ids[canon_sorted]
        assert np.array_equal(got["canon"][idx].astype(np.int64), want_canon)
        assert np.array_equal((got["flags"][idx] & 4) != 0, want_canon == idx)
        assert (want_canon != idx).sum() > 50  # the batch really had duplicates
    lists.free()
    fr.free()
    env.close()


def test_post_lists_without_velocity_limit_and_without_identity(engine, oracle_lib):
    wl = _small_world(engine, 2, 0x03, seed=4242, n_nodes=40)
    env = engine_env(engine, wl)
    fr = env.upload_frontier(wl.nodes)
    lists = env.alloc_lists(wl.n_nodes, want_state=True, stride=wl.U.shape[0])
    env.expand_lists_resident(fr, lists)
    env.synchronize()
    L = lists.download()
    idx = _emitted_indices(L)
    goal = np.zeros(10)
    goal[:2] = [2.0, 1.0]
    got = env.post_lists(lists, goal, w=3.0, v_max=0.0, want_canon=False)  # v_max <= 0: w * |.|_inf (env_base.h:62)
    for g in idx[::5]:
        assert got["heur"][g] == oracle_lib.heur(2, 0x03, 3.0, 0.0, L["state"][:, g], goal)
    assert "canon" not in got
    lists.free()
    fr.free()
    env.close()


def test_post_on_packed_lists_equals_post_on_the_strided_lists(engine):
    """mplx_post_packed_device (the consumer of the multi-GPU gather) against mplx_post_lists_device on the same
    successors: same heuristic and flags entry for entry, and the same first occurrences (canon indices translate
    through the packing)."""
    wl = engine.workloads.make("C4", scale=0.25, n_nodes=900)
    env = engine.EnvMap(wl.dim, 0)
    wl.apply(env)
    fr = env.upload_frontier(wl.nodes)
    lists = env.alloc_lists(wl.n_nodes, want_state=True, want_iters=False)
    env.expand_lists_resident(fr, lists)
    env.synchronize()
    goal = wl.nodes[:, 5].copy()
    strided = env.post_lists(lists, goal, tol_pos=0.6)
    host = lists.download()
    pk = engine.pack_host_lists(host, wl.n_nodes)
    S = lists.stride
    src = (np.repeat(np.arange(wl.n_nodes, dtype=np.int64) * S - pk["offs"][:-1], pk["count"]) + np.arange(pk["total"]))
    packed = env.alloc_packed(wl.n_nodes)
    env.pack_lists(lists, packed)
    got = env.post_packed(packed, wl.n_nodes, goal, tol_pos=0.6)
    assert got["total"] == pk["total"] > 10000
    assert np.array_equal(got["heur"].view(np.uint64), strided["heur"][src].view(np.uint64))
    assert np.array_equal(got["flags"], strided["flags"][src])
    # canon: packed index of the first occurrence == packed position of the strided first occurrence
    to_packed = np.full(host["action"].size, -1, np.int64)
    to_packed[src] = np.arange(pk["total"])
    assert np.array_equal(got["canon"].astype(np.int64), to_packed[strided["canon"][src]])
    assert 0 < np.count_nonzero(got["flags"] & 4) <= pk["total"]
    for b in (packed, lists, fr):
        b.free()
    env.close()


def _want_canon(h, idx):
    order = np.lexsort((idx, h))
    hs, ids = h[order], idx[order]
    first = np.concatenate([[True], hs[1:] != hs[:-1]])
    canon_sorted = np.maximum.accumulate(np.where(first, np.arange(ids.size), 0))
    want = np.empty(idx.size, np.int64)
    want[order] = ids[canon_sorted]
    return want


@pytest.mark.parametrize("knobs", [
    {"_form": ("claimed", "claimed+exact")},                   # automatic: two partition levels at this size, claimed buckets
    {"MPLX_POST_CLAIMED": "0", "_form": ("exact",)},           # ... by histograms + prefix sums
    {"MPLX_POST_CAP": "4096,64", "_form": ("claimed+exact",)}, # fine buckets of 64 pairs: overflow -> the exact form runs after
    {"MPLX_POST_CAP": "64,1024", "_form": ("claimed+exact",)}, # (bucket, shard) segments of 64 pairs: overflow at level 1
    {"MPLX_POST_BITS": "2,3", "_form": ("claimed",)},          # 4 x 8 claimed buckets of ~70 k slots: rounds in every table
    {"MPLX_POST_BITS": "3,0"},                                 # one level, 8 buckets of ~50 k successors: many rounds per bucket
    {"MPLX_POST_BITS": "6,8", "MPLX_POST_FILL": "40"},         # finest partition, tiny tables: rounds in most buckets
    {"MPLX_POST_BITS": "0,0", "MPLX_POST_FILL": "900"},        # ONE bucket for everything
    {"MPLX_POST_PARTITION_MIN": "1000000000"},                 # the table in HBM (the small-batch route) on the same input
])
def test_partitioned_identity_equals_first_occurrence(engine, monkeypatch, knobs):
    """identity_kernel.hip (radix partition + LDS tables) against the definition: canon[g] = smallest list index with
    the same lattice hash, bit 2 of the flags = first occurrence -- on a frontier with heavy duplication (repeated
    nodes, a lattice start region) so that buckets hold far more successors than distinct keys, for partitions finer
    and coarser than the automatic one and for tables that overflow into further rounds."""
    monkeypatch.setenv("MPLX_POST_PARTITION_MIN", "0")
    knobs = dict(knobs)
    forms = knobs.pop("_form", None)
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    wl = engine.workloads.make("C4", scale=0.25, n_nodes=3000)
    rng = np.random.default_rng(9)
    wl.nodes[:, 1000:2000] = wl.nodes[:, rng.integers(0, 1000, size=1000)]   # every third node a repeat
    wl.nodes[:3, 2000:2300] = np.round(wl.nodes[:3, 2000:2001], 1)           # 300 nodes on one lattice position
    env = engine_env(engine, wl)
    fr = env.upload_frontier(wl.nodes)
    lists = env.alloc_lists(wl.n_nodes, want_state=True, want_iters=False)
    env.expand_lists_resident(fr, lists)
    env.synchronize()
    L = lists.download()
    idx = _emitted_indices(L)
    assert idx.size > 400000
    goal = wl.nodes[:, 7].copy()
    got = env.post_lists(lists, goal, tol_pos=0.6)
    if forms:
        assert env.last_identity_form() in forms, env.last_identity_form()
    want = _want_canon(L["hash"][idx], idx)
    assert np.array_equal(got["canon"][idx].astype(np.int64), want)
    assert np.array_equal((got["flags"][idx] & 4) != 0, want == idx)
    assert (want != idx).sum() > 100000
    # the packed form (one long list; what the multi-GPU merge runs on)
    packed = env.alloc_packed(wl.n_nodes)
    env.pack_lists(lists, packed)
    gp = env.post_packed(packed, wl.n_nodes, goal, tol_pos=0.6)
    hp = packed.download()["hash"]
    wantp = _want_canon(hp, np.arange(hp.size, dtype=np.int64))
    assert np.array_equal(gp["canon"].astype(np.int64), wantp)
    assert np.array_equal((gp["flags"] & 4) != 0, wantp == np.arange(hp.size))
    for b in (packed, lists, fr):
        b.free()
    env.close()


def test_partitioned_identity_full_size_c4(engine):
    """C4's whole frontier (20.4 M successors in 48 M list slots): the partitioned pass against numpy."""
    wl = engine.workloads.make("C4")
    env = engine_env(engine, wl)
    fr = env.upload_frontier(wl.nodes)
    lists = env.alloc_lists(wl.n_nodes, want_state=True, want_iters=False)
    env.expand_lists_resident(fr, lists)
    env.synchronize()
    got = env.post_lists(lists, wl.nodes[:, 0].copy())
    assert env.last_identity_form() == "claimed"
    h = lists.hash.download(np.uint64, (lists.n_slots,))
    cnt = lists.count.download(np.int32, (wl.n_nodes,))
    S = lists.stride
    valid = (np.arange(S)[None, :] < cnt[:, None]).ravel()
    idx = np.nonzero(valid)[0].astype(np.int64)
    want = _want_canon(h[idx], idx)
    assert np.array_equal(got["canon"][idx].astype(np.int64), want)
    assert np.array_equal((got["flags"][idx] & 4) != 0, want == idx)
    lists.free()
    fr.free()
    env.close()


def _unmix(m):
    """Inverse of identity_kernel.hip's mix (murmur3's 64-bit finaliser): the hash whose mixed key is m."""
    M = (1 << 64) - 1

    def unshift(x):  # inverse of x ^= x >> 33
        return x ^ (x >> 33)

    m = unshift(m)
    m = (m * pow(0xc4ceb9fe1a85ec53, -1, 1 << 64)) & M
    m = unshift(m)
    m = (m * pow(0xff51afd7ed558ccd, -1, 1 << 64)) & M
    return unshift(m)


@pytest.mark.parametrize("shape", ["distinct", "few_keys", "one_key", "zipf", "special_keys"])
@pytest.mark.parametrize("form", ["claimed", "exact"])
def test_identity_on_synthetic_hashes(engine, monkeypatch, shape, form):
    """The partitioned identity pass on hashes that no expansion produced: all distinct, a handful of keys, ONE key
    (every pair in one bucket: the claimed form must notice the overflow and fall back), a Zipf mix, and the two keys the
    tables cannot store as they are -- ~0 (the empty marker of the exact form's tables) and the hash whose MIXED key is
    ~0 (the empty marker of the claimed form's, which stores mixed keys) -- among ordinary ones.  Packed lists (one
    long list: mplx_post_packed_device), canon[] against numpy."""
    monkeypatch.setenv("MPLX_POST_PARTITION_MIN", "0")
    monkeypatch.setenv("MPLX_POST_CLAIMED", "1" if form == "claimed" else "0")
    assert _unmix(0xFFFFFFFFFFFFFFFF) != 0xFFFFFFFFFFFFFFFF
    n = 1_500_000
    rng = np.random.default_rng({"distinct": 11, "few_keys": 12, "one_key": 13, "zipf": 14, "special_keys": 15}[shape])  # (not hash(): PYTHONHASHSEED)
    if shape == "distinct":
        h = rng.permutation(np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))
    elif shape == "few_keys":
        h = rng.integers(1, 40, size=n).astype(np.uint64) * np.uint64(0x123456789ABCDEF1)
    elif shape == "one_key":
        h = np.full(n, 0xDEADBEEFCAFEF00D, dtype=np.uint64)
    elif shape == "zipf":
        h = (rng.zipf(1.3, size=n).astype(np.uint64) % np.uint64(200_000)) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(7)
    else:
        h = rng.integers(1, 1 << 62, size=n, dtype=np.int64).astype(np.uint64)
        sp = rng.choice(n - 1, size=900, replace=False) + 1
        h[sp[:300]] = np.uint64(0xFFFFFFFFFFFFFFFF)
        h[sp[300:600]] = np.uint64(_unmix(0xFFFFFFFFFFFFFFFF))
        h[sp[600:]] = h[sp[600:] - 1]  # some ordinary duplicates as well
    wl = engine.workloads.make("C2", scale=0.125, n_nodes=8)
    env = engine_env(engine, wl)
    F = 4 * wl.dim + 2
    offs = engine.env.DeviceArray(env, 16)
    offs.upload(np.array([0, n], dtype=np.int64))
    hd = engine.env.DeviceArray(env, n * 8)
    hd.upload(h)
    st = engine.env.DeviceArray(env, F * n * 8)
    st.upload(np.zeros(F * n, dtype=np.float64))
    ps = engine._abi.PackedLists()
    ps.count, ps.offs, ps.action, ps.cost, ps.hash, ps.state = None, offs.ptr, None, None, hd.ptr, st.ptr
    ps.state_stride, ps.capacity = n, n
    got = env.post_packed(ps, 1, np.zeros(F))
    used = env.last_identity_form()
    # (a key with more duplicates than a fine bucket's capacity -- one_key, few_keys, the head of the Zipf mix -- overflows)
    assert used == {"claimed": "claimed+exact" if shape in ("one_key", "few_keys", "zipf") else "claimed", "exact": "exact"}[form], used
    want = _want_canon(h, np.arange(n, dtype=np.int64))
    assert np.array_equal(got["canon"].astype(np.int64), want)
    assert np.array_equal((got["flags"] & 4) != 0, want == np.arange(n))
    for b in (offs, hd, st):
        b.free()
    env.close()


# ---------------------------------------------------------------- fused rows (ABI v8: mplx_set_goal + mplx_succ_lists::heur / flags)
def _fused_vs_pass(engine, wl, route, kernel, tols=(0.6, -1.0, -1.0, -1.0), edges_only=False, monkeypatch=None):
    """The heur / flags rows the expansion launch writes itself against mplx_post_lists_device on the same lists
    (which reads hash and state rows back): bit-identical on every emitted successor."""
    env = engine_env(engine, wl)
    if route:
        env.set_lists_route(route)
    fr = env.upload_frontier(wl.nodes)
    ref_lists = env.alloc_lists(wl.n_nodes, want_state=True)
    env.expand_lists_resident(fr, ref_lists)
    env.synchronize()
    L = ref_lists.download()
    idx = _emitted_indices(L)
    assert idx.size > 200
    goal = L["state"][:, idx[idx.size // 3]].copy()  # an emitted successor: "same lattice state" and the tolerances fire
    w, v_max = 10.0, 1.5
    want = env.post_lists(ref_lists, goal, w=w, v_max=v_max, tol_pos=tols[0], tol_vel=tols[1], tol_acc=tols[2], tol_yaw=tols[3],
                          want_canon=True)
    env.set_goal(goal, w=w, v_max=v_max, tol_pos=tols[0], tol_vel=tols[1], tol_acc=tols[2], tol_yaw=tols[3])
    lists = env.alloc_lists(wl.n_nodes, want_state=not edges_only, want_heur=True, want_flags=True)
    env.expand_lists_resident(fr, lists)
    env.synchronize()
    assert env.last_lists_route() == (route or env.last_lists_route()) and (kernel is None or env.last_grid_kernel() == kernel)
    G = lists.download()
    assert np.array_equal(G["count"], L["count"]) and np.array_equal(G["hash"][idx], L["hash"][idx])
    assert np.array_equal(G["heur"][idx].view(np.uint64), want["heur"][idx].view(np.uint64))
    assert np.array_equal(G["flags"][idx], want["flags"][idx] & 3)
    assert (G["flags"][idx] & 1).any() and (G["flags"][idx] & 2).any() and (G["heur"][idx] > 0).any()
    # node identity on lists WITHOUT state rows: canon as before, bit 2 merged into the row the launch wrote
    import ctypes as C
    from motion_primitive_library_amd import _abi
    canon = engine.env.DeviceArray(env, lists.n_slots * 4)
    g = _abi.GoalSpec()
    gr = np.ascontiguousarray(goal)
    g.goal, g.control, g.w, g.v_max = gr.ctypes.data, wl.control, w, v_max
    g.tol_pos, g.tol_vel, g.tol_acc, g.tol_yaw = tols
    o = _abi.Post()
    o.heur, o.flags, o.canon = None, lists.flags.ptr, canon.ptr
    s = lists.c_struct()
    s.state = None
    _abi.check(env._ctx, _abi.lib().mplx_post_lists_device(env._ctx, C.byref(s), wl.n_nodes, C.byref(g), C.byref(o)))
    env.synchronize()
    got_canon = canon.download(np.int32, (lists.n_slots,))
    got_flags = lists.flags.download(np.uint8, (lists.n_slots,))
    assert np.array_equal(got_canon[idx], want["canon"][idx])
    assert np.array_equal(got_flags[idx], want["flags"][idx])
    canon.free()
    # without a goal the rows are refused, loudly
    env.set_goal(None)
    with pytest.raises(engine._abi.MplxError):
        env.expand_lists_resident(fr, lists)
    for b in (lists, ref_lists, fr):
        b.free()
    env.close()


@pytest.mark.parametrize("dim,control,route,kernel", [(2, 0x03, "grid", "lex"), (3, 0x03, "grid", "lex"), (3, 0x07, "grid", "lex"),
                                                      (2, 0x01, "grid", "lex"), (3, 0x03, "tile", "none"), (2, 0x07, "tile", "none"),
                                                      (3, 0x03, "dense", "none"), (2, 0x13, "grid", "grid"), (3, 0x0F, "grid", "grid")])
def test_fused_heuristic_and_flags_equal_the_stand_alone_pass(engine, dim, control, route, kernel):
    wl = _small_world(engine, dim, control, seed=6100 + 10 * dim + control, n_nodes=150)
    wl.nodes[:, 100:150] = wl.nodes[:, 0:50]
    for tols in ((0.6, -1.0, -1.0, -1.0), (1.0, 1.0, 1.5, 0.6)):
        _fused_vs_pass(engine, wl, route, kernel, tols=tols)


def test_fused_rows_general_kernel_and_edges_only(engine, monkeypatch):
    """The general factorised kernel on a table the lexicographic one would take (MPLX_GRID_LEX=0), and lists without
    state rows (what the engine's own search asks for): the rows do not depend on the state rows being stored."""
    wl = _small_world(engine, 3, 0x03, seed=6500, n_nodes=120)
    _fused_vs_pass(engine, wl, "grid", "lex", edges_only=True)
    _fused_vs_pass(engine, wl, "tile", "none", edges_only=True)
    monkeypatch.setenv("MPLX_GRID_LEX", "0")
    _fused_vs_pass(engine, wl, "grid", "grid")
    _fused_vs_pass(engine, wl, "grid", "grid", edges_only=True)


def test_fused_rows_through_host_pointers_and_the_service(engine):
    """mplx_expand_lists on host pointers (the arena / zero-copy path and, from the second call in a row, the resident
    kernel): the heur / flags rows come back with the lists; a new goal reaches a resident kernel too."""
    import ctypes as C
    from motion_primitive_library_amd import _abi
    wl = engine.workloads.make("C4", scale=0.125, n_nodes=48)
    env = engine_env(engine, wl)
    L = _abi.lib()
    nU, n = wl.U.shape[0], wl.n_nodes
    S = (nU + 31) & ~31
    ref = env.expand_lists(wl.nodes, want_state=True, want_iters=False, stride=S)
    idx = _emitted_indices(ref)
    for trial, gi in enumerate((idx[10], idx[idx.size // 2], idx[-5])):
        goal = np.ascontiguousarray(ref["state"][:, gi])
        env.set_goal(goal, w=10.0, v_max=2.0, tol_pos=0.5)
        cnt = np.zeros(n, np.int32)
        act, cost, hsh = np.zeros(n * S, np.int32), np.zeros(n * S), np.zeros(n * S, np.uint64)
        heur, flags = np.full(n * S, -1.0), np.zeros(n * S, np.uint8)
        o = _abi.SuccLists()
        o.count, o.action, o.cost, o.hash = cnt.ctypes.data, act.ctypes.data, cost.ctypes.data, hsh.ctypes.data
        o.heur, o.flags, o.node_stride = heur.ctypes.data, flags.ctypes.data, S
        nodes = np.ascontiguousarray(wl.nodes)
        for rep in range(3):  # the third call of a row is served by the resident kernel
            _abi.check(env._ctx, L.mplx_expand_lists(env._ctx, nodes.ctypes.data, n, n, C.byref(o)))
        d = np.abs(ref["state"][:3, idx] - goal[:3, None]).max(axis=0)
        same = ref["hash"][idx] == ref["hash"][gi]
        assert np.array_equal(cnt, ref["count"]) and np.array_equal(hsh[idx], ref["hash"][idx])
        assert np.array_equal(heur[idx], np.where(same, 0.0, 10.0 * d / 2.0)), trial
        assert np.array_equal(flags[idx] & 1, (d <= 0.5).astype(np.uint8)) and np.array_equal((flags[idx] & 2) != 0, same)
    env.close()


@pytest.mark.parametrize("form", ["claimed", "claimed_overflow", "sticky", "exact"])
def test_identity_on_lists_without_state_rows_partition_forms(engine, monkeypatch, form):
    """Lists of >= 256 k slots whose flags row the expansion launch wrote, node identity by the radix partition: canon[] and
    the first-occurrence bit OR-ed into that row -- by the claimed form (the bit pass is queued before the host learns
    whether a bucket overflowed), by the claimed form that DOES overflow (bits undone, exact form, bits again) and by the
    exact form -- against the stand-alone pass on the same lists with state rows."""
    import ctypes as C
    from motion_primitive_library_amd import _abi
    if form == "exact":
        monkeypatch.setenv("MPLX_POST_CLAIMED", "0")
    elif form == "sticky":
        monkeypatch.delenv("MPLX_POST_CLAIMED", raising=False)  # the default choice, with its memory of an overflow
    else:
        monkeypatch.setenv("MPLX_POST_CLAIMED", "1")
    if form in ("claimed_overflow", "sticky"):
        monkeypatch.setenv("MPLX_POST_CAP", "64,64")
    wl = engine.workloads.make("C4", scale=0.25, n_nodes=900)
    wl.nodes[:, 450:] = wl.nodes[:, :450]  # every successor twice
    env = engine.EnvMap(wl.dim, 0)
    wl.apply(env)
    fr = env.upload_frontier(wl.nodes)
    goal = np.ascontiguousarray(wl.nodes[:, 7])
    ref_lists = env.alloc_lists(wl.n_nodes, want_state=True)
    env.expand_lists_resident(fr, ref_lists)
    env.synchronize()
    assert ref_lists.n_slots >= 1 << 18
    want = env.post_lists(ref_lists, goal, w=10.0, v_max=2.0, tol_pos=0.6)
    want_form = env.last_identity_form()
    L = ref_lists.download()
    idx = _emitted_indices(L)
    env.set_goal(goal, w=10.0, v_max=2.0, tol_pos=0.6)
    lists = env.alloc_lists(wl.n_nodes, want_state=False, want_heur=True, want_flags=True)
    canon = engine.env.DeviceArray(env, lists.n_slots * 4)
    g = _abi.GoalSpec()
    g.goal, g.control, g.w, g.v_max = goal.ctypes.data, wl.control, 10.0, 2.0
    g.tol_pos, g.tol_vel, g.tol_acc, g.tol_yaw = 0.6, -1.0, -1.0, -1.0
    o = _abi.Post()
    o.heur, o.flags, o.canon = None, lists.flags.ptr, canon.ptr
    for rep in range(2):  # (the second call of a context after an overflow takes the exact form directly)
        env.expand_lists_resident(fr, lists)
        s = lists.c_struct()
        _abi.check(env._ctx, _abi.lib().mplx_post_lists_device(env._ctx, C.byref(s), wl.n_nodes, C.byref(g), C.byref(o)))
        env.synchronize()
        got_form = env.last_identity_form()
        got_canon = canon.download(np.int32, (lists.n_slots,))
        got_flags = lists.flags.download(np.uint8, (lists.n_slots,))
        got_heur = lists.heur.download(np.float64, (lists.n_slots,))
        assert np.array_equal(got_canon[idx], want["canon"][idx]), (form, rep)
        assert np.array_equal(got_flags[idx], want["flags"][idx]), (form, rep)
        assert np.array_equal(got_heur[idx].view(np.uint64), want["heur"][idx].view(np.uint64))
        assert 0 < np.count_nonzero(got_flags[idx] & 4) <= idx.size // 2  # duplicates: at most every second is a first
        if form == "claimed":
            assert got_form == "claimed"
        elif form == "claimed_overflow":
            assert got_form == "claimed+exact", got_form
        elif form == "sticky":
            # the stand-alone pass above was this context's first call: it overflowed, and the context remembers --
            # its next calls take the exact form directly (and stay asynchronous)
            assert want_form == "claimed+exact" and got_form == "exact", (rep, want_form, got_form)
        else:
            assert got_form == "exact"
    canon.free()
    for b in (lists, ref_lists, fr):
        b.free()
    env.close()
