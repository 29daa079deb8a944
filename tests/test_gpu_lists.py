"""GPU: the list-producing entry points (mplx_expand_lists*: the factorised kernel
expand_grid_kernel.hip, the general tiled kernel expand_tile_kernel.hip and the
dense+compaction route) against the oracle."""
import numpy as np
import pytest

from helpers import (assert_lists_equal, engine_env, engine_env_from_case, golden_cases, odd_world, oracle_env,
                     oracle_env_from_case)
from test_gpu_parity import YAW_COST_RTOL, _small_world

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("control", [0x01, 0x03, 0x07, 0x0F, 0x11, 0x13, 0x17, 0x1F])
@pytest.mark.parametrize("variant", ["plain", "potential", "region", "nolimits"])
def test_lists_all_controls(engine, oracle_lib, dim, control, variant):
    """plain / region without yaw run the tiled kernel; potential, yaw and
    unbounded-velocity cases run the dense kernel + on-device compaction."""
    wl = _small_world(engine, dim, control, seed=500 * dim + control, potential=(variant == "potential"),
                      region=(variant == "region"), limits=(variant != "nolimits"), n_nodes=70)
    env = engine_env(engine, wl)
    got = env.expand_lists(wl.nodes)
    env.close()
    ref = oracle_lib.expand(oracle_env(wl), wl.nodes, threads=8)
    rtol = YAW_COST_RTOL if control & 0x10 else 0.0
    assert_lists_equal(got, ref, wl.n_nodes, wl.U.shape[0], cost_rtol=rtol,
                       what="lists dim%d ctrl0x%x %s" % (dim, control, variant))


@pytest.mark.parametrize("name,scale,n_nodes", [("C2", 0.25, 1500), ("C3", 0.25, 700), ("C4", 0.125, 300),
                                                 ("C5", 0.2, 512)])
def test_lists_baseline_configs_scaled(engine, oracle_lib, name, scale, n_nodes):
    wl = engine.workloads.make(name, scale=scale, n_nodes=n_nodes)
    env = engine_env(engine, wl)
    got = env.expand_lists(wl.nodes)
    env.close()
    ref = oracle_lib.expand(oracle_env(wl), wl.nodes, threads=8)
    rtol = YAW_COST_RTOL if wl.control & 0x10 else 0.0
    assert_lists_equal(got, ref, wl.n_nodes, wl.U.shape[0], cost_rtol=rtol, what="lists " + name)


_GOLDEN = list(golden_cases())


@pytest.mark.parametrize("name,case,exp", _GOLDEN, ids=[c[0] for c in _GOLDEN])
def test_lists_reproduce_reference_golden_vectors(engine, name, case, exp):
    env = engine_env_from_case(engine, case)
    got = env.expand_lists(case["nodes"])
    env.close()
    rtol = YAW_COST_RTOL if case["control"] & 0x10 else 0.0
    assert_lists_equal(got, exp, case["nodes"].shape[1], case["U"].shape[0], cost_rtol=rtol, what="lists " + name)


def _random_controls(dim, n, seed, n_distinct):
    """A control table that is NOT a Cartesian grid: n rows drawn from n_distinct values per axis."""
    rng = np.random.default_rng(seed)
    vals = np.linspace(-1.0, 1.0, n_distinct) if n_distinct <= 9 else rng.uniform(-1.0, 1.0, size=n_distinct)
    U = rng.choice(vals, size=(n, dim))
    if n >= n_distinct:
        U[:n_distinct, 0] = vals  # every value in use on the first axis
    return U


@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("control", [0x01, 0x03, 0x07, 0x0F])
@pytest.mark.parametrize("variant", ["plain", "region"])
def test_lists_routes_agree(engine, oracle_lib, dim, control, variant):
    """The three kernels behind mplx_expand_lists (factorised GRID, general TILE,
    DENSE + compaction) against the oracle on the same inputs."""
    wl = _small_world(engine, dim, control, seed=900 * dim + control, region=(variant == "region"), n_nodes=75)
    ref = oracle_lib.expand(oracle_env(wl), wl.nodes, threads=8)
    env = engine_env(engine, wl)
    nU = wl.U.shape[0]
    for route, stride in (("grid", None), ("tile", nU + 5), ("dense", (nU + 15) & ~15), ("grid", nU + 16)):
        env.set_lists_route(route)
        got = env.expand_lists(wl.nodes, stride=stride)  # node_stride: default nU, odd, and line-aligned
        assert env.last_lists_route() == route and got["stride"] == (stride or nU)
        assert_lists_equal(got, ref, wl.n_nodes, wl.U.shape[0], what="route %s dim%d ctrl0x%x %s" % (
            route, dim, control, variant))
    env.set_lists_route("auto")
    env.expand_lists(wl.nodes)
    assert env.last_lists_route() == "grid"
    env.close()


@pytest.mark.parametrize("dim,n_controls,n_distinct,expect", [(3, 200, 7, "grid"), (3, 200, 40, "tile"),
                                                             (2, 37, 16, "grid"), (2, 37, 17, "tile"),
                                                             (3, 1000, 12, "grid"), (3, 1, 1, "grid")])
def test_lists_irregular_control_tables(engine, oracle_lib, dim, n_controls, n_distinct, expect):
    """Control tables that are not Cartesian products: up to 16 distinct values per
    axis are factorised, more fall back to the general tiled kernel; duplicates
    and a single control are legal."""
    wl = _small_world(engine, dim, 0x03, seed=1234 + n_controls, n_nodes=90)
    wl.U = _random_controls(dim, n_controls, 4321 + n_distinct, n_distinct)
    if n_distinct > 1:
        assert max(np.unique(wl.U[:, i]).size for i in range(dim)) == n_distinct
    ref = oracle_lib.expand(oracle_env(wl), wl.nodes, threads=8)
    env = engine_env(engine, wl)
    got = env.expand_lists(wl.nodes)
    # a frontier of at most 512 nodes with 512+ controls goes to the tiled kernel (latency of one node)
    assert env.last_lists_route() == ("tile" if n_controls >= 512 else expect)
    assert_lists_equal(got, ref, wl.n_nodes, wl.U.shape[0], what="irregular U %dx%d" % (n_controls, n_distinct))
    if expect == "grid" and n_controls >= 512:
        big = np.tile(wl.nodes, (1, 6))  # 540 nodes: the factorised kernel again
        got = env.expand_lists(big)
        assert env.last_lists_route() == "grid"
        ref6 = {k: (np.tile(v, (1, 6)) if v.ndim == 2 else np.tile(v, 6)) for k, v in ref.items() if hasattr(v, "ndim")}
        assert_lists_equal(got, ref6, big.shape[1], wl.U.shape[0], what="irregular U %dx%d x6" % (n_controls, n_distinct))
    env.close()


@pytest.mark.parametrize("rmax,boxcap,dbg", [("1", None, None), (None, "8", None), ("2", "40", None), (None, None, "64")])
def test_grid_kernel_small_lds_budgets(engine, oracle_lib, monkeypatch, rmax, boxcap, dbg):
    """The factorised kernel with a starved LDS budget (tuning overrides, read when the context is created): one row of
    cell codes per pass forces the multi-pass path, a tiny box forces sampling straight from the
    blocked-bit map.  Results must not change."""
    if rmax:
        monkeypatch.setenv("MPLX_GRID_RMAX", rmax)
    if boxcap:
        monkeypatch.setenv("MPLX_GRID_BOXCAP", boxcap)
    if dbg:  # bit 64: every pass samples by direct evaluation (the path taken when a cell code leaves its range)
        monkeypatch.setenv("MPLX_TILE_DBG", dbg)
    for dim, control, region in ((3, 0x03, False), (2, 0x07, True), (3, 0x0F, True)):
        wl = _small_world(engine, dim, control, seed=3100 + dim + control, region=region, n_nodes=80)
        ref = oracle_lib.expand(oracle_env(wl), wl.nodes, threads=8)
        env = engine_env(engine, wl)
        env.set_lists_route("grid")
        got = env.expand_lists(wl.nodes)
        env.close()
        assert_lists_equal(got, ref, wl.n_nodes, wl.U.shape[0], what="rmax=%s boxcap=%s dim%d ctrl0x%x" % (
            rmax, boxcap, dim, control))


@pytest.mark.parametrize("dim,control,potential,gradient", [(2, 0x1F, False, 0.0), (3, 0x1F, True, 0.0), (3, 0x0F, True, 0.3),
                                                            (2, 0x0F, True, 0.0), (3, 0x1F, False, 0.0), (2, 0x1F, True, 0.3)])
def test_direct_evaluation_covers_potential_and_heading_costs(engine, oracle_lib, monkeypatch, dim, control, potential, gradient):
    """MPLX_TILE_DBG bit 64 sends every pass of a SNP instantiation through the direct-evaluation path (taken in production
    only when a SNP primitive's cell code leaves its range, primitive.h:158-159; compiled into the K = 4 instantiations
    alone: the velocity maxima of K <= 3 are exact): it must do everything the row paths do -- potential
    values + search region, the potential / |vel| cost (env_map.h:113-118), the heading cost (:121-129) -- which is what
    lets SNP x yaw and SNP on a potential map run on the factorised kernel at all."""
    monkeypatch.setenv("MPLX_TILE_DBG", "64")
    wl = _small_world(engine, dim, control, seed=3300 + dim + control, potential=potential, region=potential, n_nodes=90, edge=56)
    if potential:
        wl.params["gradient_weight"] = gradient
    if control & 0x10:
        rng = np.random.default_rng(dim + control)
        along = np.arctan2(wl.nodes[dim + 1], wl.nodes[dim]) + rng.uniform(-0.4, 0.4, size=wl.n_nodes)
        wl.nodes[4 * dim] = np.where(np.arange(wl.n_nodes) % 5 == 0, wl.nodes[4 * dim], along)
    ref = oracle_lib.expand(oracle_env(wl), wl.nodes, threads=8)
    assert np.count_nonzero(ref["status"] == 1) > 30 and np.count_nonzero(ref["status"] == 2) > 5
    env = engine_env(engine, wl)
    env.set_lists_route("grid")
    got = env.expand_lists(wl.nodes)
    assert env.last_lists_route() == "grid"
    env.close()
    assert_lists_equal(got, ref, wl.n_nodes, wl.U.shape[0], cost_rtol=1e-6 if control & 0x10 else 0.0,
                       what="direct evaluation dim%d ctrl0x%x pot=%s" % (dim, control, potential))


@pytest.mark.parametrize("nosat", [False, True])
def test_free_box_shortcut_and_full_sampling_agree(engine, oracle_lib, monkeypatch, nosat):
    """Nodes whose whole reach box is free skip the sample loops (summed-area table look-up); a map with a
    large free region makes most nodes take the shortcut, MPLX_GRID_NOSAT makes none take it.  Same lists,
    same iteration counts, in 2D and 3D; list tails past count[k] are never read by the comparison."""
    if nosat:
        monkeypatch.setenv("MPLX_GRID_NOSAT", "1")
    for dim, control in ((3, 0x03), (2, 0x03), (3, 0x07), (2, 0x01)):
        wl = _small_world(engine, dim, control, seed=5200 + dim + control, n_nodes=150, edge=72)
        g = wl.grid.reshape([72] * dim)
        g[tuple([slice(0, 44)] * dim)] = 0          # a free corner region: nodes there see no obstacle
        wl.nodes[:dim, :80] = np.round(np.random.default_rng(dim).uniform(1.6, 2.8, size=(dim, 80)), 2)
        ref = oracle_lib.expand(oracle_env(wl), wl.nodes, threads=8)
        assert np.count_nonzero(ref["status"] == 2) > 50  # and the rest still has blocked edges
        env = engine_env(engine, wl)
        env.set_lists_route("grid")
        got = env.expand_lists(wl.nodes, stride=(wl.U.shape[0] + 31) & ~31)
        env.close()
        assert_lists_equal(got, ref, wl.n_nodes, wl.U.shape[0], what="sat=%s dim%d ctrl0x%x" % (not nosat, dim, control))


def test_wavefront_frontier_expands_like_the_oracle(engine, oracle_lib):
    """The open list of an eps = 0 search as frontier (workloads.wavefront_frontier): clustered nodes with
    lattice velocities and many shared successors."""
    wl = engine.workloads.make("C4", scale=0.125, n_nodes=64)
    wl.nodes = engine.workloads.wavefront_frontier(wl, 400)
    assert wl.nodes.shape == (14, 400) and np.unique(wl.nodes[:3].T, axis=0).shape[0] > 50
    ref = oracle_lib.expand(oracle_env(wl), wl.nodes, threads=8)
    env = engine_env(engine, wl)
    got = env.expand_lists(wl.nodes)
    env.close()
    assert_lists_equal(got, ref, 400, wl.U.shape[0], what="wavefront frontier")


def test_node_stride_smaller_than_the_control_table_is_rejected(engine):
    wl = _small_world(engine, 2, 0x03, seed=6, n_nodes=8)
    env = engine_env(engine, wl)
    with pytest.raises(engine._abi.MplxError) as e:
        env.expand_lists(wl.nodes, stride=wl.U.shape[0] - 1)
    assert e.value.code == engine._abi.ERR_ARG
    env.close()


def test_control_tables_larger_than_a_workgroup_tile(engine, oracle_lib):
    """|U| = 11^3 = 1331 > 1024 in nested-loop order: the lexicographic kernel (round 4; up to 8 192 controls); the same
    table shuffled: neither list kernel tiles it, the dense kernel + compaction serves it;
    |U| = 1024 = 32 x 32 (2D) is the largest table the general factorised kernel takes."""
    W = engine.workloads
    wl = _small_world(engine, 3, 0x03, seed=8100, n_nodes=40)
    wl.U = W.grid_controls(np.linspace(-1.0, 1.0, 11), 3)
    ref = oracle_lib.expand(oracle_env(wl), wl.nodes, threads=8)
    env = engine_env(engine, wl)
    got = env.expand_lists(wl.nodes)
    assert env.last_lists_route() == "grid" and env.last_grid_kernel() == "lex"
    env.close()
    assert_lists_equal(got, ref, wl.n_nodes, 1331, what="|U| = 1331")
    wl.U = np.ascontiguousarray(wl.U[np.random.default_rng(8103).permutation(1331)])
    ref = oracle_lib.expand(oracle_env(wl), wl.nodes, threads=8)
    env = engine_env(engine, wl)
    got = env.expand_lists(wl.nodes)
    assert env.last_lists_route() == "dense"
    env.close()
    assert_lists_equal(got, ref, wl.n_nodes, 1331, what="|U| = 1331, shuffled")
    wl2 = _small_world(engine, 2, 0x03, seed=8101, n_nodes=40)
    rng = np.random.default_rng(8102)
    vals = np.linspace(-1.0, 1.0, 16)
    wl2.U = np.stack([rng.choice(vals, 1024), rng.choice(vals, 1024)], axis=1)
    wl2.U[:16, 0] = vals
    wl2.U[:16, 1] = vals
    ref2 = oracle_lib.expand(oracle_env(wl2), wl2.nodes, threads=8)
    env = engine_env(engine, wl2)
    env.set_lists_route("grid")  # (40 nodes: the automatic choice would be the tiled kernel, lower latency)
    got2 = env.expand_lists(wl2.nodes)
    assert env.last_lists_route() == "grid"
    env.close()
    assert_lists_equal(got2, ref2, wl2.n_nodes, 1024, what="|U| = 1024")


@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("control", [0x01, 0x03, 0x07, 0x0F])
@pytest.mark.parametrize("nosat", [False, True])
def test_potential_map_on_the_factorised_kernel(engine, oracle_lib, monkeypatch, dim, control, nosat):
    """Potential maps with gradient_weight == 0 (env_map.h:113-118 reduces to dt * w_p * value per sample) stay on
    the factorised kernel: per-sample values from the int8 map, costs accumulated in the reference's order (bit-exact),
    the free-box shortcut now meaning "zero potential in the whole reach box".  With a search region on top, and with
    gradient_weight != 0 (per-sample |vel| from velocity rows)."""
    if nosat:
        monkeypatch.setenv("MPLX_GRID_NOSAT", "1")
    wl = _small_world(engine, dim, control, seed=7100 + 10 * dim + control, potential=True, region=True, n_nodes=120,
                      edge=64)
    wl.params["gradient_weight"] = 0.0
    g = wl.potential.reshape([64] * dim)
    g[tuple([slice(8, 30)] * dim)] = 0              # a zero-potential pocket inside the region
    wl.grid = wl.potential
    wl.nodes[:dim, :40] = np.round(np.random.default_rng(dim).uniform(1.4, 2.4, size=(dim, 40)), 2)
    ref = oracle_lib.expand(oracle_env(wl), wl.nodes, threads=8)
    fin = np.isfinite(ref["cost"]) & (ref["status"] >= 1)
    assert np.count_nonzero(ref["status"] == 2) > 50 and np.count_nonzero(fin) > (80 if control == 0x0F else 200)
    env = engine_env(engine, wl)
    got = env.expand_lists(wl.nodes)
    assert env.last_lists_route() == "grid"
    assert_lists_equal(got, ref, wl.n_nodes, wl.U.shape[0], what="potential on grid dim%d ctrl0x%x" % (dim, control))
    # gradient_weight != 0 (env_map.h:116: + gradient_weight * |vel| per sample inside the field): the velocity rows of
    # every axis are built next to the cell rows; still the factorised kernel, still bit-exact
    env.set_gradient_weight(0.25)
    got_g = env.expand_lists(wl.nodes)
    assert env.last_lists_route() == "grid"
    ref_g = oracle_lib.expand(oracle_env(wl, gradient_weight=0.25), wl.nodes, threads=8)
    assert np.any(ref_g["cost"][fin] != ref["cost"][fin])  # the term really contributes
    assert_lists_equal(got_g, ref_g, wl.n_nodes, wl.U.shape[0], what="potential + gradient on grid dim%d ctrl0x%x" % (dim, control))
    # switching back to a plain occupancy query on the same context rebuilds the blocked bits
    env.set_gradient_weight(0.0)
    env.set_potential_map(None)
    got2 = env.expand_lists(wl.nodes)
    assert env.last_lists_route() == "grid"
    wl2 = engine.workloads.Workload("small", dim, control, wl.grid, [0.0] * dim, wl.res, wl.U, wl.nodes,
                                    {k: v for k, v in wl.params.items() if "weight" not in k}, potential=None,
                                    region=wl.region)
    ref2 = oracle_lib.expand(oracle_env(wl2), wl.nodes, threads=8)
    env.close()
    assert_lists_equal(got2, ref2, wl.n_nodes, wl.U.shape[0], what="potential removed dim%d ctrl0x%x" % (dim, control))


@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("control", [0x11, 0x13, 0x17, 0x1F])
@pytest.mark.parametrize("variant", ["heading_cost", "no_cost", "no_limit", "potential", "potential_gradient", "region"])
def test_yaw_controls_on_the_factorised_kernel(engine, oracle_lib, dim, control, variant):
    """Yaw controls (VELxYAW, ACCxYAW, JRKxYAW) with the yaw rate as a fourth factor of the control table: heading
    limit at both ends (primitive.h:504-525), yaw in the lattice hash and the successor state, per-sample heading
    cost (env_map.h:121-129) alone and on top of a potential map.  Same lists from the factorised kernel and from
    the lane-per-pair kernel, both against the oracle; cos / sin differ from glibc's in the last place, hence the
    cost tolerance."""
    wl = _small_world(engine, dim, control, seed=8100 + 10 * dim + control, potential=variant.startswith("potential"),
                      region=(variant == "region"), n_nodes=110, edge=56)
    if control != 0x11:
        # most headings roughly along the velocity, or the heading limit leaves almost nothing to traverse
        rng = np.random.default_rng(dim + control)
        along = np.arctan2(wl.nodes[dim + 1], wl.nodes[dim]) + rng.uniform(-0.4, 0.4, size=wl.n_nodes)
        keep = np.arange(wl.n_nodes) % 5 == 0
        wl.nodes[4 * dim] = np.where(keep, wl.nodes[4 * dim], along)
    if variant == "potential":
        wl.params["gradient_weight"] = 0.0
    if variant == "potential_gradient":
        wl.params["gradient_weight"] = 0.3
    if variant == "no_cost":
        wl.params["wyaw"] = 0.0
    if variant == "no_limit":
        wl.params["yaw_max"] = 0.0
    ref = oracle_lib.expand(oracle_env(wl), wl.nodes, threads=8)
    assert np.count_nonzero(ref["status"] == 1) > 100
    assert variant == "no_limit" or np.count_nonzero(ref["status"] == 3) > 100  # the heading limit rejects some
    env = engine_env(engine, wl)
    for route in ("grid", "dense"):
        env.set_lists_route(route)
        got = env.expand_lists(wl.nodes)
        assert env.last_lists_route() == route
        assert_lists_equal(got, ref, wl.n_nodes, wl.U.shape[0], cost_rtol=YAW_COST_RTOL,
                           what="yaw %s route %s dim%d ctrl0x%x" % (variant, route, dim, control))
    env.close()


def test_yaw_grid_and_dense_kernels_agree_bit_for_bit(engine):
    """Same device cos / sin on the same arguments in both kernels: the two routes must agree exactly, costs
    included (the tolerance above is only for glibc against the device library)."""
    wl = engine.workloads.make("C5", scale=0.25, n_nodes=2000)
    env = engine_env(engine, wl)
    out = {}
    for route in ("grid", "dense"):
        env.set_lists_route(route)
        out[route] = env.expand_lists(wl.nodes)
    env.close()
    a, b = out["grid"], out["dense"]
    assert np.array_equal(a["count"], b["count"]) and a["stride"] == b["stride"] and a["count"].sum() > 5000
    live = (np.arange(a["stride"])[None, :] < a["count"][:, None]).ravel()  # list tails are unspecified
    for name in ("action", "hash", "iters", "cost"):
        assert np.array_equal(a[name][live], b[name][live]), name
    assert np.array_equal(a["state"][:, live].view(np.uint64), b["state"][:, live].view(np.uint64))
    assert np.isfinite(a["cost"][live]).sum() > 1000


def test_host_pointer_lists_pipelined_copy_back(engine):
    """mplx_expand_lists with host arrays: the used list prefixes are packed on the device, copied in chunks through
    pinned buffers and scattered by helper threads (lists_copy_api.cpp).  Several chunks and threads here; the
    result must equal the HBM-resident lists entry by entry, also when the caller's arrays are reused."""
    wl = engine.workloads.make("C4", scale=0.125, n_nodes=1800)
    nU = wl.U.shape[0]
    env = engine_env(engine, wl)
    fr = env.upload_frontier(wl.nodes)
    lists = env.alloc_lists(wl.n_nodes, want_state=True, want_iters=True)
    env.expand_lists_resident(fr, lists)
    env.synchronize()
    ref = lists.download()
    lists.free()
    fr.free()
    assert int(ref["count"].sum()) * 136 > 2 * (32 << 20)  # three 32-MiB chunks
    live = (np.arange(ref["stride"])[None, :] < ref["count"][:, None]).ravel()
    got = env.expand_lists(wl.nodes, stride=ref["stride"])
    for rep in range(2):
        assert np.array_equal(got["count"], ref["count"])
        for name in ("action", "hash", "iters", "cost"):
            assert np.array_equal(got[name][live].view(np.uint64 if got[name].itemsize == 8 else np.int32),
                                  ref[name][live].view(np.uint64 if ref[name].itemsize == 8 else np.int32)), name
        assert np.array_equal(got["state"][:, live].view(np.uint64), ref["state"][:, live].view(np.uint64))
        for name in ("action", "hash", "iters", "cost"):
            got[name][:] = 0
        got["state"][:] = 0
        got = env.expand_lists(wl.nodes, stride=ref["stride"], out=got)
    # a frontier in which most nodes emit nothing (empty chunks, zero-length prefixes)
    far = wl.nodes.copy()
    far[:3, 5:] = -50.0  # outside the map: every primitive is blocked at its first sample, but still emitted
    far[3:6, 5:] = 9.0   # and beyond v_max: nothing emitted at all
    got = env.expand_lists(far)
    assert got["count"][5:].sum() == 0 and np.array_equal(got["count"][:5], ref["count"][:5])
    env.close()


@pytest.mark.parametrize("name", ["C2", "C3", "C4", "C5"])
def test_lexicographic_enumeration_equals_the_table_walk(engine, monkeypatch, name):
    """Nested-loop control tables let phase A of the factorised kernel enumerate only the combinations of axis
    entries that pass the limits (A.ulex); MPLX_GRID_NOLEX walks the whole table instead.  Same lists."""
    wl = engine.workloads.make(name, scale=0.125, n_nodes=600)
    out = []
    for nolex in (False, True):
        if nolex:
            monkeypatch.setenv("MPLX_GRID_NOLEX", "1")
        env = engine_env(engine, wl)
        out.append(env.expand_lists(wl.nodes))
        assert env.last_lists_route() == "grid"
        env.close()
    a, b = out
    assert np.array_equal(a["count"], b["count"]) and a["count"].sum() > 1000
    live = (np.arange(a["stride"])[None, :] < a["count"][:, None]).ravel()
    for key in ("action", "hash", "iters", "cost"):
        assert np.array_equal(a[key][live], b[key][live]), key
    assert np.array_equal(a["state"][:, live].view(np.uint64), b["state"][:, live].view(np.uint64))
    # a shuffled copy of the same table is not in nested-loop order: the table walk serves it, same successors
    perm = np.random.default_rng(3).permutation(wl.U.shape[0])
    wl.U = np.ascontiguousarray(wl.U[perm])
    monkeypatch.delenv("MPLX_GRID_NOLEX", raising=False)
    env = engine_env(engine, wl)
    c = env.expand_lists(wl.nodes)
    env.close()
    assert np.array_equal(c["count"], a["count"])
    S = a["stride"]
    for k in (0, 17, 333):
        n = a["count"][k]
        assert np.array_equal(np.sort(perm[c["action"][k * S:k * S + n]]), a["action"][k * S:k * S + n])
        order = np.argsort(perm[c["action"][k * S:k * S + n]])
        assert np.array_equal(c["hash"][k * S:k * S + n][order], a["hash"][k * S:k * S + n])


def test_forcing_a_route_outside_its_scope_fails_loudly(engine):
    # 17 distinct control values per axis in an order that is NOT the nested loops': past the general factorised kernel's
    # 16 (the lexicographic kernel takes up to 32, in nested-loop order only) -- the workgroup-per-node kernel covers it
    wl = _small_world(engine, 2, 0x03, seed=5, n_nodes=8)
    wl.U = engine.workloads.grid_controls(list(np.linspace(-1.0, 1.0, 17)), 2)
    wl.U = np.ascontiguousarray(wl.U[np.random.default_rng(6).permutation(wl.U.shape[0])])
    env = engine_env(engine, wl)
    env.set_lists_route("grid")
    with pytest.raises(engine._abi.MplxError) as e:
        env.expand_lists(wl.nodes)
    assert e.value.code == engine._abi.ERR_STATE
    env.set_lists_route("auto")
    env.expand_lists(wl.nodes)
    assert env.last_lists_route() == "tile"
    env.close()
    # SNP x YAW: the lane-per-pair kernel's alone until round 3, on the factorised kernel since
    wl = _small_world(engine, 2, 0x1F, seed=5, n_nodes=8)
    env = engine_env(engine, wl)
    env.expand_lists(wl.nodes)
    assert env.last_lists_route() == "grid"
    env.close()


def test_lists_ragged_tiles_and_resident_buffers(engine, oracle_lib):
    """Frontier sizes around the nodes-per-workgroup tiling, and the HBM-resident call."""
    wl = engine.workloads.make("C2", scale=0.25, n_nodes=200)  # |U| = 25 -> 32 nodes per workgroup
    env = engine_env(engine, wl)
    for n in (1, 31, 32, 33, 64, 199, 200):
        sub = np.ascontiguousarray(wl.nodes[:, :n])
        got = env.expand_lists(sub)
        ref = oracle_lib.expand(oracle_env(wl), sub, threads=4)
        assert_lists_equal(got, ref, n, wl.U.shape[0], what="ragged n=%d" % n)
    fr = env.upload_frontier(wl.nodes)
    lists = env.alloc_lists(wl.n_nodes, want_state=True, want_iters=True)
    env.expand_lists_resident(fr, lists)
    env.synchronize()
    got = lists.download()
    ref = oracle_lib.expand(oracle_env(wl), wl.nodes, threads=4)
    assert_lists_equal(got, ref, wl.n_nodes, wl.U.shape[0], what="resident")
    lists.free()
    fr.free()
    env.close()


def test_hoisted_division_equals_true_division(engine):
    """The mul+fma+fma quotient of the tiled kernel must be the correctly rounded
    quotient: the cell index and sample count computed with it are checked against
    the dense kernel (true `/`) on a frontier that sits on cell faces."""
    W = engine.workloads
    grid = W.box_map([64, 64, 64], 0.1, 0.15, 77)
    U = W.grid_controls(np.arange(-2, 2.01, 0.5), 3)
    nodes = W.random_frontier(grid, [0, 0, 0], 0.1, 600, 78, 0x03, 2.0, 0.5)
    nodes[:3] = np.round(nodes[:3], 1)  # positions exactly on multiples of the resolution
    for res, org in ((0.1, [0, 0, 0]), (0.05, [-0.3, 0.1, 0.0]), (0.3, [0.05, 0, 0])):
        env = engine.EnvMap(3)
        env.setMap(org, [64, 64, 64], grid, res)
        env.set_control(0x03)
        env.set_u(U)
        env.set_v_max(2.0 if res >= 0.1 else 1.5)
        dense = env.expand(nodes)
        lists = env.expand_lists(nodes)
        env.close()
        assert_lists_equal(lists, dense, 600, U.shape[0], what="res %g" % res)


@pytest.mark.parametrize("seed", range(16))
def test_irregular_parameters_all_routes(engine, oracle_lib, seed):
    """Nothing in the kernels may depend on the round numbers the BASELINE configurations use: odd resolutions and
    durations, map origins off the lattice, velocities that are not lattice values, control values like 1/3, odd
    weights.  (The hoisted divisions by res / 0.01 / 0.1, the accumulated sample times and the per-axis sample
    counts are the places where a shortcut would show.)  Every route against the oracle, bit for bit."""
    wl, control, pot = odd_world(engine, seed, 90)
    n_nodes, U = wl.n_nodes, wl.U
    ref = oracle_lib.expand(oracle_env(wl), wl.nodes, threads=8)
    st = ref["status"]
    assert np.count_nonzero(st == 1) > 30 and np.count_nonzero(st == 2) > 10, np.bincount(st, minlength=4)
    env = engine_env(engine, wl)
    rtol = YAW_COST_RTOL if control & 0x10 else 0.0
    routes = ["auto", "dense"] + (["tile"] if not (control & 0x10) and pot is None else [])
    seen = set()
    for route in routes:
        env.set_lists_route(route)
        got = env.expand_lists(wl.nodes)
        seen.add(env.last_lists_route())
        assert_lists_equal(got, ref, n_nodes, U.shape[0], cost_rtol=rtol, what="odd parameters seed %d route %s (%s)" % (
            seed, route, env.last_lists_route()))
    env.close()
    assert "dense" in seen and (control == 0x1F or "grid" in seen or control == 0x0F and pot is not None)


@pytest.mark.parametrize("dims", [[7, 5], [33, 3], [1, 40], [9, 6, 4], [65, 2, 3], [3, 3, 70]])
@pytest.mark.parametrize("control", [0x01, 0x03, 0x13])
def test_tiny_and_lopsided_maps(engine, oracle_lib, dims, control):
    """Maps narrower than one 32-cell word of the blocked-bit rows, one cell wide, or long in one axis only: every
    primitive leaves the map somewhere, boxes are clipped on all sides, the summed-area look-ups sit on the border."""
    W = engine.workloads
    dim = len(dims)
    rng = np.random.default_rng(sum(dims) + control)
    res = 0.1
    cells = (rng.uniform(size=dims[::-1]) < 0.08).astype(np.int8) * 100
    n_nodes = 80
    nodes = np.zeros((4 * dim + 2, n_nodes))
    for i in range(dim):
        nodes[i] = np.round(rng.uniform(-0.15, dims[i] * res + 0.15, size=n_nodes), 2)
    if control & 0x02:
        nodes[dim:2 * dim] = rng.choice([-0.5, 0.0, 0.5], size=(dim, n_nodes))
    if control & 0x10:
        nodes[4 * dim] = np.arctan2(nodes[dim + 1], nodes[dim]) + rng.uniform(-0.3, 0.3, size=n_nodes)
    U = W.grid_controls([-1.0, -0.5, 0.0, 0.5, 1.0] if dim == 2 else [-1.0, 0.0, 1.0], dim,
                        yaw_rates=[-0.5, 0.0, 0.5] if control & 0x10 else None)
    params = {"v_max": 1.5, "dt": 0.5}
    if control & 0x10:
        params["yaw_max"] = 0.8
    wl = W.Workload("tiny", dim, control, cells, [0.0] * dim, res, U, nodes, params)
    ref = oracle_lib.expand(oracle_env(wl), wl.nodes, threads=4)
    assert np.count_nonzero(ref["status"] == 2) > 20
    env = engine_env(engine, wl)
    for route in ("grid", "dense") + (() if control & 0x10 else ("tile",)):
        env.set_lists_route(route)
        got = env.expand_lists(wl.nodes)
        assert env.last_lists_route() == route
        assert_lists_equal(got, ref, n_nodes, U.shape[0], cost_rtol=YAW_COST_RTOL if control & 0x10 else 0.0,
                           what="tiny map %s ctrl0x%x route %s" % (dims, control, route))
    env.close()


@pytest.mark.parametrize("name,blocks,chunk", [("C2", "8", "1"), ("C2", "8", "3"), ("C2", "5", "8"), ("C5", "16", "2"),
                                               ("C3", "7", "1"), ("C2", "8", "static"), ("C2", "8", "blocked"),
                                               ("C5", "70", "blocked3")])
def test_dynamic_node_assignment_small_grids_and_chunks(engine, oracle_lib, monkeypatch, name, blocks, chunk):
    """The factorised kernel's waves claim nodes from counters (GridArgs::work): with few workgroups every wave walks
    many chunks, chunk sizes that do not divide the frontier leave ragged last chunks, and consecutive launches swap
    the two counter sets -- the lists must be the oracle's every time (and the same with static striding)."""
    monkeypatch.setenv("MPLX_GRID_BLOCKS", blocks)
    if chunk == "static":
        monkeypatch.setenv("MPLX_GRID_STATIC", "1")
    elif chunk.startswith("blocked"):  # one contiguous block of chunks per counter instead of the round-robin deal
        monkeypatch.setenv("MPLX_GRID_BLOCKED", "1")
        monkeypatch.setenv("MPLX_GRID_CHUNK", chunk[7:] or "1")
    else:
        monkeypatch.setenv("MPLX_GRID_CHUNK", chunk)
    wl = engine.workloads.make(name, scale=0.25, n_nodes=2311)  # a prime: no chunk size divides it
    nU = wl.U.shape[0]
    env = engine_env(engine, wl)
    ref = oracle_lib.expand(oracle_env(wl), wl.nodes, threads=8)
    rtol = YAW_COST_RTOL if wl.control & 0x10 else 0.0
    fr = env.upload_frontier(wl.nodes)
    lists = env.alloc_lists(wl.n_nodes, want_state=True, want_iters=True)
    for launch in range(4):  # both counter sets, twice
        env.expand_lists_resident(fr, lists)
        env.synchronize()
        assert env.last_lists_route() == "grid"
        assert_lists_equal(lists.download(), ref, wl.n_nodes, nU, cost_rtol=rtol,
                           what="%s blocks %s chunk %s launch %d" % (name, blocks, chunk, launch))
    # a shorter frontier in the same buffers right after: fewer claims than the launch before
    env.expand_lists_resident(fr, lists, n_nodes=700)
    env.synchronize()
    got = lists.download()
    got["count"] = got["count"][:700]
    sub = {k: (v[:700 * nU] if k != "state" else v[:, :700 * nU]) for k, v in ref.items() if k != "stats"}
    assert_lists_equal(got, sub, 700, nU, cost_rtol=rtol, what="%s shorter frontier" % name)
    lists.free()
    fr.free()
    env.close()


def test_state_rows_with_padding_between_them(engine, oracle_lib):
    """mplx_succ_lists::state_stride is the caller's: rows further apart than n_nodes * node_stride (Lists(state_pad))
    hold the same lists (used by profiles/micro/c4_skew.py)."""
    wl = engine.workloads.make("C2", scale=0.25, n_nodes=300)
    nU = wl.U.shape[0]
    env = engine_env(engine, wl)
    ref = oracle_lib.expand(oracle_env(wl), wl.nodes, threads=8)
    fr = env.upload_frontier(wl.nodes)
    for pad in (0, 32, 2080):
        for route in ("grid", "tile", "dense"):
            env.set_lists_route(route)
            lists = env.alloc_lists(wl.n_nodes, want_state=True, want_iters=True, state_pad=pad)
            assert lists.c_struct().state_stride == lists.n_slots + pad
            env.expand_lists_resident(fr, lists)
            env.synchronize()
            assert_lists_equal(lists.download(), ref, wl.n_nodes, nU, what="state_pad %d route %s" % (pad, route))
            half = lists.download_nodes(100, 200)
            full = lists.download()
            S = lists.stride
            assert np.array_equal(half["state"], full["state"][:, 100 * S:200 * S], equal_nan=True)
            lists.free()
    fr.free()
    env.close()
