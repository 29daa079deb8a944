"""Map preprocessing: MapPlanner::updatePotentialMap and
MapPlanner::setSearchRegion.

CPU: the oracle's restatement against the reference's own MapPlanner (compiled
from /root/reference by oracle/Makefile `ref`, where that tree exists) and
against the committed golden fixture generated from it.
GPU: the device kernels (map_prep_kernel.hip through mplx_update_potential_map /
mplx_set_search_region_path) against the oracle, bit for bit, and the expansion
that consumes their outputs."""
import os

import numpy as np
import pytest

from helpers import assert_slots_equal
from oracle import oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "map_prep_golden.npz")


def _cases():
    """(name, dim, grid, map_dim, origin, res, potential args, region args)"""
    import motion_primitive_library_amd.workloads as W
    out = []
    for dim, edge, seed in ((2, 96, 5), (3, 40, 6), (2, 61, 7), (3, 33, 8)):
        rng = np.random.default_rng(seed)
        md = [edge, edge - 7, edge - 3][:dim]
        grid = W.box_map(md, 0.1, 0.12, seed, side_m=(0.3, 1.0)).copy()
        flat = grid.ravel()
        flat[rng.integers(0, flat.size, 40)] = -1  # unknown cells stay unknown unless a mask entry reaches them
        flat[rng.integers(0, flat.size, 40)] = 37  # leftovers of an earlier potential field count as sources
        org = [0.05, -0.4, 0.2][:dim]
        centre = [org[i] + md[i] * 0.05 for i in range(dim)]
        pots = [([0.5] * dim, None, 1.0), ([0.7, 0.7, 0.3][:dim], [1.5] * dim, 2.0), ([0.35] * dim, [0.8, 0.4, 0.6][:dim], 0.5),
                ([0.0] * dim, None, 1.0), ([0.45, 0.45, 0.0][:dim], None, 1.0)]
        path = rng.uniform(0.3, min(md) * 0.1 - 0.3, size=(6, dim)) + np.array(org)
        path[2] = path[1]                      # a zero-length segment
        path[5] = np.array(org) + np.array(md) * 0.1 + 0.4  # ends outside the map
        regs = [([0.3, 0.2, 0.4][:dim], False), ([0.3, 0.2, 0.4][:dim], True), ([0.0] * dim, False)]
        out.append(("d%d_e%d" % (dim, edge), dim, grid, md, org, 0.1, centre, pots, path, regs))
    return out


CASES = _cases()


@pytest.mark.skipif(not os.path.exists(O.REF_PLANNER_SO), reason="reference build (oracle/_ref) not present")
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_restatement_matches_the_reference_map_planner(case):
    name, dim, grid, md, org, res, centre, pots, path, regs = case
    for radius, rng_, pw in pots:
        a = O.update_potential_map(grid, md, org, res, centre, radius, rng_, pw)
        b = O.update_potential_map(grid, md, org, res, centre, radius, rng_, pw, ref=True)
        assert np.array_equal(a, b), "potential %s %s %s" % (radius, rng_, pw)
    for sr, dense in regs:
        a = O.search_region(md, org, res, path, sr, dense)
        b = O.search_region(md, org, res, path, sr, dense, ref=True)
        assert np.array_equal(a, b), "region %s dense=%s" % (sr, dense)


def test_restatement_matches_the_golden_fixture():
    """tests/golden/map_prep_golden.npz was produced by the reference's MapPlanner
    (tests/golden/make_map_prep_golden.py); it travels to machines without /root/reference."""
    z = np.load(GOLDEN)
    n = 0
    for name, dim, grid, md, org, res, centre, pots, path, regs in CASES:
        for k, (radius, rng_, pw) in enumerate(pots):
            a = O.update_potential_map(grid, md, org, res, centre, radius, rng_, pw)
            assert np.array_equal(a, z["%s/pot%d" % (name, k)]), (name, k)
            n += 1
        for k, (sr, dense) in enumerate(regs):
            a = O.search_region(md, org, res, path, sr, dense)
            assert np.array_equal(np.packbits(a), z["%s/reg%d" % (name, k)]), (name, k)
            n += 1
    assert n == 32


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_device_potential_map_is_bit_identical(engine, case):
    name, dim, grid, md, org, res, centre, pots, path, regs = case
    for radius, rng_, pw in pots:
        env = engine.EnvMap(dim)
        env.setMap(org, md, grid, res)
        got = env.updatePotentialMap(centre, radius, rng_, pw)
        env.close()
        want = O.update_potential_map(grid, md, org, res, centre, radius, rng_, pw)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, "%s radius %s range %s pow %s: %d cells differ, first %s got %s want %s" % (
            name, radius, rng_, pw, bad.size, bad[:5], got[bad[:5]], want[bad[:5]])


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_device_search_region_is_bit_identical(engine, case):
    name, dim, grid, md, org, res, centre, pots, path, regs = case
    env = engine.EnvMap(dim)
    env.setMap(org, md, grid, res)
    for sr, dense in regs:
        got = env.setSearchRegion(path, sr, dense)
        want = O.search_region(md, org, res, path, sr, dense)
        assert np.array_equal(got, want), "%s region %s dense=%s" % (name, sr, dense)
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dim", [2, 3])
def test_expansion_consumes_the_device_made_potential_and_region(engine, oracle_lib, dim):
    """End to end: potential field and tunnel made on the device, then get_succ on
    them, against the oracle fed with the oracle-made potential and region."""
    from test_gpu_parity import _small_world
    wl = _small_world(engine, dim, 0x03, seed=77 + dim, n_nodes=120)
    md, org, res = wl.map_dim, wl.origin, wl.res
    radius = [0.4] * dim
    lo = [0.5] * dim
    hi = [md[i] * res - 0.5 for i in range(dim)]
    path = np.array([lo, hi])
    env = engine.EnvMap(dim)
    wl.apply(env)
    env.set_potential_weight(0.5)
    env.set_gradient_weight(0.25)
    pot = env.updatePotentialMap([0.0] * dim, radius)
    reg = env.setSearchRegion(path, [1.0] * dim)
    got = env.expand(wl.nodes)
    env.close()
    want_pot = O.update_potential_map(wl.grid, md, org, res, [0.0] * dim, radius)
    want_reg = O.search_region(md, org, res, path, [1.0] * dim)
    assert np.array_equal(pot, want_pot) and np.array_equal(reg, want_reg)
    kw = dict(wl.params)
    kw.update(potential_weight=0.5, gradient_weight=0.25)
    oenv = O.Env(dim, wl.control, wl.U, want_pot, md, org, res, potential=want_pot, region=want_reg, **kw)
    ref = oracle_lib.expand(oenv, wl.nodes, threads=8)
    assert_slots_equal(got, ref, what="device-made potential + region, dim %d" % dim)
    assert ref["stats"]["finite"] > 0


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(O.REF_PLANNER_SO), reason="reference build (oracle/_ref) not present")
@pytest.mark.parametrize("case", CASES[:2], ids=[c[0] for c in CASES[:2]])
def test_drop_in_adapter_runs_the_reference_api_on_the_device(case):
    """The reference's own call sequence (setPotentialRadius / setPotentialMapRange /
    updatePotentialMap, setSearchRadius / setSearchRegion / getSearchRegion) on
    MPL::GpuMapPlanner from include/mplx_env_map.hpp gives the reference's cells."""
    name, dim, grid, md, org, res, centre, pots, path, regs = case
    for radius, rng_, pw in pots[:3]:
        a = O.update_potential_map(grid, md, org, res, centre, radius, rng_, pw, ref="gpu")
        b = O.update_potential_map(grid, md, org, res, centre, radius, rng_, pw)
        assert np.array_equal(a, b), "adapter potential %s" % (radius,)
    for sr, dense in regs:
        a = O.search_region(md, org, res, path, sr, dense, ref="gpu")
        b = O.search_region(md, org, res, path, sr, dense)
        assert np.array_equal(a, b), "adapter region %s" % (sr,)


@pytest.mark.gpu
@pytest.mark.parametrize("dim,with_region", [(2, False), (3, False), (3, True)])
def test_edit_map_equals_a_full_upload(engine, oracle_lib, dim, with_region):
    """mplx_edit_map (a few cells of the device's map patched in place, blocked bits with them, the free-box table
    rebuilt lazily) against mplx_set_map with the edited array: the same lists from every list kernel -- small batches
    (no free-box table after an edit) and a batch large enough to rebuild it -- and the same as the oracle on the edited map."""
    from helpers import assert_lists_equal, engine_env, oracle_env
    from test_gpu_parity import _small_world
    wl = _small_world(engine, dim, 0x03, seed=8800 + dim, n_nodes=200, region=with_region)
    rng = np.random.default_rng(3)
    flat = np.ascontiguousarray(wl.grid).ravel().copy()
    env = engine_env(engine, wl)
    before = env.expand_lists(wl.nodes)            # (makes the blocked bits and the free-box table of the ORIGINAL map)
    assert env.last_lists_route() == "grid"
    # block cells around the nodes' own neighbourhoods, free some occupied ones
    occ, free = np.nonzero(flat == 100)[0], np.nonzero(flat == 0)[0]
    idx = np.concatenate([rng.choice(free, 400, replace=False), rng.choice(occ, 300, replace=False)])
    val = np.concatenate([np.full(400, 100, np.int8), np.zeros(300, np.int8)])
    env.editMap(idx, val)
    flat[idx] = val
    wl2 = _small_world(engine, dim, 0x03, seed=8800 + dim, n_nodes=200, region=with_region)
    wl2.grid = flat.reshape(np.asarray(wl.grid).shape)
    ref = oracle_lib.expand(oracle_env(wl2), wl.nodes, threads=8)
    changed = False
    for route in ("grid", "tile", "dense"):
        env.set_lists_route(route)
        got = env.expand_lists(wl.nodes)
        assert_lists_equal(got, ref, wl.n_nodes, wl.U.shape[0], what="edited map, route %s" % route)
        changed = changed or not np.array_equal(got["cost"], before["cost"]) or not np.array_equal(got["count"], before["count"])
    assert changed, "the edit did not touch a single successor"
    # a batch large enough for the free-box table to be rebuilt from the patched bits
    env.set_lists_route("grid")
    big = np.ascontiguousarray(np.tile(wl.nodes, (1, 25)))  # 5 000 nodes
    got_big = env.expand_lists(big)
    ref_big = oracle_lib.expand(oracle_env(wl2), big, threads=8)
    assert_lists_equal(got_big, ref_big, big.shape[1], wl.U.shape[0], what="edited map, 5 000 nodes (free-box table rebuilt)")
    # ... and the same context against a fresh one that got the edited array whole
    env2 = engine_env(engine, wl2)
    whole = env2.expand_lists(big)
    for k in ("count", "action", "hash", "cost"):
        assert np.array_equal(got_big[k], whole[k]), k
    env2.close()
    with pytest.raises(engine._abi.MplxError):
        env.editMap([flat.size], [0])
    env.close()
