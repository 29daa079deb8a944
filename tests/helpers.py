"""Shared helpers of the parity tests: build the same environment for the HIP
engine and for the CPU oracle, and compare dense slot results."""
import numpy as np

from oracle import oracle as O


def oracle_env(wl, **override):
    """oracle.Env for a motion_primitive_library_amd.workloads.Workload."""
    kw = dict(wl.params)
    kw.update(override)
    return O.Env(wl.dim, wl.control, wl.U, wl.grid, wl.map_dim, wl.origin, wl.res,
                 potential=wl.potential, region=wl.region, **kw)


def engine_env(m, wl, device=0):
    env = m.EnvMap(wl.dim, device)
    wl.apply(env)
    return env


def ulp_diff(a, b):
    """|a-b| in units of the last place of b, elementwise (finite inputs)."""
    ai = np.asarray(a, np.float64).view(np.int64)
    bi = np.asarray(b, np.float64).view(np.int64)
    return np.abs(ai - bi)


def assert_slots_equal(got, ref, cost_rtol=0.0, check_iters=True, what=""):
    """Bit-exact on status / hash / successor state / iteration counts; cost
    exact by default (cost_rtol > 0 for paths that use device trig)."""
    n = ref["status"].size
    assert got["status"].size == n
    bad = np.nonzero(got["status"] != ref["status"])[0]
    assert bad.size == 0, "%s status differs in %d of %d slots, first %s: got %s want %s" % (
        what, bad.size, n, bad[:5], got["status"][bad[:5]], ref["status"][bad[:5]])
    bad = np.nonzero(got["hash"] != ref["hash"])[0]
    assert bad.size == 0, "%s hash differs in %d slots, first %s" % (what, bad.size, bad[:5])
    if "state" in got and got["state"] is not None and ref.get("state") is not None:
        # bit-exact including the sign of zero
        g = got["state"].view(np.uint64)
        r = ref["state"].view(np.uint64)
        bad = np.argwhere(g != r)
        assert bad.shape[0] == 0, "%s successor state differs in %d entries, first (row, slot) %s: got %r want %r" % (
            what, bad.shape[0], bad[:3].tolist(),
            [got["state"][tuple(b)] for b in bad[:3]], [ref["state"][tuple(b)] for b in bad[:3]])
    if check_iters and "iters" in got and got["iters"] is not None:
        bad = np.nonzero(got["iters"] != ref["iters"])[0]
        assert bad.size == 0, "%s sample-loop iteration count differs in %d slots, first %s: got %s want %s" % (
            what, bad.size, bad[:5], got["iters"][bad[:5]], ref["iters"][bad[:5]])
    fin = ref["status"] == 1
    assert np.all(np.isinf(got["cost"][~fin])), "%s non-finite slots must carry +inf" % what
    gc, rc = got["cost"][fin], ref["cost"][fin]
    if cost_rtol == 0.0:
        bad = np.nonzero(gc != rc)[0]
        assert bad.size == 0, "%s finite cost differs in %d slots (max rel %g)" % (
            what, bad.size, np.max(np.abs(gc - rc) / np.abs(rc)) if bad.size else 0)
    else:
        rel = np.abs(gc - rc) / np.maximum(np.abs(rc), 1e-300)
        assert np.all(rel <= cost_rtol), "%s cost rel err %g > %g" % (what, rel.max(), cost_rtol)
