"""Checker helper (test infrastructure, like everything under oracle/): engine successor LISTS against the dense
slots the CPU oracle / the reference build produce for the same nodes.  Used by bench.py's parity check of the
timed output and by __graft_entry__.smoke(); the tests have their own asserting twin (tests/helpers.py)."""
import numpy as np


def lists_mismatches(got, ref_dense, n_nodes, nU, cost_rtol=0.0):
    """Returns a list of human-readable problems (empty = the lists are the reference's successors: same set, same
    order, bit-identical hash / state / iteration counts, cost exact or within cost_rtol)."""
    bad = []
    st = ref_dense["status"].reshape(n_nodes, nU)
    emit = (st == 1) | (st == 2)
    want_count = emit.sum(axis=1)
    if not np.array_equal(got["count"][:n_nodes], want_count):
        k = np.nonzero(got["count"][:n_nodes] != want_count)[0]
        return ["count differs for %d nodes (first %d: got %d want %d)" % (k.size, k[0], got["count"][k[0]], want_count[k[0]])]
    src = np.nonzero(emit.ravel())[0]
    node = src // nU
    first = np.concatenate([[0], np.cumsum(want_count)[:-1]])
    stride = int(got.get("stride", nU))
    dst = node * stride + (np.arange(src.size) - first[node])
    if not np.array_equal(got["action"][dst], (src % nU).astype(np.int32)):
        bad.append("action order differs")
    if got.get("hash") is not None and not np.array_equal(got["hash"][dst], ref_dense["hash"][src]):
        bad.append("lattice hash differs")
    if got.get("state") is not None and ref_dense.get("state") is not None:
        if not np.array_equal(got["state"][:, dst].view(np.uint64), ref_dense["state"][:, src].view(np.uint64)):
            bad.append("successor state differs")
    if got.get("iters") is not None and not np.array_equal(got["iters"][dst], ref_dense["iters"][src]):
        bad.append("iteration count differs")
    gc, rc = got["cost"][dst], ref_dense["cost"][src]
    fin = np.isfinite(rc)
    if not np.array_equal(np.isinf(gc), ~fin):
        bad.append("blocked / finite pattern differs")
    elif cost_rtol == 0.0:
        if not np.array_equal(gc[fin], rc[fin]):
            bad.append("finite costs differ")
    elif fin.any() and (np.abs(gc[fin] - rc[fin]) / np.maximum(np.abs(rc[fin]), 1e-300)).max() > cost_rtol:
        bad.append("cost beyond rtol %g" % cost_rtol)
    return bad
