"""Frontier sharding across GPUs.

Every (node, control) pair is independent, so the frontier is block-partitioned
by node, the map / U / parameters are replicated per rank, and the data path
needs NO collective.  The optional all-gather below is for consumers that want
the complete successor set on every rank (e.g. an on-device dedup / open-list
merge stage): it exchanges compact records only -- (global slot, cost, hash) =
24 B per *emitted* successor instead of the 129 B dense slot -- with one
count exchange followed by one padded all_gather, which RCCL runs as direct
peer-to-peer copies over xGMI (all links busy at once; a ring would be
per-link bound).  Works with any torch.distributed backend (nccl on GPUs,
gloo on CPU for the tests).
"""
import numpy as np


def partition(n_nodes, world, rank):
    """Block partition [lo, hi) of the frontier for `rank` (contiguous, sizes differ by <= 1)."""
    if not (0 <= rank < world):
        raise ValueError("rank %d not in [0, %d)" % (rank, world))
    base, rem = divmod(int(n_nodes), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def compact_records(status, cost, hash_, node_offset, nU):
    """Dense slots of one shard -> compact records of the emitted successors
    (status FINITE or BLOCKED, as the reference returns them), with GLOBAL slot
    ids (global node index * nU + control)."""
    status = np.asarray(status)
    keep = np.nonzero((status == 1) | (status == 2))[0]
    gslot = keep.astype(np.int64) + np.int64(node_offset) * np.int64(nU)
    return gslot, np.asarray(cost)[keep], np.asarray(hash_)[keep].astype(np.int64)


def all_gather_records(gslot, cost, hash_, device=None):
    """All-gather variable-length compact records from every rank.  Returns the
    concatenation ordered by rank (= ascending global slot for a block
    partition).  Needs an initialised torch.distributed process group."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    n = torch.tensor([gslot.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)
    # one [cap, 3] int64 payload per rank: slot, cost bits, hash bits
    pay = torch.zeros((cap, 3), dtype=torch.int64, device=dev)
    if gslot.shape[0]:
        pay[: gslot.shape[0], 0] = torch.as_tensor(np.ascontiguousarray(gslot), device=dev)
        pay[: gslot.shape[0], 1] = torch.as_tensor(np.ascontiguousarray(cost).view(np.int64), device=dev)
        pay[: gslot.shape[0], 2] = torch.as_tensor(np.ascontiguousarray(hash_).view(np.int64), device=dev)
    out = torch.empty((world * cap, 3), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(out, pay)
    out = out.cpu().numpy().reshape(world, cap, 3)
    parts = [out[r, : counts[r]] for r in range(world)]
    allp = np.concatenate(parts, axis=0) if parts else np.zeros((0, 3), np.int64)
    return allp[:, 0].copy(), allp[:, 1].copy().view(np.float64), allp[:, 2].copy().view(np.uint64)
