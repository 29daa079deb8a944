"""Frontier sharding across GPUs.

Every (node, control) pair is independent (env_map<Dim>::get_succ, reference
include/mpl_planner/env/env_map.h:147-172, reads nothing but its arguments), so
the frontier is block-partitioned by node, the map / U / parameters are
replicated per rank, and the expansion itself needs NO collective.

The all-gather below is for consumers that want the complete successor set on
every rank (an on-device dedup / open-list merge stage; north_star: "only when
the open set exceeds a single GPU's launch").  It moves the successor LISTS in
their packed form (mplx_pack_lists_device: node k owns entries
[offs[k], offs[k+1]) of every row -- only emitted successors, no padding between
nodes), device to device: one exchange of the (nodes, entries) pair of every
rank, then one all_gather_into_tensor per row, padded to the largest rank and
compacted on the device.  The tensors are the engine's own buffers (TorchArray
hands torch-owned HBM to the C ABI through data_ptr), so nothing crosses the
host.  Works with any torch.distributed backend: nccl (= RCCL over xGMI) on
GPUs, gloo on CPU for the tests.  A C host uses mplx_comm_allgather_lists
(include/mplx.h), which does the same with exact-size all-pairs ncclSend/Recv.
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


class TorchArray:
    """Memory owned by a torch tensor (HBM on a GPU rank, host memory under gloo) with the interface the engine
    expects of a DeviceArray (.ptr, .nbytes, .download, .free): lets torch.distributed move the very buffers the
    kernels wrote."""

    def __init__(self, nbytes, device):
        import torch

        self.nbytes = int(nbytes)
        self.t = torch.zeros((self.nbytes + 7) // 8 + 1, dtype=torch.int64, device=device)
        if self.t.is_cuda:
            # the fill runs on TORCH's stream; the engine launches on its own: without this the zeroing can land after a
            # kernel of the engine has written the buffer (seen as a merge that reported 1 first occurrence of 20 M)
            torch.cuda.current_stream(self.t.device).synchronize()
        self.ptr = self.t.data_ptr()

    def view(self, dtype, n=None, offset=0):
        """The first n elements of `dtype` starting `offset` bytes in, as a tensor sharing the memory."""
        import torch

        raw = self.t.view(torch.uint8)[int(offset):self.nbytes]
        v = raw[: (raw.numel() // dtype.itemsize) * dtype.itemsize].view(dtype)
        return v if n is None else v[: int(n)]

    def download(self, dtype, shape, offset=0):
        import torch

        td = {np.dtype(np.int32): torch.int32, np.dtype(np.int64): torch.int64, np.dtype(np.float64): torch.float64,
              np.dtype(np.uint64): torch.int64, np.dtype(np.uint8): torch.uint8}[np.dtype(dtype)]
        n = int(np.prod(shape))
        out = self.view(td, n, offset).cpu().numpy()
        return out.view(dtype).reshape(shape).copy()

    def free(self):
        self.t = None
        self.ptr = None


def torch_alloc(device):
    """`alloc` argument of EnvMap.alloc_lists / alloc_packed: buffers owned by torch on `device`."""
    return lambda nbytes: TorchArray(nbytes, device)


def all_gather_packed(count, offs, rows, n_local, group=None):
    """All-gather of packed successor lists, device to device.

    count : int32 tensor [>= n_local]   per-node counts of this rank's shard
    offs  : int64 tensor [>= n_local+1] their exclusive prefix sums (offs[n_local] = entries of this rank)
    rows  : dict name -> tensor; 1-D rows are [capacity], a 2-D row (the state) is [F, capacity]
    Returns (count_all, offs_all, rows_all, node_offs, entry_offs): the concatenation in rank order (= ascending
    global node index for the block partition); node_offs / entry_offs are the per-rank offsets, [world + 1].
    Needs an initialised torch.distributed process group; every tensor lives on the backend's device.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    dev = count.device
    total = offs[n_local:n_local + 1]  # stays on the device
    meta = torch.cat([torch.tensor([n_local], dtype=torch.int64, device=dev), total.to(torch.int64)])
    metas = torch.empty(world * 2, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(metas, meta, group=group)
    m = metas.cpu().numpy().reshape(world, 2)  # the one host round trip: sizes
    node_offs = np.concatenate([[0], np.cumsum(m[:, 0])]).astype(np.int64)
    entry_offs = np.concatenate([[0], np.cumsum(m[:, 1])]).astype(np.int64)
    cap_n, cap_e = max(int(m[:, 0].max()), 1), max(int(m[:, 1].max()), 1)

    def gather(x, cap, sizes):
        """x: [..., >= own size] -> concatenation over ranks of x[..., :size_r]."""
        lead = x.shape[:-1]
        if x.dim() == 1 and x.shape[0] >= cap:
            own = x[:cap]  # a contiguous prefix of the engine's own buffer: no staging copy
        else:
            own = torch.zeros(lead + (cap,), dtype=x.dtype, device=dev)
            k = min(cap, x.shape[-1])
            own[..., :k] = x[..., :k]
        out = torch.empty((world,) + lead + (cap,), dtype=x.dtype, device=dev)
        dist.all_gather_into_tensor(out.view(-1), own.view(-1), group=group)
        return torch.cat([out[r][..., : int(sizes[r])] for r in range(world)], dim=-1)

    count_all = gather(count, cap_n, m[:, 0])
    rows_all = {k: gather(v, cap_e, m[:, 1]) for k, v in rows.items() if v is not None}
    offs_all = torch.zeros(int(node_offs[-1]) + 1, dtype=torch.int64, device=dev)
    offs_all[1:] = torch.cumsum(count_all.to(torch.int64), dim=0)
    return count_all, offs_all, rows_all, node_offs, entry_offs


def packed_views(packed, n_local, total=None):
    """Tensor views of an env.PackedLists whose buffers are TorchArrays (for all_gather_packed)."""
    import torch

    cap = packed.capacity if total is None else int(total)
    rows = {"action": packed.action.view(torch.int32, cap), "cost": packed.cost.view(torch.float64, cap)}
    if packed.hash is not None:
        rows["hash"] = packed.hash.view(torch.int64, cap)
    if packed.state is not None:
        rows["state"] = packed.state.view(torch.float64, packed.n_fields * packed.capacity).view(packed.n_fields, packed.capacity)[:, :cap]
    return packed.count.view(torch.int32, max(n_local, 1))[:n_local], packed.offs.view(torch.int64, n_local + 1), rows


def packed_struct(count_all, offs_all, rows_all):
    """The C struct (_abi.PackedLists) of gathered packed lists held in torch tensors (what all_gather_packed returns):
    lets the engine run on them where they are, e.g. EnvMap.post_packed -- the on-device merge of the whole frontier's
    successors (heuristic, goal flags, first occurrences) that is the reason to gather at all.  Keep the tensors alive
    while the struct is in use."""
    from . import _abi

    total = int(offs_all.shape[0]) and int(rows_all["action"].shape[-1])
    ps = _abi.PackedLists()
    ps.count, ps.offs = count_all.data_ptr(), offs_all.data_ptr()
    ps.action, ps.cost = rows_all["action"].data_ptr(), rows_all["cost"].data_ptr()
    ps.hash = rows_all["hash"].data_ptr() if rows_all.get("hash") is not None else None
    st = rows_all.get("state")
    if st is not None:
        assert st.is_contiguous() and st.shape[-1] == total
        ps.state = st.data_ptr()
    ps.state_stride = total
    ps.capacity = max(total, 1)
    return ps


def comm_schedule(world, rank, meta, n_fields):
    """mplx_comm_schedule (include/mplx.h): the all-pairs exchange mplx_comm_allgather_lists executes, as data.

    meta : int64 [world][_abi.COMM_META] -- what the ranks all-gather first.
    Returns (ops, node_offs, entry_offs); ops is a list of dicts (kind, peer, row, elem, src_off, dst_off, bytes).
    Raises _abi.MplxError with the collective verdict when the meta is inconsistent (every rank raises the same).
    Pure host arithmetic inside libmplx.so: needs neither a GPU nor RCCL."""
    import ctypes as C

    from . import _abi

    L = _abi.lib()
    meta = np.ascontiguousarray(meta, dtype=np.int64).reshape(world, _abi.COMM_META)
    noff = np.zeros(world + 1, np.int64)
    eoff = np.zeros(world + 1, np.int64)
    n = L.mplx_comm_schedule(world, rank, meta.ctypes.data, n_fields, None, 0, noff.ctypes.data, eoff.ctypes.data)
    if n < 0:
        raise _abi.MplxError(int(n), "mplx_comm_schedule: the ranks' meta records are inconsistent")
    ops = (_abi.CommOp * max(int(n), 1))()
    n2 = L.mplx_comm_schedule(world, rank, meta.ctypes.data, n_fields, ops, int(n), None, None)
    assert n2 == n
    return ([{f: getattr(ops[i], f) for f, _ in _abi.CommOp._fields_} for i in range(int(n))], noff, eoff)
