"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" = RCCL over
xGMI on ROCm, "gloo" on CPU for tests).  Chromosomes are independent units
(cLoops/pipe.py:117 `Parallel(... for f in fs)`), so the data path needs NO collective;
the only exchange is the per-step gather of the small candidate-loop tables
(cLoops/pipe.py:119-127 merges the workers' pickled results the same way)."""
import numpy as np


def lpt_assign(sizes, n_parts):
    """Longest-processing-time assignment of work units (chromosomes) to ranks.
    sizes: list of work sizes; returns list of lists of unit indices per rank."""
    order = sorted(range(len(sizes)), key=lambda i: -sizes[i])
    load = [0] * n_parts
    parts = [[] for _ in range(n_parts)]
    for i in order:
        r = min(range(n_parts), key=lambda k: (load[k], k))
        parts[r].append(i)
        load[r] += sizes[i]
    return parts


_PINNED = {}


def _pinned(name, rows, cols):
    """a cached pinned host tensor of at least rows x cols int32 (device path only: pageable staging of the 58 MB table of the
    200 M-PET sweep costs 10 ms per direction, pinned 1.5 ms)"""
    import torch
    t = _PINNED.get(name)
    if t is None or t.shape[0] < rows or t.shape[1] != cols:
        t = torch.empty((max(rows + rows // 4, 1024), cols), dtype=torch.int32, pin_memory=True)
        _PINNED[name] = t
    return t


def gather_tables(table, device=None, group=None, dst=None, copy=True):
    """Gather variable-length int32 [K_r, C] tables (cluster boxes) from every rank.

    Two collectives: the row counts (all_gather of one int64 each), then one padded gather of the rows.
    dst=None: all-gather -- the list of per-rank numpy arrays on every rank.
    dst=r: only rank r receives (the reference merges its workers' results in the parent, cLoops/pipe.py:119-127); the
    other ranks get a list of empty tables and do not pay for the device-to-host copy of everybody's rows.
    `table` may be a list of tables (taken as their concatenation)."""
    import torch
    import torch.distributed as dist
    # a LIST of tables (one per chromosome) is taken as their concatenation -- on the device path its parts go straight into the
    # pinned staging buffer (one host copy instead of np.concatenate + staging)
    parts = None
    if isinstance(table, (list, tuple)):
        alltabs = [np.asarray(t) for t in table]
        if not alltabs or any(t.ndim != 2 for t in alltabs) or len({t.shape[1] for t in alltabs}) > 1:
            raise ValueError("tables must be a non-empty list of [K, C] arrays with one C")
        ncols = alltabs[0].shape[1]
        parts = [np.ascontiguousarray(t, dtype=np.int32) for t in alltabs if len(t)]
        if device is None or torch.device(device).type == "cpu":
            table = np.concatenate(parts) if parts else np.zeros((0, ncols), np.int32)
            parts = None
        else:
            table = np.zeros((sum(len(t) for t in parts), ncols), np.int32) if not parts else None
    if parts:
        nrows, c_in = sum(len(t) for t in parts), parts[0].shape[1]
    else:
        table = np.ascontiguousarray(table, dtype=np.int32)
        if table.ndim != 2:
            raise ValueError("table must be [K, C]")
        nrows, c_in = table.shape
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    on_dev = device is not None and torch.device(device).type != "cpu"
    dev = torch.device("cpu") if device is None else device
    k = torch.tensor([nrows], dtype=torch.int64, device=dev)
    ks = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(ks, k, group=group)
    ks = ks.cpu().tolist()
    kmax = max(max(ks), 1)
    c = c_in
    pad = torch.zeros((kmax, c), dtype=torch.int32, device=dev)
    if nrows:
        if on_dev:
            stage = _pinned("h2d", nrows, c)
            view = stage.numpy()
            a = 0
            for t in (parts if parts else [table]):
                view[a: a + len(t)] = t
                a += len(t)
            pad[:nrows].copy_(stage[:nrows], non_blocking=True)
        else:
            pad[:nrows] = torch.from_numpy(table)
    receiver = dst is None or rank == dst
    if dst is None:
        out = torch.empty((world * kmax, c), dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(out, pad, group=group)
    else:
        out = torch.empty((world * kmax, c), dtype=torch.int32, device=dev) if receiver else None
        dist.gather(pad, list(out.view(world, kmax, c).unbind(0)) if receiver else None, dst=dst, group=group)
    if not receiver:
        return [np.zeros((0, c), np.int32) for _ in range(world)]
    if on_dev:
        host = _pinned("d2h", world * kmax, c)
        host[: world * kmax].copy_(out, non_blocking=True)
        torch.cuda.synchronize()
        out = host[: world * kmax].numpy().reshape(world, kmax, c)
    else:
        out = out.numpy().reshape(world, kmax, c)
    # copy=False: views of the staging buffer, valid until the next call
    return [out[r, : ks[r]].copy() if copy else out[r, : ks[r]] for r in range(world)]


def make_allsum(device=None, group=None):
    """-> allsum(np.ndarray) = element-wise sum over all ranks (int64 or float64, any small shape):
    the only per-step exchange of the chained sweep (cloops_amd.pipe.runSweepFast).
    device=None: host tensors (a gloo group); a GPU device: the array goes through a cached pinned buffer to the device, is
    all-reduced over RCCL on torch's current stream and comes back the same way (one stream synchronisation)."""
    import torch
    import torch.distributed as dist
    on_dev = device is not None and torch.device(device).type != "cpu"
    dev = torch.device("cpu") if device is None else torch.device(device)
    cache = {}

    def allsum(a):
        a = np.ascontiguousarray(a)
        if not on_dev:
            t = torch.from_numpy(a.copy())
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            return t.numpy()
        key = (a.dtype.str, a.size)
        if key not in cache:
            tdt = torch.from_numpy(np.zeros(1, a.dtype)).dtype
            cache[key] = (torch.empty(a.size, dtype=tdt, pin_memory=True), torch.empty(a.size, dtype=tdt, device=dev))
        pin, d = cache[key]
        pin.numpy()[:] = a.ravel()
        d.copy_(pin, non_blocking=True)
        dist.all_reduce(d, op=dist.ReduceOp.SUM, group=group)
        pin.copy_(d, non_blocking=True)
        torch.cuda.current_stream(dev).synchronize()
        return pin.numpy().reshape(a.shape).copy()
    return allsum


def shard_chromosomes(sizes, rank=None, world=None):
    """indices of the chromosomes this rank owns (LPT by PET count)."""
    import torch.distributed as dist
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    return sorted(lpt_assign(list(sizes), world)[rank])
