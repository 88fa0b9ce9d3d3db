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


def gather_tables(table, device=None, group=None):
    """All-gather variable-length int32 [K_r, C] tables (cluster boxes) from every rank.

    Two collectives: the row counts (all_gather of one int64 each), then one padded
    all_gather of the rows.  Returns the list of per-rank numpy arrays (on every rank)."""
    import torch
    import torch.distributed as dist
    table = np.ascontiguousarray(table, dtype=np.int32)
    if table.ndim != 2:
        raise ValueError("table must be [K, C]")
    world = dist.get_world_size(group)
    dev = torch.device("cpu") if device is None else device
    k = torch.tensor([table.shape[0]], dtype=torch.int64, device=dev)
    ks = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(ks, k, group=group)
    ks = ks.cpu().tolist()
    kmax = max(max(ks), 1)
    c = table.shape[1]
    pad = torch.zeros((kmax, c), dtype=torch.int32, device=dev)
    if table.shape[0]:
        pad[: table.shape[0]] = torch.from_numpy(table).to(dev)
    out = torch.empty((world * kmax, c), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(out, pad, group=group)
    out = out.cpu().numpy().reshape(world, kmax, c)
    return [out[r, : ks[r]].copy() for r in range(world)]


def make_allsum(device=None, group=None):
    """-> allsum(np.ndarray) = element-wise sum over all ranks (int64 or float64, any small shape):
    the only per-step exchange of the chained sweep (cloops_amd.pipe.runSweepFast)."""
    import torch
    import torch.distributed as dist
    dev = torch.device("cpu") if device is None else device

    def allsum(a):
        a = np.ascontiguousarray(a)
        t = torch.from_numpy(a.copy()).to(dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t.cpu().numpy()
    return allsum


def shard_chromosomes(sizes, rank=None, world=None):
    """indices of the chromosomes this rank owns (LPT by PET count)."""
    import torch.distributed as dist
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    return sorted(lpt_assign(list(sizes), world)[rank])
