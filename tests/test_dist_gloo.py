"""CPU, world_size 2, gloo: the N>1 plumbing (LPT sharding + variable-length table gather)."""
import os
import socket
import sys

import numpy as np
import pytest

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from cloops_amd.dist import gather_tables
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(100 + rank)
    k = 0 if rank == 1 else 7                       # one rank contributes an EMPTY table
    tab = rng.integers(0, 1000, (k, 5)).astype(np.int32)
    got = gather_tables(tab)
    ok = len(got) == world
    for r in range(world):
        rr = np.random.default_rng(100 + r)
        kk = 0 if r == 1 else 7
        ok = ok and np.array_equal(got[r], rr.integers(0, 1000, (kk, 5)).astype(np.int32))
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_gather_tables_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def test_lpt_assign_balances_hg38():
    from cloops_amd.dist import lpt_assign
    from cloops_amd.synth import chrom_sizes
    sizes = [n for _, _, n in chrom_sizes(200000000)]
    for g, bound in ((2, 0.51), (4, 0.26), (8, 0.135)):
        parts = lpt_assign(sizes, g)
        assert sorted(i for p in parts for i in p) == list(range(23))
        assert max(sum(sizes[i] for i in p) for p in parts) / sum(sizes) <= bound
