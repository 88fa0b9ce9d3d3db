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
    # gather to one rank (what bench.py does: the reference merges in the parent): the others receive nothing
    got0 = gather_tables(tab, dst=0)
    ok = ok and len(got0) == world
    for r in range(world):
        rr = np.random.default_rng(100 + r)
        kk = 0 if r == 1 else 7
        want = rr.integers(0, 1000, (kk, 5)).astype(np.int32) if rank == 0 else np.zeros((0, 5), np.int32)
        ok = ok and np.array_equal(got0[r], want)
    # a list of tables counts as their concatenation (what a rank's per-chromosome tables are)
    got2 = gather_tables([tab[:3], tab[3:3], tab[3:]])
    for r in range(world):
        ok = ok and np.array_equal(got2[r], got[r])
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


def _sweep_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import fake_backend
    from cloops_amd import api, pipe
    from cloops_amd.dist import make_allsum, shard_chromosomes, gather_tables
    from cloops_amd.synth import synth_chrom
    api.Chromosome = fake_backend.FakeChromosome
    api.device_count = lambda: 1
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    names = ["chrA", "chrB", "chrC"]
    data = {n: synth_chrom(12000 + 3000 * k, 3000000, 50 + k) for k, n in enumerate(names)}
    mine = shard_chromosomes([len(data[n][0]) for n in names])
    fs = [pipe.CACHE.put_arrays("%s-%s" % (names[i], names[i]), *data[names[i]]) for i in mine]
    dataI, cut, cuts, steps = pipe.runSweepFast(fs, [1000, 2000], [6, 4], cut=0, allsum=make_allsum())
    tabs = gather_tables(np.concatenate([v["boxes"] for v in dataI.values()]).astype(np.int32).reshape(-1, 4)
                         if dataI else np.zeros((0, 4), np.int32))
    q.put((rank, cut, [s.get("cut_out") for s in steps], [s["n_in"] for s in steps],
           {k[0]: v["boxes"].tolist() for k, v in dataI.items()}, sum(len(t) for t in tabs)))
    dist.destroy_process_group()


def test_distributed_chained_sweep_equals_single_process():
    """2 ranks (gloo), chromosomes sharded by LPT, cut chained through all-reduced statistics ==
    one process over all chromosomes (CPU oracle backend for both)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fake_backend
    from cloops_amd import api, pipe
    from cloops_amd.synth import synth_chrom
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sweep_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs])
    for p in procs:
        p.join(60)
    # single-process reference with the same backend
    old = api.Chromosome, api.device_count
    api.Chromosome, api.device_count = fake_backend.FakeChromosome, (lambda: 1)
    try:
        pipe.CACHE.clear()
        names = ["chrA", "chrB", "chrC"]
        fs = [pipe.CACHE.put_arrays("%s-%s" % (n, n), *synth_chrom(12000 + 3000 * k, 3000000, 50 + k)) for k, n in enumerate(names)]
        dataI, cut, cuts, steps = pipe.runSweepFast(fs, [1000, 2000], [6, 4], cut=0)
    finally:
        api.Chromosome, api.device_count = old
        pipe.CACHE.clear()
    want_boxes = {k[0]: v["boxes"].tolist() for k, v in dataI.items()}
    got_boxes = {}
    for r in res:
        assert r[1] == cut and r[2] == [s.get("cut_out") for s in steps] and r[3] == [s["n_in"] for s in steps]
        got_boxes.update(r[4])
        assert r[5] == sum(len(v) for v in want_boxes.values())      # the final gather sees every candidate
    assert got_boxes == want_boxes
    assert any(c is not None for c in res[0][2])
