"""libcloops_comm.so / cloops_amd.comm: the torch-free RCCL layer of the multi-GPU path.

CPU: the library loads and exports every symbol include/cloops_comm.h declares; the unique-id hand-over between processes
(world 2 and 3) delivers rank 0's bytes to everyone.  GPU (world 1 -- one MI355X per box): init, all-reduce (sum, max),
all-gather / gather of tables through the device, barrier."""
import multiprocessing as mp
import os
import re

import numpy as np
import pytest

from cloops_amd import comm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_the_header():
    lib = comm.load()
    hdr = open(os.path.join(ROOT, "include", "cloops_comm.h")).read()
    declared = set(re.findall(r"\b(cl_comm_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(comm.SYMBOLS)
    for s in declared:
        assert hasattr(lib, s)


def _id_worker(args):
    rank, world, tag, d = args
    return comm.exchange_id(rank, world, lambda: bytes(bytearray((7 * k + rank) % 251 for k in range(comm.ID_BYTES))), tag=tag,
                            timeout=30.0, directory=d)


@pytest.mark.parametrize("world", [2, 3])
def test_unique_id_reaches_every_rank(tmp_path, world):
    tag = "t%d_%d" % (os.getpid(), world)
    with mp.get_context("fork").Pool(world) as pool:
        # the non-zero ranks start first and have to wait for rank 0's file
        order = list(range(1, world)) + [0]
        got = pool.map(_id_worker, [(r, world, tag, str(tmp_path)) for r in order], chunksize=1)
    want = bytes(bytearray((7 * k) % 251 for k in range(comm.ID_BYTES)))
    assert all(g == want for g in got)


@pytest.mark.gpu
def test_collectives_at_world_one():
    c = comm.Comm(0, 1, 0)
    try:
        assert c.rank == 0 and c.world == 1
        c.barrier()
        a = np.arange(6000, dtype=np.int64) * 3 - 5
        assert np.array_equal(c.allsum(a), a) and c.allsum(a).dtype == np.int64
        f = np.linspace(-1.0, 1.0, 77)
        assert np.array_equal(c.allsum(f), f)
        assert c.allmax(3.25) == 3.25
        rng = np.random.RandomState(3)
        t1, t2 = rng.randint(0, 1 << 30, (1000, 4)).astype(np.int32), rng.randint(0, 1 << 30, (7, 4)).astype(np.int32)
        for dst in (None, 0):
            out = c.gather_tables([t1, t2], dst=dst)
            assert len(out) == 1 and np.array_equal(out[0], np.concatenate([t1, t2]))
        assert c.gather_tables(np.zeros((0, 4), np.int32), dst=0)[0].shape == (0, 4)
        big = rng.randint(0, 1 << 30, (3000000, 4)).astype(np.int32)       # 48 MB: the staging buffers grow
        assert np.array_equal(c.gather_tables(big, dst=0, copy=False)[0], big)
    finally:
        c.close()
