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


@pytest.mark.gpu
def test_device_resident_exchanges_at_world_one():
    """cl_comm_gather_device / cl_comm_allreduce_f64_device: the candidate tables go to the merging rank from where
    cl_cand_finish_device left them, the statistics are reduced in device memory -- at world size 1 (one MI355X per box)"""
    import ctypes
    from cloops_amd import api, pipe
    from cloops_amd.synth import synth_chrom
    pipe.CACHE.clear()
    fs = [pipe.CACHE.put_arrays("chr%d-chr%d" % (k, k), *synth_chrom(n, 30000000, 60 + k)) for k, n in enumerate((90000, 60000, 30000), 1)]
    c = comm.Comm(0, 1, 0)
    try:
        host = pipe.runSweepFast(fs, [1000, 2000], [6, 4], cut=0)[0]
        dev = pipe.runSweepFast(fs, [1000, 2000], [6, 4], cut=0, finish_device=True)[0]
        assert list(host) == list(dev) and all(v["boxes"] is None and v["n_rows"] == len(host[k]["boxes"]) for k, v in dev.items())
        out = c.gather_device([v["dev_rows"] for v in dev.values()], [v["n_rows"] for v in dev.values()], dst=0)
        assert len(out) == 1 and np.array_equal(out[0], np.concatenate([host[k]["boxes"] for k in host]))
        assert c.gather_device([], [], dst=0)[0].shape == (0, 4)
        # the all-reduce in place in device memory: a double buffer of the HIP runtime's own
        hip = ctypes.CDLL("libamdhip64.so")
        ptr = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(ptr), ctypes.c_size_t(8 * 1000)) == 0
        try:
            a = np.linspace(-3.0, 5.0, 1000)
            assert hip.hipMemcpy(ptr, a.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(8000), 1) == 0
            c.allsum_device(ptr.value, 1000)
            b = np.zeros(1000)
            assert hip.hipMemcpy(b.ctypes.data_as(ctypes.c_void_p), ptr, ctypes.c_size_t(8000), 2) == 0
            assert np.array_equal(a, b)
        finally:
            hip.hipFree(ptr)
    finally:
        c.close()
        pipe.CACHE.clear()


# ---- the Python side of Comm at world size > 1, over a stand-in for libcloops_comm.so ---------------------------------
class _Hub(object):
    """what RCCL does, in one process: every rank deposits its buffer, a barrier, everybody reads what it needs"""
    def __init__(self, world):
        import threading
        self.world = world
        self.bar = threading.Barrier(world)
        self.slots = [None] * world


class _FakeCommLib(object):
    """the C ABI of include/cloops_comm.h on host memory (ctypes pointers in, ctypes pointers out), one instance per rank"""
    def __init__(self, hub, rank):
        self.hub, self.rank = hub, rank

    def cl_comm_last_error(self):
        return b""

    def cl_comm_unique_id(self, buf):
        return 0

    def cl_comm_init(self, blob, rank, world, device, out):
        return 0

    def cl_comm_destroy(self, h):
        pass

    def _exchange(self, arr):
        self.hub.slots[self.rank] = arr.copy()
        self.hub.bar.wait()
        allv = [s.copy() for s in self.hub.slots]
        self.hub.bar.wait()
        return allv

    def _view(self, ptr, n, ctype):
        import ctypes
        addr = ptr.value if hasattr(ptr, "value") else ctypes.cast(ptr, ctypes.c_void_p).value
        return np.ctypeslib.as_array((ctype * int(n)).from_address(addr))

    def cl_comm_allreduce_f64(self, h, ptr, n):
        import ctypes
        v = self._view(ptr, n, ctypes.c_double)
        v[:] = np.sum(self._exchange(v), axis=0)
        return 0

    def cl_comm_allreduce_max_f64(self, h, ptr, n):
        import ctypes
        v = self._view(ptr, n, ctypes.c_double)
        v[:] = np.max(self._exchange(v), axis=0)
        return 0

    def cl_comm_allgather_i32(self, h, pin, n, pout):
        import ctypes
        allv = self._exchange(self._view(pin, n, ctypes.c_int32))
        self._view(pout, n * self.hub.world, ctypes.c_int32)[:] = np.concatenate(allv)
        return 0

    def cl_comm_gather_i32(self, h, pin, n, root, pout):
        import ctypes
        allv = self._exchange(self._view(pin, n, ctypes.c_int32))
        if self.rank == root:
            self._view(pout, n * self.hub.world, ctypes.c_int32)[:] = np.concatenate(allv)
        return 0

    def cl_comm_barrier(self, h):
        self.hub.bar.wait()
        return 0

    def cl_comm_host_alloc(self, nbytes):
        import ctypes
        self._bufs = getattr(self, "_bufs", {})
        b = (ctypes.c_char * int(nbytes))()
        self._bufs[ctypes.addressof(b)] = b
        return ctypes.addressof(b)

    def cl_comm_host_free(self, p):
        self._bufs.pop(p.value if hasattr(p, "value") else p, None)

    def cl_comm_gather_i32_pinned(self, h, pin, n, root, pout):
        return self.cl_comm_allgather_i32(h, pin, n, pout) if root < 0 else self.cl_comm_gather_i32(h, pin, n, root, pout)


@pytest.mark.parametrize("world", [2, 3])
def test_comm_python_side_at_world_n(monkeypatch, world):
    """allsum (ints come back as ints), allmax, gather_tables (ragged tables, an empty one, dst / all-gather) with the ranks as
    threads over the stand-in library"""
    import threading
    hub = _Hub(world)
    rng = np.random.RandomState(11)
    tables = [rng.randint(0, 1 << 30, (k, 4)).astype(np.int32) for k in ([5, 0, 1200][:world])]
    out = [None] * world
    comms = [None] * world          # (kept until the checks are done: copy=False hands out views of the communicator's buffers)

    def rank_main(r):
        c = comms[r] = comm.Comm.__new__(comm.Comm)
        c._lib, c._h, c.rank, c.world, c.device = _FakeCommLib(hub, r), 1, r, world, 0
        res = {}
        res["sum_i"] = c.allsum(np.arange(10, dtype=np.int64) * (r + 1))
        res["sum_f"] = c.allsum(np.asarray([0.5 * (r + 1), -1.0]))
        res["max"] = c.allmax(float(r))
        res["all"] = c.gather_tables(tables[r], dst=None)
        res["root"] = c.gather_tables([tables[r], tables[r][:2]], dst=0, copy=False)
        out[r] = res
    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(60)
    tri = world * (world + 1) // 2
    for r in range(world):
        res = out[r]
        assert res is not None
        assert res["sum_i"].dtype == np.int64 and np.array_equal(res["sum_i"], np.arange(10) * tri)
        assert np.allclose(res["sum_f"], [0.5 * tri, -1.0 * world])
        assert res["max"] == float(world - 1)
        assert all(np.array_equal(a, b) for a, b in zip(res["all"], tables))
        if r == 0:
            assert all(np.array_equal(a, np.concatenate([t, t[:2]])) for a, t in zip(res["root"], tables))
        else:
            assert all(len(a) == 0 for a in res["root"])


def test_id_file_names_one_launcher_instance_and_one_communicator(monkeypatch):
    """the id file is found by NAME only (no clock heuristics: a rank may import late): the name carries the launcher's pid AND
    its start time (a leftover of an earlier launcher that had the same pid is another file), MASTER_PORT, the elastic run id /
    restart count and the number of communicators the process has formed so far (a second communicator of the same ranks)"""
    t1 = comm.default_tag()
    assert str(os.getppid()) in t1 and comm._parent_start() in t1 and comm._parent_start() != "0"
    monkeypatch.setattr(comm, "_SEQ", comm._SEQ + 1)
    t2 = comm.default_tag()
    assert t1 != t2
    monkeypatch.setenv("MASTER_PORT", "29999")
    monkeypatch.setenv("TORCHELASTIC_RESTART_COUNT", "1")
    assert comm.default_tag() not in (t1, t2)
    monkeypatch.setattr(comm, "_parent_start", lambda: "12345")
    assert "12345" in comm.default_tag()
    assert comm.id_path("x", "/d") == "/d/cloops_comm_id_x"


def test_stale_id_file_fails_fast_and_a_nonce_names_the_file_exactly(tmp_path, monkeypatch):
    """a full-size id file much older than this process: reported after the grace period instead of being waited for until the timeout;
    with CLOOPS_COMM_NONCE in the environment of every rank the file is found by name alone (no clock heuristics)"""
    import time
    monkeypatch.delenv("CLOOPS_COMM_NONCE", raising=False)
    monkeypatch.setattr(comm, "STALE_ID_GRACE_S", 0.3)
    path = comm.id_path("stale", str(tmp_path))
    with open(path, "wb") as fh:
        fh.write(b"x" * comm.ID_BYTES)
    old = time.time() - 3600
    os.utime(path, (old, old))
    t0 = time.time()
    with pytest.raises(comm.CommError, match="CLOOPS_COMM_NONCE"):
        comm.exchange_id(1, 2, lambda: b"", tag="stale", timeout=30.0, directory=str(tmp_path))
    assert time.time() - t0 < 10
    monkeypatch.setenv("CLOOPS_COMM_NONCE", "launch42")
    assert "launch42" in comm.default_tag()
    path2 = comm.id_path(None, str(tmp_path))
    with open(path2, "wb") as fh:
        fh.write(b"y" * comm.ID_BYTES)
    os.utime(path2, (old, old))
    assert comm.exchange_id(1, 2, lambda: b"", timeout=5.0, directory=str(tmp_path)) == b"y" * comm.ID_BYTES
