"""cloops_amd/csrc/cloops_comm.cpp at world sizes > 1 without GPUs: the file is compiled UNCHANGED with g++ and linked against
tests/c/fake_rccl_hip.cpp (host memory, threads as ranks) instead of libamdhip64 / librccl.  One MI355X per lease means RCCL itself
never runs with two ranks in this project's GPU tests; what CAN be pinned on a CPU is the library's own control flow -- above all
that no rank is ever left inside a send when another rank leaves with an error (the hang of round 5's cl_comm_gather_device).
Reference shape: the parent's merge of its workers' results, cLoops/pipe.py:117-127."""
import ctypes
import os
import subprocess
import threading
import time

import numpy as np
import pytest

from cloops_amd import comm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROCM_INC = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "include")


@pytest.fixture(scope="module")
def fake(tmp_path_factory):
    if not os.path.exists(os.path.join(ROCM_INC, "rccl", "rccl.h")):
        pytest.skip("no ROCm headers here")
    so = str(tmp_path_factory.mktemp("fakecomm") / "libcloops_comm_fake.so")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I" + ROCM_INC,
                           os.path.join(ROOT, "cloops_amd", "csrc", "cloops_comm.cpp"), os.path.join(ROOT, "tests", "c", "fake_rccl_hip.cpp"),
                           "-o", so, "-lpthread"])
    return comm._declare(ctypes.CDLL(so))


def _ranks(lib, world, body, timeout=60):
    """run body(comm_object, rank) on `world` threads over one fake communicator -> list of results (exceptions re-raised)"""
    blob = (ctypes.c_char * comm.ID_BYTES)()
    assert lib.cl_comm_unique_id(blob) == 0
    out, err = [None] * world, [None] * world

    def main(r):
        try:
            h = ctypes.c_void_p()
            assert lib.cl_comm_init(blob, r, world, 0, ctypes.byref(h)) == 0
            c = comm.Comm.__new__(comm.Comm)
            c._lib, c._h, c.rank, c.world, c.device = lib, h, r, world, 0
            try:
                out[r] = body(c, r)
            finally:
                lib.cl_comm_destroy(h)
        except BaseException as e:       # noqa: BLE001 -- handed to the main thread
            err[r] = e
    ts = [threading.Thread(target=main, args=(r,)) for r in range(world)]
    t0 = time.time()
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout)
    assert not any(t.is_alive() for t in ts), "a rank hangs"
    for e in err:
        if e is not None:
            raise e
    return out, time.time() - t0


@pytest.mark.parametrize("world", [2, 3, 8])
def test_exchanges_at_world_n(fake, monkeypatch, world):
    monkeypatch.setattr(comm, "_lib", fake)
    rng = np.random.default_rng(world)
    tabs = [[rng.integers(0, 1 << 30, (int(rng.integers(0, 50)), 4)).astype(np.int32) for _ in range(3)] for _ in range(world)]
    tabs[world - 1] = [np.zeros((0, 4), np.int32)] * 3                    # a rank without candidates

    def body(c, r):
        res = {}
        res["sum"] = c.allsum(np.arange(6, dtype=np.float64) * (r + 1))
        res["max"] = c.allmax(float(r))
        res["host"] = c.gather_tables(tabs[r], dst=0)
        res["dev"] = c.gather_device([t.ctypes.data for t in tabs[r]], [len(t) for t in tabs[r]], cols=4, dst=1 % world)
        d = np.full(5, float(r + 1))
        c.allsum_device(d.ctypes.data, 5)
        res["dsum"] = d
        return res
    out, _ = _ranks(fake, world, body)
    tri = world * (world + 1) / 2
    want = [np.concatenate(t) for t in tabs]
    for r, res in enumerate(out):
        assert np.allclose(res["sum"], np.arange(6) * tri) and res["max"] == world - 1 and np.allclose(res["dsum"], tri)
        if r == 0:
            assert all(np.array_equal(a, b) for a, b in zip(res["host"], want))
        if r == 1 % world:
            assert all(np.array_equal(a, b) for a, b in zip(res["dev"], want))
        else:
            assert all(len(a) == 0 for a in res["dev"])


@pytest.mark.parametrize("case", ["too_small", "null_buffer", "bad_table"])
def test_gather_device_errors_are_collective(fake, monkeypatch, case):
    """every rank returns the same error, and none is left waiting in a send: the size / buffer checks are made by all ranks from
    one all-gathered record BEFORE any send or receive is posted (round 5: the root returned after the counts, its peers blocked)"""
    monkeypatch.setattr(comm, "_lib", fake)
    world, root = 3, 0
    tabs = [np.arange(4 * (5 + r), dtype=np.int32).reshape(-1, 4) for r in range(world)]
    total = sum(len(t) for t in tabs)

    def body(c, r):
        tp = (ctypes.c_void_p * 1)(tabs[r].ctypes.data if not (case == "bad_table" and r == 2) else None)
        tr = (ctypes.c_int64 * 1)(len(tabs[r]))
        per = (ctypes.c_int64 * world)()
        out = np.zeros((total, 4), np.int32)
        cap = total - 1 if case == "too_small" else total
        outp = None if (case == "null_buffer" or r != root) else out.ctypes.data_as(ctypes.c_void_p)
        rc = fake.cl_comm_gather_device(c._h, tp, tr, 1, 4, root, outp, cap if r == root else 0, per)
        msg = (fake.cl_comm_last_error() or b"").decode()
        # the communicator is still usable afterwards: a correct gather on the same handle
        ok = c.gather_device([tabs[r].ctypes.data], [len(tabs[r])], cols=4, dst=root)
        return rc, msg, ok
    out, took = _ranks(fake, world, body)
    assert took < 15, "a rank waited for a peer that had left"
    word = {"too_small": "too small", "null_buffer": "null receive buffer", "bad_table": "bad table"}[case]
    for r, (rc, msg, ok) in enumerate(out):
        assert rc != 0 and word in msg, (r, rc, msg)
        if r == root:
            assert all(np.array_equal(a, b) for a, b in zip(ok, tabs))
