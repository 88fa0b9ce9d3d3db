"""RCCL without PyTorch: the multi-GPU exchanges of the sweep over libcloops_comm.so (include/cloops_comm.h).

One process per GPU, launched by anything that sets RANK / WORLD_SIZE / LOCAL_RANK (torch.distributed.run does; it is only
the launcher -- this module imports no torch).  The ncclUniqueId travels from rank 0 to the others through a file in /tmp
named after the launcher instance (pid + start time), MASTER_PORT and the communicator's number (one node, which is what the
path is specified for).

    comm = Comm.from_env()
    allsum = comm.make_allsum()          # for cloops_amd.pipe.runSweepFast(..., allsum=allsum)
    tables = comm.gather_tables(rows, dst=0)

The reference shape: joblib workers + merge in the parent (cLoops/pipe.py:113-127), the cut estimated from all chromosomes'
distance lists between the steps (pipe.py:247-275)."""
import ctypes
import os
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libcloops_comm.so")
SYMBOLS = ("cl_comm_last_error", "cl_comm_rccl_version", "cl_comm_unique_id", "cl_comm_init", "cl_comm_destroy", "cl_comm_rank", "cl_comm_world",
           "cl_comm_allreduce_f64", "cl_comm_allreduce_max_f64", "cl_comm_allgather_i32", "cl_comm_gather_i32", "cl_comm_barrier",
           "cl_comm_host_alloc", "cl_comm_host_free", "cl_comm_gather_i32_pinned", "cl_comm_gather_device", "cl_comm_allreduce_f64_device")
ID_BYTES = 128
ID_DIR = "/tmp"
_lib = None
_SEQ = 0               # communicators formed by this process so far (every rank forms them in the same order)


def _parent_start():
    """start time of the launcher process (clock ticks since boot, /proc/<ppid>/stat field 22): with its pid it names ONE
    launcher instance, so a file left behind by an earlier launch that happened to get the same pid is never read"""
    try:
        with open("/proc/%d/stat" % os.getppid()) as fh:
            return fh.read().rsplit(")", 1)[1].split()[19]
    except Exception:
        return "0"


def default_tag():
    """names the id file of one communicator of one launch: the launcher's pid and start time (all local ranks share the
    parent), MASTER_PORT, under torch.distributed.run its run id and restart count, and the number of communicators this
    process has formed before -- neither concurrent launches, nor a restarted worker group of the same agent, nor a second
    communicator of the same ranks meet each other's file"""
    nonce = os.environ.get("CLOOPS_COMM_NONCE", "")
    return "%s_%s_%s_%s_%s_%d%s" % (os.getppid(), _parent_start(), os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"),
                                    os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"), _SEQ, ("_" + nonce) if nonce else "")


class CommError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError("libcloops_comm.so is missing (%s): build it with `python -m cloops_amd.build`" % SO_PATH)
    _lib = _declare(ctypes.CDLL(SO_PATH))
    return _lib


def _declare(lib):
    """the prototypes of include/cloops_comm.h on a loaded library (tests load a build of cloops_comm.cpp against a host-memory
    test double of HIP / RCCL through this, tests/test_comm_fake_world.py)"""
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    lib.cl_comm_last_error.restype = ctypes.c_char_p
    lib.cl_comm_rccl_version.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    lib.cl_comm_unique_id.argtypes = [vp]
    lib.cl_comm_init.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(vp)]
    lib.cl_comm_destroy.argtypes = [vp]
    lib.cl_comm_destroy.restype = None
    lib.cl_comm_rank.argtypes = [vp]
    lib.cl_comm_world.argtypes = [vp]
    lib.cl_comm_allreduce_f64.argtypes = [vp, vp, i64]
    lib.cl_comm_allreduce_max_f64.argtypes = [vp, vp, i64]
    lib.cl_comm_allgather_i32.argtypes = [vp, vp, i64, vp]
    lib.cl_comm_gather_i32.argtypes = [vp, vp, i64, ctypes.c_int, vp]
    lib.cl_comm_barrier.argtypes = [vp]
    lib.cl_comm_host_alloc.restype = vp
    lib.cl_comm_host_alloc.argtypes = [i64]
    lib.cl_comm_host_free.restype = None
    lib.cl_comm_host_free.argtypes = [vp]
    lib.cl_comm_gather_i32_pinned.argtypes = [vp, vp, i64, ctypes.c_int, vp]
    lib.cl_comm_gather_device.argtypes = [vp, vp, vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int, vp, i64, vp]
    lib.cl_comm_allreduce_f64_device.argtypes = [vp, vp, i64, vp]
    return lib


def _check(rc):
    if rc != 0:
        raise CommError((load().cl_comm_last_error() or b"").decode("utf-8", "replace"))


def id_path(tag=None, directory=ID_DIR):
    return os.path.join(directory, "cloops_comm_id_%s" % (tag or default_tag()))


def rccl_versions():
    """-> (built_with, loaded) RCCL version codes of libcloops_comm.so / of the librccl mapped into this process"""
    lib = load()
    b, l = ctypes.c_int(0), ctypes.c_int(0)
    _check(lib.cl_comm_rccl_version(ctypes.byref(b), ctypes.byref(l)))
    return int(b.value), int(l.value)


STALE_ID_SKEW_S = 120.0
STALE_ID_GRACE_S = 30.0


def _process_start_time():
    """wall-clock start of this process (from /proc; now, if that cannot be read)"""
    try:
        with open("/proc/self/stat") as fh:
            ticks = float(fh.read().rsplit(")", 1)[1].split()[19])
        with open("/proc/uptime") as fh:
            up = float(fh.read().split()[0])
        return time.time() - (up - ticks / os.sysconf("SC_CLK_TCK"))
    except Exception:
        return time.time()


def exchange_id(rank, world, make_id, tag=None, timeout=300.0, directory=ID_DIR):
    """rank 0 makes the id (make_id() -> bytes) and publishes it; every rank returns the same bytes.  The file name carries
    default_tag(), which names this launcher instance and this communicator: whatever is found under it was written for it."""
    if world == 1:
        return make_id()
    path = id_path(tag, directory)
    if rank == 0:
        try:
            os.remove(path)                                # a file an earlier launch under the same tag left behind (it died between
        except OSError:                                    # publishing and the unlink behind its first barrier)
            pass
        blob = make_id()
        tmp = path + ".tmp%d" % os.getpid()
        with open(tmp, "wb") as fh:
            fh.write(blob)
        os.replace(tmp, path)                              # atomic: readers see nothing or all of it
        return blob
    t0 = time.time()
    # a file much older than THIS process cannot be for it (launchers whose ranks are children of a long-lived parent -- a shell
    # loop, a notebook -- give two successive launches the same tag): the window is generous because ranks may start seconds apart
    # A launcher that exports CLOOPS_COMM_NONCE (any string unique to the launch, the same on every rank) names its file exactly: no
    # clock is consulted.  Without one the age check applies, and a full-size file that STAYS too old is reported after STALE_ID_GRACE_S
    # instead of being waited for until the timeout (rank 0 replaces a leftover as the first thing it does: a file still stale after
    # that long is either a leftover of a launch whose rank 0 never came, or this rank started minutes after rank 0 published).
    exact = bool(os.environ.get("CLOOPS_COMM_NONCE")) and tag is None
    oldest = _process_start_time() - STALE_ID_SKEW_S
    stale_since = None
    while True:
        try:
            with open(path, "rb") as fh:
                blob = fh.read()
                fresh = exact or os.fstat(fh.fileno()).st_mtime >= oldest
            if len(blob) == ID_BYTES and fresh:
                return blob
            if len(blob) == ID_BYTES:
                stale_since = stale_since or time.time()
                if time.time() - stale_since > STALE_ID_GRACE_S:
                    raise CommError("the id file %s is older than this process by more than %.0f s and rank 0 has not replaced it within %.0f s: a leftover "
                                    "of an earlier launch under the same tag, or this rank started long after rank 0 -- export CLOOPS_COMM_NONCE=<unique per "
                                    "launch> on every rank to name the file exactly" % (path, STALE_ID_SKEW_S, STALE_ID_GRACE_S))
        except (IOError, OSError):
            stale_since = None
        if time.time() - t0 > timeout:
            raise CommError("no unique id from rank 0 within %.0f s (%s)" % (timeout, path))
        time.sleep(0.01)


class Comm(object):
    def __init__(self, rank, world, device, tag=None):
        lib = load()
        self._lib = lib
        self.rank, self.world, self.device = int(rank), int(world), int(device)

        def make_id():
            buf = ctypes.create_string_buffer(ID_BYTES)
            _check(lib.cl_comm_unique_id(buf))
            return buf.raw
        global _SEQ
        tag = tag or default_tag()
        self._id_path = id_path(tag)
        _SEQ += 1
        built, loaded = rccl_versions()
        self.rccl_built, self.rccl_loaded = built, loaded
        if self.world > 1 and built // 100 != loaded // 100:
            # (decided from facts every rank sees alike, before anything is exchanged: all ranks raise together)
            raise CommError("libcloops_comm.so was built against RCCL %d but the process has librccl %d mapped (another copy was "
                            "loaded first, e.g. by PyTorch): refusing to form a %d-rank communicator across that skew" % (built, loaded, self.world))
        blob = exchange_id(self.rank, self.world, make_id, tag)
        h = ctypes.c_void_p()
        _check(lib.cl_comm_init(ctypes.c_char_p(blob), self.rank, self.world, self.device, ctypes.byref(h)))
        self._h = h
        if self.world > 1:
            self.barrier()
            if self.rank == 0:                             # everyone has joined: the id file has done its job
                try:
                    os.remove(self._id_path)
                except OSError:
                    pass

    @classmethod
    def from_env(cls, device=None):
        rank = int(os.environ.get("RANK", "0"))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        return cls(rank, world, local if device is None else device)

    def close(self):
        """frees the communicator and its page-locked buffers -- views handed out by gather_tables(copy=False) die with them"""
        for k in ("_pin_send", "_pin_recv"):
            buf = getattr(self, k, None)
            if buf is not None:
                self._lib.cl_comm_host_free(ctypes.c_void_p(buf[0]))
                setattr(self, k, None)
        for p in getattr(self, "_retired", []):
            self._lib.cl_comm_host_free(ctypes.c_void_p(p))
        self._retired = []
        if getattr(self, "_h", None):
            self._lib.cl_comm_destroy(self._h)
            self._h = None

    def _pinned(self, which, nbytes):
        """a page-locked int32 buffer of at least nbytes, kept with the communicator (grown on demand) -> numpy view"""
        buf = getattr(self, which, None)
        if buf is None or buf[1] < nbytes:
            if buf is not None:
                # views of the old buffer may still be out (gather_tables(copy=False)): it is kept until close()
                if not hasattr(self, "_retired"):
                    self._retired = []
                self._retired.append(buf[0])
            want = int(nbytes + nbytes // 4 + 4096)
            p = self._lib.cl_comm_host_alloc(want)
            if not p:
                raise CommError("cl_comm_host_alloc(%d) failed" % want)
            buf = (p, want)
            setattr(self, which, buf)
        return np.ctypeslib.as_array(ctypes.cast(buf[0], ctypes.POINTER(ctypes.c_int32)), shape=(buf[1] // 4,))

    def barrier(self):
        _check(self._lib.cl_comm_barrier(self._h))

    def allmax(self, x):
        a = np.asarray([x], dtype=np.float64)
        _check(self._lib.cl_comm_allreduce_max_f64(self._h, a.ctypes.data_as(ctypes.c_void_p), 1))
        return float(a[0])

    def allsum(self, a):
        """element-wise sum over the ranks of a small int64 / float64 array (counts ride as float64: exact below 2^53)"""
        a = np.asarray(a)
        f = np.ascontiguousarray(a, dtype=np.float64).ravel().copy()
        _check(self._lib.cl_comm_allreduce_f64(self._h, f.ctypes.data_as(ctypes.c_void_p), f.size))
        f = f.reshape(a.shape)
        return np.rint(f).astype(a.dtype) if a.dtype.kind in "iu" else f.astype(a.dtype, copy=False)

    def make_allsum(self):
        return self.allsum

    def gather_device(self, ptrs, rows, cols=4, dst=0, copy=True):
        """DEVICE-RESIDENT tables (api.Chromosome.cand_finish_device: pointer + rows, one per chromosome of this rank) to rank
        `dst`: exact sizes, ranks in order, one device-to-host copy at `dst` (cl_comm_gather_device) -> list of per-rank int32
        [K_r, cols] arrays on `dst` (empty tables elsewhere).  copy=False: views of the communicator's page-locked buffer."""
        keep = [(int(p), int(k)) for p, k in zip(ptrs, rows) if int(k) > 0]
        nt = len(keep)
        tp = (ctypes.c_void_p * max(nt, 1))(*[p for p, _ in keep])
        tr = (ctypes.c_int64 * max(nt, 1))(*[k for _, k in keep])
        per = (ctypes.c_int64 * self.world)()
        mine = sum(k for _, k in keep)
        # (the receive buffer must exist before the counts are known on this rank: sized by a first exchange of the counts when it
        #  is too small -- the library all-gathers them again, a few microseconds)
        recv = self.rank == dst
        need_rows = mine if self.world == 1 else None
        if need_rows is None:
            cnt = np.zeros(self.world, dtype=np.int32)
            _check(self._lib.cl_comm_allgather_i32(self._h, np.asarray([mine], np.int32).ctypes.data_as(ctypes.c_void_p), 1, cnt.ctypes.data_as(ctypes.c_void_p)))
            need_rows = int(cnt.astype(np.int64).sum())
        out = self._pinned("_pin_recv", max(need_rows, 1) * cols * 4) if recv else None
        outp = out.ctypes.data_as(ctypes.c_void_p) if recv else None
        _check(self._lib.cl_comm_gather_device(self._h, tp, tr, nt, int(cols), int(dst), outp, int(need_rows), per))
        if not recv:
            return [np.zeros((0, cols), np.int32) for _ in range(self.world)]
        res, at = [], 0
        for r in range(self.world):
            k = int(per[r])
            v = out[at * cols:(at + k) * cols].reshape(k, cols)
            res.append(v.copy() if copy else v)
            at += k
        return res

    def allsum_device(self, dev_ptr, n, stream=None):
        """element-wise sum over the ranks of n float64 in place in DEVICE memory (cl_comm_allreduce_f64_device)"""
        _check(self._lib.cl_comm_allreduce_f64_device(self._h, ctypes.c_void_p(dev_ptr), int(n), ctypes.c_void_p(stream)))

    def gather_tables(self, table, dst=None, copy=True):
        """variable-length int32 [K_r, C] tables from every rank -> list of per-rank arrays (on `dst` only when given; the
        other ranks get empty tables).  `table` may be a list of tables (their concatenation).  Two collectives: the row
        counts, then the rows padded to the longest table.  copy=False hands out VIEWS of the communicator's page-locked
        receive buffer: the next gather overwrites them, close() frees them.)"""
        tabs = [np.asarray(t) for t in table] if isinstance(table, (list, tuple)) else [np.asarray(table)]
        if not tabs or any(t.ndim != 2 for t in tabs) or len({t.shape[1] for t in tabs}) > 1:
            raise ValueError("tables must be a non-empty list of [K, C] arrays with one C")
        k, c = sum(len(t) for t in tabs), tabs[0].shape[1]
        mine = np.asarray([k, c], dtype=np.int32)
        ks = np.zeros(2 * self.world, dtype=np.int32)
        _check(self._lib.cl_comm_allgather_i32(self._h, mine.ctypes.data_as(ctypes.c_void_p), 2, ks.ctypes.data_as(ctypes.c_void_p)))
        ks = ks.reshape(self.world, 2)
        if len({int(x) for x in ks[:, 1]}) != 1:
            raise ValueError("the ranks' tables differ in their number of columns")
        kmax = max(int(ks[:, 0].max()), 1)
        # the tables go straight into a page-locked send buffer (one copy, no concatenation), the rows of all ranks arrive in a
        # page-locked receive buffer (views of it are handed out with copy=False: valid until the next gather)
        pad = self._pinned("_pin_send", kmax * c * 4)[: kmax * c].reshape(kmax, c)
        at = 0
        for t in tabs:
            if len(t):
                pad[at:at + len(t)] = t
                at += len(t)
        pad[at:] = 0
        recv = dst is None or dst == self.rank
        out = self._pinned("_pin_recv", self.world * kmax * c * 4)[: self.world * kmax * c].reshape(self.world, kmax, c) if recv else None
        outp = out.ctypes.data_as(ctypes.c_void_p) if recv else None
        _check(self._lib.cl_comm_gather_i32_pinned(self._h, pad.ctypes.data_as(ctypes.c_void_p), kmax * c, -1 if dst is None else int(dst), outp))
        if not recv:
            return [np.zeros((0, c), np.int32) for _ in range(self.world)]
        return [out[r, : int(ks[r, 0])].copy() if copy else out[r, : int(ks[r, 0])] for r in range(self.world)]
