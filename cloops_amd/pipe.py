"""Per-chromosome dispatch of the clustering hot path -- drop-in for the reference's
cLoops/pipe.py:52-174 (`singleDBSCAN`, `runDBSCAN`, `filterClusterByDis`, `checkSameLoop`,
`combineTwice`) plus the (eps, minPts) sweep loop of cLoops/pipe.py:241-281 (`runSweep`).

Same names, argument meaning, return shapes and stderr report lines as the reference.
What differs is where the work happens:

  * a chromosome's .jd matrix is loaded ONCE and stays resident in HBM across the sweep
    (the reference re-reads the file in every step, io.py:206-217 via pipe.py:58);
  * the cut filter (pipe.py:59-63), the clustering (pipe.py:70) and the per-cluster
    bounding boxes (the O(K*N) pandas loop of pipe.py:78-102) run on the GPU;
  * `runDBSCAN` spreads chromosomes over the visible GPUs (one host thread per GPU) instead
    of joblib worker processes (pipe.py:117).
"""
import os
import sys
import threading
import time
from collections import OrderedDict
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import api
from . import _roctx
from .cDBSCAN2 import cDBSCAN as DBSCAN          # pipe.py:42  (production variant)
from .dist import lpt_assign
from .ests import estIntSelCutFrag, estIntSelCutFrag_from_stats

#: clustering variant used by singleDBSCAN; "block" mirrors the alternative import at pipe.py:43
DBSCAN_VARIANT = "v2"


def parseJd(f, cut=0):
    """cLoops/io.py:206-217: `.jd` = joblib-pickled int64 [n,3] rows [id, X, Y]; key from the
    file name 'chrA-chrB.jd'."""
    import joblib
    key = os.path.split(f)[1].replace(".jd", "")
    key = tuple(key.split("-"))
    mat = joblib.load(f)
    if cut > 0:
        d = mat[:, 2] - mat[:, 1]
        p = np.where(d >= cut)[0]
        mat = mat[p, :]
    return key, mat


# ---------------------------------------------------------------------------------------
# resident chromosomes
# ---------------------------------------------------------------------------------------
#: The chromosomes of one GPU share this many HIP streams (0: every chromosome gets a stream of its own).  A sweep step
#: enqueues every chromosome's run and then collects them; how many of those runs execute side by side is what the streams
#: decide.  One stream per chromosome leaves that to how the HIP runtime maps 23+ streams onto its hardware queues
#: (GPU_MAX_HW_QUEUES, default 4): measured on the 200 M-PET mode-3 sweep 1 queue 0.270 s, 2: 0.190, 3: 0.175, 4: 0.195,
#: 8: 0.174, 12+: 0.24-0.32 -- and the forms that copy labels to the host every run fell off a cliff at 8 (DESIGN.md
#: section 8).  Three shared streams, filled largest chromosome first, are the measured optimum made explicit: the same
#: schedule whatever the environment says.
SWEEP_STREAMS = 3


class _StreamPool(object):
    """per device: up to SWEEP_STREAMS library-made streams (cl_stream_create), handed out to the least loaded one; they
    live as long as the process (handles may outlive the cache entry that made them)"""

    def __init__(self):
        self._by_dev = {}
        self._lock = threading.Lock()

    def pick(self, device, n):
        if SWEEP_STREAMS <= 0:
            return None, None
        from . import _lib
        with self._lock:
            slots = self._by_dev.setdefault(device, [])
            if len(slots) < SWEEP_STREAMS:
                ptr = _lib.load().cl_stream_create(int(device))
                if not ptr:                               # (no such device: the handle's own creation reports it)
                    return None, None
                slots.append([ptr, 0])
            slot = min(slots, key=lambda s: s[1])
            slot[1] += max(1, int(n))
            return slot[0], slot

    @staticmethod
    def release(slot, n):
        if slot is not None:
            slot[1] -= max(1, int(n))


    def rebalance(self, chroms):
        """`chroms` = the idle handles of one sweep [(chromosome made by _make_chrom, n PETs)]: dealt again over their device's
        shared streams, largest first (LPT).  A handle is bound to a stream when it is uploaded -- in arrival order, before anyone
        knows which chromosomes a sweep will take: chr1 .. chrX of a 200 M genome arrive as 72 / 63 / 65 M PETs on the three
        streams, and a step lasts as long as its fullest stream."""
        if SWEEP_STREAMS <= 0:
            return
        by_dev = {}
        for ch, n in chroms:
            slot = getattr(ch.close, "slot", None)
            if slot is not None:
                by_dev.setdefault(ch.device, []).append((ch, max(1, int(n)), slot))
        with self._lock:
            for dev, items in by_dev.items():
                slots = self._by_dev.get(dev, [])
                if len(slots) < 2:
                    continue
                load = {id(s): 0 for s in slots}
                for ch, n, slot in sorted(items, key=lambda t: -t[1]):
                    best = min(slots, key=lambda s: load[id(s)])
                    load[id(best)] += n
                    if best is not slot:
                        ch.set_stream(best[0])
                        slot[1] -= n
                        best[1] += n
                        ch.close.slot = best


STREAMS = _StreamPool()


def _make_chrom(X, Y, device):
    """a resident chromosome on one of the device's shared streams (its close() also gives the stream's slot back) -> chromosome"""
    stream, slot = STREAMS.pick(device, len(X))
    try:
        ch = api.Chromosome(X, Y, device=device, stream=stream) if stream else api.Chromosome(X, Y, device=device)
    except Exception:
        _StreamPool.release(slot, len(X))
        raise
    close = ch.close

    def close_and_release():
        close()
        _StreamPool.release(close_and_release.slot, len(X))
        close_and_release.slot = None
    close_and_release.slot = slot
    ch.close = close_and_release
    return ch
class _Resident(object):
    """a chromosome resident in HBM + its host copies.  `ids` (the row ids of cLoops/io.py:181-183) and `d` (the distances Y - X the
    reference-shaped wrappers hand out, pipe.py:59-63) are made on FIRST USE: the sweep on the device needs neither, and for a 200 M-PET
    genome they are 3.2 GB of int64 that took as long to fill as the upload itself."""
    __slots__ = ("key", "_ids", "X", "Y", "_d", "chrom", "device", "stamp", "lock", "pins", "sweep_lock", "replaced")

    @property
    def ids(self):
        if self._ids is None:
            self._ids = np.arange(len(self.X), dtype=np.int64)
        return self._ids

    @ids.setter
    def ids(self, v):
        self._ids = v

    @property
    def d(self):
        if self._d is None:
            self._d = self.Y.astype(np.int64) - self.X.astype(np.int64)
        return self._d

    @d.setter
    def d(self, v):
        self._d = v

    def __len__(self):
        return len(self.X)


class ChromCache(object):
    """path -> chromosome resident in HBM (+ host copies of ids and distances).

    `max_items` bounds the number of IDLE residents kept around; a resident that is pinned (a sweep holds it,
    `pinned()`) or whose lock is held is never evicted, so a sweep over any number of chromosomes keeps all of
    its handles alive (the reference has no such limit: it re-reads every .jd in every step)."""

    def __init__(self, max_items=64):
        self._items = OrderedDict()
        self._lock = threading.Lock()
        self.max_items = max_items

    def _evict_idle(self):
        """called with self._lock held: drop the oldest residents that nobody uses while over the limit"""
        if len(self._items) <= self.max_items:
            return []
        out = []
        for f in list(self._items.keys()):
            if len(self._items) - len(out) <= self.max_items:
                break
            r = self._items[f]
            if r.pins > 0 or r.lock.locked():
                continue
            out.append(f)
        return [self._items.pop(f) for f in out]

    def pinned(self, fs, devices=None):
        """context manager: the residents of `fs` (loaded on demand, file k on devices[k % len]; None = wherever it
        already is), pinned against eviction until the block ends"""
        cache = self

        class _Pinned(object):
            def __enter__(self_inner):
                self_inner.rs = []
                try:
                    for k, f in enumerate(fs):
                        dev = None if devices is None else devices[k % len(devices)]
                        self_inner.rs.append(cache.get(f, dev, _pin=True))
                except Exception:
                    self_inner.__exit__(None, None, None)
                    raise
                return self_inner.rs

            def __exit__(self_inner, *exc):
                with cache._lock:
                    for r in self_inner.rs:
                        r.pins -= 1
                    victims = cache._evict_idle()
                    # a resident that was replaced in the cache while a sweep held it is closed by its last user
                    victims += [r for r in self_inner.rs if r.replaced and r.pins == 0]
                    for r in victims:
                        r.replaced = False
                for v in victims:
                    v.chrom.close()
                return False
        return _Pinned()

    def put_arrays(self, name, X, Y, device=0, ids=None):
        """Register an in-memory chromosome under the pseudo path 'mem://<chrA>-<chrB>' (no .jd
        file, no disk): what a direct BEDPE -> HBM loader hands to the sweep."""
        f = "mem://" + name
        r = _Resident()
        r.key = tuple(name.split("-")) if "-" in name else (name, name)
        r.stamp, r.device = ("mem", len(X)), device
        r.lock = threading.Lock()
        r.sweep_lock = threading.Lock()
        r.replaced = False
        r.pins = 0
        r.X = np.ascontiguousarray(X)
        r.Y = np.ascontiguousarray(Y)
        r.ids = None if ids is None else np.asarray(ids)
        r.d = None
        r.chrom = _make_chrom(r.X, r.Y, device)
        r.chrom.set_device_labels(False)        # runs without a host destination (the sweep) skip the row-order scatter
        with self._lock:
            old = self._items.pop(f, None)
            self._items[f] = r
            if old is not None and old.pins > 0:
                old.replaced, old = True, None            # still held by a sweep: its last user closes it
        if old is not None:
            old.chrom.close()
        return f

    def put_chrom(self, name, chrom, X, Y, ids=None, key=None, device=0):
        """register a chromosome that is ALREADY resident (e.g. api.Chromosome.subsample of another resident) under the
        pseudo path `name` ('mem://...'); X, Y: its host copies (distances, ids for the reference-shaped wrappers)"""
        r = _Resident()
        tail = name[len("mem://"):].split("/")[-1]
        r.key = tuple(key) if key is not None else (tuple(tail.split("-")) if "-" in tail else (tail, tail))
        r.stamp, r.device = ("mem", len(X)), device
        r.lock = threading.Lock()
        r.sweep_lock = threading.Lock()
        r.replaced = False
        r.pins = 0
        r.X = np.ascontiguousarray(X)
        r.Y = np.ascontiguousarray(Y)
        r.ids = None if ids is None else np.asarray(ids)
        r.d = None
        r.chrom = chrom
        r.chrom.set_device_labels(False)
        with self._lock:
            old = self._items.pop(name, None)
            self._items[name] = r
            if old is not None and old.pins > 0:
                old.replaced, old = True, None
        if old is not None:
            old.chrom.close()
        return name

    def drop(self, f):
        """forget one resident (closed at once unless a sweep still holds it)"""
        with self._lock:
            r = self._items.pop(f, None)
            if r is not None and r.pins > 0:
                r.replaced, r = True, None
        if r is not None:
            r.chrom.close()

    def get(self, f, device=None, _pin=False):
        """the resident of `f`; `device=None` takes it wherever it already lives (GPU 0 if it has to be loaded),
        an explicit device reloads a chromosome that lives on another GPU"""
        if f.startswith("mem://"):
            with self._lock:
                r = self._items[f]
                r.pins += 1 if _pin else 0
                return r
        st = os.stat(f)
        stamp = (st.st_mtime_ns, st.st_size)
        with self._lock:
            r = self._items.get(f)
            if r is not None and r.stamp == stamp and (device is None or r.device == device):
                self._items.move_to_end(f)
                r.pins += 1 if _pin else 0
                return r
        device = 0 if device is None else device
        key, mat = parseJd(f, cut=0)
        mat = np.asarray(mat)
        r = _Resident()
        r.key, r.stamp, r.device = key, stamp, device
        r.lock = threading.Lock()
        r.sweep_lock = threading.Lock()
        r.replaced = False
        r.pins = 1 if _pin else 0
        if len(mat):
            r.ids = mat[:, 0]
            r.X = np.ascontiguousarray(mat[:, 1])
            r.Y = np.ascontiguousarray(mat[:, 2])
        else:
            r.ids = r.X = r.Y = np.zeros(0, np.int64)
        r.d = r.Y - r.X
        r.chrom = _make_chrom(r.X, r.Y, device)
        r.chrom.set_device_labels(False)
        with self._lock:
            old = self._items.pop(f, None)
            self._items[f] = r
            victims = self._evict_idle()
            if old is not None and old.pins > 0:
                old.replaced, old = True, None            # still held by a sweep: its last user closes it
        for v in victims:
            v.chrom.close()
        if old is not None:
            old.chrom.close()
        return r

    def clear(self):
        with self._lock:
            for r in self._items.values():
                r.chrom.close()
            self._items.clear()


CACHE = ChromCache()


def _cluster_arrays(res_chrom, eps, minPts, cut, variant=None):
    """One chromosome, one (eps, minPts, cut): the array-level equivalent of pipe.py:57-110.

    Returns (dataI_boxes int64[kI,4], dataS_boxes int64[kS,4], dis float64[], dss float64[],
    n_readI, n_readS, n_in).  Boxes are (minX, maxX, minY, maxY) in ascending cluster id
    (the iteration order of `set(labels.values)`, pipe.py:76-78)."""
    r = res_chrom
    variant = variant or DBSCAN_VARIANT
    d = r.d
    dss_parts = []
    n_in = len(d)
    if cut > 0:                                           # pipe.py:59-63
        short = d < cut
        dss_parts.append(d[short].astype(np.float64))
        n_in = int(len(d) - short.sum())
    empty4 = np.zeros((0, 4), np.int64)
    if n_in == 0:                                         # pipe.py:64-65
        return empty4, empty4, np.zeros(0), (dss_parts[0] if dss_parts else np.zeros(0)), 0, 0, 0
    with r.lock:
        res = r.chrom.cluster(variant, eps, minPts, cut, pinned=True)
        lab = res.labels
        b = res.boxes
        K = len(b)
        cls = np.zeros(K + 1, np.int8)                    # 0 skip, 1 inter, 2 self; slot K for noise
        if K:
            ok = (b["count"] > 0) & (b["min_x"] != b["max_x"]) & (b["min_y"] != b["max_y"])   # pipe.py:83-85
            inter = ok & (b["max_x"] < b["min_y"])                                             # pipe.py:97
            cls[:K][inter] = 1
            cls[:K][ok & ~inter] = 2
        pc = cls[np.where(lab >= 0, lab, K)]
        selI = pc == 1
        selS = pc == 2
        dis = d[selI].astype(np.float64)                  # pipe.py:106-107 (row order)
        dss_parts.append(d[selS].astype(np.float64))      # pipe.py:108-109
    box = np.stack([b["min_x"], b["max_x"], b["min_y"], b["max_y"]], 1).astype(np.int64) if K else empty4
    dataI = box[cls[:K] == 1]
    dataS = box[cls[:K] == 2]
    dss = np.concatenate(dss_parts) if len(dss_parts) > 1 else dss_parts[0]
    return dataI, dataS, dis, dss, int(selI.sum()), int(selS.sum()), n_in


def _records(key, boxes):
    return [[key[0], int(x0), int(x1), key[1], int(y0), int(y1)] for x0, x1, y0, y1 in boxes]


def singleDBSCAN(f, eps, minPts, cut=0, device=None):
    """Run DBSCAN to detect interactions for one chromosome (cLoops/pipe.py:52-110).

    Returns (key, f, dataI, dataS, dis, dss) exactly shaped like the reference: record rows
    [chrA, minX, maxX, chrB, minY, maxY]; dis/dss are lists of float distances."""
    r = CACHE.get(f, device)
    key = r.key
    if cut > 0 and not (r.d >= cut).any() or len(r.d) == 0:
        dss = list(r.d[r.d < cut].astype(np.float64)) if cut > 0 else []
        return key, f, [], [], [], dss
    report = "Clustering %s and %s using eps as %s, minPts as %s,pre-set distance cutoff as > %s\n" % (
        key[0], key[1], eps, minPts, cut)
    sys.stderr.write(report)
    dataI, dataS, dis, dss, nI, nS, _ = _cluster_arrays(r, eps, minPts, cut)
    report = "Clustering %s and %s finished. Estimated %s self-ligation reads and %s inter-ligation reads\n" % (
        key[0], key[1], nS, nI)
    sys.stderr.write(report)
    return key, f, _records(key, dataI), _records(key, dataS), list(dis), list(dss)


def _devices():
    env = os.environ.get("CLOOPS_DEVICES")
    if env:
        return [int(x) for x in env.split(",") if x != ""]
    return list(range(max(1, api.device_count())))


def _run_many(fs, eps, minPts, cut, fn):
    """fn(f, eps, minPts, cut, device) for every file: chromosomes LPT-assigned to the
    visible GPUs, one host thread per GPU; results in `fs` order."""
    devs = _devices()
    if len(devs) <= 1 or len(fs) <= 1:
        return [fn(f, eps, minPts, cut, devs[0]) for f in fs]
    sizes = [len(CACHE.get(f)) if f.startswith("mem://") else os.path.getsize(f) for f in fs]
    parts = lpt_assign(sizes, len(devs))
    out = [None] * len(fs)

    def work(k):
        for i in parts[k]:
            out[i] = fn(fs[i], eps, minPts, cut, devs[k])
    with ThreadPoolExecutor(max_workers=len(devs)) as ex:
        list(ex.map(work, range(len(devs))))
    return out


def runDBSCAN(fs, eps, minPts, cut=0, cpu=1):
    """Run DBSCAN to detect interactions for all chromosomes (cLoops/pipe.py:113-127).
    `cpu` is accepted for signature compatibility; parallelism is over GPUs."""
    ds = _run_many(fs, eps, minPts, cut, singleDBSCAN)
    dataI, dataS, dis, dss = {}, [], [], []
    for d in ds:
        if len(d[2]) == 0:
            continue
        dataI[d[0]] = {"f": d[1], "records": d[2]}
        dataS.extend(d[3])
        dis.extend(d[4])
        dss.extend(d[5])
    return dataI, dataS, dis, dss


def filterClusterByDis(data, cut):
    """Filter inter-ligation clusters by distances (cLoops/pipe.py:130-143): keep a record when the distance
    between its anchor mid-points is >= cut.  The reference is Python 2: `/` on the int mid-points is FLOOR
    division (pipe.py:138).  Mutates and returns `data` like the reference."""
    def mid_distance(rec):
        return (rec[4] + rec[5]) // 2 - (rec[1] + rec[2]) // 2
    for entry in data.values():
        entry["records"] = [rec for rec in entry["records"] if mid_distance(rec) >= cut]
    return data


def checkSameLoop(ra, rb):
    """check if two anchors are exact same (cLoops/pipe.py:146-152)."""
    return (ra[1], ra[2], ra[4], ra[5]) == (rb[1], rb[2], rb[4], rb[5])


def combineTwice(dataI, dataI_2):
    """Combining multiple clustering result (cLoops/pipe.py:155-174): records of `dataI_2` whose exact box is not
    already in `dataI` are appended (the set of known boxes is taken BEFORE appending, so duplicates inside
    `dataI_2` all come through, like the reference)."""
    box = lambda rec: (rec[1], rec[2], rec[4], rec[5])
    for key, new in dataI_2.items():
        have = dataI.get(key)
        if have is None:
            dataI[key] = {"f": new["f"], "records": new["records"]}
            continue
        known = {box(rec) for rec in have["records"]}
        have["records"].extend(rec for rec in new["records"] if box(rec) not in known)
    return dataI


def _single_arrays(f, eps, minPts, cut, device):
    r = CACHE.get(f, device)
    if len(r.d) == 0:
        e4 = np.zeros((0, 4), np.int64)
        return r.key, f, e4, e4, np.zeros(0), np.zeros(0), 0
    dataI, dataS, dis, dss, nI, nS, n_in = _cluster_arrays(r, eps, minPts, cut)
    return r.key, f, dataI, dataS, dis, dss, n_in


def _boxes_classified(r, res):
    """host part of pipe.py:78-102 on the K-row cluster table: (dataI boxes, dataS boxes) as
    int32 [k,4] arrays (minX, maxX, minY, maxY) in ascending cluster id"""
    b = res.boxes
    K = len(b)
    empty4 = np.zeros((0, 4), np.int32)
    if K == 0:
        return empty4, empty4
    t = b.view(np.int32).reshape(K, 5)                     # rows {minX, maxX, minY, maxY, count}
    ok = (t[:, 4] > 0) & (t[:, 0] != t[:, 1]) & (t[:, 2] != t[:, 3])      # pipe.py:83-85
    inter = ok & (t[:, 1] < t[:, 2])                                       # pipe.py:97
    return t[inter, :4], t[ok & ~inter, :4]              # copies (fancy index): the pinned view may be reused


def _combine_steps(step_boxes, _hash_bits=None):
    """combineTwice (pipe.py:155-174) applied over all steps at once, on arrays: a box is kept
    in the step where it FIRST appears (duplicates inside one step are all kept, like the
    reference, whose `ds` set is built before the loop).  step_boxes: list of int64[k,4].

    One value sort of `hash << ibits | position` (numpy's SIMD sort, no argsort): equal boxes become
    one run whose first element carries the smallest position, i.e. the first step."""
    step_boxes = [b for b in step_boxes if len(b)]
    if not step_boxes:
        return np.zeros((0, 4), np.int64)
    if len(step_boxes) == 1:
        return step_boxes[0]
    rows = np.concatenate(step_boxes)
    n = len(rows)
    step = np.repeat(np.arange(len(step_boxes), dtype=np.int32), [len(b) for b in step_boxes])
    ibits = max(1, int(n - 1).bit_length())
    u = rows.astype(np.uint64)
    h = (u[:, 0] * np.uint64(0x9E3779B97F4A7C15)) ^ (u[:, 1] * np.uint64(0xC2B2AE3D27D4EB4F)) \
        ^ (u[:, 2] * np.uint64(0x165667B19E3779F9)) ^ (u[:, 3] * np.uint64(0xD6E8FEB86659FD93))
    h ^= h >> np.uint64(29)
    if _hash_bits:                                                        # tests: force hash collisions
        h = (h & np.uint64((1 << _hash_bits) - 1)) << np.uint64(64 - _hash_bits)
    key = ((h >> np.uint64(ibits)) << np.uint64(ibits)) | np.arange(n, dtype=np.uint64)
    key.sort()
    pos = (key & np.uint64((1 << ibits) - 1)).astype(np.int64)          # original positions, grouped by hash
    hs = key >> np.uint64(ibits)
    head = np.empty(n, dtype=bool)
    head[0] = True
    np.not_equal(hs[1:], hs[:-1], out=head[1:])
    first_sorted = np.maximum.accumulate(np.where(head, np.arange(n), 0))    # index (in sorted order) of the run head
    first_pos = pos[first_sorted]                                        # smallest position of the run
    dup = np.flatnonzero(~head)                                          # elements of runs longer than 1
    keep = np.ones(n, dtype=bool)
    if len(dup) == 0:
        return rows
    keep[pos[dup]] = step[pos[dup]] == step[first_pos[dup]]
    # exact check: an element whose row differs from its run head shares only the (truncated) hash with
    # it.  Those few runs are redone exactly; everything else is settled.
    mism = dup[(rows[first_pos[dup]] != rows[pos[dup]]).any(axis=1)]
    if len(mism):
        bad = np.flatnonzero(np.isin(first_sorted, np.unique(first_sorted[mism])))
        bpos = pos[bad]
        _, inv = np.unique(rows[bpos], axis=0, return_inverse=True)
        inv = inv.ravel()
        first_step = np.full(int(inv.max()) + 1, np.iinfo(np.int32).max, np.int32)
        np.minimum.at(first_step, inv, step[bpos])
        keep[bpos] = step[bpos] == first_step[inv]
    return rows[keep]


def _pmap(pool, fn, items):
    """fn over items on the host thread pool (ctypes calls and numpy kernels release the GIL, so the
    per-chromosome waits / small synchronous statistics calls overlap instead of adding up)."""
    if pool is None or len(items) <= 1:
        return [fn(x) for x in items]
    return list(pool.map(fn, items))


def _select_kth(chroms, cut, loghist, ranks, allsum=None, pool=None, fine=None):
    """Exact order statistics (0-based `ranks`, ascending) of the self group's |d| over the union of the chromosomes
    (of all ranks).  `loghist` = the genome-wide log-binned histogram of cl_dist_summary (bins monotone in |d|): it
    locates the bin of a rank; the bin is then refined with 2048-bin histograms of (|d| - lo) >> shift summed over the
    chromosomes (and, with `allsum`, over the ranks) until single distances are resolved -- one refinement pass for
    bins up to 2048 distances wide (|d| < 2^18), two beyond."""
    from .ests import logbin_range, logbin
    cum = np.cumsum(np.asarray(loghist, dtype=np.int64))
    out, cache = [], {}
    if fine is not None and fine[0] >= 1:
        # `fine` = (lo, exact histogram of lo <= |d| < lo + 2048), lo a lower bin edge: a rank inside it needs no further pass
        b0 = logbin(fine[0])
        below = int(cum[b0 - 1]) if b0 > 0 else 0
        fcum = np.cumsum(fine[1])
    for rank in ranks:
        if fine is not None and fine[0] >= 1 and below <= rank < below + int(fcum[-1]):
            out.append(fine[0] + int(np.searchsorted(fcum, rank - below, side="right")))
            continue
        b = int(np.searchsorted(cum, rank, side="right"))
        rem = rank - (int(cum[b - 1]) if b > 0 else 0)
        lo, hi = logbin_range(b)
        while hi - lo > 1:
            shift = 0
            while ((hi - lo + (1 << shift) - 1) >> shift) > 2048:
                shift += 1
            key = (lo, hi, shift)
            if key not in cache:
                h = np.zeros(2048, dtype=np.int64)
                for hh in _pmap(pool, lambda r: r.chrom.dist_bin_hist(cut, lo, hi, shift), chroms):
                    h += hh
                cache[key] = allsum(h) if allsum is not None else h
            c = np.cumsum(cache[key])
            k = int(np.searchsorted(c, rem, side="right"))
            rem -= int(c[k - 1]) if k > 0 else 0
            lo, hi = lo + (k << shift), min(hi, lo + ((k + 1) << shift))
        out.append(lo)
    return out


SWEEP_THREADS = 8
#: enqueue the chromosomes of a sweep step from the pool's threads instead of one after the other from the calling thread.
#: Measured on the 200 M-PET mode-3 sweep: 0.212 s against 0.208 s -- the serial order (largest chromosome first) is worth more
#: than the 2 ms of host time the 23 enqueues take, so it stays off.
PARALLEL_ENQUEUE = False
STEP_TRACE = None        # developer hook (tools/step_gap.py): a list -> (step, t of the last wait that returned, t of the next step's first enqueue)


def _lib_logbins():
    from ._lib import DIST_LOGBINS
    return DIST_LOGBINS

#: |2**cut - nearest integer| below which runSweepFast re-derives the cut from the distance lists (ests.py:57 truncates)
CUT_RECHECK_MARGIN = 1e-6


class _DataI(dict):
    """runSweepFast's candidate tables; `.gathered` = what `device_consumer` returned"""
    gathered = None


def runSweepFast(fs, eps, minPts, cut=0, max_cut=False, log=None, variant=None, allsum=None, probe=None, forced_cuts=None, finish_device=False,
                 device_consumer=None):
    """runSweep with the per-step statistics reduced on the GPUs: neither labels nor distance
    lists come back to the host -- per chromosome only the K-row cluster table, a few sums, and
    the 256-bin histograms of an exact radix select for the median (all additive over chromosomes
    and over GPUs).  Same chain, same cuts, same candidate boxes as runSweep / the reference's
    pipe.py:241-281; candidate boxes stay numpy arrays:

    `allsum` (optional): element-wise global sum of a small numpy array over all ranks
    (cloops_amd.dist.make_allsum) -- with it every rank passes ITS chromosomes as `fs` and the
    chained cut is estimated from the genome-wide statistics; the per-step exchange is a few
    hundred bytes (6 sums, 2 squared deviations, 4-8 histograms of 256 bins).

    `probe` (optional): called as probe(f, eps, minPts, cut_in, result) for every completed run (bench.py reads
    the HIP-event kernel timings of profiled handles through it).

    `forced_cuts` (optional): one cut per step that REPLACES the cut estimated by that step (every step still does all
    of its work, the estimate included) -- bench.py replays the genome-wide chain on one emulated rank's share of the
    chromosomes with it.

    `finish_device`: the candidate tables stay on the device ({"f": f, "dev_rows": pointer, "n_rows": k, "boxes": None} per
    chromosome, valid until the handle's next sweep finishes): a multi-rank run hands them to comm.Comm.gather_device, which sends
    exact sizes to the merging rank without a detour through the host (cLoops/pipe.py:119-127 merges in the parent).

    `device_consumer` (with finish_device): called as device_consumer(dataI) WHILE the residents are still pinned in the cache and
    the sweep locks are held -- the device pointers cannot be evicted, freed or overwritten by another sweep before it returns; what
    it returns is `dataI.gathered` (bench.py: the RCCL gather of the tables to the merging rank).  Without a consumer the caller
    must use the pointers before anything else touches these handles.

    returns (dataI {key: {"f": f, "boxes": int32[k,4]}} of the local chromosomes, cut, cuts, steps)."""
    variant = variant or DBSCAN_VARIANT
    gsum = allsum if allsum is not None else (lambda a: a)
    devs = _devices()
    with CACHE.pinned(fs, devs) as res_all:
        return _sweep_fast(fs, res_all, eps, minPts, cut, max_cut, log, variant, allsum, gsum, probe, forced_cuts, finish_device, device_consumer)


def _sweep_fast(fs, res_all, eps, minPts, cut, max_cut, log, variant, allsum, gsum, probe=None, forced_cuts=None, finish_device=False,
                device_consumer=None):
    cuts = [cut]
    steps = []
    live = [(f, r) for f, r in zip(fs, res_all) if len(r)]
    by_size = sorted(live, key=lambda fr: -len(fr[1]))   # enqueue order: largest first (results are collected in file order)
    appended = {}                                        # f -> inter-ligation boxes appended on the device so far
    pool = ThreadPoolExecutor(max_workers=SWEEP_THREADS) if len(live) > 1 else None
    # the candidate buffer, the layout and the sort index of a handle are state of ONE sweep: a second sweep over the same
    # residents waits (locks taken in one global order, so two sweeps over overlapping sets cannot deadlock)
    held = sorted({id(r): r for _, r in live}.values(), key=id)
    for r in held:
        r.sweep_lock.acquire()
    try:
        # (a synchronous call on a handle -- cluster(), sig_counts() -- holds its r.lock: no stream is swapped under one)
        for r in held:
            r.lock.acquire()
        try:
            STREAMS.rebalance([(r.chrom, len(r)) for _, r in live])
        finally:
            for r in held:
                r.lock.release()
        for f, r in live:
            r.chrom.cand_reset()
            # the region query of the first run at an eps serves the later runs at that eps (their minPts are smaller: the
            # reference sorts them descending, pipe.py:316-320) -- its counts have to tell `count >= m` for every m of the list
            # (cl_sweep_plan: with several eps the q index pays from the first sort on and, if the values share a divisor, one fine
            #  sort serves all of them; the counts of an eps are bracketed for the whole minPts list)
            r.chrom.sweep_plan(eps, minPts)
        step_no = 0
        # where the summary should look for the next median (see _select_kth).  The first step has no earlier median to go by:
        # it counts the distances 1 .. 2048 exactly (self-ligation distances are a few hundred bp: ests.py's cut model) and falls
        # back to the refinement passes if the median lies beyond
        fine_lo = 1
        for ep in eps:
            for m in minPts:
                step_cut = cut
                this_step = step_no
                step_no += 1
                t_step0 = time.perf_counter()
                # one roctx range per step, from the enqueue to this rank's statistics on the host (rocprofv3 --marker-trace)
                rng = _roctx.range_("cloops sweep step %d: eps %d minPts %d cut %d" % (this_step, ep, m, step_cut))
                rng.__enter__()

                # chromosomes are independent inside a step: enqueue them all (each handle has its own
                # streams), then collect -- the kernels of different chromosomes overlap on the GPU
                # and the host-side collection runs on the pool
                def enqueue(fr):
                    f, r = fr
                    if STEP_TRACE is not None and fr is by_size[0]:
                        STEP_TRACE.append((this_step, "enqueue", time.perf_counter()))
                    r.lock.acquire()
                    try:
                        r.chrom.step_async(variant, ep, m, step_cut, this_step, fine_lo)
                    except Exception as e:
                        r.lock.release()
                        return e
                    return None

                errs = _pmap(pool if PARALLEL_ENQUEUE else None, enqueue, by_size)
                if any(e is not None for e in errs):
                    # a failed enqueue must not leave the other chromosomes locked with a run in flight
                    for (f, r), e in zip(by_size, errs):
                        if e is None:
                            try:
                                r.chrom.wait()
                            except Exception:
                                pass
                            r.lock.release()
                    rng.__exit__(None, None, None)
                    raise [e for e in errs if e is not None][0]

                def collect(fr):
                    f, r = fr
                    try:
                        res = r.chrom.wait()
                        if STEP_TRACE is not None:
                            STEP_TRACE.append((this_step, "waited", time.perf_counter()))
                        if probe is not None:
                            probe(f, ep, m, step_cut, res)
                        # the run carried its own tail on the device: the table classified (pipe.py:83-97), its
                        # inter-ligation boxes appended to the chromosome's candidate buffer under this step's number,
                        # the distance statistics reduced -- everything is on the host with the run's completion
                        nI, nS, s1 = r.chrom.step_result()
                        n_in = r.chrom.last_n_in()
                    finally:
                        r.lock.release()
                    return f, r, nI, nS, n_in, s1

                used = []
                nI_tot = nS = n_in = 0
                tot = {"n_all": [0, 0], "n_pos": [0, 0], "sumx": [0.0, 0.0], "sumxx": [0.0, 0.0]}
                loghist = np.zeros(_lib_logbins(), dtype=np.int64)
                fine = np.zeros(2048, dtype=np.int64)
                xshift = 0.0
                # (results are taken in file order AS THEY ARRIVE: the sums of the chromosomes that finished early are done while
                #  the others still run -- behind the step's last wait only that chromosome's share is left)
                for f, r, nI, ndS, nin, s1 in (pool.map(collect, live) if (pool is not None and len(live) > 1) else map(collect, live)):
                    nS += ndS
                    n_in += nin
                    if nI == 0:                               # runDBSCAN skips such chromosomes entirely (pipe.py:121-122)
                        continue
                    nI_tot += nI
                    appended[f] = appended.get(f, 0) + nI
                    used.append(r)
                    for gg in (0, 1):
                        for kk in ("n_all", "n_pos", "sumx", "sumxx"):
                            tot[kk][gg] += s1[kk][gg]
                    loghist += s1["loghist"]                  # (integer histograms: exact in any order)
                    if s1.get("fine") is not None:
                        fine += s1["fine"]
                    xshift = s1["xshift"]
                if STEP_TRACE is not None:
                    STEP_TRACE.append((this_step, "collected", time.perf_counter()))
                # the genome-wide statistics: everything is additive over chromosomes and ranks -- two small exchanges per
                # step (one integer vector, one float vector), then the histograms of the median's refinement
                gi = np.concatenate([np.asarray([nI_tot, nS, n_in, len(used)] + tot["n_all"] + tot["n_pos"], dtype=np.int64), loghist, fine])
                gf = np.asarray(tot["sumx"] + tot["sumxx"] + [xshift if used else 0.0, 1.0 if used else 0.0], dtype=np.float64)
                if STEP_TRACE is not None:
                    STEP_TRACE.append((this_step, "reduced", time.perf_counter()))
                if allsum is not None:
                    # ONE exchange per step: the counts ride as float64 next to the sums (every count and every sum of counts
                    # stays far below 2^53, so they come back exact)
                    both = gsum(np.concatenate([gi.astype(np.float64), gf]))
                    gi, gf = np.rint(both[:len(gi)]).astype(np.int64), both[len(gi):]
                g, loghist, fine = gi[:4], gi[8:8 + len(loghist)], gi[8 + len(loghist):]
                st = {"eps": ep, "minPts": m, "cut_in": int(cut), "n_inter": int(g[0]), "n_self": int(g[1]), "n_in": int(g[2])}
                steps.append(st)
                st["wall_s"] = time.perf_counter() - t_step0   # enqueue .. statistics of this rank on the host (+ the exchange); the cut follows
                rng.__exit__(None, None, None)
                if STEP_TRACE is not None:
                    STEP_TRACE.append((this_step, "stats", time.perf_counter()))
                if int(g[3]) == 0:                            # pipe.py:251-255
                    if log:
                        log("ERROR: no inter-ligation PETs detected for eps %s minPts %s,can't model the distance cutoff,continue anyway" % (ep, m))
                    if forced_cuts is not None and forced_cuts[this_step] is not None:
                        cut = int(forced_cuts[this_step])
                        cuts.append(cut)
                        st["cut_out"] = cut
                    continue
                xshift = float(gf[4]) / max(float(gf[5]), 1.0)            # the library constant (ranks without a chromosome report 0)
                tot = {"n_all": [int(gi[4]), int(gi[5])], "n_pos": [int(gi[6]), int(gi[7])]}
                if tot["n_all"][0] > 0 and tot["n_all"][1] > 0:      # pipe.py:256-259
                    if tot["n_pos"][0] == 0 or tot["n_pos"][1] == 0:
                        raise ValueError("cannot convert float NaN to integer")      # what int(2 ** nan) raises in ests.py:57
                    # sum log2|d| and the sum of squared deviations from the group mean, from sum x and sum x^2 (x = log2|d| - xshift)
                    sumlog = [float(gf[0]) + xshift * tot["n_pos"][0], float(gf[1]) + xshift * tot["n_pos"][1]]
                    sq = [float(gf[2]) - float(gf[0]) ** 2 / tot["n_pos"][0], float(gf[3]) - float(gf[1]) ** 2 / tot["n_pos"][1]]
                    tot["sumlog"] = sumlog
                    n1 = tot["n_pos"][1]
                    med = _select_kth(used, cut, loghist, sorted({(n1 - 1) // 2, n1 // 2}), allsum, pool,
                                      fine=(fine_lo, fine) if fine_lo >= 1 else None)
                    # the next step's summary also counts the distances around this median exactly (the median moves little
                    # from step to step): lower edge of the log bin 1024 below it, so that the ranks below it are known
                    from .ests import logbin, logbin_range
                    fine_lo = logbin_range(logbin(max(1, med[0] - 1024)))[0]
                    cut_2, frags, margin = estIntSelCutFrag_from_stats(tot["n_pos"], tot["sumlog"], sq, (med[0], med[-1]), with_margin=True)
                    if margin < CUT_RECHECK_MARGIN and allsum is None:
                        # 2**cut sits on an integer boundary within the rounding noise of the reduction order: settle
                        # it the reference's way, from the distance lists with numpy's own sums (rare; one extra run
                        # per chromosome with labels on the host)
                        parts = [_cluster_arrays(r, ep, m, step_cut, variant) for r in used]
                        cut_2, frags = estIntSelCutFrag(np.concatenate([p[2] for p in parts]), np.concatenate([p[3] for p in parts]))
                        st["cut_rechecked"] = True
                    if log:
                        log("Estimated inter-ligation and self-ligation distance cutoff as %s for eps=%s,minPts=%s" % (cut_2, ep, m))
                    if forced_cuts is not None and forced_cuts[this_step] is not None:
                        cut_2 = int(forced_cuts[this_step])
                    st["cut_out"] = int(cut_2)
                    st["frags"] = int(frags)
                    cuts.append(cut_2)
                    cut = cut_2                               # pipe.py:274
                    if STEP_TRACE is not None:
                        STEP_TRACE.append((this_step, "cut", time.perf_counter()))
                elif forced_cuts is not None and forced_cuts[this_step] is not None:
                    cut = int(forced_cuts[this_step])
                    cuts.append(cut)
                    st["cut_out"] = cut
        pos = [c for c in cuts if c > 0]
        if pos:
            cut = int(np.max(pos)) if max_cut else int(np.min(pos))     # pipe.py:276-280
        else:
            raise ValueError("zero-size array to reduction operation minimum which has no identity")
        # combineTwice over all steps (pipe.py:257,275), then filterClusterByDis (pipe.py:130-143, floor division): on the
        # device, per chromosome; only the surviving boxes cross PCIe
        final_cut = cut

        def finish(fr):
            f, r = fr
            with r.lock:
                if finish_device:
                    ptr, k = r.chrom.cand_finish_device(final_cut)
                    return r.key, {"f": f, "boxes": None, "dev_rows": ptr, "n_rows": k}
                b = r.chrom.cand_finish(final_cut, appended[f])
            return r.key, {"f": f, "boxes": b}                # int32 [k, 4] rows (minX, maxX, minY, maxY), append order

        # key order = first appearance over the steps, file order inside a step: what combineTwice's dict inserts
        # (pipe.py:155-174) and runStat later walks -- `appended` was filled in exactly that order
        res_of = dict(live)
        dataI = _DataI(_pmap(pool, finish, [(f, res_of[f]) for f in appended]))
        if finish_device and device_consumer is not None:
            dataI.gathered = device_consumer(dataI)           # (residents pinned, sweep locks held: the pointers are alive)
    finally:
        if pool is not None:
            pool.shutdown(wait=True)
        for f, r in live:
            # the announced minPts list belongs to this sweep: a later one-off run on the handle serves its own minPts only
            try:
                r.chrom.sweep_plan([], [])
            except Exception:
                pass
        for r in held:
            r.sweep_lock.release()
    return dataI, cut, cuts, steps


def runSweep(fs, eps, minPts, cut=0, cpu=1, max_cut=False, log=None):
    """The (eps, minPts) sweep of cLoops/pipe.py:241-281 with its chained distance cutoff:
    the cut estimated from step k (`cut = cut_2`, pipe.py:274) pre-filters step k+1.

    eps: list ascending, minPts: list descending (the order `main` establishes,
    pipe.py:310-324).  Returns (dataI, cut, cuts, steps): dataI after `combineTwice` and
    `filterClusterByDis`, the final cut (min or max of the positive cuts, pipe.py:276-280),
    every cut seen, and one dict per executed step."""
    dataI = {}
    cuts = [cut]
    steps = []
    for ep in eps:
        for m in minPts:
            rs = _run_many(fs, ep, m, cut, _single_arrays)
            dataI_2, dis_2, dss_2, nS, n_in = {}, [], [], 0, 0
            for key, f, dI, dS, dis, dss, nin in rs:     # runDBSCAN merge, pipe.py:119-127
                n_in += nin
                if len(dI) == 0:
                    continue
                dataI_2[key] = {"f": f, "records": _records(key, dI)}
                nS += len(dS)
                dis_2.append(dis)
                dss_2.append(dss)
            st = {"eps": ep, "minPts": m, "cut_in": int(cut), "n_inter": sum(len(v["records"]) for v in dataI_2.values()),
                  "n_self": nS, "n_in": n_in}
            steps.append(st)
            if len(dataI_2) == 0:                         # pipe.py:251-255
                if log:
                    log("ERROR: no inter-ligation PETs detected for eps %s minPts %s,can't model the distance cutoff,continue anyway" % (ep, m))
                continue
            dis_2 = np.concatenate(dis_2) if dis_2 else np.zeros(0)
            dss_2 = np.concatenate(dss_2) if dss_2 else np.zeros(0)
            if len(dis_2) == 0 or len(dss_2) == 0:       # pipe.py:256-257
                dataI = combineTwice(dataI, dataI_2)
            else:
                cut_2, frags = estIntSelCutFrag(dis_2, dss_2)
                if log:
                    log("Estimated inter-ligation and self-ligation distance cutoff as %s for eps=%s,minPts=%s" % (cut_2, ep, m))
                st["cut_out"] = int(cut_2)
                st["frags"] = int(frags)
                cuts.append(cut_2)
                cut = cut_2                               # pipe.py:274
                dataI = combineTwice(dataI, dataI_2)
    pos = [c for c in cuts if c > 0]
    if pos:
        cut = int(np.max(pos)) if max_cut else int(np.min(pos))     # pipe.py:276-280
    else:
        # np.min([]) raises in the reference; keep that behaviour observable
        raise ValueError("zero-size array to reduction operation minimum which has no identity")
    dataI = filterClusterByDis(dataI, cut)
    return dataI, cut, cuts, steps


# ---------------------------------------------------------------------------------------
# the whole flow of cLoops/pipe.py:206-295 (minus the viewer converters and plots, SURVEY 2 #9,#11)
# ---------------------------------------------------------------------------------------
MODES = {1: ([500, 1000, 2000], [5], 0), 2: ([1000, 2000, 5000], [5], 0),
         3: ([5000, 7500, 10000], [50, 40, 30, 20], 1), 4: ([2500, 5000, 7500, 10000], [30, 20], 1)}   # pipe.py:329-344


def pipe(fs, fout, eps, minPts, chroms="", cpu=1, tmp=0, hic=0, washU=0, juice=0, cut=0, plot=0, max_cut=False,
         log=None):
    """cLoops/pipe.py:206-295: BEDPE -> per-chromosome PETs -> (eps, minPts) sweep with the chained
    distance cutoff on the GPU(s) -> candidate loops -> significance -> `<fout>.loop`.

    Differences to the reference, all outside the hot path: no washU / juicebox conversion and no
    plots (`washU`, `juice`, `plot` are accepted and ignored).  `eps == 0` estimates eps from the distances of the
    PETs mapped to different strands (io.py:62-129, ests.py:23-33)."""
    import shutil
    from . import io as cio
    from . import cModel
    if chroms == "":
        chroms = []
    else:
        chroms = set(chroms.split(","))
    if os.path.isdir(fout):                                   # pipe.py:225-228
        if log:
            log("working directory %s exists, return." % fout)
        return
    os.mkdir(fout)
    if eps == 0 or eps == [0]:                                # pipe.py:231-239: eps from the data
        from .ests import estFragSize
        cfs, ds = cio.parseRawBedpe(fs, fout, chroms, cut)
        cfs = [cio.txt2jd(f) for f in cfs]
        eps = [estFragSize(ds) * 2]
    else:
        cfs = [cio.txt2jd(f) for f in cio.parseRawBedpe2(fs, fout, chroms, cut)]
    dataI, cut, cuts, steps = runSweepFast(cfs, eps, minPts, cut=cut, max_cut=max_cut, log=log)
    records = {key: {"f": v["f"], "records": _records(key, v["boxes"])} for key, v in dataI.items()}
    e = cModel.runStat(records, minPts, 0, cpu, fout, hic)    # pipe.py:284 passes cut = 0
    if e:
        shutil.rmtree(fout)
        return
    if tmp == False:                                          # noqa: E712  (pipe.py:294)
        shutil.rmtree(fout)
    return steps


def main(argv=None):
    """`python -m cloops_amd -f a.bedpe.gz -o out -m 1` -- the flags of cLoops/utils.py:73-204 that
    drive the hot path (same names; -w / -j / -plot are accepted and ignored)."""
    import argparse
    ap = argparse.ArgumentParser(prog="cloops_amd")
    ap.add_argument("-f", dest="fnIn", required=True)
    ap.add_argument("-o", dest="fnOut", required=True)
    ap.add_argument("-m", dest="mode", type=int, default=0, choices=[0, 1, 2, 3, 4])
    ap.add_argument("-eps", dest="eps", default="0")
    ap.add_argument("-minPts", dest="minPts", default="0")
    ap.add_argument("-p", dest="cpu", type=int, default=1)
    ap.add_argument("-c", dest="chroms", default="")
    ap.add_argument("-w", dest="washU", action="store_true")
    ap.add_argument("-j", dest="juice", action="store_true")
    ap.add_argument("-s", dest="tmp", action="store_true")
    ap.add_argument("-hic", dest="hic", action="store_true")
    ap.add_argument("-cut", dest="cut", type=int, default=0)
    ap.add_argument("-plot", dest="plot", action="store_true")
    ap.add_argument("-max_cut", dest="max_cut", action="store_true")
    op = ap.parse_args(argv)
    if op.mode == 0:                                          # pipe.py:306-327
        eps = sorted(int(x) for x in str(op.eps).split(","))
        minPts = sorted((int(x) for x in str(op.minPts).split(",")), reverse=True)
        if minPts == [0]:
            sys.stderr.write("minPts not assigned!\n")
            return 1
        hic = int(op.hic)
    else:
        eps, minPts, hic = MODES[op.mode]
    sys.stderr.write("mode:%s\t eps:%s\t minPts:%s\t hic:%s\t\n" % (op.mode, eps, minPts, hic))
    pipe(op.fnIn.split(","), op.fnOut, eps, minPts, op.chroms, op.cpu, op.tmp, hic, op.washU, op.juice, op.cut,
         op.plot, op.max_cut, log=lambda m: sys.stderr.write(m + "\n"))
    return 0
