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
from collections import OrderedDict
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import api
from .cDBSCAN2 import cDBSCAN as DBSCAN          # pipe.py:42  (production variant)
from .dist import lpt_assign
from .ests import estIntSelCutFrag

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
class _Resident(object):
    __slots__ = ("key", "ids", "X", "Y", "d", "chrom", "device", "stamp", "lock")


class ChromCache(object):
    """path -> chromosome resident in HBM (+ host copies of ids and distances)."""

    def __init__(self, max_items=64):
        self._items = OrderedDict()
        self._lock = threading.Lock()
        self.max_items = max_items

    def get(self, f, device=0):
        st = os.stat(f)
        stamp = (st.st_mtime_ns, st.st_size)
        with self._lock:
            r = self._items.get(f)
            if r is not None and r.stamp == stamp and r.device == device:
                self._items.move_to_end(f)
                return r
        key, mat = parseJd(f, cut=0)
        mat = np.asarray(mat)
        r = _Resident()
        r.key, r.stamp, r.device = key, stamp, device
        r.lock = threading.Lock()
        if len(mat):
            r.ids = mat[:, 0]
            r.X = np.ascontiguousarray(mat[:, 1])
            r.Y = np.ascontiguousarray(mat[:, 2])
        else:
            r.ids = r.X = r.Y = np.zeros(0, np.int64)
        r.d = r.Y - r.X
        r.chrom = api.Chromosome(r.X, r.Y, device=device)
        with self._lock:
            old = self._items.pop(f, None)
            self._items[f] = r
            while len(self._items) > self.max_items:
                _, ev = self._items.popitem(last=False)
                ev.chrom.close()
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


def singleDBSCAN(f, eps, minPts, cut=0, device=0):
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
    sizes = [os.path.getsize(f) for f in fs]
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
    """Filter inter-ligation clusters by distances (cLoops/pipe.py:130-143).  The reference
    is Python 2: `/` on the int mid-points is FLOOR division (pipe.py:138)."""
    for key in data:
        nr = []
        for r in data[key]["records"]:
            d = (r[4] + r[5]) // 2 - (r[1] + r[2]) // 2
            if d >= cut:
                nr.append(r)
        data[key]["records"] = nr
    return data


def checkSameLoop(ra, rb):
    """check if two anchors are exact same (cLoops/pipe.py:146-152)."""
    if ra[1] == rb[1] and ra[2] == rb[2] and ra[4] == rb[4] and ra[5] == rb[5]:
        return True
    return False


def combineTwice(dataI, dataI_2):
    """Combining multiple clustering result (cLoops/pipe.py:155-174): exact-box dedup."""
    for key in dataI_2.keys():
        if key not in dataI:
            dataI[key] = {"f": dataI_2[key]["f"], "records": dataI_2[key]["records"]}
        else:
            ds = set()
            for r in dataI[key]["records"]:
                ds.add((r[1], r[2], r[4], r[5]))
            for r in dataI_2[key]["records"]:
                if (r[1], r[2], r[4], r[5]) not in ds:
                    dataI[key]["records"].append(r)
    return dataI


def _single_arrays(f, eps, minPts, cut, device):
    r = CACHE.get(f, device)
    if len(r.d) == 0:
        e4 = np.zeros((0, 4), np.int64)
        return r.key, f, e4, e4, np.zeros(0), np.zeros(0)
    dataI, dataS, dis, dss, nI, nS, n_in = _cluster_arrays(r, eps, minPts, cut)
    return r.key, f, dataI, dataS, dis, dss


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
            dataI_2, dis_2, dss_2, nS = {}, [], [], 0
            for key, f, dI, dS, dis, dss in rs:          # runDBSCAN merge, pipe.py:119-127
                if len(dI) == 0:
                    continue
                dataI_2[key] = {"f": f, "records": _records(key, dI)}
                nS += len(dS)
                dis_2.append(dis)
                dss_2.append(dss)
            st = {"eps": ep, "minPts": m, "cut_in": int(cut), "n_inter": sum(len(v["records"]) for v in dataI_2.values()),
                  "n_self": nS}
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
