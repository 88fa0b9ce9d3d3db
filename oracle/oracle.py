"""ctypes front-end of the C oracle (oracle/cloops_oracle.c) + numpy restatements of the
host-side pieces of the hot path.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference lines restated here:
  * labels(...)         -> cLoops/cDBSCAN.py:6-205, cLoops/cDBSCAN2.py:7-383,
                           cLoops/blockDBSCAN.py:6-239 (via the C restatement)
  * single_dbscan(...)  -> cLoops/pipe.py:52-110 (singleDBSCAN), written against arrays
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcloops_oracle.so")
_lib = None

VARIANTS = {"v1": 1, "v2": 2, "block": 3}


def build(force=False):
    """Compile the C oracle with gcc (no GPU, no reference needed)."""
    src = os.path.join(_HERE, "cloops_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libcloops_oracle.so"])
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(_SO)
        i64p = ctypes.POINTER(ctypes.c_int64)
        i32p = ctypes.POINTER(ctypes.c_int32)
        for name in ("cl_oracle_v1", "cl_oracle_v2", "cl_oracle_block"):
            fn = getattr(lib, name)
            fn.argtypes = [i64p, i64p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, i32p]
            fn.restype = ctypes.c_int
        lib.cl_oracle_neighbor_counts.argtypes = [i64p, i64p, ctypes.c_int64, ctypes.c_int64, i32p]
        lib.cl_oracle_neighbor_counts.restype = ctypes.c_int
        _lib = lib
    return _lib


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def labels(variant, X, Y, eps, minPts):
    """Oracle cluster labels aligned to rows (-1 = not in the reference's .labels dict).

    Raises IndexError on empty input for v1/block, like the reference
    (cDBSCAN.py:77, blockDBSCAN.py:74)."""
    lib = _load()
    X = _i64(X)
    Y = _i64(Y)
    n = X.shape[0]
    out = np.full(n, -1, dtype=np.int32)
    fn = {"v1": lib.cl_oracle_v1, "v2": lib.cl_oracle_v2, "block": lib.cl_oracle_block}[variant]
    i64p = ctypes.POINTER(ctypes.c_int64)
    rc = fn(X.ctypes.data_as(i64p), Y.ctypes.data_as(i64p), n, int(eps), int(minPts),
            out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    if rc == -1:
        raise IndexError("index 0 is out of bounds for axis 0 with size 0")
    if rc != 0:
        raise MemoryError("oracle rc=%d" % rc)
    return out


def neighbor_counts(X, Y, eps):
    """Brute-force |{q: L1(p,q) <= eps}| per row (self included); O(n^2)."""
    lib = _load()
    X = _i64(X)
    Y = _i64(Y)
    out = np.zeros(X.shape[0], dtype=np.int32)
    i64p = ctypes.POINTER(ctypes.c_int64)
    lib.cl_oracle_neighbor_counts(X.ctypes.data_as(i64p), Y.ctypes.data_as(i64p), X.shape[0], int(eps),
                                  out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    return out


def single_dbscan(variant, X, Y, eps, minPts, cut=0):
    """cLoops/pipe.py:52-110 restated on arrays, driven by the oracle's labels.

    Returns dict(labels=<int32[n] aligned to the *unfiltered* rows, -1 noise/filtered>,
                 dataI=[[minX,maxX,minY,maxY],...], dataS=[...] (ascending cluster id,
                 pipe.py:76-102), dis=<float64 array, multiset of pipe.py:107>,
                 dss=<float64 array, multiset of pipe.py:63 + :109>)."""
    X = _i64(X)
    Y = _i64(Y)
    n = X.shape[0]
    d = Y - X
    keep = np.arange(n)
    dss = []
    if cut > 0:                                   # pipe.py:59-63
        keep = np.where(d >= cut)[0]
        dss.append(d[d < cut].astype(np.float64))
    full = np.full(n, -1, dtype=np.int32)
    res = dict(labels=full, dataI=[], dataS=[], dis=np.zeros(0), dss=np.zeros(0))
    if len(keep) == 0:                            # pipe.py:64-65
        res["dss"] = np.concatenate(dss) if dss else np.zeros(0)
        return res
    lab = labels(variant, X[keep], Y[keep], eps, minPts)
    full[keep] = lab
    xs, ys = X[keep], Y[keep]
    dis = []
    # per-cluster bounding boxes by one sort of the labelled points (the reference loops over the clusters with a
    # boolean mask each, pipe.py:78-102: O(K * N))
    sel = np.flatnonzero(lab >= 0)
    if len(sel):
        order = sel[np.argsort(lab[sel], kind="stable")]
        ls = lab[order]
        starts = np.flatnonzero(np.concatenate([[True], ls[1:] != ls[:-1]]))
        xo, yo = xs[order], ys[order]
        x0, x1 = np.minimum.reduceat(xo, starts), np.maximum.reduceat(xo, starts)
        y0, y1 = np.minimum.reduceat(yo, starts), np.maximum.reduceat(yo, starts)
        ok = (x0 != x1) & (y0 != y1)                  # pipe.py:83-85
        inter = ok & (x1 < y0)                        # pipe.py:97
        selfl = ok & ~inter
        res["dataI"] = np.stack([x0, x1, y0, y1], 1)[inter].astype(np.int64).tolist()      # ascending cluster id
        res["dataS"] = np.stack([x0, x1, y0, y1], 1)[selfl].astype(np.int64).tolist()
        cls_of_point = np.repeat(np.where(inter, 1, np.where(selfl, 2, 0)), np.diff(np.concatenate([starts, [len(ls)]])))
        dd = (yo - xo).astype(np.float64)
        dis.append(dd[cls_of_point == 1])
        dss.append(dd[cls_of_point == 2])
    res["dis"] = np.concatenate(dis) if dis else np.zeros(0)
    res["dss"] = np.concatenate(dss) if dss else np.zeros(0)
    return res
