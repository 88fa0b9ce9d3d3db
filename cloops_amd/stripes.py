"""Candidate stripes: the clustering step of scripts/callStripes (reference file:line cited per function).

callStripes clusters a chromosome's PETs with variant 1 (`cLoops.cDBSCAN`) after stretching ONE axis
by `ext` (default 50), so that eps-balls become thin rectangles and line-like pile-ups along the other
axis come out as clusters.  Here the stretch is a weighted metric inside the kernels
(`cl_cluster_weighted`, 64-bit rotated coordinates) -- no scaled copy of the matrix exists."""
import sys

import numpy as np

from . import api
from .pipe import CACHE


def singleStripDBSCAN(f, eps, minPts, extx=1, exty=1, device=0):
    """scripts/callStripes:37-72.  Returns (key, dataI) with records
    [chrA, minX, maxX, chrB, minY, maxY, nPETs] in ascending cluster id (the iteration order of
    `set(labels.values)`, as in pipe.singleDBSCAN); coordinates unscaled like callStripes:59-66
    (`int(min / ext)`)."""
    r = CACHE.get(f, device)
    key = r.key
    sys.stdout.write("Clustering %s and %s using eps as %s, minPts as %s\n" % (key[0], key[1], eps, minPts))
    if len(r.d) == 0:
        raise IndexError("index 0 is out of bounds for axis 0 with size 0")          # cDBSCAN.py:77 on an empty mat
    with r.lock:
        res = r.chrom.cluster_weighted(eps, minPts, extx, exty)
    b = res.boxes
    live = np.flatnonzero(b["count"] > 0) if len(b) else np.zeros(0, np.int64)       # variant 1 keeps id gaps
    dataI = [[key[0], int(b["min_x"][k]), int(b["max_x"][k]), key[1], int(b["min_y"][k]), int(b["max_y"][k]), int(b["count"][k])]
             for k in live]
    sys.stdout.write("Clustering %s and %s finished.\n" % (key[0], key[1]))
    return key, dataI


def filterCandidateStripes(rs, pets=200, lengthFoldDiff=20):
    """scripts/callStripes:75-86 (py2 integer `/` on ints is floor division; a zero-length side raises
    ZeroDivisionError there too)."""
    for key in list(rs.keys()):
        nds = []
        for r in rs[key]:
            if r[6] < pets:
                continue
            xlen = r[2] - r[1]
            ylen = r[5] - r[4]
            if (xlen // ylen > lengthFoldDiff) or (ylen // xlen > lengthFoldDiff):
                nds.append(r)
        rs[key] = nds
    return rs
