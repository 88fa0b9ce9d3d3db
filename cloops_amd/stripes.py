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


# ---- significance of candidate stripes (host side, scipy; scripts/callStripes:89-269) ---------
def getNearbyStripes(iva, ivb, win=5):
    """scripts/callStripes:89-120 with Python-2 integer arithmetic: the longer anchor stays, the shorter
    one slides by its half-length; equal lengths return None (the script then fails unpacking it)."""
    lena = iva[1] - iva[0]
    lenb = ivb[1] - ivb[0]
    ivas, ivbs = [], []
    ca = sum(iva) // 2
    cb = sum(ivb) // 2
    sa = (iva[1] - iva[0]) // 2
    sb = (ivb[1] - ivb[0]) // 2
    if lena > lenb:
        step = sb
        for i in range(0 - win, win + 1):
            if i == 0:
                continue
            ivas.append(iva)
            ivbs.append([max([0, cb + i * step - sb]), max([0, cb + i * step + sb])])
        return ivas, ivbs
    if lena < lenb:
        step = sa
        for i in range(0 - win, win + 1):
            if i == 0:
                continue
            ivas.append([max([0, ca + i * step - sa]), max([0, ca + i * step + sa])])
            ivbs.append(ivb)
        return ivas, ivbs
    return None


def getStripePsFdr(iva, ivb, model, N, win=5):
    """scripts/callStripes:123-185 -> ra, rb, rab, es, es_ra, es_rb, fdr, pop, nbp"""
    from scipy.stats import binom, poisson
    from .cModel import getPETsforRegions
    ra, rb, rab = getPETsforRegions(iva, ivb, model)
    ivas, ivbs = getNearbyStripes(iva, ivb, win=win)       # TypeError for equal lengths, like the script
    nras = [model.region(na) for na in ivas]
    nrbs = [model.region(nb) for nb in ivbs]
    rabs, nbps = [], []
    for nra in nras:
        nralen = float(len(nra))
        for nrb in nrbs:
            nrblen = len(nrb)
            nrab = float(len(np.intersect1d(nra, nrb, assume_unique=True)))
            if nrab > 0:
                rabs.append(nrab)
                nbps.append(nrab / (nralen * nrblen))
            else:
                nbps.append(0.0)
                rabs.append(0.0)
    rabs = np.array(rabs)
    fdr = len(rabs[rabs > rab]) / float(len(rabs))
    mrabs = float(np.mean(rabs))
    es = rab / np.mean(rabs[rabs > 0]) if mrabs > 0 else np.inf
    pop = max([1e-300, poisson.sf(rab - 1.0, mrabs)])
    bp = np.mean(nbps) * ra * rb / N
    nbp = max([1e-300, binom.sf(rab - 1.0, N - rab, bp)])
    return ra, rb, rab, es, rab / float(ra), rab / float(rb), fdr, pop, nbp


def estStripeSig(f, records, device=0):
    """scripts/callStripes:188-233: one row per candidate stripe; None without PETs or records"""
    import pandas as pd
    from .cModel import CoverageModel
    r0 = CACHE.get(f, device)
    N = len(r0.d)
    if N < 2:                                             # getGenomeCoverage returns (None, 0) (cModel.py:53-54)
        return None
    model = CoverageModel(np.stack([np.asarray(r0.ids), np.asarray(r0.X), np.asarray(r0.Y)], 1))
    ds = {}
    for i, r in enumerate(records):
        chrom = r[0]
        key = "%s-%s-%s" % (r[0], r[3], i)
        iva = [max(0, r[1]), r[2]]
        ivb = [max(0, r[4]), r[5]]
        ra, rb, rab, es, es_ra, es_rb, fdr, pop, nbp = getStripePsFdr(iva, ivb, model, N)
        ds[key] = {
            "ra": ra, "rb": rb, "rab": rab, "ES": es, "ES_ra": es_ra, "ES_rb": es_rb, "FDR": fdr,
            "poisson_p-value": pop, "binomial_p-value": nbp,
            "iva": "%s:%s-%s" % (chrom, iva[0], iva[1]), "ivb": "%s:%s-%s" % (chrom, ivb[0], ivb[1]),
        }
    if len(ds) == 0:
        return None
    return pd.DataFrame(ds).T


def markStripeSig(ds, escut=2.0, fdrcut=0.1, ppcut=1e-5, es_cut=0.2):
    """scripts/callStripes:236-269 as one boolean mask: enrichment, local FDR and Poisson cuts, and at least one
    anchor holding the fraction `es_cut` of its PETs in the stripe; `significant` is 1.0 / 0.0 like the script's."""
    col = lambda name: ds[name].astype(float)
    sig = (col("ES") >= escut) & (col("FDR") <= fdrcut) & (col("poisson_p-value") <= ppcut) \
        & ((col("ES_ra") >= es_cut) | (col("ES_rb") >= es_cut))
    ds["significant"] = sig.astype(float)
    return ds


def callStripes(fs, fout, eps=20000, minPts=5, pets=100, ext=50, lengthFoldDiff=50):
    """scripts/callStripes:285-372 for a list of .jd files (or 'mem://' chromosomes of pipe.CACHE): horizontal
    (X stretched) and vertical (Y stretched) stripes -> `fout_x_horizontal.stripe`, `fout_y_vertical.stripe`.
    The juicebox converter (`-j`) is a viewer format and not part of this package."""
    import pandas as pd
    out = {}
    for name, kw in (("x_horizontal", {"extx": ext}), ("y_vertical", {"exty": ext})):
        ds = dict(singleStripDBSCAN(f, eps, minPts, **kw) for f in fs)
        key2f = {CACHE.get(f).key: f for f in fs}
        ds = filterCandidateStripes(ds, pets=pets, lengthFoldDiff=lengthFoldDiff)
        tabs = [estStripeSig(key2f[key], ds[key]) for key in ds.keys()]
        tabs = [t for t in tabs if t is not None]
        if len(tabs) > 0:
            tab = markStripeSig(pd.concat(tabs))
            tab.to_csv(fout + "_%s.stripe" % name, sep="\t", index_label="stripeId")
            out[name] = tab
    return out


def main(argv=None):
    """`python -m cloops_amd.stripes -d <dir of .jd> -o <prefix>`: the flags of scripts/callStripes:375-459
    (`-j` accepted and ignored: viewer format; `-p` accepted: parallelism is over GPUs)."""
    import argparse
    import os
    from glob import glob
    ap = argparse.ArgumentParser(description="Call stripes (scripts/callStripes) on MI355X")
    ap.add_argument("-d", dest="d", required=True, type=str)
    ap.add_argument("-o", dest="output", required=True, type=str)
    ap.add_argument("-eps", dest="eps", default=20000, type=int)
    ap.add_argument("-minPts", dest="minPts", default=5, type=int)
    ap.add_argument("-ext", dest="ext", default=50, type=int)
    ap.add_argument("-pets", dest="pets", default=200, type=int)
    ap.add_argument("-lenFold", dest="lengthFoldDiff", default=50, type=int)
    ap.add_argument("-c", dest="chroms", default="", type=str)
    ap.add_argument("-j", dest="juice", action="store_true")
    ap.add_argument("-p", dest="cpu", default=1, type=int)
    op = ap.parse_args(argv)
    chroms = op.chroms.split(",")
    fs = []
    for f in sorted(glob(os.path.join(op.d, "*.jd"))):
        c = tuple(os.path.splitext(os.path.split(f)[-1])[0].split("-"))
        if chroms == [""] or (c[0] in chroms and c[1] in chroms):          # scripts/callStripes:300-306
            fs.append(f)
    callStripes(fs, op.output, eps=op.eps, minPts=op.minPts, pets=op.pets, ext=op.ext, lengthFoldDiff=op.lengthFoldDiff)
    return 0


if __name__ == "__main__":
    sys.exit(main())
