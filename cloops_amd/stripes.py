"""Candidate stripes: the clustering step of scripts/callStripes (reference file:line cited per function).

callStripes clusters a chromosome's PETs with variant 1 (`cLoops.cDBSCAN`) after stretching ONE axis
by `ext` (default 50), so that eps-balls become thin rectangles and line-like pile-ups along the other
axis come out as clusters.  Here the stretch is a weighted metric inside the kernels
(`cl_cluster_weighted`, 64-bit rotated coordinates) -- no scaled copy of the matrix exists."""
import sys

import numpy as np

from . import api
from .pipe import CACHE


def singleStripDBSCAN(f, eps, minPts, extx=1, exty=1, device=None):
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
    """scripts/callStripes:75-86: keep clusters with >= `pets` PETs whose bounding box is line-like -- one side more
    than `lengthFoldDiff` times the other (Python-2 integer `/` = floor division; a zero-length side raises
    ZeroDivisionError there too).  Mutates and returns `rs` like the script."""
    def line_like(rec):
        w, h = rec[2] - rec[1], rec[5] - rec[4]
        return max(w // h, h // w) > lengthFoldDiff
    for key, recs in rs.items():
        rs[key] = [rec for rec in recs if rec[6] >= pets and line_like(rec)]
    return rs


# ---- significance of candidate stripes (scripts/callStripes:89-233) on the K8 interval counts ---------
def _stripe_windows(records, win=5):
    """The 22 count windows of every candidate stripe, vectorised: window 0 of each side is the anchor itself,
    1..10 its permuted background (scripts/callStripes:89-120 in Python-2 integer arithmetic): the LONGER anchor
    stays in place, the shorter one slides by multiples of its own half-length.  Equal lengths have no
    background in the script (`getNearbyStripes` returns None and the caller fails unpacking it): TypeError.
    -> (iva int64[R,2], ivb int64[R,2], windows int32[R,44] = lo[22] then hi[22]; A0..A10 then B0..B10)"""
    r = np.asarray([[x[1], x[2], x[4], x[5]] for x in records], dtype=np.int64).reshape(-1, 4)
    iva = np.stack([np.maximum(0, r[:, 0]), r[:, 1]], 1)
    ivb = np.stack([np.maximum(0, r[:, 2]), r[:, 3]], 1)
    lens = np.stack([iva[:, 1] - iva[:, 0], ivb[:, 1] - ivb[:, 0]], 1)
    if (lens[:, 0] == lens[:, 1]).any():
        raise TypeError("cannot unpack non-iterable NoneType object")       # scripts/callStripes:135 on equal anchor lengths
    slide_b = lens[:, 0] > lens[:, 1]                       # anchor a is the longer one: b slides
    lo = np.empty((len(r), 22), np.int64)
    hi = np.empty((len(r), 22), np.int64)
    for side, iv, slides in ((0, iva, ~slide_b), (11, ivb, slide_b)):
        centre = iv.sum(1) // 2
        half = (iv[:, 1] - iv[:, 0]) // 2
        lo[:, side], hi[:, side] = iv[:, 0], iv[:, 1]
        shifts = [i for i in range(-win, win + 1) if i != 0]
        for k, i in enumerate(shifts, start=1):
            lo[:, side + k] = np.where(slides, np.maximum(0, centre + i * half - half), iv[:, 0])
            hi[:, side + k] = np.where(slides, np.maximum(0, centre + i * half + half), iv[:, 1])
    return iva, ivb, np.concatenate([lo, hi], 1).astype(np.int32)


def estStripeSig(f, records, device=None):
    """scripts/callStripes:123-233: one row per candidate stripe; None without PETs or records.  The interval
    counting is kernel K8 on the resident chromosome (`cl_sig_counts`: |region(A_k)|, |region(B_l)|, the
    direct X-in-A / Y-in-B count and all |region(A_k) & region(B_l)|); the statistics of all stripes are array
    expressions on that table -- same integers, same numpy / scipy calls on the same operands as the script."""
    import pandas as pd
    from scipy.stats import binom, poisson
    r0 = CACHE.get(f, device)
    if len(r0.d) < 2 or len(records) == 0:                  # getGenomeCoverage -> (None, 0) (cModel.py:53-54)
        return None
    iva, ivb, wins = _stripe_windows(records)
    with r0.lock:
        counts, N = r0.chrom.sig_counts(wins, 0)
    c = counts.astype(np.int64)
    R = len(records)
    ra, rb, rab = c[:, 0], c[:, 11], c[:, 22]
    grid = c[:, 23:].reshape(R, 11, 11)[:, 1:, 1:].astype(np.float64)       # |region(A_k) & region(B_l)|, k, l = 1..10
    rabs = grid.reshape(R, 100)
    with np.errstate(divide="ignore", invalid="ignore"):
        share = np.where(grid > 0, grid / (c[:, 1:11].astype(np.float64)[:, :, None] * c[:, 12:22][:, None, :]), 0.0).reshape(R, 100)
    lam = rabs.mean(axis=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        es = np.where(lam > 0, rab / (rabs.sum(1) / np.maximum((rabs > 0).sum(1), 1)), np.inf)
    fdr = (rabs > rab[:, None]).sum(1) / float(100)
    pop = np.maximum(1e-300, poisson.sf(rab - 1.0, lam))
    nbp = np.maximum(1e-300, binom.sf(rab - 1.0, N - rab, share.mean(axis=1) * ra * rb / N))
    rows = {}
    for i, r in enumerate(records):
        rows["%s-%s-%s" % (r[0], r[3], i)] = {
            "ra": int(ra[i]), "rb": int(rb[i]), "rab": int(rab[i]), "ES": float(es[i]), "ES_ra": rab[i] / float(ra[i]),
            "ES_rb": rab[i] / float(rb[i]), "FDR": float(fdr[i]), "poisson_p-value": float(pop[i]), "binomial_p-value": float(nbp[i]),
            "iva": "%s:%s-%s" % (r[0], iva[i, 0], iva[i, 1]), "ivb": "%s:%s-%s" % (r[0], ivb[i, 0], ivb[i, 1]),
        }
    return pd.DataFrame(rows).T


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
