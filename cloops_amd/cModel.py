"""Significance of candidate loops against the local permuted background, and the `.loop` table.

Host-side restatement of cLoops/cModel.py (getGenomeCoverage :45-57, getCounts :60-70,
getPETsforRegions :73-80, getNearbyPairRegions :83-105, getMultiplePsFdr :108-161, getBonPvalues
:164-171, checkOneEndOverlap/checkOverlap :174-195, removeDup :198-259, getIntSig :262-331,
markIntSig :334-362, markIntSigHic :365-386) and of runStat (cLoops/pipe.py:177-203).

Semantics pinned: the reference is Python 2 -- `/` on the integer
interval ends in getNearbyPairRegions is FLOOR division (:89-93) and dicts iterate in insertion
order under the Python-3 conversion the goldens were made with (removeDup :206-259 depends on
that order).  The PET index sets of the reference (Python sets of row positions) are sorted
index arrays here; every float is produced by the same scipy / numpy call on the same operands in
the same order, so the `.loop` rows are bit-identical to the converted reference.
"""
import numpy as np
import pandas as pd
from scipy.stats import hypergeom, binom, poisson

from .pipe import parseJd


def parseIv(iv):
    """cLoops/io.py parseIv: 'chr1:100-200' -> ['chr1', 100, 200]"""
    chrom, rest = iv.split(":")
    a, b = rest.split("-")
    return [chrom, int(a), int(b)]


class CoverageModel(object):
    """getGenomeCoverage (cModel.py:45-57): sorted X and Y coordinates with the row positions."""

    def __init__(self, mat):
        X = np.asarray(mat[:, 1])
        Y = np.asarray(mat[:, 2])
        self.N = len(X)
        self.xo = np.argsort(X, kind="stable")
        self.yo = np.argsort(Y, kind="stable")
        self.xs = X[self.xo]
        self.ys = Y[self.yo]

    def side(self, iv, axis):
        """getCounts (cModel.py:60-70): row positions of the PETs whose X (axis 0) / Y (axis 1) lies in
        [iv[0], iv[1]] (both ends inclusive), as a sorted unique array."""
        keys, order = (self.xs, self.xo) if axis == 0 else (self.ys, self.yo)
        l = np.searchsorted(keys, iv[0], side="left")
        r = np.searchsorted(keys, iv[1], side="right")
        return np.sort(order[l:r])

    def region(self, iv):
        """S_X(iv) | S_Y(iv)"""
        return np.union1d(self.side(iv, 0), self.side(iv, 1))


def getPETsforRegions(iva, ivb, model):
    """cModel.py:73-80"""
    ra = len(model.region(iva))
    rb = len(model.region(ivb))
    rab = len(np.intersect1d(model.side(iva, 0), model.side(ivb, 1), assume_unique=True))
    return ra, rb, rab


def getNearbyPairRegions(iva, ivb, win=5):
    """cModel.py:83-105 with Python-2 integer arithmetic"""
    ivas, ivbs = [], []
    ca = sum(iva) // 2
    cb = sum(ivb) // 2
    sa = (iva[1] - iva[0]) // 2
    sb = (ivb[1] - ivb[0]) // 2
    step = (sa + sb) // 2
    for i in range(0 - win, win + 1):
        if i == 0:
            continue
        ivas.append([max([0, ca + i * step - sa]), max([0, ca + i * step + sa])])
        ivbs.append([max([0, cb + i * step - sb]), max([0, cb + i * step + sb])])
    return ivas, ivbs


def getMultiplePsFdr(iva, ivb, model, N, win=5):
    """cModel.py:108-161 -> ra, rb, rab, es, fdr, hyp, pop, nbp"""
    ra, rb, rab = getPETsforRegions(iva, ivb, model)
    hyp = max([1e-300, hypergeom.sf(rab - 1.0, N, ra, rb)])
    ivas, ivbs = getNearbyPairRegions(iva, ivb, win=win)
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
    if len(rabs) == 0:
        return ra, rb, rab, np.inf, 0.0, hyp, 0.0, 1e-300, 1e-300,
    rabs = np.array(rabs)
    fdr = len(rabs[rabs > rab]) / float(len(rabs))
    mrabs = float(np.mean(rabs))
    if mrabs > 0:
        es = rab / np.mean(rabs[rabs > 0])
    else:
        es = np.inf
    lam = mrabs
    pop = max([1e-300, poisson.sf(rab - 1.0, lam)])
    bp = np.mean(nbps) * ra * rb / N
    nbp = max([1e-300, binom.sf(rab - 1.0, N - rab, bp)])
    return ra, rb, rab, es, fdr, hyp, pop, nbp


def getBonPvalues(ps):
    """cModel.py:164-171"""
    ps = np.array(ps)
    ps = ps * len(ps)
    ps[ps > 1.0] = 1.0
    return ps


def checkOneEndOverlap(xa, xb, ya, yb):
    """cModel.py:174-182"""
    if (ya <= xa <= yb) or (ya <= xb <= yb) or (ya <= xa <= xb <= yb):
        return True
    if (xa <= ya <= xb) or (xa <= yb <= xb) or (xa <= ya <= yb <= xb):
        return True
    return False


def checkOverlap(ivai, ivbi, ivaj, ivbj):
    """cModel.py:185-195"""
    if ivai[0] != ivaj[0] or ivbi[0] != ivbj[0]:
        return
    if checkOneEndOverlap(ivai[1], ivai[2], ivaj[1], ivaj[2]) and checkOneEndOverlap(ivbi[1], ivbi[2], ivbj[1], ivbj[2]):
        return True
    return False


def removeDup(ds, bpcut=1e-5):
    """cModel.py:198-259: overlapped loops -> keep, among those with binomial p <= bpcut, the one with
    the highest rab / ra / rb; iteration in dict insertion order."""
    uniqueds = {}
    reds = {}
    rekeys = set()
    keys = list(ds.keys())
    ivs = {k: (parseIv(ds[k]["iva"]), parseIv(ds[k]["ivb"])) for k in keys}
    for i in range(len(keys) - 1):
        keyi = keys[i]
        if keyi in rekeys:
            continue
        ivai, ivbi = ivs[keyi]
        flag = 1
        for j in range(i + 1, len(keys)):
            keyj = keys[j]
            if keyj in rekeys:
                continue
            ivaj, ivbj = ivs[keyj]
            if checkOverlap(ivai, ivbi, ivaj, ivbj):
                if keyi not in reds:
                    reds[keyi] = [keyi]
                    rekeys.add(keyi)
                reds[keyi].append(keyj)
                rekeys.add(keyj)
                flag = 0
        if flag:
            uniqueds[keyi] = ds[keyi]
    for key in reds.keys():
        ts = {}
        for t in reds[key]:
            if ds[t]["binomial_p-value"] > bpcut:
                continue
            ts[t] = float(ds[t]["rab"]) / ds[t]["ra"] / ds[t]["rb"]
        if len(ts) == 0:
            continue
        ts = pd.Series(ts)
        ts.sort_values(inplace=True, ascending=False)
        uniqueds[ts.index[0]] = ds[ts.index[0]]
    return uniqueds


def getIntSigFromMat(mat, records, minPts, discut, name=""):
    """getIntSig (cModel.py:262-331) on an in-memory [n,3] matrix (already cut-filtered by the caller
    exactly like parseJd(f, discut), io.py:213-216)."""
    j = mat.shape[0]
    if j < 2:                                  # getGenomeCoverage :52-54
        return None
    model = CoverageModel(mat)
    N = j
    ds = {}
    i = 0
    for r in records:
        chrom = r[0]
        key = "%s-%s-%s" % (r[0], r[3], i)
        iva = [max(0, r[1]), r[2]]
        ivb = [max(0, r[4]), r[5]]
        distance = abs(sum(ivb) / 2.0 - sum(iva) / 2.0)
        if distance < discut:
            continue
        ra, rb, rab = getPETsforRegions(iva, ivb, model)
        if rab < max(minPts):
            continue
        i += 1
        ra, rb, rab, es, fdr, hyp, pop, nbp = getMultiplePsFdr(iva, ivb, model, N)
        ds[key] = {
            "distance": distance, "ra": ra, "rb": rb, "rab": rab, "ES": es, "FDR": fdr,
            "hypergeometric_p-value": hyp, "poisson_p-value": pop, "binomial_p-value": nbp,
            "iva": "%s:%s-%s" % (chrom, iva[0], iva[1]), "ivb": "%s:%s-%s" % (chrom, ivb[0], ivb[1]),
        }
    if len(ds.keys()) == 0:
        return None
    ds = removeDup(ds)
    if len(ds.keys()) == 0:
        return None
    ds = removeDup(ds)
    if len(ds.keys()) == 0:
        return None
    ds = pd.DataFrame(ds).T
    ds["poisson_p-value_corrected"] = getBonPvalues(ds["poisson_p-value"])
    ds["binomial_p-value_corrected"] = getBonPvalues(ds["binomial_p-value"])
    ds["hypergeometric_p-value_corrected"] = getBonPvalues(ds["hypergeometric_p-value"])
    return ds


def _windows(records):
    """iva, ivb, distance and the 22 windows of every record (cModel.py:276-279 + getNearbyPairRegions
    :83-105 with Python-2 integer arithmetic), vectorised: -> (iva int64[R,2], ivb, distance float64[R],
    windows int32[R,44] = lo[22] then hi[22], A0..A10 then B0..B10)."""
    r = np.asarray([[x[1], x[2], x[4], x[5]] for x in records], dtype=np.int64).reshape(-1, 4)
    iva = np.stack([np.maximum(0, r[:, 0]), r[:, 1]], 1)
    ivb = np.stack([np.maximum(0, r[:, 2]), r[:, 3]], 1)
    distance = np.abs(ivb.sum(1) / 2.0 - iva.sum(1) / 2.0)
    ca, cb = iva.sum(1) // 2, ivb.sum(1) // 2
    sa, sb = (iva[:, 1] - iva[:, 0]) // 2, (ivb[:, 1] - ivb[:, 0]) // 2
    step = (sa + sb) // 2
    lo = np.zeros((len(r), 22), np.int64)
    hi = np.zeros((len(r), 22), np.int64)
    lo[:, 0], hi[:, 0], lo[:, 11], hi[:, 11] = iva[:, 0], iva[:, 1], ivb[:, 0], ivb[:, 1]
    k = 1
    for i in range(-5, 6):
        if i == 0:
            continue
        lo[:, k] = np.maximum(0, ca + i * step - sa)
        hi[:, k] = np.maximum(0, ca + i * step + sa)
        lo[:, 11 + k] = np.maximum(0, cb + i * step - sb)
        hi[:, 11 + k] = np.maximum(0, cb + i * step + sb)
        k += 1
    return iva, ivb, distance, np.concatenate([lo, hi], 1).astype(np.int32)


def _overlap_lists(iv, chrom):
    """for every loop i the ascending array of j > i whose two anchors both overlap i's
    (checkOverlap, cModel.py:174-195).  Candidate pairs come from a sorted sweep over the left anchors
    (two well-formed intervals can only satisfy checkOneEndOverlap if they intersect); the reference's
    exact predicate is then evaluated on the candidates, vectorised."""
    L = len(iv)

    def one_end(xa, xb, ya, yb):           # checkOneEndOverlap, cModel.py:174-182
        t1 = ((ya <= xa) & (xa <= yb)) | ((ya <= xb) & (xb <= yb)) | ((ya <= xa) & (xa <= xb) & (xb <= yb))
        t2 = ((xa <= ya) & (ya <= xb)) | ((xa <= yb) & (yb <= xb)) | ((xa <= ya) & (ya <= yb) & (yb <= xb))
        return t1 | t2
    _, cid = np.unique(chrom, return_inverse=True)
    a0, a1 = iv[:, 0], iv[:, 1]
    length = a1 - a0
    # a few very long anchors would blow up the sweep window: they are paired with everything instead
    lim = max(1, int(np.percentile(length, 99)) * 4) if L else 1
    long_idx = np.nonzero(length > lim)[0]
    short = length <= lim
    order = np.argsort(a0, kind="stable")
    S = a0[order]
    lo = np.searchsorted(S, a0 - lim, side="left")      # sorted positions whose start could still reach a0
    hi = np.searchsorted(S, a1, side="right")           # ... and do not start after a1
    cnt = np.maximum(hi - lo, 0)
    ii = np.repeat(np.arange(L), cnt)
    off = np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt)
    jj = order[np.repeat(lo, cnt) + off]
    keep = short[jj]                                    # long partners are added below
    ii, jj = ii[keep], jj[keep]
    if len(long_idx):
        ii = np.concatenate([ii, np.repeat(np.arange(L), len(long_idx))])
        jj = np.concatenate([jj, np.tile(long_idx, L)])
    m = jj > ii
    ii, jj = ii[m], jj[m]
    ok = (cid[ii] == cid[jj]) & one_end(iv[ii, 0], iv[ii, 1], iv[jj, 0], iv[jj, 1]) & \
        one_end(iv[ii, 2], iv[ii, 3], iv[jj, 2], iv[jj, 3])
    ii, jj = ii[ok], jj[ok]
    o2 = np.lexsort((jj, ii))
    ii, jj = ii[o2], jj[o2]
    if len(ii):                                         # a long anchor can be produced by both routes
        uniq = np.ones(len(ii), bool)
        uniq[1:] = (ii[1:] != ii[:-1]) | (jj[1:] != jj[:-1])
        ii, jj = ii[uniq], jj[uniq]
    bounds = np.searchsorted(ii, np.arange(L + 1))
    return [jj[bounds[i]:bounds[i + 1]] for i in range(L)]


def _remove_dup_fast(ds, bpcut=1e-5):
    """removeDup (cModel.py:198-259) with the overlap tests vectorised; same visiting order, same
    `rekeys` bookkeeping, same result."""
    keys = list(ds.keys())
    L = len(keys)
    if L == 0:
        return {}
    iv = np.asarray([parseIv(ds[k]["iva"])[1:] + parseIv(ds[k]["ivb"])[1:] for k in keys], dtype=np.int64)
    chrom = np.asarray([ds[k]["iva"].split(":")[0] + "|" + ds[k]["ivb"].split(":")[0] for k in keys])
    adj = _overlap_lists(iv, chrom)
    removed = np.zeros(L, bool)
    uniqueds, reds = {}, {}
    for i in range(L - 1):
        if removed[i]:
            continue
        hit = adj[i]
        if len(hit):
            hit = hit[~removed[hit]]
        if len(hit):
            reds[keys[i]] = [keys[i]] + [keys[j] for j in hit.tolist()]
            removed[i] = True
            removed[hit] = True
        else:
            uniqueds[keys[i]] = ds[keys[i]]
    for key in reds.keys():
        ts = {}
        for t in reds[key]:
            if ds[t]["binomial_p-value"] > bpcut:
                continue
            ts[t] = float(ds[t]["rab"]) / ds[t]["ra"] / ds[t]["rb"]
        if len(ts) == 0:
            continue
        ts = pd.Series(ts)
        ts.sort_values(inplace=True, ascending=False)
        uniqueds[ts.index[0]] = ds[ts.index[0]]
    return uniqueds


def getIntSigFromCounts(records, counts, N, minPts, discut):
    """getIntSig (cModel.py:262-331) fed with the interval counts of kernel K8 (cl_sig_counts) instead of
    Python sets: the set sizes are the SAME integers, every float is produced by the same numpy /
    scipy call on the same operands in the same order, so the rows are bit-identical."""
    if N < 2:
        return None
    iva, ivb, distance, _ = _windows(records)
    kept = []
    i = 0
    for n, r in enumerate(records):
        if distance[n] < discut:
            continue
        if counts[n, 22] < max(minPts):
            continue
        kept.append((n, "%s-%s-%s" % (r[0], r[3], i)))
        i += 1
    if not kept:
        return None
    idx = np.asarray([n for n, _ in kept])
    c = counts[idx].astype(np.int64)
    ra, rb, rab = c[:, 0], c[:, 11], c[:, 22]
    # the 10 x 10 permuted background of every record at once (cModel.py:121-158).  rabs are integer
    # counts, so their sums are exact in any order; nbps (non-integers) are reduced along the last axis,
    # which numpy sums exactly like the reference's 1-D np.mean of the same 100 values in the same order.
    nra = c[:, 1:11].astype(np.float64)                      # nralen = float(len(nra))
    nrb = c[:, 12:22]                                        # nrblen = len(nrb)  (int)
    cab = c[:, 23:].reshape(-1, 11, 11)[:, 1:, 1:].astype(np.float64)
    rabs = cab.reshape(len(idx), 100)
    den = nra[:, :, None] * nrb[:, None, :]
    with np.errstate(divide="ignore", invalid="ignore"):
        nbps = np.where(cab > 0, cab / den, 0.0).reshape(len(idx), 100)
    fdr_l = (rabs > rab[:, None]).sum(1) / float(100)
    lam_l = rabs.mean(axis=1)
    npos = (rabs > 0).sum(1)
    with np.errstate(divide="ignore", invalid="ignore"):
        es_l = np.where(lam_l > 0, rab / (rabs.sum(1) / np.maximum(npos, 1)), np.inf)
    bp_l = nbps.mean(axis=1) * ra * rb / N
    hyp = np.maximum(1e-300, hypergeom.sf(rab - 1.0, N, ra, rb))
    pop = np.maximum(1e-300, poisson.sf(rab - 1.0, np.asarray(lam_l)))
    nbp = np.maximum(1e-300, binom.sf(rab - 1.0, N - rab, np.asarray(bp_l)))
    ds = {}
    for q, (n, key) in enumerate(kept):
        chrom = records[n][0]
        ds[key] = {
            "distance": float(distance[n]), "ra": int(ra[q]), "rb": int(rb[q]), "rab": int(rab[q]), "ES": float(es_l[q]), "FDR": float(fdr_l[q]),
            "hypergeometric_p-value": float(hyp[q]), "poisson_p-value": float(pop[q]), "binomial_p-value": float(nbp[q]),
            "iva": "%s:%s-%s" % (chrom, iva[n, 0], iva[n, 1]), "ivb": "%s:%s-%s" % (chrom, ivb[n, 0], ivb[n, 1]),
        }
    ds = _remove_dup_fast(ds)
    if len(ds.keys()) == 0:
        return None
    ds = _remove_dup_fast(ds)
    if len(ds.keys()) == 0:
        return None
    ds = pd.DataFrame(ds).T
    ds["poisson_p-value_corrected"] = getBonPvalues(ds["poisson_p-value"])
    ds["binomial_p-value_corrected"] = getBonPvalues(ds["binomial_p-value"])
    ds["hypergeometric_p-value_corrected"] = getBonPvalues(ds["hypergeometric_p-value"])
    return ds


def getIntSig(f, records, minPts, discut):
    """cModel.py:262-331; `f` is a .jd path or a 'mem://' chromosome of cloops_amd.pipe.CACHE.  The
    interval counting runs on the GPU (kernel K8) on the chromosome resident in HBM."""
    from .pipe import CACHE
    if len(records) == 0:
        return None
    r = CACHE.get(f)
    _, _, _, wins = _windows(records)
    with r.lock:
        counts, N = r.chrom.sig_counts(wins, discut)
    return getIntSigFromCounts(records, counts, N, minPts, discut)


def _mark(ds, sig):
    ds["significant"] = sig.astype(float)                 # 1.0 / 0.0, the column the reference writes
    return ds


def markIntSig(ds, escut=2.0, fdrcut=1e-2, bpcut=1e-3, ppcut=1e-5, hypcut=1e-10):
    """cModel.py:334-362 as one boolean mask: every cut must hold (enrichment, local FDR, the three p-values)"""
    col = lambda name: ds[name].astype(float)
    return _mark(ds, (col("ES") >= escut) & (col("FDR") <= fdrcut) & (col("hypergeometric_p-value") <= hypcut)
                 & (col("poisson_p-value") <= ppcut) & (col("binomial_p-value") <= bpcut))


def markIntSigHic(ds, escut=2.0, fdrcut=0.01, bpcut=1e-5, ppcut=1e-5):
    """cModel.py:365-386 (Hi-C / HiChIP cuts: strict `<` on the FDR, no hypergeometric cut)"""
    col = lambda name: ds[name].astype(float)
    return _mark(ds, (col("ES") >= escut) & (col("FDR") < fdrcut) & (col("poisson_p-value") <= ppcut)
                 & (col("binomial_p-value") <= bpcut))


def runStat(dataI, minPts, cut, cpu, fout, hichip=0):
    """cLoops/pipe.py:177-203: significance for every chromosome, mark, write `<fout>.loop`."""
    ds = [getIntSig(dataI[key]["f"], dataI[key]["records"], minPts, cut) for key in dataI.keys()]
    ds = [d for d in ds if d is not None]
    if len(ds) == 0:
        return 1
    ds = pd.concat(ds)
    try:
        ds = markIntSigHic(ds) if hichip else markIntSig(ds)
        ds.to_csv(fout + ".loop", sep="\t", index_label="loopId")
    except Exception:
        ds.to_csv(fout + "_raw.loop", sep="\t", index_label="loopId")
    return 0
