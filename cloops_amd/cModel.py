"""Significance of candidate loops against the local permuted background, and the `.loop` table.

What cLoops/cModel.py computes per candidate loop (getIntSig :262-331 with getPETsforRegions :73-80,
getNearbyPairRegions :83-105, getMultiplePsFdr :108-161, getBonPvalues :164-171, removeDup :198-259,
markIntSig :334-362, markIntSigHic :365-386) and runStat (cLoops/pipe.py:177-203), restructured: the
interval counting (getGenomeCoverage :45-57, getCounts :60-70 and every set union / intersection) is
kernel K8 on the chromosome resident in HBM (`cl_sig_counts`), the statistics of ALL records of a
chromosome are evaluated as array expressions on the count table, and the duplicate removal works on a
sorted sweep instead of the reference's O(L^2) pair loop.  There is no per-record Python path here; the
cross-check against the reference's own functions lives in tests/ (tests/refload.py).

Semantics pinned: the reference is Python 2 -- `/` on the integer
interval ends in getNearbyPairRegions is FLOOR division (:89-93) and dicts iterate in insertion
order under the Python-3 conversion the goldens were made with (removeDup :206-259 depends on
that order).  The set sizes of the reference are the integers K8 returns; every float is produced by
the same scipy / numpy call on the same operands in the same order, so the `.loop` rows are
bit-identical to the converted reference.
"""
import numpy as np
import pandas as pd
from scipy.stats import hypergeom, binom, poisson



def parseIv(iv):
    """cLoops/io.py parseIv: 'chr1:100-200' -> ['chr1', 100, 200]"""
    chrom, rest = iv.split(":")
    a, b = rest.split("-")
    return [chrom, int(a), int(b)]


def getBonPvalues(ps):
    """Bonferroni correction (cModel.py:164-171): p * number of tests, capped at 1"""
    ps = np.asarray(ps, dtype=float)
    return np.minimum(ps * len(ps), 1.0)


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
    for members in reds.values():
        best = _best_of_group(ds, members, bpcut)
        if best is not None:
            uniqueds[best] = ds[best]
    return uniqueds


def _best_of_group(ds, members, bpcut):
    """the survivor of one group of overlapping loops (cModel.py:247-258): among the members whose binomial
    p-value passes `bpcut`, the one with the largest rab / ra / rb -- picked with the same pandas sort as the
    reference so that ties resolve identically; None when no member passes"""
    cand = [t for t in members if not ds[t]["binomial_p-value"] > bpcut]
    if not cand:
        return None
    score = pd.Series({t: float(ds[t]["rab"]) / ds[t]["ra"] / ds[t]["rb"] for t in cand})
    return score.sort_values(ascending=False).index[0]


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
