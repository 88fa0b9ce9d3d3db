"""Re-sampling of one chromosome and loop calling per sample -- the second caller of cDBSCAN (variant 1) in the reference,
scripts/jd2saturation, on chromosomes resident in HBM.

    generateSamplingData   scripts/jd2saturation:32-55   rows drawn exactly like the script (one index array shuffled again and
                                                         again by numpy's legacy global generator, the first int(n * depth) of it
                                                         per sample); the sample is gathered ON THE DEVICE from the resident
                                                         chromosome (cl_chrom_subsample) -- no .jd files, only the row list moves
    singleDBSCAN, runDBSCAN :56-127                      the script's own copy of the dispatch: `parseJd(f, cut)` drops the short
                                                         PETs BEFORE clustering and, unlike cLoops/pipe.py:59-63, does NOT add
                                                         their distances to `dss`
    getLoops               :154-178                      one clustering run per eps at a FIXED cut (no chain), every run's boxes
                                                         filtered by that run's own estimated cut, then combineTwice
    getSets, getSaturation :181-214

The script cannot run in the reference as shipped: `from cLoops.pipe import checkOverlap` (:25) names a function
cLoops/pipe.py does not have (the one it means is cLoops/bk.py:12-19), its runStat hands cModel.getIntSig an int where
cModel.py:290 takes `max(minPts)` of a list, and :246 joins a tuple -- so its `main` has no behaviour to pin; the functions above are
pinned by golden vectors made from the script's OWN functions wired to the real cDBSCAN class
(tests/golden/make_golden_saturation.py).
"""
import os

import numpy as np

from . import pipe
from .ests import estIntSelCutFrag

VARIANT = "v1"            # scripts/jd2saturation:23  `from cLoops.cDBSCAN import cDBSCAN as DBSCAN`


def _sample_name(depth, rep, key):
    return "mem://depth_%s_rep_%s/%s-%s" % (depth, rep, key[0], key[1])


def generateSamplingData(jd, repeats, step, cut=0):
    """scripts/jd2saturation:32-55.  `jd`: a .jd path or a 'mem://' chromosome of pipe.CACHE.  Seed numpy's legacy generator
    (np.random.seed) before the call for a reproducible draw, like any user of the script would.  Returns the names of the
    samples, registered in pipe.CACHE (close them with pipe.CACHE.clear() or drop())."""
    r = pipe.CACHE.get(jd)
    keep = np.arange(len(r.d)) if cut <= 0 else np.where(r.d >= cut)[0]          # parseJd(jd, cut), io.py:213-216
    n = len(keep)
    fs = []
    ss = np.arange(n)
    for i in np.arange(1.0 / step, 1.0, 1.0 / step):
        m = int(n * i)
        for rep in range(repeats):
            np.random.shuffle(ss)
            rows = keep[ss[:m]]                                                   # mat[ns, :] of the FILTERED matrix
            with r.lock:
                chrom = r.chrom.subsample(rows)
            name = _sample_name(i, rep, r.key)
            pipe.CACHE.put_chrom(name, chrom, r.X[rows], r.Y[rows], ids=r.ids[rows], key=r.key, device=r.device)
            fs.append(name)
    return fs


def singleDBSCAN(f, eps, minPts, cut=0):
    """scripts/jd2saturation:56-108 -> (key, f, dataI, dataS, dis, dss), record rows [chrA, minX, maxX, chrB, minY, maxY]"""
    r = pipe.CACHE.get(f)
    n_short = int((r.d < cut).sum()) if cut > 0 else 0
    if len(r.d) - n_short == 0:                                                   # :63-64
        return r.key, f, [], [], [], []
    dataI, dataS, dis, dss, _, _, _ = pipe._cluster_arrays(r, eps, minPts, cut, VARIANT)
    dss = dss[n_short:]                          # the short PETs were dropped by parseJd(f, cut): they are in no list here
    dis = list(dis) if len(dataI) else []                                         # :103-106
    dss = list(dss) if len(dataS) else []
    return r.key, f, pipe._records(r.key, dataI), pipe._records(r.key, dataS), dis, dss


def runDBSCAN(fs, eps, minPts, cut):
    """scripts/jd2saturation:111-127"""
    dataI, dataS, dis, dss = {}, [], [], []
    for f in fs:
        d = singleDBSCAN(f, eps, minPts, cut)
        if len(d[2]) == 0:
            continue
        dataI[d[0]] = {"f": d[1], "records": d[2]}
        dataS.extend(d[3])
        dis.extend(d[4])
        dss.extend(d[5])
    return dataI, dataS, dis, dss


def sample_depth(jd):
    """the depth the script reads back from a sample's directory name (:161-163)"""
    return float(jd.split("/")[-2].split("_")[1])


def callLoops(jd, eps, minPts, cut, cd=1):
    """the clustering part of getLoops (:154-176) -> (dataI, cut, minPts used, cuts).  cd: scale minPts by the sampling depth."""
    if cd:
        minPts = int(sample_depth(jd) * minPts)
    dataI = {}
    cuts = []
    for ep in eps:
        dataI_2, dataS_2, dis_2, dss_2 = runDBSCAN([jd], ep, minPts, cut)
        if len(dataI_2) == 0 or len(dataS_2) == 0:
            continue
        cut_2, frags = estIntSelCutFrag(np.array(dis_2), np.array(dss_2))
        cuts.append(cut_2)
        dataI_2 = pipe.filterClusterByDis(dataI_2, cut_2)
        dataI = pipe.combineTwice(dataI, dataI_2)
    return dataI, min(cuts), minPts, cuts


def getLoops(jd, eps, minPts, hic, cut, fout, cd=1):
    """scripts/jd2saturation:154-178: loops of one (re-sampled) chromosome -> `<fout>.loop`"""
    from . import cModel
    floop = fout + ".loop"
    if os.path.isfile(floop):
        return floop
    dataI, cut, minPts, _ = callLoops(jd, eps, minPts, cut, cd)
    # :130-151.  (The script hands getIntSig its int minPts, where cModel.py:290 takes max(minPts) of a list -- one more reason
    # it cannot run against the library it ships with; a one-element list is what that line evidently wants.)
    cModel.runStat(dataI, [minPts], cut, 1, fout, hic)
    return floop


def checkOneEndOverlap(xa, xb, ya, yb):
    """cLoops/bk.py:1-9"""
    return (ya <= xa <= yb) or (ya <= xb <= yb) or (xa <= ya <= xb) or (xa <= yb <= xb)


def checkOverlap(ra, rb):
    """cLoops/bk.py:12-19 (what scripts/jd2saturation:25 means to import): both anchors overlap"""
    return checkOneEndOverlap(ra[1], ra[2], rb[1], rb[2]) and checkOneEndOverlap(ra[4], ra[5], rb[4], rb[5])


def getSets(f):
    """scripts/jd2saturation:181-192: the significant loops of a `.loop` table as [chrA, startA, endA, chrB, startB, endB]"""
    import pandas as pd
    from .cModel import parseIv
    mat = pd.read_csv(f, sep="\t", index_col=0)
    s = mat["significant"]
    s = s[s > 0]
    rs = []
    for i in s.index:
        a = parseIv(mat.loc[i, "iva"])
        b = parseIv(mat.loc[i, "ivb"])
        rs.append([a[0], a[1], a[2], b[0], b[1], b[2]])
    return rs


def getSaturation(fa, fbs, fout):
    """scripts/jd2saturation:195-214: per depth and replicate, the share of the full data's significant loops that a sample
    recovers (an overlapping significant loop) -> `<fout>_ResamplingRatios.txt`"""
    import pandas as pd
    ds = {}
    rsa = getSets(fa)
    for f in fbs:
        n = os.path.splitext(os.path.split(f)[-1])[0].split("_")
        d, r = float(n[1]), int(n[-1])
        rsb = getSets(f)
        c = sum(1 for ra in rsa if any(checkOverlap(ra, rb) for rb in rsb))
        ds.setdefault(d, {})[r] = c
    ds = pd.DataFrame(ds) / len(rsa) * 100
    ds.to_csv(fout + "_ResamplingRatios.txt", sep="\t", index_label="replicates")
    return ds


def jd2saturation(jd, fout, eps, minPts, repeats, step, cpu=1, hic=0, cut=0):
    """scripts/jd2saturation:217-247 (cpu is accepted for the signature: the samples run one after the other on the GPU)"""
    if os.path.isdir(fout):
        return None
    os.mkdir(fout)
    floop = getLoops(jd, eps, minPts, hic, cut, os.path.join(fout, os.path.basename(fout)), cd=0)
    fs = generateSamplingData(jd, repeats, step, cut)
    out = []
    for f in fs:
        out.append(getLoops(f, eps, minPts, hic, cut, os.path.join(fout, f.split("/")[-2]), cd=1))
        pipe.CACHE.drop(f)
    return getSaturation(floop, out, os.path.join(fout, os.path.basename(fout)))
