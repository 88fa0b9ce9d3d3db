"""Distance-cutoff estimation between self-ligation and inter-ligation PETs.

Host-side restatement of cLoops/ests.py:36-61 (`estIntSelCutFrag`) -- the tiny float step
that closes the (eps, minPts) sweep chain of cLoops/pipe.py:247-275.  numpy float64, same
operations in the same order as the reference (abs, drop NaN, drop <= 0, log2, median +
3 sigma vs the sigma-weighted mean of the two means, take the smaller, 2**cut truncated)."""
import numpy as np


def estFragSize(ds, top=500):
    """cLoops/ests.py:23-33: the fragment size behind `eps = 0` (pipe.py:237-239: eps = [2 * frags]) -- the median of the
    `top` most frequent distances between PETs mapped to different strands.  Same pandas calls as the reference (a
    Series of the distance counts, `sort_values(ascending=False)` with its default sort, the first `top` index values):
    which of several equally frequent distances make the cut is whatever pandas' sort does, here as there."""
    from collections import Counter
    import pandas as pd
    ds = pd.Series(Counter(ds))
    ds.sort_values(inplace=True, ascending=False)
    ds = ds[:top]
    return int(np.median(ds.index))


def estIntSelCutFrag(di, ds, log=1):
    """di: distances of PETs in inter-ligation clusters; ds: of self-ligation PETs.
    Returns (rcut, rfrags) as Python ints (ests.py:57,60)."""
    di = np.abs(np.asarray(di, dtype=np.float64))
    ds = np.abs(np.asarray(ds, dtype=np.float64))
    di = di[~np.isnan(di)]
    ds = ds[~np.isnan(ds)]
    di = di[di > 0]
    ds = ds[ds > 0]
    if log:
        di = np.log2(di)
        ds = np.log2(ds)
    ds_std, di_std = ds.std(), di.std()
    cut1 = np.median(ds) + 3 * ds_std
    cut2 = (ds.mean() * ds_std + di.mean() * di_std) / (ds_std + di_std)
    cut = min([cut1, cut2])
    rcut = int(2 ** cut)
    frags = np.median(ds)
    rfrags = int(2 ** frags)
    return rcut, rfrags


def estIntSelCutFrag_from_stats(n_pos, sumlog, sqdev, median_pair, with_margin=False):
    """The same estimator from pre-reduced statistics (cl_dist_summary / cl_dist_bin_hist
    of the GPU library) instead of the raw distance lists:
      n_pos      [inter, self]  number of non-zero distances
      sumlog     [inter, self]  sum of log2|d|
      sqdev      [inter, self]  sum of (log2|d| - mean)^2
      median_pair (lo, hi)      the two middle order statistics of the self-group |d| (equal for odd n)
    Same formulas, same order of operations as ests.py:49-60; only the summation order inside
    mean / std differs from numpy's pairwise sums (the result is truncated to int)."""
    di_mean = sumlog[0] / n_pos[0]
    ds_mean = sumlog[1] / n_pos[1]
    di_std = np.sqrt(sqdev[0] / n_pos[0])
    ds_std = np.sqrt(sqdev[1] / n_pos[1])
    lo, hi = median_pair
    ds_median = (np.log2(np.float64(lo)) + np.log2(np.float64(hi))) / 2 if lo != hi else np.log2(np.float64(lo))
    cut1 = ds_median + 3 * ds_std
    cut2 = (ds_mean * ds_std + di_mean * di_std) / (ds_std + di_std)
    cut = min([cut1, cut2])
    rcut = int(2 ** cut)
    rfrags = int(2 ** ds_median)
    if with_margin:
        # distance of 2**cut from the nearest integer: the sums behind `cut` are reduced in a different order
        # than numpy's pairwise sums, so a value this close to an integer could truncate differently
        raw = float(2 ** cut)
        return rcut, rfrags, abs(raw - round(raw))
    return rcut, rfrags


# ---- the log-binned first level of the exact median (cl_dist_summary of include/cloops_hip.h) -----------------
def logbin(d):
    """bin of a distance d >= 1: floor(log2 d) * 128 + the 7 bits below the leading one (monotone in d)"""
    d = int(d)
    e = d.bit_length() - 1
    m = ((d >> (e - 7)) if e >= 7 else (d << (7 - e))) & 127
    return e * 128 + m


def logbin_range(b):
    """[lo, hi) of the distances that fall into log bin b"""
    e, m = divmod(int(b), 128)
    if e >= 7:
        return (128 + m) << (e - 7), (128 + m + 1) << (e - 7)
    lo = (128 + m) >> (7 - e)
    return lo, lo + 1
