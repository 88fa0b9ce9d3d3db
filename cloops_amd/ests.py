"""Distance-cutoff estimation between self-ligation and inter-ligation PETs.

Host-side restatement of cLoops/ests.py:36-61 (`estIntSelCutFrag`) -- the tiny float step
that closes the (eps, minPts) sweep chain of cLoops/pipe.py:247-275.  numpy float64, same
operations in the same order as the reference (abs, drop NaN, drop <= 0, log2, median +
3 sigma vs the sigma-weighted mean of the two means, take the smaller, 2**cut truncated)."""
import numpy as np


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
