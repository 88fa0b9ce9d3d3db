"""Developer model (CPU, numpy): which PETs the region query can settle without walking candidates, on a slice of chr1 of the 200 M-PET genome at the
sweep's three (eps, cut) settings and minPts list {20,30,40,50}: round 5's bracket (own strip + both q windows) against class counters over K
thirds / quarters of the strip with the upper window ends searched `cap` entries deep.   python tools/k2_bounds_model.py"""
import sys, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from cloops_amd.synth import synth_chrom, chrom_sizes
name, length, n = chrom_sizes(200000000)[0]
X, Y = synth_chrom(n, length, 3000)
q_all = (Y.astype(np.int64) - X); p_all = (X.astype(np.int64) + Y)
thr = np.array([20, 30, 40, 50])
def nxt(c):  # smallest threshold > c, else 10**9
    idx = np.searchsorted(thr, c, side="right")
    return np.where(idx < len(thr), thr[np.minimum(idx, len(thr) - 1)], 10**9)
for eps, cut in ((5000, 0), (7500, 5711), (10000, 5517)):
    keep = q_all >= cut
    q = q_all[keep]; p = p_all[keep]
    strip = p // eps; r = p % eps
    S0, NS = 40000 * 5000 // eps, 600
    sel = (strip >= S0 - 1) & (strip <= S0 + NS)
    q, strip, r = q[sel], strip[sel], r[sel]
    o = np.lexsort((q, strip)); q, strip, r = q[o], strip[o], r[o]
    ss = np.searchsorted(strip, np.arange(S0 - 1, S0 + NS + 2))
    tot = dict(n=0, easy=0, own50=0, ph2=0, settled_now=0, defer_now=0)
    cls = {(K, cap): dict(settled=0, defer=0, walk=0) for K in (2, 3, 4) for cap in (31, 63, 127)}
    walk_now = 0
    for k in range(1, NS + 1):
        b, e = ss[k], ss[k + 1]; tb, te = ss[k - 1], ss[k + 2]
        if e == b: continue
        qs = q[b:e]; rs = r[b:e]
        lo = np.searchsorted(qs, qs - eps, "left"); hi = np.searchsorted(qs, qs + eps, "right")
        idx = np.arange(e - b)
        c = hi - lo
        easy = ((hi - idx - 1) >= 49) | ((idx - lo) >= 49)
        tot["n"] += e - b; tot["easy"] += easy.sum()
        h1 = ~easy
        own50 = h1 & (c >= 50); tot["own50"] += own50.sum()
        h2 = h1 & (c < 50); tot["ph2"] += h2.sum()
        qa, ra = q[tb:b], r[tb:b]; qb, rb = q[e:te], r[e:te]
        ja = np.searchsorted(qa, qs - eps, "left"); ka = np.searchsorted(qa, qs + eps, "right")
        jb = np.searchsorted(qb, qs - eps, "left"); kb = np.searchsorted(qb, qs + eps, "right")
        nA = np.minimum(ka - ja, 31); nB = np.minimum(kb - jb, 31)
        ub = c + nA + nB
        sett = ub < nxt(c)
        tot["settled_now"] += (h2 & sett).sum(); d = h2 & ~sett; tot["defer_now"] += d.sum()
        walk_now += ((ka - ja) + (kb - jb))[d].sum()
        for (K, cap) in cls:
            phi = rs * K // eps; pa = ra * K // eps; pb = rb * K // eps
            for i in np.nonzero(h2)[0]:
                wa = pa[ja[i]:min(ka[i], ja[i] + cap)]; wb = pb[jb[i]:min(kb[i], jb[i] + cap)]
                capped = (ka[i] - ja[i] >= cap) or (kb[i] - jb[i] >= cap)
                lowr = c[i] + (wa > phi[i]).sum() + (wb < phi[i]).sum()
                upr = lowr + (wa == phi[i]).sum() + (wb == phi[i]).sum()
                if lowr >= 50 or (not capped and upr < nxt(lowr)): cls[(K, cap)]["settled"] += 1
                else:
                    cls[(K, cap)]["defer"] += 1; cls[(K, cap)]["walk"] += (ka[i] - ja[i]) + (kb[i] - jb[i])
    print("eps %d cut %d: n %d easy(ph0) %.3f own>=50 %.3f phase2 %.3f | now: settled %.3f deferred %.3f avg walk %.1f" % (
        eps, cut, tot["n"], tot["easy"] / tot["n"], tot["own50"] / tot["n"], tot["ph2"] / tot["n"], tot["settled_now"] / tot["n"], tot["defer_now"] / tot["n"], walk_now / max(1, tot["defer_now"])))
    for K in cls:
        print("   classes K=%s: settled %.3f deferred %.3f avg walk %.1f" % (K, cls[K]["settled"] / tot["n"], cls[K]["defer"] / tot["n"], cls[K]["walk"] / max(1, cls[K]["defer"])))
