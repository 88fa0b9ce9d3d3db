"""Developer check of the list form (k_lists.hip): the same runs at every traversal level against the oracle, on a
mid-size synthetic chromosome, with and without a cut, re-used words included.  python tools/lists_check.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from cloops_amd import api
from cloops_amd.synth import synth_chrom

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
oracle.build()
X, Y = synth_chrom(n, 46709983, 11)
bad = 0
for variant in ("v2", "v1"):
    for (eps, m, cut) in ((2000, 5, 0), (5000, 20, 0), (5000, 10, 3000), (7500, 30, 4000), (500, 3, 0)):
        keep = (Y.astype(np.int64) - X >= cut)
        want = np.full(n, -1, np.int32)
        want[keep] = oracle.labels(variant, X[keep], Y[keep], eps, m)
        for level in (0, 1, 2, 3, 4):
            ch = api.Chromosome(X, Y)
            ch.set_traversal(level)
            t0 = time.time()
            got = ch.cluster(variant, eps, m, cut).labels
            ok = np.array_equal(want, got)
            if cut > 0:
                # a second run at another cut re-uses the first run's words (band query, words read in place)
                cut2 = cut + eps // 3
                keep2 = (Y.astype(np.int64) - X >= cut2)
                want2 = np.full(n, -1, np.int32)
                want2[keep2] = oracle.labels(variant, X[keep2], Y[keep2], eps, m)
                got2 = ch.cluster(variant, eps, m, cut2).labels
                ok2 = np.array_equal(want2, got2)
            else:
                ok2 = True
            ch.close()
            print("%s eps=%d minPts=%d cut=%d level=%d: %s %s (%d labelled, %d differ)" % (
                variant, eps, m, cut, level, "ok" if ok else "MISMATCH", "ok" if ok2 else "MISMATCH(reuse)",
                int((want >= 0).sum()), int((want != got).sum())), flush=True)
            bad += (not ok) + (not ok2)
print("lists_check: %d mismatches" % bad)
sys.exit(1 if bad else 0)
