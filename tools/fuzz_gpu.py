"""One-off GPU-vs-oracle fuzz (developer tool): many seeded random cases, all variants, with and
without a cut, small and medium sizes, sparse and dense strips.  python tools/fuzz_gpu.py [seed] [ncases]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle
from cloops_amd import api
import cases

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 600
rng = np.random.default_rng(seed)
oracle.build()
bad = 0
t0 = time.time()
for k in range(ncases):
    kind = k % 4
    if kind == 0:
        ids, X, Y, eps, minPts = cases.adversarial_case(rng, k)
    elif kind == 1:
        ids, X, Y, eps, minPts = cases.clumpy_case(rng, k)
    elif kind == 2:
        # dense diagonal band: long strips, many chains, big components
        n = int(rng.integers(2000, 60000))
        eps = int(rng.choice([200, 1000, 5000]))
        minPts = int(rng.choice([3, 5, 20, 50]))
        L = int(rng.integers(20, 400)) * eps
        X = rng.integers(0, L, n)
        Y = X + np.abs(rng.normal(0, float(rng.choice([0.5, 2, 10])) * eps, n)).astype(np.int64)
        ids = np.arange(n)
    else:
        n = int(rng.integers(1000, 40000))
        eps = int(rng.choice([50, 300, 2000]))
        minPts = int(rng.choice([2, 4, 5, 10]))
        L = int(rng.integers(50, 3000)) * eps
        X = rng.integers(0, L, n)
        Y = X + rng.integers(0, L // 2 + 1, n)
        # pile-ups
        m = n // 10
        src = rng.integers(0, n, m)
        X[:m] = X[src] + rng.integers(-eps, eps + 1, m) // 4
        X = np.abs(X)
        Y[:m] = np.maximum(X[:m], Y[src] + rng.integers(-eps, eps + 1, m) // 4)
        ids = np.arange(n)
    X = np.ascontiguousarray(X, dtype=np.int32); Y = np.ascontiguousarray(Y, dtype=np.int32)
    cut = 0 if k % 3 else int(rng.integers(0, 4 * eps))
    ch = api.Chromosome(X, Y)
    keep = (Y.astype(np.int64) - X >= cut) if cut > 0 else np.ones(len(X), bool)
    for variant in ("v1", "v2", "block"):
        if keep.sum() == 0:
            continue
        want = np.full(len(X), -1, np.int32)
        want[keep] = oracle.labels(variant, X[keep], Y[keep], eps, minPts)
        got = ch.cluster(variant, eps, minPts, cut).labels
        if not np.array_equal(want, got):
            bad += 1
            print("MISMATCH case %d kind %d variant %s n=%d eps=%d minPts=%d cut=%d: %d rows differ" % (
                k, kind, variant, len(X), eps, minPts, cut, int((want != got).sum())))
        if variant != "block" and cut > 0:
            # the same handle at another cut: the run re-uses the K2 words of the first (band query, words read in place)
            cut2 = cut + int(rng.integers(1, 2 * eps))
            keep2 = Y.astype(np.int64) - X >= cut2
            if keep2.sum():
                want2 = np.full(len(X), -1, np.int32)
                want2[keep2] = oracle.labels(variant, X[keep2], Y[keep2], eps, minPts)
                got2 = ch.cluster(variant, eps, minPts, cut2).labels
                if not np.array_equal(want2, got2):
                    bad += 1
                    print("MISMATCH (re-used words, mode %d) case %d kind %d variant %s n=%d eps=%d minPts=%d cut=%d -> %d: %d rows differ" % (
                        ch.last_region_mode(), k, kind, variant, len(X), eps, minPts, cut, cut2, int((want2 != got2).sum())))
    ch.close()
print("fuzz seed %d: %d cases x 3 variants, %d mismatches, %.1f s" % (seed, ncases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
