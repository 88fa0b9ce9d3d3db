"""One-off GPU fuzz of the count cache with an announced minPts list (cl_set_count_thresholds; developer tool): seeded random
dense / clumpy data, random lists, runs in random order (values of the list, values outside it, cuts moving up and down)
on a handle with the cache against a handle without, a sample against the sequential oracle.
python tools/fuzz_lists.py [seed] [ncases]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle
from cloops_amd import api

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 120
rng = np.random.default_rng(seed)
oracle.build()
bad = runs = 0
t0 = time.time()
for k in range(ncases):
    eps = int(rng.choice([300, 1000, 5000, 7500]))
    n = int(rng.integers(20000, 400000))
    nstrips = int(rng.integers(8, 2000))                                  # mean strip population 10 .. tens of thousands
    L = max(4 * eps, nstrips * eps // 2)
    # clusters of random size around random anchors + a background whose distance is log-uniform
    ncl = int(rng.integers(1, max(2, n // 60)))
    size = rng.integers(5, 140, ncl)
    which = np.repeat(np.arange(ncl), size)[: n // 2]
    ax = rng.integers(0, L, ncl); span = rng.integers(0, 20 * eps, ncl)
    sig = float(rng.choice([0.05, 0.2, 0.6])) * eps
    cx = ax[which] + rng.normal(0, sig, len(which)); cy = ax[which] + span[which] + rng.normal(0, sig, len(which))
    nb = n - len(which)
    bx = rng.integers(0, L, nb); by = bx + np.exp(rng.uniform(np.log(10), np.log(L), nb))
    X = np.abs(np.concatenate([cx, bx])).astype(np.int64); Y = np.abs(np.concatenate([cy, by])).astype(np.int64)
    X, Y = np.minimum(X, Y), np.maximum(X, Y)
    p = rng.permutation(len(X))
    X = np.ascontiguousarray(X[p], dtype=np.int32); Y = np.ascontiguousarray(Y[p], dtype=np.int32)
    served = sorted({int(v) for v in rng.integers(2, 129, int(rng.integers(1, 7)))}, reverse=True)
    seq = [(served[0], 0)]
    for _ in range(int(rng.integers(4, 10))):
        m = int(rng.choice(served)) if rng.random() < 0.8 else int(rng.integers(2, 129))
        seq.append((m, int(rng.integers(0, 6 * eps)) if rng.random() < 0.8 else 0))
    variant = "v2" if k % 2 == 0 else "v1"
    a = api.Chromosome(X, Y); b = api.Chromosome(X, Y)
    b.set_count_reuse(False)
    a.set_count_thresholds(served)
    modes = []
    for j, (m, cut) in enumerate(seq):
        ra = a.cluster(variant, eps, m, cut); modes.append(a.last_region_mode())
        rb = b.cluster(variant, eps, m, cut)
        runs += 1
        if not (np.array_equal(ra.labels, rb.labels) and np.array_equal(ra.boxes, rb.boxes)):
            bad += 1
            print("MISMATCH case %d run %d %s n=%d eps=%d served=%s (minPts %d, cut %d) modes %s: %d rows differ" % (
                k, j, variant, len(X), eps, served, m, cut, modes, int((ra.labels != rb.labels).sum())))
        if j == len(seq) - 1 and k % 4 == 0:
            want = oracle.single_dbscan(variant, X, Y, eps, m, cut)["labels"]
            if not np.array_equal(ra.labels, want):
                bad += 1
                print("ORACLE MISMATCH case %d %s n=%d eps=%d minPts %d cut %d" % (k, variant, len(X), eps, m, cut))
    a.close(); b.close()
    if k % 20 == 19:
        print("case %d: %d runs, %d bad, %.0f s; last modes %s" % (k, runs, bad, time.time() - t0, modes), flush=True)
print("done: %d cases, %d runs, %d bad, %.0f s" % (ncases, runs, bad, time.time() - t0))
sys.exit(1 if bad else 0)
