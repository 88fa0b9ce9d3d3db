"""One-off GPU-vs-oracle fuzz of cl_cluster_weighted (developer tool): python tools/fuzz_weighted.py [seed] [ncases]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from cloops_amd import api

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rng = np.random.default_rng(seed)
oracle.build()
bad = 0
t0 = time.time()
for k in range(ncases):
    n = int(rng.integers(50, 30000))
    wx, wy = [(50, 1), (1, 50), (1, 1), (int(rng.integers(1, 200)), int(rng.integers(1, 200)))][k % 4]
    eps = int(rng.choice([300, 5000, 20000, 200000]))
    minPts = int(rng.choice([2, 3, 5, 12]))
    x0 = int(rng.choice([-2 ** 28, 0, 10 ** 8, 2 ** 29 - 10 ** 7]))
    spanx = int(rng.integers(1, 400)) * max(1, eps // wx)
    spany = int(rng.integers(1, 400)) * max(1, eps // wy)
    X = x0 + rng.integers(0, min(spanx, 5 * 10 ** 6) + 1, n)
    Y = X + rng.integers(0, min(spany, 5 * 10 ** 6) + 1, n)
    if k % 3 == 0:
        m = n // 3
        if k % 2: X[:m] = X[0] + rng.integers(-2, 3, m)
        else: Y[:m] = Y[0] + rng.integers(-2, 3, m)
    X = np.clip(X, -2 ** 29 + 1, 2 ** 29 - 1).astype(np.int32); Y = np.clip(Y, -2 ** 29 + 1, 2 ** 29 - 1).astype(np.int32)
    ch = api.Chromosome(X, Y)
    got = ch.cluster_weighted(eps, minPts, wx, wy).labels
    ch.close()
    want = oracle.labels("v1", X.astype(np.int64) * wx, Y.astype(np.int64) * wy, eps, minPts)
    if not np.array_equal(got, want):
        bad += 1
        print("MISMATCH case %d n=%d eps=%d minPts=%d w=(%d,%d): %d rows differ" % (k, n, eps, minPts, wx, wy, int((got != want).sum())))
print("weighted fuzz seed %d: %d cases, %d mismatches, %.1f s" % (seed, ncases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
