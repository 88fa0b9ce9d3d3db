"""One-off GPU fuzz of the layouts merged from the fine layout (cl_set_eps_list -> k_layout_from_fine; developer tool): random eps lists
with a common divisor (k = 1 .. 8 runs per strip), data with dense blobs, gaps of empty strips, pile-ups and large distances (many
fine strips per tile, long runs: the sorted-key searches and their fall-backs), every eps of the list on a handle with the list
announced against a handle that sorts every layout, a sample against the oracle.   python tools/fuzz_layouts.py [seed] [ncases]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle
from cloops_amd import api

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(seed)
oracle.build()
bad = runs = 0
t0 = time.time()
for k in range(ncases):
    w = int(rng.choice([16, 100, 500, 2500]))
    ks = sorted({int(v) for v in rng.integers(1, 9, int(rng.integers(2, 5)))})
    if len(ks) < 2:
        ks = [1, 2]
    eps_list = [w * v for v in ks]
    n = int(rng.integers(20000, 300000))
    nstrips = int(rng.integers(20, 6000))
    L = max(8 * w, nstrips * w // 2)
    parts_x, parts_y = [], []
    nb = n // 2
    bx = rng.integers(0, L, nb)
    by = bx + np.exp(rng.uniform(np.log(5), np.log(max(max(L, 10) * float(rng.choice([0.01, 1.0])), 6.0)), nb)).astype(np.int64)
    parts_x.append(bx); parts_y.append(by)
    ncl = max(1, (n - nb) // int(rng.integers(10, 300)))
    ax = rng.integers(0, L, ncl); span = rng.integers(0, 40 * w, ncl)
    which = rng.integers(0, ncl, n - nb)
    sig = float(rng.choice([0.05, 0.5, 3.0])) * w
    parts_x.append(np.abs(ax[which] + rng.normal(0, sig, n - nb)).astype(np.int64))
    parts_y.append(np.abs(ax[which] + span[which] + rng.normal(0, sig, n - nb)).astype(np.int64))
    if rng.random() < 0.5:                                                  # a pile-up inside one fine strip
        m = int(rng.integers(500, 6000))
        px = int(rng.integers(0, L)) + rng.integers(0, max(w // 2, 1) + 1, m)
        parts_x.append(px); parts_y.append(px + rng.integers(0, 3 * w + 1, m))
    if rng.random() < 0.4:                                                  # a second blob behind a gap of empty strips
        off = L + int(rng.integers(100, 9000)) * w
        m = n // 4
        gx = off + rng.integers(0, max(L // 4, 1), m)
        parts_x.append(gx); parts_y.append(gx + rng.integers(0, 20 * w, m))
    X = np.concatenate(parts_x); Y = np.concatenate(parts_y)
    X, Y = np.minimum(X, Y), np.maximum(X, Y)
    p = rng.permutation(len(X))
    X = np.ascontiguousarray(X[p], dtype=np.int32); Y = np.ascontiguousarray(Y[p], dtype=np.int32)
    a = api.Chromosome(X, Y); b = api.Chromosome(X, Y)
    a.set_sort_index(1); a.set_eps_list(eps_list)
    b.set_sort_index(-1)
    variant = "v2" if k % 2 == 0 else "v1"
    minPts = int(rng.choice([3, 5, 20]))
    for j, ep in enumerate(eps_list):
        cut = int(rng.integers(0, 3 * ep)) if rng.random() < 0.5 else 0
        ra = a.cluster(variant, ep, minPts, cut); rb = b.cluster(variant, ep, minPts, cut)
        runs += 1
        ok = np.array_equal(ra.labels, rb.labels)
        if ok and j == len(eps_list) - 1 and k % 3 == 0:
            ok = np.array_equal(ra.labels, oracle.single_dbscan(variant, X, Y, ep, minPts, cut)["labels"])
        if not ok:
            bad += 1
            print("MISMATCH case %d %s n=%d w=%d eps=%d (list %s) minPts %d cut %d" % (k, variant, len(X), w, ep, eps_list, minPts, cut), flush=True)
    a.close(); b.close()
print("layout fuzz seed %d: %d cases, %d runs, %d mismatches, %.0f s" % (seed, ncases, runs, bad, time.time() - t0))
sys.exit(1 if bad else 0)
