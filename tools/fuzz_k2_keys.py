"""One-off GPU fuzz aimed at the sorted-key region query (k_region_keys, round 6; developer tool): dense data (mean strip population
above 40, where the kernel is picked) built to reach its rare paths -- dense blobs separated by thousands of EMPTY strips (the
strip_rel clamp: keys that cannot tell a PET's strips apart), pile-ups of many hundreds of PETs per strip (neighbour strips that leave
the staged window: clipping and the global-memory continuation), windows far beyond the 6-step upper-bound searches, announced minPts
lists with gaps of every size, minPts up to 128, cuts at both traversal levels that run the kernel (4: base layout; 3: compacted
copy with a filtered tail), both rotated variants.  Every run against the sequential oracle.
python tools/fuzz_k2_keys.py [seed] [ncases]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle
from cloops_amd import api

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rng = np.random.default_rng(seed)
oracle.build()
bad = runs = 0
t0 = time.time()
for k in range(ncases):
    eps = int(rng.choice([500, 2000, 5000, 10000]))
    nblob = int(rng.integers(1, 5))
    per = int(rng.integers(15000, 60000))
    gap = int(rng.choice([0, 50, 3000, 6000, 20000]))                      # empty strips between the blobs
    width = int(rng.integers(60, 600))                                     # strips per blob: 25 .. 1000 PETs per strip
    if k % 5 == 4:
        # the strip_rel clamp: two blobs dense enough that the MEAN strip population stays above 40 across a gap of more than 4095 strips
        nblob, per, gap, width = 2, int(rng.integers(120000, 200000)), int(rng.integers(4100, 5200)), int(rng.integers(100, 300))
    xs, ys = [], []
    x0 = int(rng.integers(0, 5 * eps))
    for bI in range(nblob):
        L = width * eps // 2
        n1 = per // 2
        bx = x0 + rng.integers(0, max(L, 1), per - n1)
        by = bx + np.exp(rng.uniform(np.log(10), np.log(max(40 * eps, 100)), per - n1)).astype(np.int64)
        ncl = max(1, n1 // int(rng.integers(20, 200)))
        ax = x0 + rng.integers(0, max(L, 1), ncl); span = rng.integers(0, 30 * eps, ncl)
        which = rng.integers(0, ncl, n1)
        sig = float(rng.choice([0.02, 0.1, 0.5])) * eps
        cx = ax[which] + rng.normal(0, sig, n1); cy = ax[which] + span[which] + rng.normal(0, sig, n1)
        xs += [bx, np.abs(cx).astype(np.int64)]; ys += [by, np.abs(cy).astype(np.int64)]
        if rng.random() < 0.5:                                             # a pile-up: hundreds to thousands of PETs in one or two strips
            m = int(rng.integers(300, 3000))
            px = x0 + int(rng.integers(0, max(L, 1))) + rng.integers(0, eps // 2 + 1, m)
            xs.append(px); ys.append(px + rng.integers(0, int(rng.choice([eps, 20 * eps])), m))
        x0 += L + gap * eps // 2
    X = np.concatenate(xs); Y = np.concatenate(ys)
    X, Y = np.minimum(X, Y), np.maximum(X, Y)
    if Y.max() >= (1 << 28) - 2 * eps:
        continue
    p = rng.permutation(len(X))
    X = np.ascontiguousarray(X[p], dtype=np.int32); Y = np.ascontiguousarray(Y[p], dtype=np.int32)
    served = sorted({int(v) for v in rng.integers(2, 129, int(rng.integers(1, 6)))}, reverse=True)
    variant = "v2" if k % 2 == 0 else "v1"
    level = 4 if k % 3 else 3
    a = api.Chromosome(X, Y)
    a.set_traversal(level)
    a.set_count_thresholds(served)
    seq = [(served[0], int(rng.integers(0, 3 * eps)) if rng.random() < 0.5 else 0)]
    for _ in range(int(rng.integers(1, 4))):
        seq.append((int(rng.choice(served)), int(rng.integers(0, 4 * eps))))
    for m, cut in seq:
        ra = a.cluster(variant, eps, m, cut)
        want = oracle.single_dbscan(variant, X, Y, eps, m, cut)["labels"]
        runs += 1
        if not np.array_equal(ra.labels, want):
            bad += 1
            print("MISMATCH case %d %s level %d n=%d eps=%d served=%s minPts %d cut %d gap %d width %d: %d rows differ" % (
                k, variant, level, len(X), eps, served, m, cut, gap, width, int((ra.labels != want).sum())), flush=True)
            if os.environ.get("FUZZ_DUMP"):
                np.savez_compressed(os.path.join(os.environ["FUZZ_DUMP"], "k2fuzz_%d_%d.npz" % (seed, k)), X=X, Y=Y, eps=eps, m=m, cut=cut, served=np.asarray(served),
                                    got=ra.labels, want=want, variant=variant, level=level)
    a.close()
    if k % 10 == 9:
        print("case %d: %d runs, %d bad, %.0f s" % (k, runs, bad, time.time() - t0), flush=True)
if os.environ.get("CLOOPS_DEVEL_LIB") == "1" and int(os.environ.get("CLOOPS_DBG", "0")) & 4096:
    import ctypes
    from cloops_amd import _lib
    st = (ctypes.c_ulonglong * 8)()
    _lib.load().cl_debug_k2stats(st)
    print("k_region_keys paths: launches %d, PETs in phase 2 %d, beyond the strip_rel clamp %d, through global memory %d, clipped windows %d, "
          "deferred walks %d, walks in place %d, capped windows %d" % tuple(int(v) for v in st))
print("k2 keys fuzz seed %d: %d cases, %d runs, %d mismatches, %.0f s" % (seed, ncases, runs, bad, time.time() - t0))
sys.exit(1 if bad else 0)
