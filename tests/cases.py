"""Seeded input generators shared by the golden script and the tests."""
import numpy as np


def adversarial_case(rng, k):
    """One case of the adversarial family of SURVEY.md Appendix A item 2:
    n ~ U{50..799}, span ~ U{200..5999}, eps in {7,20,50,100}, minPts in {3,4,5,8},
    X ~ U{1000..1000+span}, Y = X + U{0..span}; every 3rd case gets exact duplicates,
    every 5th case loses a random quarter of its rows (non-contiguous ids).
    Returns (ids, X, Y, eps, minPts)."""
    n = int(rng.integers(50, 800))
    span = int(rng.integers(200, 6000))
    eps = int(rng.choice([7, 20, 50, 100]))
    minPts = int(rng.choice([3, 4, 5, 8]))
    X = rng.integers(1000, 1000 + span + 1, n)
    Y = X + rng.integers(0, span + 1, n)
    if k % 3 == 0:
        m = n // 4
        src = rng.integers(0, n, m)
        X[:m] = X[src]
        Y[:m] = Y[src]
    ids = np.arange(n)
    if k % 5 == 0:
        keep = np.sort(rng.permutation(n)[: n - n // 4])
        ids, X, Y = ids[keep], X[keep], Y[keep]
    return ids.astype(np.int64), X.astype(np.int64), Y.astype(np.int64), eps, minPts


def plain_case(rng, k):
    """Appendix A item 3: same family, no duplicates, no row dropping."""
    n = int(rng.integers(50, 800))
    span = int(rng.integers(200, 6000))
    eps = int(rng.choice([7, 20, 50, 100]))
    minPts = int(rng.choice([3, 4, 5, 8]))
    X = rng.integers(1000, 1000 + span + 1, n)
    Y = X + rng.integers(0, span + 1, n)
    return np.arange(n, dtype=np.int64), X.astype(np.int64), Y.astype(np.int64), eps, minPts


def clumpy_case(rng, k):
    """Dense blobs + background so that crowded cells, staircase (Pareto) neighbours and
    contested border points are frequent -- the regime of real PET data around anchors."""
    n_blobs = int(rng.integers(3, 30))
    eps = int(rng.choice([10, 25, 60, 150]))
    minPts = int(rng.choice([3, 5, 8, 12, 20]))
    span = int(rng.integers(20, 120)) * eps
    pts = []
    for _ in range(n_blobs):
        cx = rng.integers(1000, 1000 + span)
        cy = cx + rng.integers(0, span)
        m = int(rng.integers(2, 60))
        s = float(rng.choice([0.2, 0.5, 1.0, 2.0])) * eps
        pts.append(np.stack([cx + rng.normal(0, s, m), cy + rng.normal(0, s, m)], 1))
    nb = int(rng.integers(20, 400))
    bx = rng.integers(1000, 1000 + span, nb)
    pts.append(np.stack([bx, bx + rng.integers(0, span, nb)], 1).astype(float))
    P = np.rint(np.concatenate(pts)).astype(np.int64)
    P = P[rng.permutation(len(P))]
    X = np.minimum(P[:, 0], P[:, 1])
    Y = np.maximum(P[:, 0], P[:, 1])
    X = np.maximum(X, 0)
    return np.arange(len(X), dtype=np.int64), X, Y, eps, minPts
