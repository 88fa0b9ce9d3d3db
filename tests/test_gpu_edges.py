"""GPU edge cases against the CPU oracle (bit-exact): pile-ups larger than the LDS window,
degenerate grids, extreme minPts, tiny inputs, everything filtered, negative coordinates (v1)."""
import numpy as np
import pytest

import oracle
from cloops_amd import api, _lib

pytestmark = pytest.mark.gpu
ALL = ["v2", "v1", "block"]


def run_all(X, Y, eps, minPts, cut=0, variants=ALL):
    ch = api.Chromosome(X, Y)
    try:
        for v in variants:
            got = ch.cluster(v, eps, minPts, cut)
            want = oracle.single_dbscan(v, X, Y, eps, minPts, cut)["labels"]
            assert np.array_equal(got.labels, want), (v, eps, minPts, cut, int((got.labels != want).sum()))
            assert got.n_clusters == len(np.unique(want[want >= 0]))
    finally:
        ch.close()


def test_pileups_exceed_lds_window():
    """thousands of PETs inside one eps window: the staged halo (128) and span (120) overflow"""
    rng = np.random.default_rng(1)
    parts = []
    for c, m in ((50000, 3000), (52000, 700), (300000, 1500)):
        parts.append(np.stack([c + rng.integers(-40, 41, m), c + 9000 + rng.integers(-40, 41, m)], 1))
    bg = rng.integers(0, 600000, 4000)
    parts.append(np.stack([bg, bg + rng.integers(0, 50000, 4000)], 1))
    P = np.concatenate(parts)
    P = P[rng.permutation(len(P))]
    for eps, minPts in ((100, 5), (30, 20), (2000, 50)):
        run_all(P[:, 0], P[:, 1], eps, minPts)


def test_exact_duplicates_heavy():
    rng = np.random.default_rng(2)
    base = rng.integers(1000, 200000, 300)
    reps = rng.integers(1, 40, 300)
    X = np.repeat(base, reps)
    Y = X + np.repeat(rng.integers(0, 5000, 300), reps)
    p = rng.permutation(len(X))
    run_all(X[p], Y[p], 50, 5)
    run_all(X[p], Y[p], 1, 3)


@pytest.mark.parametrize("eps", [1, 3, 10 ** 6, 2 * 10 ** 8])
def test_degenerate_grid_sizes(eps):
    """eps = 1 (maximal strip table for this extent) .. eps larger than the whole extent (one strip)"""
    rng = np.random.default_rng(eps % 97)
    n = 4000
    X = rng.integers(0, 200000, n)
    Y = X + rng.integers(0, 100000, n)
    run_all(X, Y, eps, 4)


@pytest.mark.parametrize("minPts", [1, 2, 1000, 10 ** 6])
def test_extreme_minpts(minPts):
    rng = np.random.default_rng(7)
    n = 5000
    X = rng.integers(0, 100000, n)
    Y = X + rng.integers(0, 3000, n)
    run_all(X, Y, 500, minPts)


def test_tiny_inputs():
    for n in (1, 2, 3, 7):
        X = np.arange(n) * 3 + 100
        Y = X + 10
        run_all(X, Y, 5, 1)
        run_all(X, Y, 5, 2)
        run_all(X, Y, 100, 3)


def test_all_identical_points():
    X = np.full(2000, 12345)
    Y = np.full(2000, 54321)
    run_all(X, Y, 10, 5)
    run_all(X, Y, 10, 5000)


def test_cut_filters_everything_and_almost_everything():
    rng = np.random.default_rng(9)
    X = rng.integers(0, 100000, 3000)
    Y = X + rng.integers(0, 2000, 3000)
    run_all(X, Y, 300, 4, cut=10 ** 7)          # nothing survives: pipe.py:64-65
    run_all(X, Y, 300, 4, cut=1990)             # a handful survive
    run_all(X, Y, 300, 4, cut=1)                # only d == 0 rows are dropped


def test_negative_and_reversed_coordinates_v1_block():
    """cDBSCAN (v1) and blockDBSCAN accept any integers; cDBSCAN2 needs 0 <= X <= Y."""
    rng = np.random.default_rng(11)
    n = 6000
    X = rng.integers(-50000, 50000, n)
    Y = rng.integers(-50000, 50000, n)
    run_all(X, Y, 700, 4, variants=["v1", "block"])
    ch = api.Chromosome(X, Y)
    with pytest.raises(_lib.CloopsHipError) as ei:
        ch.cluster("v2", 700, 4)
    assert ei.value.code == _lib.CL_ERR_DOMAIN
    ch.close()


def test_coordinate_domain_limit():
    with pytest.raises(_lib.CloopsHipError) as ei:
        api.Chromosome(np.array([0, 1 << 29]), np.array([5, (1 << 29) + 5]))
    assert ei.value.code == _lib.CL_ERR_DOMAIN


def test_large_eps_giant_component_16M():
    """mode-3/4 regime: eps 5000, minPts 20 on 16 M PETs -- the self-ligation diagonal is one
    component of millions of PETs (two-level reductions, chain scans)."""
    from cloops_amd.synth import synth_chrom
    X, Y = synth_chrom(16000000, 248956422, 4242)
    ch = api.Chromosome(X, Y)
    try:
        got = ch.cluster("v2", 5000, 20, pinned=True)
        want = oracle.labels("v2", X, Y, 5000, 20)
        assert np.array_equal(got.labels, want)
        # cluster table against a host recomputation for the biggest clusters
        big = np.argsort(-got.boxes["count"])[:5]
        for c in big:
            sel = want == c
            b = got.boxes[c]
            assert (int(b["count"]), int(b["min_x"]), int(b["max_x"]), int(b["min_y"]), int(b["max_y"])) == (
                int(sel.sum()), int(X[sel].min()), int(X[sel].max()), int(Y[sel].min()), int(Y[sel].max()))
    finally:
        ch.close()


@pytest.mark.parametrize("n,L,eps,minPts", [(300000, 3000000, 20000, 30), (300000, 3000000, 20000, 5),
                                            (600000, 3000000, 50000, 100), (200000, 1000000, 3000, 8)])
def test_strips_longer_than_the_lds_window(n, L, eps, minPts):
    """dense data at large eps: a strip holds ~1000 PETs, tiles lie inside one strip and the neighbour
    strips are staged as separate (capped) far windows -- every source of a window is exercised:
    main LDS window, far window, capped far window + global continuation, plain global"""
    from cloops_amd.synth import synth_chrom
    X, Y = synth_chrom(n, L, 99)
    run_all(X, Y, eps, minPts)
    run_all(X, Y, eps, minPts, cut=eps // 2, variants=["v2"])


@pytest.mark.parametrize("variant", ["v2", "v1"])
def test_hybrid_sort_equals_full_radix_sort(variant, monkeypatch):
    """short strips everywhere -> the strip-level radix sort + in-strip ranking path; CLOOPS_DBG=256
    forces the full radix sort: same labels, same table, and both equal the oracle.  A second
    chromosome with a pile-up (a strip of thousands of PETs) must pick the full sort by itself."""
    from cloops_amd.synth import synth_chrom
    X, Y = synth_chrom(400000, 20000000, 77)
    want = oracle.labels(variant, X, Y, 2000, 5)
    for dbg in ("0", "256"):
        monkeypatch.setenv("CLOOPS_DBG", dbg)
        ch = api.Chromosome(X, Y)                       # fresh handle: the sort plan is per handle
        try:
            for cut in (0, 3000, 0):
                got = ch.cluster(variant, 2000, 5, cut)
                if cut == 0:
                    assert np.array_equal(got.labels, want), (variant, dbg)
                else:
                    keep = Y.astype(np.int64) - X >= cut
                    w2 = np.full(len(X), -1, np.int32)
                    w2[keep] = oracle.labels(variant, X[keep], Y[keep], 2000, 5)
                    assert np.array_equal(got.labels, w2), (variant, dbg, cut)
        finally:
            ch.close()
    monkeypatch.setenv("CLOOPS_DBG", "0")
    rng = np.random.default_rng(5)
    Xp = np.concatenate([X, 5000000 + rng.integers(-300, 301, 5000).astype(np.int32)])
    Yp = np.concatenate([Y, 5600000 + rng.integers(-300, 301, 5000).astype(np.int32)])
    run_all(Xp, Yp, 2000, 5, variants=[variant])


def test_failed_arena_reservation_is_harmless():
    """the one-allocation workspace reservation at upload is best effort: when it fails (memory pressure) the handle must
    work -- buffers allocated one by one -- and no stale HIP error may surface in the next run"""
    from cloops_amd import _lib as L
    rng = np.random.default_rng(3)
    n = 20000
    X = rng.integers(0, 400000, n)
    Y = X + rng.integers(0, 30000, n)
    lib = L.load()
    lib.cl_debug_arena_overcommit(1 << 60)
    try:
        ch = api.Chromosome(X, Y)
    finally:
        lib.cl_debug_arena_overcommit(0)
    try:
        for v in ALL:
            got = ch.cluster(v, 500, 4, 300)
            want = oracle.single_dbscan(v, X, Y, 500, 4, 300)["labels"]
            assert np.array_equal(got.labels, want), v
    finally:
        ch.close()


def test_walkers_beside_clusters_they_do_not_touch():
    """the border rule's queue (k_border_q): pairs of sparse PETs whose windows one strip away run along dense clusters they are
    NOT adjacent to -- nearly every walker has a walk left over after the capped steps, more than the queue of a tile holds
    (the rest finishes in place) -- next to walkers that are adjacent to two and three clusters (contested: release records)"""
    rng = np.random.default_rng(11)
    eps = 1000
    parts = []
    for k in range(40):
        c = 200000 + 6000 * k                                   # dense diagonal blobs: cores
        parts.append(np.stack([c + rng.integers(-150, 151, 700), c + 9000 + rng.integers(-150, 151, 700)], 1))
        # walkers in pairs, one strip further along the diagonal (X + Y larger by ~1.2-1.9 eps), same distance range: inside the q
        # window of the blob's strip but out of reach of its cores -- and some just within reach
        m = 400
        bx = c + rng.integers(-150, 151, m) + rng.integers(600, 950, m)
        by = bx + 9000 + rng.integers(-300, 301, m)
        parts.append(np.stack([bx, by], 1))
        parts.append(np.stack([bx + rng.integers(1, 40, m), by + rng.integers(1, 40, m)], 1))
    P = np.concatenate(parts)
    P = P[rng.permutation(len(P))]
    for minPts in (30, 8):
        run_all(P[:, 0], P[:, 1], eps, minPts, variants=["v2", "v1"])
        run_all(P[:, 0], P[:, 1], eps, minPts, cut=8800, variants=["v2"])


def _dense_blob(rng, x0, width, per, eps):
    """half log-uniform background, half clusters, on `width` strips from x0 on (mean strip population per / width)"""
    L = width * eps // 2
    n1 = per // 2
    bx = x0 + rng.integers(0, L, per - n1)
    by = bx + np.exp(rng.uniform(np.log(10), np.log(40 * eps), per - n1)).astype(np.int64)
    ncl = max(1, n1 // 60)
    ax = x0 + rng.integers(0, L, ncl)
    span = rng.integers(0, 30 * eps, ncl)
    which = rng.integers(0, ncl, n1)
    cx = np.abs(ax[which] + rng.normal(0, 0.1 * eps, n1)).astype(np.int64)
    cy = np.abs(ax[which] + span[which] + rng.normal(0, 0.1 * eps, n1)).astype(np.int64)
    return np.concatenate([bx, cx]), np.concatenate([by, cy]), L


def _from_rotated(P, Q):
    """(p = X + Y, q = Y - X) of equal parity -> X, Y"""
    P = np.asarray(P, np.int64); Q = np.asarray(Q, np.int64)
    Q = Q + ((P + Q) & 1)
    return (P - Q) // 2, (P + Q) // 2


@pytest.mark.parametrize("variant", ["v2", "v1"])
def test_region_query_keys_across_a_gap_of_thousands_of_empty_strips(variant):
    """the sorted-key region query packs `strip - first strip of the window` into 12 bits.  A dense blob (the MEAN strip population stays
    above 40: the kernel's regime), more than 4095 empty strips, then 30 consecutive strips of 20 PETs each at nearly ONE distance: every
    PET there has 20 neighbours in its strip and at most 40 in the two strips beside it -- nobody is core at minPts 70 -- but at the
    clamp of the strip field no key says "same strip", and a test on the keys alone sees 600 PETs at one distance (found by
    tools/fuzz_k2_keys.py; they are counted through the strip table)"""
    rng = np.random.default_rng(29)
    eps = 500
    x1, y1, L = _dense_blob(rng, 1000, 250, 250000, eps)
    p0 = ((2 * (1000 + L) + 80 * eps) // eps + 4300) * eps                 # first strip behind the gap (strips: p // eps)
    t = np.repeat(np.arange(30), 20)
    P = p0 + t * eps + 250 + rng.integers(0, 10, len(t))
    x2, y2 = _from_rotated(P, 1000 + rng.integers(0, 10, len(t)))
    X = np.concatenate([x1, x2]); Y = np.concatenate([y1, y2])
    X, Y = np.minimum(X, Y), np.maximum(X, Y)
    assert (250000 % 1024) + len(t) > 256                                   # the tile that holds the blob's last PETs holds most of them
    p = rng.permutation(len(X))
    X, Y = X[p], Y[p]
    behind = (X + Y) >= p0
    ch = api.Chromosome(X, Y)
    try:
        ch.set_count_thresholds([70, 35])
        for m, cut in ((70, 0), (35, 0), (70, 400)):
            got = ch.cluster(variant, eps, m, cut)
            want = oracle.single_dbscan(variant, X, Y, eps, m, cut)["labels"]
            if m == 70:
                assert (want[behind] < 0).all()                            # (the construction: nobody behind the gap is clustered)
            assert np.array_equal(got.labels, want), (m, cut)
    finally:
        ch.close()


@pytest.mark.parametrize("level", [4, 3])
def test_region_query_keys_beside_a_pile_up(level):
    """a strip of 1300 PETs beside a strip of 48 PETs high in their strip (their neighbours one strip below are the pile-up's PETs that
    lie higher still: a sixth of it), minPts 91: every one of the 48 needs its candidates walked, their q windows in the pile-up hold
    300 to 1300 PETs -- walks deferred to the workgroup's list (window sizes of up to 1023 fit an entry) next to walks that must run
    in place, in ONE wave: the slots of the list must not leave gaps (found by tools/fuzz_k2_keys.py)"""
    rng = np.random.default_rng(26)
    eps = 5000
    x, y, L = _dense_blob(rng, 20000, 450, 60000, eps)
    sp = (2 * (20000 + L) + 80 * eps) // eps + 8                          # the pile-up's strip: a few empty strips behind the blob
    xp, yp = _from_rotated(sp * eps + rng.integers(0, eps, 1300), rng.integers(0, eps, 1300))
    xq, yq = _from_rotated((sp + 1) * eps + rng.integers(int(0.7 * eps), int(0.95 * eps), 48), np.linspace(0.2 * eps, 1.9 * eps, 48).astype(np.int64))
    X = np.concatenate([x, xp, xq]); Y = np.concatenate([y, yp, yq])
    X, Y = np.minimum(X, Y), np.maximum(X, Y)
    p = rng.permutation(len(X))
    X, Y = X[p], Y[p]
    ch = api.Chromosome(X, Y)
    try:
        ch.set_traversal(level)
        ch.set_count_thresholds([91, 86])
        for m, cut in ((91, 0), (86, 0), (91, 1500)):
            got = ch.cluster("v2", eps, m, cut)
            assert np.array_equal(got.labels, oracle.single_dbscan("v2", X, Y, eps, m, cut)["labels"]), (m, cut)
    finally:
        ch.close()


def test_region_query_falls_back_when_a_distance_leaves_the_key_field():
    """the sorted-key region query holds q in 28 bits of its keys; a PET whose distance reaches 2^28 - eps (coordinates go up to 2^29)
    sends the dense shapes back to the pair-predicate kernel (k_region_core): eps 2^20 keeps the mean strip population above 40 over
    the 256 strips such a PET opens up -- same labels either way"""
    rng = np.random.default_rng(5)
    eps = 1 << 20
    x, y, _ = _dense_blob(rng, 5000, 40, 30000, eps)
    X = np.concatenate([x, [10, 20]]); Y = np.concatenate([y, [10 + (1 << 28) - 1000, 20 + (1 << 28) - 900]])
    X, Y = np.minimum(X, Y), np.maximum(X, Y)
    assert Y.max() < (1 << 29) and (Y - X).max() + eps + 1 >= (1 << 28)
    p = rng.permutation(len(X))
    run_all(X[p], Y[p], eps, 50, variants=["v2", "v1"])
    run_all(X[p], Y[p], eps, 50, cut=300000, variants=["v2"])
