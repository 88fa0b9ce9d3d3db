"""GPU: region-query reuse inside one eps (cl_set_count_reuse / cl_set_count_floor / cl_set_count_thresholds, include/cloops_hip.h).

The first run at an eps keeps its per-PET neighbour-count words; later runs at that eps (any cut, a minPts between the
announced floor and the first run's) take the words of the PETs outside the cut band as they are and run the region query
on the band only.  Whatever the order of (minPts, cut), the labels and tables must equal those of a handle with the
cache switched off -- and the sequential C oracle (cDBSCAN2.py:333-334 / cDBSCAN.py:186-205 count test, pipe.py:59-63
filter before clustering, pipe.py:247-250 loop order)."""
import numpy as np
import pytest

import golden_util as G
import oracle
from cloops_amd import api
from cloops_amd.synth import synth_chrom

pytestmark = pytest.mark.gpu


def _check_seq(X, Y, variant, floor, seq, oracle_at=(), modes=None, served=None):
    a = api.Chromosome(X, Y)
    b = api.Chromosome(X, Y)
    b.set_count_reuse(False)
    if served is not None:
        a.set_count_thresholds(served)          # the minPts values themselves (what the sweep driver announces)
    else:
        a.set_count_floor(floor)
    got_modes = []
    try:
        for k, (eps, m, cut) in enumerate(seq):
            ra = a.cluster(variant, eps, m, cut)
            got_modes.append(a.last_region_mode())
            rb = b.cluster(variant, eps, m, cut)
            assert b.last_region_mode() == 0
            assert np.array_equal(ra.labels, rb.labels), (variant, k, eps, m, cut, int((ra.labels != rb.labels).sum()))
            assert ra.n_clusters == rb.n_clusters and np.array_equal(ra.boxes, rb.boxes)
            if k in oracle_at:
                want = oracle.single_dbscan(variant, X, Y, eps, m, cut)["labels"]
                assert np.array_equal(ra.labels, want), (variant, k, eps, m, cut)
    finally:
        a.close()
        b.close()
    if modes is not None:
        assert got_modes == modes, got_modes
    return got_modes


@pytest.mark.parametrize("variant", ["v2", "v1"])
def test_dense_chain_like_mode3(variant):
    """dense strips (the tile shape with deferred candidate loops): the cut moving up and down like pipe()'s chain"""
    X, Y = synth_chrom(1500000, 20000000, 77)
    seq = [(5000, 50, 0), (5000, 40, 4536), (5000, 30, 6098), (5000, 20, 6306),          # make, remap x 3
           (5000, 20, 6306), (5000, 50, 6306), (5000, 30, 5000),                          # remap (words made at cut 0), ...
           (5000, 20, 0), (5000, 35, 0),                                                  # same cut as the run that made them
           (5000, 10, 3000),                                                              # below the floor: a new set of words
           (5000, 10, 3000), (5000, 12, 4000), (5000, 50, 4000),                          # same / remap / above its cap: new
           (7500, 50, 5711), (7500, 40, 3871), (7500, 30, 5004), (7500, 20, 5256)]        # words made on a cut layout, cuts below and above
    modes = [0, 2, 2, 2, 2, 2, 2, 1, 1, 0, 1, 0, 0, 0, 2, 2, 2]
    if api.TRAVERSAL_OVERRIDE is None or api.TRAVERSAL_OVERRIDE >= 4:
        # traversal level 4 makes the words of an eps on the base layout itself (threshold 0) also when the making run has a cut: the
        # second run at (10, 3000) re-uses them under "another" cut (band query) instead of as they are
        modes[10] = 2
        # ... and a run without a cut cannot take words made under a cut as they are (the PETs below that cut got none): (5000, 20, 0)
        # behind the chain of cuts still can (the words of eps 5000 were made at cut 0); nothing else changes in this sequence
    # (5000, 12, 4000): the words of (5000, 10, 3000) have cap 10 < 12 -> a new set with cap 12, floor 12 ... then 50 > 12
    _check_seq(X, Y, variant, 20, seq, oracle_at=(1, 3, 7, 11, 14, 16), modes=modes)


@pytest.mark.parametrize("variant", ["v2", "v1"])
def test_dense_chain_announced_list(variant):
    """the same data with the minPts LIST announced (cl_set_count_thresholds): counts are kept only as exact as the tests of
    those values need; a minPts outside the list makes a new set of words"""
    X, Y = synth_chrom(1500000, 20000000, 77)
    seq = [(5000, 50, 0), (5000, 40, 4536), (5000, 30, 6098), (5000, 20, 6306),          # make, remap x 3
           (5000, 20, 6306), (5000, 50, 6306), (5000, 30, 5000), (5000, 40, 0),           # remap ... same cut as the making run
           (5000, 35, 0),                                                                 # not announced: new words (cap 35; serves 20, 30, 35)
           (5000, 30, 2000), (5000, 20, 0), (5000, 40, 0),                                # remap, same, above the cap: new
           (7500, 50, 5711), (7500, 40, 3871), (7500, 30, 5004), (7500, 20, 5256),        # words made on a cut layout
           (10000, 40, 5517), (10000, 50, 5517), (10000, 20, 4896), (10000, 30, 5977)]    # made at 40 (serves 20, 30, 40); 50: new words
    modes = [0, 2, 2, 2, 2, 2, 2, 1, 0, 2, 1, 0, 0, 2, 2, 2, 0, 0, 2, 2]
    _check_seq(X, Y, variant, 0, seq, oracle_at=(1, 3, 7, 9, 13, 15, 19), modes=modes, served=[50, 40, 30, 20])
    # an odd list, values next to each other and far apart
    seq = [(7500, 64, 0), (7500, 63, 4000), (7500, 33, 5000), (7500, 32, 5200), (7500, 5, 6000), (7500, 2, 0), (7500, 64, 3000)]
    _check_seq(X, Y, variant, 0, seq, oracle_at=(1, 2, 4), modes=[0, 2, 2, 2, 2, 1, 2], served=[2, 5, 32, 33, 63, 64])


@pytest.mark.parametrize("variant", ["v2", "v1"])
def test_sparse_chr21_small_minpts(variant):
    """the example data, the sparse tile shape, minPts down to 2"""
    X, Y = G.chr21_xy()
    seq = [(500, 8, 0), (500, 5, 4601), (500, 3, 300), (500, 2, 13532), (500, 6, 0),
           (2000, 5, 11103), (2000, 5, 0), (2000, 4, 20000), (2000, 2, 500), (2000, 5, 10 ** 9)]
    _check_seq(X, Y, variant, 2, seq, oracle_at=(1, 3, 6, 8))


def test_pileups_and_duplicates():
    """windows that leave the staged range (words without hints), exact duplicates, a cut inside a pile-up"""
    rng = np.random.default_rng(5)
    parts = []
    for c, d, m in ((50000, 9000, 3000), (52000, 9100, 700), (300000, 400, 1500), (300100, 8000, 900)):
        parts.append(np.stack([c + rng.integers(-40, 41, m), c + d + rng.integers(-40, 41, m)], 1))
    bg = rng.integers(0, 600000, 6000)
    parts.append(np.stack([bg, bg + rng.integers(0, 50000, 6000)], 1))
    P = np.concatenate(parts)
    P = np.concatenate([P, P[rng.integers(0, len(P), 800)]])
    P = P[rng.permutation(len(P))]
    X, Y = np.ascontiguousarray(P[:, 0]), np.ascontiguousarray(P[:, 1])
    for variant in ("v2", "v1"):
        for eps in (100, 2000):
            seq = [(eps, 20, 0), (eps, 10, 390), (eps, 5, 8990), (eps, 5, 9050), (eps, 3, 200), (eps, 20, 9000)]
            _check_seq(X, Y, variant, 3, seq, oracle_at=(1, 2, 4))


def test_fuzz_orders():
    """seeded random data / orders: every run against the cache-less handle, a sample against the oracle"""
    rng = np.random.default_rng(11)
    for case in range(12):
        n = int(rng.integers(3000, 80000))
        eps = int(rng.choice([200, 1000, 5000]))
        L = int(rng.integers(20, 400)) * eps
        X = rng.integers(0, L, n)
        spread = float(rng.choice([0.5, 2, 10])) * eps
        Y = X + np.abs(rng.normal(0, spread, n)).astype(np.int64)
        X = np.ascontiguousarray(X, dtype=np.int32)
        Y = np.ascontiguousarray(Y, dtype=np.int32)
        ms = sorted({int(v) for v in rng.choice([2, 3, 5, 8, 20, 33, 50, 64, 100, 128], 4)}, reverse=True)
        seq = [(eps, m, int(rng.integers(0, 3 * spread)) if k else 0) for k, m in enumerate(ms + ms[::-1])]
        variant = "v2" if case % 2 == 0 else "v1"
        _check_seq(X, Y, variant, min(ms), seq, oracle_at=(1, len(seq) - 2), served=ms if case % 3 == 0 else None)


def test_async_step_form_matches_sync():
    """the sweep driver's asynchronous form: two result slots, words re-used across in-flight runs"""
    X, Y = synth_chrom(800000, 12000000, 31)
    a = api.Chromosome(X, Y)
    b = api.Chromosome(X, Y)
    b.set_count_reuse(False)
    a.set_count_floor(20)
    runs = [(7500, 50, 0), (7500, 40, 4100), (7500, 30, 5200), (7500, 20, 4800), (10000, 50, 4800), (10000, 20, 6000)]
    try:
        for k in range(0, len(runs), 2):
            for eps, m, cut in runs[k:k + 2]:
                a.cluster_async("v2", eps, m, cut)
            ra = [a.wait(copy=True) for _ in runs[k:k + 2]]
            for (eps, m, cut), r in zip(runs[k:k + 2], ra):
                rb = b.cluster("v2", eps, m, cut)
                assert np.array_equal(r.labels, rb.labels), (eps, m, cut)
                assert np.array_equal(r.boxes, rb.boxes)
    finally:
        a.close()
        b.close()
