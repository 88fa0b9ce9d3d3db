"""GPU, BASELINE.json's full sizes: properties that need no oracle (the oracle takes minutes there).

 * the cluster table is the exact per-label count / bounding box of the labels (pipe.py:78-102),
 * runs are deterministic (lock-free union-find, atomics, two result slots),
 * a run with `cut` equals a run on the pre-filtered chromosome (pipe.py:59-63),
 * neighbour counts: every pair is counted from both ends, counts grow with eps.
plus, because the C oracle does 2 M PETs/s, the bit-exact comparison of configs[1] itself: the labels of all 5 M PETs
against the sequential oracle for the three variants (test_cfg2_labels_equal_oracle)."""
import numpy as np
import pytest

from cloops_amd import api
from cloops_amd.synth import synth_chrom

pytestmark = pytest.mark.gpu
CHR1 = 248956422


def check_table(X, Y, res):
    lab = res.labels
    b = res.boxes
    K = len(b)
    sel = lab >= 0
    assert (lab[sel] < K).all()
    cnt = np.bincount(lab[sel], minlength=K)
    assert np.array_equal(cnt, b["count"])
    assert res.n_clusters == int((cnt > 0).sum())
    for name, arr, fn, init in (("min_x", X, np.minimum, np.iinfo(np.int32).max), ("max_x", X, np.maximum, np.iinfo(np.int32).min),
                                ("min_y", Y, np.minimum, np.iinfo(np.int32).max), ("max_y", Y, np.maximum, np.iinfo(np.int32).min)):
        want = np.full(K, init, np.int64)
        fn.at(want, lab[sel], arr[sel])
        live = cnt > 0
        assert np.array_equal(want[live], b[name][live].astype(np.int64)), name


@pytest.fixture(scope="module")
def cfg2():
    X, Y = synth_chrom(5000000, CHR1, 2000)             # BASELINE.json configs[1]
    return X, Y


@pytest.mark.parametrize("variant", ["v2", "v1", "block"])
def test_cfg2_table_and_determinism(cfg2, variant):
    X, Y = cfg2
    ch = api.Chromosome(X, Y)
    try:
        r1 = ch.cluster(variant, 2000, 5)
        l1 = r1.labels.copy()
        check_table(X, Y, r1)
        r2 = ch.cluster(variant, 2000, 5)
        assert np.array_equal(l1, r2.labels)
        assert r1.n_clusters > 50000 and (l1 >= 0).mean() > 0.5          # the workload really clusters
    finally:
        ch.close()


@pytest.mark.parametrize("variant", ["v2", "v1", "block"])
def test_cfg2_labels_equal_oracle(cfg2, variant):
    """BASELINE.json configs[1] at full size, bit-exact: 5 M PETs, eps 2000, minPts 5 -- every label against the
    sequential C oracle (cDBSCAN2.py:7-383 / cDBSCAN.py:6-205 / blockDBSCAN.py:6-239 restated)"""
    import oracle
    X, Y = cfg2
    ch = api.Chromosome(X, Y)
    try:
        got = ch.cluster(variant, 2000, 5).labels
    finally:
        ch.close()
    want = oracle.labels(variant, X, Y, 2000, 5)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("variant", ["v2", "v1", "block"])
def test_cfg2_cut_equals_prefiltered(cfg2, variant):
    X, Y = cfg2
    cut = 4601                                          # a cut of the golden chain
    keep = Y.astype(np.int64) - X >= cut
    a = api.Chromosome(X, Y)
    b = api.Chromosome(np.ascontiguousarray(X[keep]), np.ascontiguousarray(Y[keep]))
    try:
        ra = a.cluster(variant, 2000, 5, cut)
        rb = b.cluster(variant, 2000, 5, 0)
        assert (ra.labels[~keep] == -1).all()
        assert np.array_equal(ra.labels[keep], rb.labels)
        assert a.last_n_in() == int(keep.sum())
        assert np.array_equal(ra.boxes, rb.boxes)
    finally:
        a.close(); b.close()


def test_cfg2_neighbor_count_properties(cfg2):
    X, Y = cfg2
    ch = api.Chromosome(X, Y)
    try:
        c1 = ch.neighbor_counts(1000)
        c2 = ch.neighbor_counts(2000)
        assert (c1 >= 1).all() and (c2 >= c1).all()
        assert int((c1.astype(np.int64) - 1).sum()) % 2 == 0 and int((c2.astype(np.int64) - 1).sum()) % 2 == 0
        # the clustering path's saturated core test agrees with the exact counts
        lab = ch.cluster("v1", 2000, 5).labels
        assert (lab[c2 < 5] >= -1).all() and (c2[lab >= 0] >= 1).all()
    finally:
        ch.close()


def test_cfg3_dense_regime_40M():
    """40 M PETs, eps 10000, minPts 50 (BASELINE.json configs[2] regime: strips of ~800 PETs, one
    giant diagonal component): table consistency and determinism of the production variant."""
    X, Y = synth_chrom(40000000, CHR1, 3000)
    ch = api.Chromosome(X, Y)
    try:
        r1 = ch.cluster("v2", 10000, 50)
        l1 = r1.labels.copy()
        check_table(X, Y, r1)
        r2 = ch.cluster("v2", 10000, 50)
        assert np.array_equal(l1, r2.labels)
    finally:
        ch.close()


@pytest.mark.parametrize("wx,wy", [(50, 1), (1, 50)])
def test_cfg2_weighted_metric_table_and_determinism(cfg2, wx, wy):
    """callStripes' stretched metric on a whole chromosome: scaled coordinates reach 1.2e10 (64-bit path)"""
    X, Y = cfg2
    ch = api.Chromosome(X, Y)
    try:
        r1 = ch.cluster_weighted(20000, 5, wx, wy)
        check_table(X, Y, r1)
        r2 = ch.cluster_weighted(20000, 5, wx, wy)
        assert np.array_equal(r1.labels, r2.labels)
        assert r1.n_clusters > 50000
    finally:
        ch.close()
