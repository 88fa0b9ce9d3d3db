"""CPU: the C oracle reproduces every committed golden vector (made from the real
reference by tests/golden/make_golden.py).  Runs anywhere (no GPU, no reference)."""
import numpy as np
import pytest

import golden_util as G
import oracle


@pytest.mark.parametrize("variant", ["v1", "v2", "block"])
@pytest.mark.parametrize("eps,minPts", [(500, 5), (1000, 5), (2000, 5), (5000, 20)])
def test_chr21_unchained(variant, eps, minPts):
    X, Y = G.chr21_xy()
    gold = G.chr21_labels(variant, eps, minPts)
    got = oracle.labels(variant, X, Y, eps, minPts)
    assert np.array_equal(gold, got)
    m = G.meta()["chr21_%s_%d_%d" % (variant, eps, minPts)]
    assert int((got >= 0).sum()) == m["labelled"]
    assert len(np.unique(got[got >= 0])) == m["clusters"]


@pytest.mark.parametrize("variant", ["v2", "v1"])
def test_chr21_mode1_chain(variant):
    """pipe.py:247-275 chain on config 1: labels and the derived counts per step."""
    X, Y = G.chr21_xy()
    for step in G.meta()["chr21_chain_" + variant]:
        r = oracle.single_dbscan(variant, X, Y, step["eps"], step["minPts"], step["cut_in"])
        assert np.array_equal(r["labels"], G.chr21_chain_labels(variant, step["eps"]))
        assert len(r["dataI"]) == step["inter"] and len(r["dataS"]) == step["self"]
        assert len(r["dis"]) == step["n_dis"] and len(r["dss"]) == step["n_dss"]


@pytest.mark.parametrize("family", ["adversarial", "plain", "clumpy"])
def test_families(family):
    regen = G.regenerate_family(family)
    for (k, ids, X, Y, eps, minPts, gold), rg in zip(G.family_cases(family), regen):
        # the committed inputs are what tests/cases.py regenerates from the seed
        assert np.array_equal(ids, rg[0]) and np.array_equal(X, rg[1]) and np.array_equal(Y, rg[2])
        assert (eps, minPts) == (rg[3], rg[4])
        for variant in ("v1", "v2", "block"):
            got = oracle.labels(variant, X, Y, eps, minPts)
            assert np.array_equal(gold[variant], got), (family, k, variant)


@pytest.mark.parametrize("variant", ["v1", "v2", "block"])
def test_synth150k(variant):
    X, Y, z = G.synth150k()
    for eps, minPts in ((2000, 5), (5000, 20)):
        got = oracle.labels(variant, X, Y, eps, minPts)
        assert np.array_equal(z["%s_%d_%d" % (variant, eps, minPts)], got)


@pytest.mark.parametrize("variant", ["v2", "v1", "block"])
def test_dense400k_headline_regime(variant):
    """the oracle against the REAL classes at the headline's density and settings (minPts 20-50 on strips of hundreds of PETs:
    cDBSCAN2.py:194-302 crowded cells / Pareto edge points), cut filter of pipe.py:59-63 in front"""
    X, Y, z, m = G.dense400k()
    for eps, minPts, cut in G.DENSE_SETTINGS:
        key = "%s_%d_%d_%d" % (variant, eps, minPts, cut)
        got = oracle.single_dbscan(variant, X, Y, eps, minPts, cut)["labels"]
        assert np.array_equal(got, z[key]), key
        assert int((got >= 0).sum()) == m["runs"][key]["labelled"]


def test_empty_input_behaviour():
    e = np.zeros(0, np.int64)
    assert len(oracle.labels("v2", e, e, 100, 5)) == 0          # cDBSCAN2: {}
    for v in ("v1", "block"):                                    # cDBSCAN.py:77 / blockDBSCAN.py:74
        with pytest.raises(IndexError):
            oracle.labels(v, e, e, 100, 5)


def test_neighbor_counts_bruteforce():
    rng = np.random.default_rng(5)
    X = rng.integers(0, 500, 300)
    Y = X + rng.integers(0, 500, 300)
    c = oracle.neighbor_counts(X, Y, 40)
    d = np.abs(X[:, None] - X[None, :]) + np.abs(Y[:, None] - Y[None, :])
    assert np.array_equal(c, (d <= 40).sum(1))


def test_oracle_v1_on_scaled_coordinates_matches_the_real_class():
    """scripts/callStripes:44-46 stretches one axis by ext = 50 before cDBSCAN (variant 1): the oracle takes
    64-bit coordinates, and on the x50-scaled chr21 matrix it reproduces the labels of the REAL class
    (tests/golden/make_golden_stripes.py) -- which makes it the checker of cl_cluster_weighted."""
    import os
    X, Y = G.chr21_xy()
    z = np.load(os.path.join(G.GOLD, "chr21_stripes_labels.npz"))
    for name, wx, wy in (("x50", 50, 1), ("y50", 1, 50)):
        lab = oracle.labels("v1", X.astype(np.int64) * wx, Y.astype(np.int64) * wy, 20000, 5)
        assert np.array_equal(lab, z[name]), name
